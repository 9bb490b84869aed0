// panel_shape.hip's feature ladder for the FUSED kernel's tile (32 columns per wave) in two instruction shapes:
//   K = 16:  13 row tiles x 2 column tiles of v_mfma_f32_16x16x4_f32   (panel_elbo_kernel<13,2>: 104 MFMAs of 32 cycles per chunk)
//   K = 32:  7 row tiles x 1 column tile of v_mfma_f32_32x32x2_f32     (56 MFMAs of 64 cycles per chunk, 224 rows)
// F = 0 register operands; 1 + A fragments from LDS; 2 + barrier per chunk; 3 + LDS-DMA staging + counted wait;
// 4 + per-chunk B operands.  One wave per SIMD, 256 workgroups.
// hipcc -O3 --offload-arch=gfx950 panel_shape2.hip -o panel_shape2 && ./panel_shape2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t" ::"v"(g), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}

template <int F>
__global__ void __launch_bounds__(256, 1) shape16(const float* __restrict__ P, const float* __restrict__ X, float* out, int nl) {
  constexpr int MB = 13, NCT = 2, NPW = 4;
  __shared__ __attribute__((aligned(16))) float lds[3][NPW * 4 * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 3 * NPW * 4 * 256; i += 256) (&lds[0][0])[i] = P[i];
  __syncthreads();
  float xb[NCT][MB][4];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int t = 0; t < MB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[ct][t][r] = X[((ct * MB + t) * 4 + r) * 64 + lane];
  f32x4 acc[MB][NCT];
#pragma unroll
  for (int rt = 0; rt < MB; ++rt)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int buf = 0;
  const float* src = P + lane * 4;
  float sink[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 sink4[4];
  if (F == 6) asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr(&lds[2][w * 256]))) : "m0");
  for (int l = 0; l < nl; ++l) {
#pragma unroll
    for (int kc = 0; kc < MB; ++kc) {
      float bv[NCT][4];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[ct][r] = F >= 4 ? xb[ct][kc][r] : xb[ct][0][r];
      const float* base = &lds[buf][lane * 4];
      float4 a_nxt = *reinterpret_cast<const float4*>(base);
#pragma unroll
      for (int rt = 0; rt < MB; ++rt) {
        const float4 a4 = F >= 1 ? a_nxt : make_float4(xb[0][rt][0], xb[0][rt][1], xb[0][rt][2], xb[0][rt][3]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[ct][r], acc[rt][ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (F == 9 || F == 10) {
            // F = 9: the DMA (with its m0 write) FIRST, while no LDS read of this wave is outstanding, the next fragment's
            // read one K step later.  F = 10: m0 written once per chunk (rt == 0), the pieces through the immediate offset
            if (r == 0 && rt < NPW) {
              const int piece = rt * 4 + w;
              const float* gp = src + ((l * MB + kc) & 63) * 4096 + piece * 256;
              if (F == 9) {
                glds16(gp, __builtin_amdgcn_readfirstlane(lds_addr(&lds[buf == 0 ? 2 : buf - 1][piece * 256])));
              } else {
                if (rt == 0)
                  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr(&lds[buf == 0 ? 2 : buf - 1][w * 1024]))) : "memory", "m0");
                if (rt == 0) asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp) : "memory");
                if (rt == 1) asm volatile("global_load_lds_dwordx4 %0, off offset:1024" ::"v"(gp) : "memory");
                if (rt == 2) asm volatile("global_load_lds_dwordx4 %0, off offset:2048" ::"v"(gp) : "memory");
                if (rt == 3) asm volatile("global_load_lds_dwordx4 %0, off offset:3072" ::"v"(gp) : "memory");
              }
            }
            if (r == 1 && rt + 1 < MB) a_nxt = *reinterpret_cast<const float4*>(base + (rt + 1) * 256);
          } else if (r == 0 && F >= 1) {
            if (rt + 1 < MB) a_nxt = *reinterpret_cast<const float4*>(base + (rt + 1) * 256);
          } else if (r == 1 && F >= 3) {
            if (rt < NPW) {
              const int piece = rt * 4 + w;
              const float* gp = src + ((l * MB + kc) & 63) * 4096 + piece * 256;
              if (F == 5) {          // a plain 4-byte load into a register (no LDS involved)
                asm volatile("global_load_dword %0, %1, off" : "=&v"(sink[rt]) : "v"(gp) : "memory");
              } else if (F == 6) {   // LDS-DMA without touching m0 (set once, same target every time)
                asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp) : "memory");
              } else if (F == 7) {   // only the m0 set-up
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr(&lds[buf == 0 ? 2 : buf - 1][piece * 256]))) : "memory", "m0");
              } else if (F == 8) {   // a 16-byte load into registers
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(sink4[rt]) : "v"(gp) : "memory");
              } else {
                glds16(gp, __builtin_amdgcn_readfirstlane(lds_addr(&lds[buf == 0 ? 2 : buf - 1][piece * 256])));
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (F >= 3 && F != 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (F >= 2) __syncthreads();
      buf = (buf == 2) ? 0 : buf + 1;
    }
  }
  if (F >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = sink[0] + sink[1] + sink[2] + sink[3];
  if (F == 8) s += sink4[0][0] + sink4[1][1] + sink4[2][2] + sink4[3][3];
#pragma unroll
  for (int rt = 0; rt < MB; ++rt)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s += acc[rt][ct][0] + acc[rt][ct][1] + acc[rt][ct][2] + acc[rt][ct][3];
  out[blockIdx.x * 256 + tid] = s;
}

// 32x32x2: a chunk = 16 k of 224 rows = 14 pieces of 1 KiB (piece rt2: rows 32 rt2 .. +31 would be 2 KiB; here a row tile of
// 32 rows x 16 k = 2 pieces, read as two ds_read_b128: lane (i = lane % 32, h = lane / 32) holds A[i][8 h' + ...])
template <int F>
__global__ void __launch_bounds__(256, 1) shape32(const float* __restrict__ P, const float* __restrict__ X, float* out, int nl) {
  constexpr int MR = 7, NKC = 13, NPW = 4;  // 7 row tiles of 32; 13 chunks of 16 k (208)
  __shared__ __attribute__((aligned(16))) float lds[3][NPW * 4 * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 3 * NPW * 4 * 256; i += 256) (&lds[0][0])[i] = P[i];
  __syncthreads();
  float xb[NKC][8];  // B operand: 8 K steps of 2 per chunk
#pragma unroll
  for (int t = 0; t < NKC; ++t)
#pragma unroll
    for (int r = 0; r < 8; ++r) xb[t][r] = X[(t * 8 + r) * 64 + lane];
  f32x16 acc[MR];
#pragma unroll
  for (int rt = 0; rt < MR; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[rt][i] = 0.f;
  int buf = 0;
  const float* src = P + lane * 4;
  for (int l = 0; l < nl; ++l) {
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      float bv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) bv[r] = F >= 4 ? xb[kc][r] : xb[0][r];
      const float* base = &lds[buf][lane * 4];
      float4 a0n = *reinterpret_cast<const float4*>(base), a1n = *reinterpret_cast<const float4*>(base + 256);
#pragma unroll
      for (int rt = 0; rt < MR; ++rt) {
        const float4 a0 = F >= 1 ? a0n : make_float4(xb[0][0], xb[0][1], xb[0][2], xb[0][3]);
        const float4 a1 = F >= 1 ? a1n : make_float4(xb[1][0], xb[1][1], xb[1][2], xb[1][3]);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], bv[r], acc[rt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (r == 0 && F >= 1) {
            if (rt + 1 < MR) a0n = *reinterpret_cast<const float4*>(base + (2 * rt + 2) * 256);
          } else if (r == 1 && F >= 1) {
            if (rt + 1 < MR) a1n = *reinterpret_cast<const float4*>(base + (2 * rt + 3) * 256);
          } else if (r == 2 && F >= 3) {
            if (rt < NPW) {
              const int piece = rt * 4 + w;
              glds16(src + ((l * NKC + kc) & 63) * 4096 + piece * 256,
                     __builtin_amdgcn_readfirstlane(lds_addr(&lds[buf == 0 ? 2 : buf - 1][piece * 256])));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (F >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (F >= 2) __syncthreads();
      buf = (buf == 2) ? 0 : buf + 1;
    }
  }
  if (F >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int rt = 0; rt < MR; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[rt][i];
  out[blockIdx.x * 256 + tid] = s;
}

template <typename L>
static void run(const char* name, L launch, double flops) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-72s %8.3f ms  %7.1f TF  %.3f of 157.3\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}
int main() {
  float *P, *X, *out;
  (void)hipMalloc(&P, 64 * 4096 * 4 + 65536);
  (void)hipMemset(P, 0, 64 * 4096 * 4 + 65536);
  (void)hipMalloc(&X, 3 * 13 * 8 * 64 * 4);
  (void)hipMemset(X, 0, 3 * 13 * 8 * 64 * 4);
  (void)hipMalloc(&out, 256 * 256 * 4);
  const int nl = 200;
  setvbuf(stdout, nullptr, _IONBF, 0);
  const double f16 = 2048.0 * 13 * 13 * 8 * nl * 4.0 * 256, f32 = 4096.0 * 13 * 7 * 8 * nl * 4.0 * 256;
#define R16(F, NAME) run("16x16x4, 13x2 tiles: " NAME, [&] { shape16<F><<<256, 256>>>(P, X, out, nl); }, f16)
#define R32(F, NAME) run("32x32x2,  7x1 tiles: " NAME, [&] { shape32<F><<<256, 256>>>(P, X, out, nl); }, f32)
  R16(0, "register operands");
  R16(1, "+ A fragments from LDS");
  R16(2, "+ barrier per chunk");
  R16(3, "+ LDS-DMA staging, counted wait");
  R16(4, "+ per-chunk B operands");
  R16(5, "F=3 with plain global_load_dword into a VGPR instead");
  R16(6, "F=3 without the m0 set-up (m0 set once)");
  R16(9, "F=3 with the DMA + m0 write BEFORE the fragment read");
  R16(10, "F=3 with ONE m0 write per chunk, pieces by immediate offset");
  R16(7, "F=2 + only the m0 set-up (no load)");
  R32(0, "register operands");
  R32(1, "+ A fragments from LDS");
  R32(2, "+ barrier per chunk");
  R32(3, "+ LDS-DMA staging, counted wait");
  R32(4, "+ per-chunk B operands");
  return 0;
}
