// Where do the ~15 % between a pure-MFMA loop (0.97 of peak) and panel_mfma_kernel<13,3,QUAD> (0.83) go?
// The kernel's inner loop rebuilt feature by feature (one wave per SIMD, 256 workgroups of 256 threads):
//   F = 0  39 accumulators (13 row tiles x 3 column tiles), 4-step K chains per tile, operands in registers
//   F = 1  + the A fragment of every row tile read from LDS (ds_read_b128, one tile ahead)
//   F = 2  + one workgroup barrier per chunk (13 row tiles)
//   F = 3  + LDS-DMA staging of the next-but-one chunk (4 x 16-byte global_load_lds per wave and chunk) and the
//            counted vmcnt wait in front of the barrier
//   F = 4  + B operands that change per chunk (13 x 12 registers, as the kernel's xb slab)
// hipcc -O3 --offload-arch=gfx950 panel_shape.hip -o panel_shape && ./panel_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off\n\t" ::"v"(g),
      "s"(lds)
      : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}

template <int F>
__global__ void __launch_bounds__(256, 1) shape(const float* __restrict__ P, const float* __restrict__ X, float* out,
                                                 int nl) {
  constexpr int MB = 13, NCT = 3, NPW = 4;
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
  for (int l = 0; l < nl; ++l) {
#pragma unroll
    for (int kc = 0; kc < MB; ++kc) {
      float bv[NCT][4];
      float4 treg[NPW];
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
          if (r == 0 && F >= 1) {
            if (rt + 1 < MB) a_nxt = *reinterpret_cast<const float4*>(base + (rt + 1) * 256);
          } else if (F == 6 && r == 1) {
            if (rt < NPW)
              treg[rt] = *reinterpret_cast<const float4*>(src + ((l * MB + kc) & 63) * 4096 + (rt * 4 + w) * 256);
            if (rt >= 8 && rt < 8 + NPW)
              *reinterpret_cast<float4*>(&lds[buf == 0 ? 2 : buf - 1][((rt - 8) * 4 + w) * 256 + lane * 4]) = treg[rt - 8];
          } else if (F == 10 && r == 1) {
            if (rt < NPW) {
              const int piece = rt * 4 + w;
              asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\t" ::"v"(src + ((l * MB + kc) & 63) * 4096 + piece * 256),
                  "s"(__builtin_amdgcn_readfirstlane(lds_addr(&lds[buf == 0 ? 2 : buf - 1][piece * 256]))) : "memory", "m0");
            }
          } else if (F == 11 && r == 1) {
            if (rt < NPW) {  // saddr form: uniform 64-bit base in SGPRs, one constant 32-bit lane offset
              const int piece = rt * 4 + w;
              const float* ub = P + ((l * MB + kc) & 63) * 4096 + piece * 256;
              const unsigned long long ubi = (unsigned long long)ub;
              const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ubi), hi = __builtin_amdgcn_readfirstlane((unsigned)(ubi >> 32));
              const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
              asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t" ::"v"(lane * 16), "s"(sb),
                  "s"(__builtin_amdgcn_readfirstlane(lds_addr(&lds[buf == 0 ? 2 : buf - 1][piece * 256]))) : "memory", "m0");
            }
          } else if (F == 7 && r == 1) {
            if (rt == 0) {
#pragma unroll
              for (int pc = 0; pc < NPW; ++pc) {
                const int piece = pc * 4 + w;
                glds16(src + ((l * MB + kc) & 63) * 4096 + piece * 256,
                       __builtin_amdgcn_readfirstlane(lds_addr(&lds[buf == 0 ? 2 : buf - 1][piece * 256])));
              }
            }
          } else if ((F == 5 ? r == 3 : r == 1) && F >= 3) {
            if (F == 8 ? (rt == 0) : (F == 9 ? (rt >= 6 && rt < 6 + NPW) : rt < NPW)) {
              const int piece = (F == 9 ? rt - 6 : rt) * 4 + w;
              glds16(src + ((l * MB + kc) & 63) * 4096 + piece * 256,
                     __builtin_amdgcn_readfirstlane(lds_addr(&lds[buf == 0 ? 2 : buf - 1][piece * 256])));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (F == 8) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else if (F >= 3 && F != 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (F >= 2) __syncthreads();
      buf = (buf == 2) ? 0 : buf + 1;
    }
  }
  if (F >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int rt = 0; rt < MB; ++rt)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s += acc[rt][ct][0] + acc[rt][ct][1] + acc[rt][ct][2] + acc[rt][ct][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int F>
static void run(const char* name, const float* P, const float* X, float* out, int nl) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  shape<F><<<256, 256>>>(P, X, out, nl);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  shape<F><<<256, 256>>>(P, X, out, nl);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2048.0 * 13 * 13 * 12 * nl * 4.0 * 256;
  printf("F=%d %-60s %8.3f ms  %7.1f TF  %.3f of 157.3\n", F, name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}
int main() {
  float *P, *X, *out;
  hipMalloc(&P, 64 * 4096 * 4 + 65536);
  hipMemset(P, 0, 64 * 4096 * 4 + 65536);
  hipMalloc(&X, 3 * 13 * 4 * 64 * 4);
  hipMemset(X, 0, 3 * 13 * 4 * 64 * 4);
  hipMalloc(&out, 256 * 256 * 4);
  const int nl = 200;
  setvbuf(stdout, nullptr, _IONBF, 0);
  run<0>("39 accumulators, K chains of 4, register operands", P, X, out, nl);
  run<1>("+ A fragments from LDS (one tile ahead)", P, X, out, nl);
  run<2>("+ barrier per chunk", P, X, out, nl);
  run<3>("+ LDS-DMA staging and counted wait", P, X, out, nl);
  run<4>("+ per-chunk B operands (156-register slab)", P, X, out, nl);
  run<5>("F=3 with the DMA issued after the LAST K step of a tile", P, X, out, nl);
  run<6>("F=2 + classic staging: global_load_dwordx4 -> ds_write_b128", P, X, out, nl);
  run<10>("F=3 with global_load_lds_dword (a quarter of the bytes)", P, X, out, nl);
  run<11>("F=3 with SGPR base + constant lane offset (saddr form)", P, X, out, nl);
  run<7>("F=3 with the 4 DMA issues in one clump", P, X, out, nl);
  run<8>("F=3 with ONE DMA issue per wave and chunk", P, X, out, nl);
  run<9>("F=3 with the DMA issues in row tiles 6..9", P, X, out, nl);
  return 0;
}
