// Would TWO waves per SIMD help the fused kernel's loop?  (round 6; a question DESIGN.md section 9 leaves open)
// panel_elbo_kernel runs one wave per SIMD (the whole 512-register file: 13 x 2 accumulators twice + the alpha slab);
// whenever that wave waits - the counted vmcnt and the barrier per chunk, an LDS fragment - the SIMD's matrix pipe idles:
// its loop reaches 0.91 of the MFMA rate in isolation (panel_shape2.hip, F = 3).  Here the SAME tile (13 row tiles x 2
// column tiles per SIMD and chunk, A fragments from LDS, LDS-DMA staging of the next chunks, one counted wait + barrier
// per chunk) is run
//   SOLO: by one wave per SIMD (256 threads), all 13 row tiles;
//   DUO:  by two waves per SIMD (512 threads): the same 32 columns in both (the B operand is held twice), row tiles 0..6
//         in one, 7..12 in the other - 56 / 48 MFMAs per chunk each, 13 x 2 x 4 = 104 per SIMD as before.
// hipcc -O3 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 panel_duo.hip -o panel_duo && ./panel_duo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MB = 13, NCT = 2;

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t" ::"v"(g), "s"(lds) : "memory", "m0");
}

// one wave's share of a chunk: row tiles RT0 .. RT1 - 1; NW waves deal the 13 pieces of the next-but-one chunk
template <int RT0, int RT1, int NW>
__device__ __forceinline__ void chunk(const float* lbase, f32x4 (&acc)[MB][NCT], const float (&bv)[NCT][4], const float* gsrc,
                                      unsigned lds_dst, int w) {
  float4 a_nxt = *reinterpret_cast<const float4*>(lbase + RT0 * 256);
#pragma unroll
  for (int rt = RT0; rt < RT1; ++rt) {
    const float4 a4 = a_nxt;
    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[ct][r], acc[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (r == 0 && rt + 1 < RT1) a_nxt = *reinterpret_cast<const float4*>(lbase + (rt + 1) * 256);
      if (r == 1) {  // this wave's pieces of the stage: piece w (+ NW while it exists), one per row tile
        const int i = rt - RT0, piece = w + NW * i;
        if (i < (MB + NW - 1) / NW && piece < MB) glds16(gsrc + piece * 256, lds_dst + piece * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <bool DUO>
__global__ void __launch_bounds__(DUO ? 512 : 256, 1) loop(const float* __restrict__ P, const float* __restrict__ X, float* out, int nl) {
  constexpr int NW = DUO ? 8 : 4;
  __shared__ __attribute__((aligned(16))) float lds[3][16 * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = DUO ? (w >> 2) : 0;
  for (int i = tid; i < 3 * 16 * 256; i += NW * 64) (&lds[0][0])[i] = P[i];
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
  constexpr int NPW = (MB + NW - 1) / NW;  // DMA operations a wave may have in flight per stage
  for (int l = 0; l < nl; ++l) {
#pragma unroll
    for (int kc = 0; kc < MB; ++kc) {
      float bv[NCT][4];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[ct][r] = xb[ct][kc][r];
      const float* lbase = &lds[buf][lane * 4];
      const float* g = src + ((l * MB + kc) & 63) * 4096;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(&lds[buf == 0 ? 2 : buf - 1][0]));
      if (!DUO) chunk<0, MB, 4>(lbase, acc, bv, g, dst, w);
      else if (h == 0) chunk<0, 7, 8>(lbase, acc, bv, g, dst, w);
      else chunk<7, MB, 8>(lbase, acc, bv, g, dst, w);
      if (NPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      __syncthreads();
      buf = (buf == 2) ? 0 : buf + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int rt = 0; rt < MB; ++rt)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s += acc[rt][ct][0] + acc[rt][ct][1] + acc[rt][ct][2] + acc[rt][ct][3];
  out[blockIdx.x * (NW * 64) + tid] = s;
}

template <typename F>
static void run(const char* name, F launch, int nl) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  // MFMAs per SIMD: nl outputs x 13 chunks x 104; executed flops per MFMA 2048
  const double flops = (double)nl * MB * 104 * 2048.0 * 4 * 256;
  printf("%-62s %8.3f ms  %6.1f TF executed = %.3f of 157.3\n", name, best, flops / best / 1e9, flops / best / 1e9 / 157.3);
}

int main() {
  float *P, *X, *out;
  (void)hipMalloc(&P, 64 * 4096 * 4 + 65536);
  (void)hipMalloc(&X, NCT * MB * 4 * 64 * 4);
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMemset(P, 0, 64 * 4096 * 4 + 65536);
  (void)hipMemset(X, 0, NCT * MB * 4 * 64 * 4);
  const int nl = 150;
  run("SOLO: one wave per SIMD, 13 row tiles (the kernel's loop)", [&] { loop<false><<<256, 256>>>(P, X, out, nl); }, nl);
  run("DUO:  two waves per SIMD, 7 + 6 row tiles, same columns", [&] { loop<true><<<256, 512>>>(P, X, out, nl); }, nl);
  run("SOLO again", [&] { loop<false><<<256, 256>>>(P, X, out, nl); }, nl);
  run("DUO again", [&] { loop<true><<<256, 512>>>(P, X, out, nl); }, nl);
  return 0;
}
