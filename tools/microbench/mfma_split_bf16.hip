// EXPLORATORY (round-5 verdict item 9): loop-level rate of the three-way bf16 split of the fp32 contraction against the
// fp32 MFMA loop of panel_elbo_kernel, same tile shape: one wave per SIMD, 13 row tiles x 2 column tiles of 16 x 16
// accumulators, the B operand (the wave's 32 columns of alpha) resident in registers, the A operand (Omega_l) read from
// LDS as fragments.  fp32: K = 208 as 52 v_mfma_f32_16x16x4_f32 per tile, one ds_read_b128 per 8 MFMAs.  split: K = 224
// (7 blocks of 32) as 7 x 6 v_mfma_f32_16x16x32_bf16 per tile - the products a_i b_j with i + j <= 4 of
// a = a1 + a2 + a3 -, three ds_read_b128 (one per plane) per 12 MFMAs.  LDS holds a fixed image (no refills): this is
// the MFMA + LDS-read ceiling, not a kernel.  hipcc -O3 --offload-arch=gfx950 mfma_split_bf16.hip -o m && ./m
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int MB = 13, NCT = 2;

__global__ void __launch_bounds__(256, 1) loop_f32(float* out, int outputs, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[MB * 256];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < MB * 256; i += 256) lds[i] = seed * (float)(i % 7);
  __syncthreads();
  float xb[NCT][MB][4];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int kc = 0; kc < MB; ++kc)
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[ct][kc][r] = seed + (float)(ct + kc + r + lane);
  f32x4 acc[MB][NCT];
  float keep = 0.f;
  for (int l = 0; l < outputs; ++l) {
#pragma unroll
    for (int kc = 0; kc < MB; ++kc) {
#pragma unroll
      for (int rt = 0; rt < MB; ++rt) {
        const float4 a4 = *reinterpret_cast<const float4*>(&lds[rt * 256 + lane * 4]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                av[r], xb[ct][kc][r], (kc == 0 && r == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[rt][ct], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < MB; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) keep += acc[rt][ct][0] + acc[rt][ct][3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = keep;
}

template <int NPROD>  // 6: i + j <= 4 of three pieces; 4 / 3: two pieces, all four products / without a2 b2
__global__ void __launch_bounds__(256, 1) loop_split(float* out, int outputs, float seed) {
  constexpr int NP = NPROD == 6 ? 3 : 2, KB = 7;
  __shared__ __attribute__((aligned(16))) unsigned short lds[MB * NP * 512];  // [rt][plane][64 lanes x 8 bf16]
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < MB * NP * 512; i += 256) lds[i] = (unsigned short)(0x3f80 + (i % 5));
  __syncthreads();
  bf16x8 xb[NP][NCT][KB];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) xb[p][ct][kb][e] = (__bf16)(seed + (float)(p + ct + kb + e + (lane & 3)));
  f32x4 acc[MB][NCT];
  float keep = 0.f;
  for (int l = 0; l < outputs; ++l) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int rt = 0; rt < MB; ++rt) {
        bf16x8 av[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) av[p] = *reinterpret_cast<const bf16x8*>(&lds[(rt * NP + p) * 512 + lane * 8]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          f32x4 c = (kb == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[rt][ct];
          // smallest products first
          if (NPROD == 6) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2], xb[0][ct][kb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], xb[1][ct][kb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], xb[2][ct][kb], c, 0, 0, 0);
          }
          if (NPROD == 4) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], xb[1][ct][kb], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], xb[0][ct][kb], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], xb[1][ct][kb], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], xb[0][ct][kb], c, 0, 0, 0);
          acc[rt][ct] = c;
        }
      }
    }
#pragma unroll
    for (int rt = 0; rt < MB; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) keep += acc[rt][ct][0] + acc[rt][ct][3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = keep;
}

template <typename F>
static double run(const char* name, F launch, int outputs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  // useful flops: the M = 200 product of one output for a 128-column workgroup tile: 2 * 200 * 200 * 128
  const double flops = 2.0 * 200 * 200 * 128 * (double)outputs * 256;
  printf("%-58s %8.3f ms  %7.1f TF (fp32-equivalent, M = 200)  %6.2f us per output and workgroup\n", name, best,
         flops / best / 1e9, best * 1e3 / outputs);
  return best;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * 4);
  const int outputs = 400;
  const double a = run("fp32: 52 x v_mfma_f32_16x16x4_f32 per tile (the kernel's loop)", [&] { loop_f32<<<256, 256>>>(out, outputs, 1.f); }, outputs);
  const double b = run("bf16 x 3, six products: 42 x v_mfma_f32_16x16x32_bf16 per tile", [&] { loop_split<6><<<256, 256>>>(out, outputs, 1.f); }, outputs);
  const double c = run("bf16 x 2, four products: 28 x 16x16x32 per tile", [&] { loop_split<4><<<256, 256>>>(out, outputs, 1.f); }, outputs);
  const double d = run("bf16 x 2, three products: 21 x 16x16x32 per tile", [&] { loop_split<3><<<256, 256>>>(out, outputs, 1.f); }, outputs);
  printf("ratio fp32 / split: six products %.2f, four %.2f, three %.2f\n", a / b, a / c, a / d);
  return 0;
}
