// Pure-MFMA issue rate on gfx950: v_mfma_f32_16x16x4_f32 vs v_mfma_f32_32x32x2_f32, independent accumulators,
// 1 or 2 waves per SIMD.  hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static void run(const char* name, F launch, double flop_per_wave_iter, int iters, int wgs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = flop_per_wave_iter * iters * 4.0 * wgs;
  printf("%-44s %8.3f ms  %7.1f TF\n", name, ms, flops / ms / 1e9);
}
int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 20000;
  for (int wpc = 1; wpc <= 2; ++wpc) {
    const int wgs = 256 * wpc;
    char nm[128];
    snprintf(nm, 128, "16x16x4 f32, 16 accumulators, %d WG/CU", wpc);
    run(nm, [&] { k16<16><<<wgs, 256>>>(out, iters, 1.f, 2.f); }, 16 * 2048.0, iters, wgs);
    snprintf(nm, 128, "16x16x4 f32,  4 accumulators, %d WG/CU", wpc);
    run(nm, [&] { k16<4><<<wgs, 256>>>(out, iters * 4, 1.f, 2.f); }, 4 * 2048.0, iters * 4, wgs);
    snprintf(nm, 128, "32x32x2 f32,  4 accumulators, %d WG/CU", wpc);
    run(nm, [&] { k32<4><<<wgs, 256>>>(out, iters * 2, 1.f, 2.f); }, 4 * 4096.0, iters * 2, wgs);
    snprintf(nm, 128, "32x32x2 f32,  2 accumulators, %d WG/CU", wpc);
    run(nm, [&] { k32<2><<<wgs, 256>>>(out, iters * 4, 1.f, 2.f); }, 2 * 4096.0, iters * 4, wgs);
  }
  return 0;
}
