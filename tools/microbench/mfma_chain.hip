// Dependent-issue latency of v_mfma_f32_16x16x4_f32 on gfx950: NACC independent accumulator chains per wave, one wave per
// SIMD (256 workgroups x 4 waves).  A chain's next MFMA needs the previous result as SrcC: with few chains the matrix pipe
// waits.  hipcc -O3 --offload-arch=gfx950 mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256, 1) k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(float* out) {
  const int iters = 160000 / (8 * NACC) * 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k16<NACC><<<256, 256>>>(out, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k16<NACC><<<256, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 8 * NACC;  // MFMAs per wave
  printf("chains %d: %8.3f ms  %7.1f TF  %6.1f ns per MFMA per wave\n", NACC, ms, n * 2048.0 * 1024 / ms / 1e9, ms * 1e6 / n);
}
int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<6>(out); run<8>(out); run<16>(out);
  return 0;
}
