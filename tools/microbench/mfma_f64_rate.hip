// What v_mfma_f64_16x16x4_f64 really delivers on gfx950: NACC independent accumulators per wave, WPS waves per SIMD
// (WPS = 1: 256 workgroups of 4 waves; WPS = 2: 512 workgroups), register operands only.  The projection kernel
// (whiten_mfma_kernel) is priced against this, not against the data-sheet 78.6 TF.
// hipcc -O3 --offload-arch=gfx950 mfma_f64_rate.hip -o mfma_f64_rate && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256, 2) k64(double* out, int iters, double a, double b) {
  f64x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(double* out, int wgs) {
  const int iters = 40000 / (4 * NACC) + 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k64<NACC><<<wgs, 256>>>(out, iters, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k64<NACC><<<wgs, 256>>>(out, iters, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 4 * NACC;  // MFMAs per wave
  printf("accumulators %2d, %d workgroups: %8.3f ms  %6.1f TF fp64  %6.1f ns per MFMA per wave\n", NACC, wgs, ms,
         n * 2048.0 * 4 * wgs / ms / 1e9, ms * 1e6 / n);
}
int main() {
  double* out;
  hipMalloc(&out, 1024 * 256 * 8);
  run<1>(out, 256); run<2>(out, 256); run<4>(out, 256); run<13>(out, 256);
  run<1>(out, 512); run<4>(out, 512); run<13>(out, 512);
  return 0;
}
