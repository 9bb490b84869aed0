// Does v_mfma_f32_16x16x4_f32 run at the same rate with its B operand in an AGPR and its accumulator in arch VGPRs?
// 8 independent accumulators per wave, one wave per SIMD.  hipcc -O3 --offload-arch=gfx950 mfma_operand.hip && ./a.out
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: acc AGPR, B VGPR (compiler default)   1: acc VGPR, B VGPR   2: acc VGPR, B AGPR   3: acc AGPR, B AGPR
__global__ void __launch_bounds__(256, 1) k(float* out, int iters, float a, float b) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bb[i] = b + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(bb[i]));
        if (MODE == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(bb[i]));
        if (MODE == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(bb[i]));
        if (MODE == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "a"(bb[i]));
      }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, float* out) {
  const int iters = 5000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<256, 256>>>(out, iters, 1.f, 2.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<256, 256>>>(out, iters, 1.f, 2.f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 32;
  printf("%-40s %8.3f ms  %7.1f TF\n", name, ms, n * 2048.0 * 1024 / ms / 1e9);
}
int main() {
  float* out;
  (void)hipMalloc(&out, 4096 * 256 * 4);
  run<0>("acc AGPR, B VGPR", out);
  run<1>("acc VGPR, B VGPR", out);
  run<2>("acc VGPR, B AGPR", out);
  run<3>("acc AGPR, B AGPR", out);
  return 0;
}
