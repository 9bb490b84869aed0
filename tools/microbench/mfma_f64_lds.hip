// The projection kernel's inner loop in isolation (gfx950): 13 fp64 MFMAs per K step whose A operands are 13
// ds_read_b64 of a lane-linear LDS image, 4 K steps per chunk, one barrier per chunk; 2 workgroups per CU.
//   MODE 0: register operands only            MODE 1: + A fragments from LDS, read in the K step that uses them
//   MODE 2: + barrier per chunk               MODE 3: as 2, the next K step's fragments read before this step's MFMAs
// hipcc -O3 --offload-arch=gfx950 mfma_f64_lds.hip -o mfma_f64_lds && ./mfma_f64_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int MB = 13;

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(double* out, int chunks, double a0, double b0) {
  __shared__ double lds[2][4 * MB * 64];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2 * 4 * MB * 64; i += 256) (&lds[0][0])[i] = a0 + 1e-9 * i;
  __syncthreads();
  f64x4 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double b = b0;
  double nxt[MB];
  if (MODE == 3) {
#pragma unroll
    for (int rt = 0; rt < MB; ++rt) nxt[rt] = lds[0][rt * 64 + lane];
  }
  for (int c = 0; c < chunks; ++c) {
    const double* base = &lds[c & 1][lane];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double a[MB];
      if (MODE == 0) {
#pragma unroll
        for (int rt = 0; rt < MB; ++rt) a[rt] = a0;
      } else if (MODE == 3) {
#pragma unroll
        for (int rt = 0; rt < MB; ++rt) a[rt] = nxt[rt];
        const double* nb = (ks < 3) ? base + (ks + 1) * MB * 64 : &lds[(c + 1) & 1][lane];
#pragma unroll
        for (int rt = 0; rt < MB; ++rt) nxt[rt] = nb[rt * 64];
      } else {
#pragma unroll
        for (int rt = 0; rt < MB; ++rt) a[rt] = base[(ks * MB + rt) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < MB; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], b, acc[rt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE >= 2) __syncthreads();
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < MB; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(double* out, int wgs) {
  const int chunks = 13 * 40;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<wgs, 256>>>(out, chunks, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<wgs, 256>>>(out, chunks, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)chunks * 4 * MB;
  printf("mode %d, %4d workgroups: %8.3f ms  %6.1f TF fp64 (%.2f of 77.2)\n", MODE, wgs, ms, n * 2048.0 * 4 * wgs / ms / 1e9,
         n * 2048.0 * 4 * wgs / ms / 1e9 / 77.2);
}
int main() {
  double* out;
  hipMalloc(&out, 2048 * 256 * 8);
  for (int wgs : {256, 512}) {
    run<0>(out, wgs); run<1>(out, wgs); run<2>(out, wgs); run<3>(out, wgs);
  }
  return 0;
}
