// On-device Lloyd iterations for the inducing-point initialisation.
// Replaces sklearn.cluster.KMeans at gpsa/models/vgpsa.py:74-76, 90-92 when the coordinates already live
// in HBM.  Deterministic: no atomics — per-block partial sums in a fixed order, then a fixed-order sum.
#include "common.hpp"

namespace gpsa {

constexpr int KM_CHUNK = 256;  // centres staged in LDS per pass

// assign[n] = argmin_k |x_n - c_k|^2 (ties -> lowest k); d2[n] = that distance (may be NULL)
__global__ void __launch_bounds__(256)
kmeans_assign_kernel(const float* __restrict__ X, long long N, int D, const float* __restrict__ Cc,
                     int K, int* __restrict__ assign, float* __restrict__ d2) {
  __shared__ float cs[KM_CHUNK][MAXD];
  const long long n = blockIdx.x * 256LL + threadIdx.x;
  float x[MAXD];
#pragma unroll
  for (int d = 0; d < MAXD; ++d) x[d] = (n < N && d < D) ? X[n * D + d] : 0.f;
  float best = 3.0e38f;
  int bi = 0;
  for (int k0 = 0; k0 < K; k0 += KM_CHUNK) {
    const int kc = min(KM_CHUNK, K - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < KM_CHUNK * MAXD; i += 256) {
      const int r = i / MAXD, d = i % MAXD;
      cs[r][d] = (r < kc && d < D) ? Cc[(long long)(k0 + r) * D + d] : 0.f;
    }
    __syncthreads();
    for (int r = 0; r < kc; ++r) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < MAXD; ++d) {
        const float u = x[d] - cs[r][d];
        s += u * u;
      }
      if (s < best) {
        best = s;
        bi = k0 + r;
      }
    }
  }
  if (n < N) {
    assign[n] = bi;
    if (d2) d2[n] = best;
  }
}

// part[blk][k][0..D) = sum of the block's points assigned to k, part[blk][k][D] = their count
__global__ void __launch_bounds__(256)
kmeans_partial_kernel(const float* __restrict__ X, const int* __restrict__ assign, long long N, int D,
                      int K, double* __restrict__ part) {
  __shared__ float xs[256][MAXD];
  __shared__ int as[256];
  const long long n = blockIdx.x * 256LL + threadIdx.x;
  as[threadIdx.x] = (n < N) ? assign[n] : -1;
#pragma unroll
  for (int d = 0; d < MAXD; ++d) xs[threadIdx.x][d] = (n < N && d < D) ? X[n * D + d] : 0.f;
  __syncthreads();
  double* prow = part + (long long)blockIdx.x * K * (D + 1);
  for (int k = threadIdx.x; k < K; k += 256) {
    double s[MAXD] = {0.0, 0.0, 0.0, 0.0};
    double cnt = 0.0;
    for (int p = 0; p < 256; ++p)
      if (as[p] == k) {
#pragma unroll
        for (int d = 0; d < MAXD; ++d) s[d] += (double)xs[p][d];
        cnt += 1.0;
      }
    for (int d = 0; d < D; ++d) prow[(long long)k * (D + 1) + d] = s[d];
    prow[(long long)k * (D + 1) + D] = cnt;
  }
}

// centres[k] = sum / count (an empty cluster keeps its previous centre); counts[k] = count
__global__ void kmeans_finish_kernel(const double* __restrict__ part, long long nblk, int D, int K,
                                     float* __restrict__ Cc, int* __restrict__ counts) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  double s[MAXD] = {0.0, 0.0, 0.0, 0.0};
  double cnt = 0.0;
  for (long long b = 0; b < nblk; ++b) {
    const double* p = part + (b * K + k) * (D + 1);
    for (int d = 0; d < D; ++d) s[d] += p[d];
    cnt += p[D];
  }
  if (cnt > 0.0)
    for (int d = 0; d < D; ++d) Cc[(long long)k * D + d] = (float)(s[d] / cnt);
  if (counts) counts[k] = (int)cnt;
}

}  // namespace gpsa

extern "C" {

long long gpsa_kmeans_workspace(long long N, int D, int K) {
  return cdiv(N, 256) * (long long)K * (D + 1) * 8;
}

int gpsa_kmeans_assign(const float* X, long long N, int D, const float* centres, int K, int* assign,
                       float* d2, void* stream) {
  if (N < 1 || K < 1 || D < 1 || D > gpsa::MAXD) return GPSA_EINVAL;
  gpsa::kmeans_assign_kernel<<<(unsigned)cdiv(N, 256), 256, 0, as_stream(stream)>>>(X, N, D, centres, K,
                                                                                   assign, d2);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_kmeans_update(const float* X, const int* assign, long long N, int D, int K, float* centres,
                       int* counts, void* workspace, long long workspace_bytes, void* stream) {
  if (N < 1 || K < 1 || D < 1 || D > gpsa::MAXD) return GPSA_EINVAL;
  const long long nblk = cdiv(N, 256);
  if (workspace_bytes < nblk * (long long)K * (D + 1) * 8) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* part = (double*)workspace;
  gpsa::kmeans_partial_kernel<<<(unsigned)nblk, 256, 0, st>>>(X, assign, N, D, K, part);
  gpsa::kmeans_finish_kernel<<<(unsigned)cdiv(K, 128), 128, 0, st>>>(part, nblk, D, K, centres, counts);
  GPSA_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
