// KL( N(delta_l, Omega_l) || N(mu_l, K) ) for l = 1..L and its backward, fp64, from the factorisations
// the step already holds (K^-1, logdet K, Omega_l^-1, logdet Omega_l).  Replaces
// torch.distributions.kl.kl_divergence(MultivariateNormal, MultivariateNormal) at
// gpsa/models/vgpsa.py:506-516, 520-530 and its autograd.  Two small fused kernels instead of ~30
// elementwise launches per KL term.
#include "common.hpp"

namespace gpsa {

// one 1024-thread block per l:  kl[l] = 0.5 (logdetK - logdetO[l] + <Kinv, Omega_l> + d_l^T KD_l - M)
// with KD = Kinv d (computed by the caller with the dense-product kernel)
__global__ void __launch_bounds__(1024)
mvn_kl_fwd_kernel(const double* __restrict__ Kinv, const double* __restrict__ logdetK,
                  const double* __restrict__ Omega, long long omega_stride,
                  const double* __restrict__ logdetO, long long logdet_stride,
                  const double* __restrict__ Dm, const double* __restrict__ KD, int M, int L,
                  double* __restrict__ kl) {
  __shared__ double red[16];
  const int l = blockIdx.x;
  const double* Om = Omega + (long long)l * omega_stride;
  double acc = 0.0;
  for (int e = threadIdx.x; e < M * M; e += 1024) acc += Kinv[e] * Om[e];
  for (int m = threadIdx.x; m < M; m += 1024) acc += KD[(long long)m * L + l] * Dm[(long long)m * L + l];
  const double t = block_sum(acc, red);
  if (threadIdx.x == 0)
    kl[l] = 0.5 * (logdetK[0] - logdetO[(long long)l * logdet_stride] + t - (double)M);
}

// thread per (i,j):  dOmega[l][i][j] = 0.5 g_l (Kinv - Oinv_l)[i][j]
//                    Sp[i][j] = (sum_l g_l) K[i][j] - sum_l g_l (Omega_l[i][j] + d[i,l] d[j,l])
// so that dK = 0.5 Kinv Sp Kinv;   block 0 also writes dDm[m,l] = g_l KD[m,l].
__global__ void __launch_bounds__(256)
mvn_kl_bwd_kernel(const double* __restrict__ Kuu, const double* __restrict__ Kinv,
                  const double* __restrict__ Omega, long long omega_stride,
                  const double* __restrict__ Oinv, long long oinv_stride,
                  const double* __restrict__ Dm, const double* __restrict__ KD,
                  const double* __restrict__ g, int M, int L, double* __restrict__ dOmega,
                  double* __restrict__ dDm, double* __restrict__ Sp) {
  const long long e = blockIdx.x * 256LL + threadIdx.x;
  if (blockIdx.x == 0)
    for (int t = threadIdx.x; t < M * L; t += 256) dDm[t] = g[t % L] * KD[t];
  if (e >= (long long)M * M) return;
  const int i = (int)(e / M), j = (int)(e % M);
  const double kin = Kinv[e];
  double s = 0.0, gs = 0.0;
  for (int l = 0; l < L; ++l) {
    const double gl = g[l];
    gs += gl;
    const double om = Omega[(long long)l * omega_stride + e];
    dOmega[(long long)l * M * M + e] = 0.5 * gl * (kin - Oinv[(long long)l * oinv_stride + e]);
    s += gl * (om + Dm[(long long)i * L + l] * Dm[(long long)j * L + l]);
  }
  Sp[e] = gs * Kuu[e] - s;
}

}  // namespace gpsa

extern "C" {

int gpsa_mvn_kl_fwd(const double* Kinv, const double* logdetK, const double* Omega,
                    long long omega_stride, const double* logdetO, long long logdet_stride,
                    const double* Dm, const double* KD, int M, int L, double* kl, void* stream) {
  if (M < 1 || L < 1) return GPSA_EINVAL;
  gpsa::mvn_kl_fwd_kernel<<<L, 1024, 0, as_stream(stream)>>>(Kinv, logdetK, Omega, omega_stride, logdetO,
                                                             logdet_stride, Dm, KD, M, L, kl);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_mvn_kl_bwd(const double* Kuu, const double* Kinv, const double* Omega, long long omega_stride,
                    const double* Oinv, long long oinv_stride, const double* Dm, const double* KD,
                    const double* g, int M, int L, double* dOmega, double* dDm, double* Sp,
                    void* stream) {
  if (M < 1 || L < 1) return GPSA_EINVAL;
  const unsigned nb = (unsigned)cdiv((long long)M * M, 256);
  gpsa::mvn_kl_bwd_kernel<<<nb, 256, 0, as_stream(stream)>>>(Kuu, Kinv, Omega, omega_stride, Oinv,
                                                             oinv_stride, Dm, KD, g, M, L, dOmega, dDm, Sp);
  GPSA_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
