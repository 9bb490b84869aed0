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

// ---- grouped form: ALL KL terms of a step in one launch each way -----------------------------------
// The step's M x M matrices live in one batch (priors first, then every Omega): term t pairs the
// covariance mats[om_idx[t]] with the prior mats[pr_idx[t]] (pr_idx < 0: the term is absent, kl = 0);
// inverses and log-determinants come from the same batch.  D [T,M]: d_t = delta_t - mu_t.
// Round 6: a term's rows are dealt to gridDim.y workgroups (one workgroup per term took 21 us at M = 200 whatever the
// number of terms: a single CU pulling two 320 KB matrices).  Slice q takes the row groups q, q + gridDim.y, ... of 64;
// its partial sum goes to part[t][q], and the slice that arrives LAST (a counter per term, zeroed by the caller -
// the step's step_prep_kernel - before the launch and left at zero again) adds the partials in the fixed order
// q = 0, 1, ...: bitwise repeatable whichever slice that is.  gridDim.y == 1: no counter, no partials (part / cnt unused).
__global__ void __launch_bounds__(1024)
mvn_kl_grouped_fwd_kernel(const double* __restrict__ mats, const double* __restrict__ inv,
                          const double* __restrict__ logdet, const int* __restrict__ om_idx,
                          const int* __restrict__ pr_idx, const double* __restrict__ D, int M,
                          double* __restrict__ kl, double* __restrict__ KD, double* __restrict__ kl_copy,
                          double* part, int* cnt) {
  __shared__ double red[16];
  const int t = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = blockIdx.y, NS = gridDim.y;
  const int p = pr_idx[t], o = om_idx[t];
  if (p < 0) {  // uniform
    if (q != 0) return;
    if (threadIdx.x == 0) {
      kl[t] = 0.0;
      if (kl_copy != nullptr) kl_copy[t] = 0.0;
    }
    for (int m = threadIdx.x; m < M; m += 1024) KD[(long long)t * M + m] = 0.0;
    return;
  }
  const double* Ki = inv + (long long)p * M * M;
  const double* Om = mats + (long long)o * M * M;
  const double* d = D + (long long)t * M;
  double acc = 0.0;
  // one wave per row: trace term and (K^-1 d)_i in the same pass.  Four rows x four 64-element pieces of both
  // matrices are requested before anything is added (round 4: a row at a time was one memory latency per row,
  // thirteen in a chain per wave at M = 200 - 25 us for 54 terms); out-of-range pieces read a clamped address
  // and count as zero.
  for (int i0 = w + 64 * q; i0 < M; i0 += 64 * NS) {
    double tr[4] = {0.0, 0.0, 0.0, 0.0}, kd[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j0 = 0; j0 < M; j0 += 256) {
      double kv[4][4], ov[4][4], dv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = j0 + lane + 64 * q;
        dv[q] = j < M ? d[j < M ? j : M - 1] : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + 16 * r;
          const long long e = (long long)(i < M ? i : M - 1) * M + (j < M ? j : M - 1);
          kv[r][q] = Ki[e];
          ov[r][q] = Om[e];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = i0 + 16 * r < M && j0 + lane + 64 * q < M;
          const double k = ok ? kv[r][q] : 0.0;
          tr[r] += k * ov[r][q];
          kd[r] += k * dv[q];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 16 * r;
      if (i < M) {  // (uniform)
        const double s = wave_sum(kd[r]);
        if (lane == 0) {
          KD[(long long)t * M + i] = s;
          acc += s * d[i];
        }
        acc += tr[r];
      }
    }
  }
  double tot = block_sum(acc, red);
  if (threadIdx.x == 0) {
    if (NS > 1) {
      part[(long long)t * NS + q] = tot;
      __threadfence();
      if (atomicAdd(&cnt[t], 1) != NS - 1) return;
      __threadfence();
      tot = 0.0;
      for (int i = 0; i < NS; ++i) tot += __hip_atomic_load(&part[(long long)t * NS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cnt[t] = 0;
    }
    const double v = 0.5 * (logdet[p] - logdet[o] + tot - (double)M);
    kl[t] = v;
    if (kl_copy != nullptr) kl_copy[t] = v;  // (the step's cache for passes that reuse the M x M stage)
  }
}

// grid (elements / 256, P + 1): blockIdx.y = prior group pg (P: the pseudo-group of the absent terms).  A thread owns
// one element e of the M x M matrices and walks the terms of its group, eight at a time (their index loads are
// block-uniform, the eight terms' matrix loads are in flight together):
//   dOmega[t][e] (+)= 0.5 g_t (K_p^-1 - Omega_t^-1)[e]                       (zero for an absent term)
//   S[pg][e] = (sum g_t) K_p[e] - sum g_t (Omega_t[e] + d_t[i] d_t[j])         ( dK_p = 0.5 K_p^-1 S_p K_p^-1 )
// and block x == 0 writes dD[t] = g_t K_p^-1 d_t.  (History: one thread walking 50 terms serially was 80 us for
// 57 matrices of 200 x 200; one block per (term, 256 elements) with a parallel prior lookup 39-53 us - 8500 tiny
// blocks with two barriers each; this form: one pass over the terms per element, 471 blocks.)
__global__ void __launch_bounds__(256)
mvn_kl_grouped_bwd_kernel(const double* __restrict__ mats, const double* __restrict__ inv,
                          const int* __restrict__ om_idx,
                          const int* __restrict__ pr_list, const int* __restrict__ grp_off,
                          const int* __restrict__ order, const double* __restrict__ D,
                          const double* __restrict__ KD, const double* __restrict__ g, int M, int T, int P,
                          double* __restrict__ dOmega, double* __restrict__ dD, double* __restrict__ S,
                          int accumulate) {
  const long long mm = (long long)M * M, e = blockIdx.x * 256LL + threadIdx.x;
  const int pg = blockIdx.y;
  const int t0 = grp_off[pg], t1 = grp_off[pg + 1];
  if (pg == P) {  // pseudo-group of the absent terms: zero gradients
    if (blockIdx.x == 0)
      for (int q = t0; q < t1; ++q)
        for (int m = threadIdx.x; m < M; m += 256) dD[(long long)order[q] * M + m] = 0.0;
    if (!accumulate && e < mm)
      for (int q = t0; q < t1; ++q) dOmega[(long long)order[q] * mm + e] = 0.0;
    return;
  }
  const int p = pr_list[pg];
  if (blockIdx.x == 0)
    for (int q = t0; q < t1; ++q) {
      const int t = order[q];
      for (int m = threadIdx.x; m < M; m += 256) dD[(long long)t * M + m] = g[t] * KD[(long long)t * M + m];
    }
  if (e >= mm) return;
  const int i = (int)(e / M), j = (int)(e % M);
  const double kinv = inv[(long long)p * mm + e];
  double s0 = 0.0, s1 = 0.0, gs = 0.0;
  for (int q = t0; q < t1; q += 8) {
    double ga[8], om[8], oi[8], di[8], dj[8], old[8];
    int tt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool on = q + u < t1;
      const int ta = order[on ? q + u : t0];
      tt[u] = on ? ta : -1;
      ga[u] = on ? g[ta] : 0.0;
      const long long mo = (long long)om_idx[ta] * mm + e;
      om[u] = mats[mo];
      oi[u] = inv[mo];
      di[u] = D[(long long)ta * M + i];
      dj[u] = D[(long long)ta * M + j];
      old[u] = (accumulate && on) ? dOmega[(long long)ta * mm + e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (tt[u] >= 0) dOmega[(long long)tt[u] * mm + e] = old[u] + 0.5 * ga[u] * (kinv - oi[u]);
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      gs += ga[u] + ga[u + 1];
      s0 += ga[u] * (om[u] + di[u] * dj[u]);
      s1 += ga[u + 1] * (om[u + 1] + di[u + 1] * dj[u + 1]);
    }
  }
  S[(long long)pg * mm + e] = gs * mats[(long long)p * mm + e] - (s0 + s1);
}

// gpsa_mvn_kl_grouped_fwd with a second copy of the terms (the step engine's cache) written by the same launch
// part [T][slices] doubles and cnt [T] ints ZEROED by the caller (both may be null: one workgroup per term)
int mvn_kl_grouped_fwd_slices(int M) { return M > 64 ? (M + 63) / 64 < 8 ? (M + 63) / 64 : 8 : 1; }
int mvn_kl_grouped_fwd_copy(const double* mats, const double* inv, const double* logdet, const int* om_idx,
                            const int* pr_idx, const double* D, int M, int T, double* kl, double* KD,
                            double* kl_copy, double* part, int* cnt, hipStream_t st) {
  if (M < 1 || T < 1) return GPSA_EINVAL;
  const int ns = (part != nullptr && cnt != nullptr) ? mvn_kl_grouped_fwd_slices(M) : 1;
  mvn_kl_grouped_fwd_kernel<<<dim3((unsigned)T, (unsigned)ns), 1024, 0, st>>>(mats, inv, logdet, om_idx, pr_idx, D, M, kl,
                                                                             KD, kl_copy, part, cnt);
  GPSA_LAUNCH_CHECK();
  return 0;
}

}  // namespace gpsa

extern "C" {

int gpsa_mvn_kl_grouped_fwd(const double* mats, const double* inv, const double* logdet,
                            const int* om_idx, const int* pr_idx, const double* D, int M, int T,
                            double* kl, double* KD, void* stream) {
  if (M < 1 || T < 1) return GPSA_EINVAL;
  gpsa::mvn_kl_grouped_fwd_kernel<<<T, 1024, 0, as_stream(stream)>>>(mats, inv, logdet, om_idx, pr_idx, D,
                                                                     M, kl, KD, nullptr, nullptr, nullptr);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_mvn_kl_grouped_bwd_acc(const double* mats, const double* inv, const int* om_idx,
                                const int* pr_list, const int* grp_off, const int* order, const double* D,
                                const double* KD, const double* g, int M, int T, int P, double* dOmega,
                                double* dD, double* S, int accumulate, void* stream) {
  if (M < 1 || T < 1 || P < 1) return GPSA_EINVAL;
  dim3 grid((unsigned)cdiv((long long)M * M, 256), (unsigned)(P + 1));
  gpsa::mvn_kl_grouped_bwd_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      mats, inv, om_idx, pr_list, grp_off, order, D, KD, g, M, T, P, dOmega, dD, S, accumulate);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_mvn_kl_grouped_bwd(const double* mats, const double* inv, const int* om_idx,
                            const int* pr_list, const int* grp_off, const int* order, const double* D,
                            const double* KD, const double* g, int M, int T, int P, double* dOmega,
                            double* dD, double* S, void* stream) {
  return gpsa_mvn_kl_grouped_bwd_acc(mats, inv, om_idx, pr_list, grp_off, order, D, KD, g, M, T, P, dOmega,
                                     dD, S, 0, stream);
}

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
