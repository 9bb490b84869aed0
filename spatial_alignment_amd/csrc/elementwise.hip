// Reparameterised Gaussian sampling, its backward, and the Gaussian likelihood sum.
// HBM-bound elementwise / reduction kernels: coalesced on both the sample-major ([C,L]) API tensors
// and the output-major ([L,C]) internal ones via 32x32 LDS tile transposes; wave-shuffle + two-pass
// (deterministic) reductions.  Reference: gpsa/models/vgpsa.py:186-204, 334-351, 423-426, 532-538.
#include <stdlib.h>

#include "common.hpp"
#include "internal.hpp"

namespace gpsa {

constexpr double TWO_JITTER = 2e-5;  // diagonal_offset added twice (vgpsa.py:191/201 and :204)

// out[0] = (TO)( sum(part[0..n)) * cscale * (expo ? exp(*expo) : 1) * (mul ? *mul : 1) )
template <typename TE, typename TO>
__global__ void sum_scale_kernel(const double* __restrict__ part, long long n,
                                 const TE* __restrict__ expo, const double* __restrict__ mul,
                                 double cscale, TO* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    double r = s * cscale;
    if (expo) r *= exp((double)expo[0]);
    if (mul) r *= mul[0];
    out[0] = (TO)r;
  }
}

// ---- data GP ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
data_sample_fwd_kernel(const float* __restrict__ meanT, const float* __restrict__ v,
                       const double* __restrict__ q, const float* __restrict__ var_u,
                       const float* __restrict__ eps, long long C, int L, float* __restrict__ F,
                       float* __restrict__ Sigma) {
  __shared__ float tm[32][33], tv[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long c0 = (long long)blockIdx.x * 32;
  const int l0 = blockIdx.y * 32;
  const double var0 = exp((double)var_u[0]);
  {
    const long long c = c0 + tx;
    // sigma^2 - q formed in fp64 before rounding: it cancels to ~1e-3 sigma^2 for dense inducing sets
    const float resid = (c < C) ? (float)(var0 - q[c]) : 0.f;
#pragma unroll
    for (int ly = 0; ly < 32; ly += 8) {
      const int l = l0 + ty + ly;
      if (c < C && l < L) {
        const long long o = (long long)l * C + c;
        const float var = resid + v[o] + (float)TWO_JITTER;
        Sigma[o] = var;
        tv[ty + ly][tx] = var;
        tm[ty + ly][tx] = meanT[o];
      }
    }
  }
  __syncthreads();
  {
    const int l = l0 + tx;
#pragma unroll
    for (int cy = 0; cy < 32; cy += 8) {
      const long long c = c0 + ty + cy;
      if (c < C && l < L) {
        const long long o = c * L + l;
        F[o] = tm[tx][ty + cy] + sqrtf(tv[tx][ty + cy]) * eps[o];
      }
    }
  }
}

// one block per 32-column tile, loops over all L
__global__ void __launch_bounds__(256)
data_sample_bwd_kernel(const float* __restrict__ dF, const float* __restrict__ eps,
                       const float* __restrict__ Sigma, long long C, int L, float* __restrict__ g,
                       float* __restrict__ dmeanT, float* __restrict__ qbar,
                       double* __restrict__ part) {
  __shared__ float td[32][33], te[32][33];
  __shared__ float colacc[8][32];
  __shared__ double red[4];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long c0 = (long long)blockIdx.x * 32;
  float gsum = 0.f;
  for (int l0 = 0; l0 < L; l0 += 32) {
    __syncthreads();
    {
      const int l = l0 + tx;
#pragma unroll
      for (int cy = 0; cy < 32; cy += 8) {
        const long long c = c0 + ty + cy;
        float a = 0.f, b = 0.f;
        if (c < C && l < L) {
          a = dF[c * L + l];
          b = eps[c * L + l];
        }
        td[ty + cy][tx] = a;
        te[ty + cy][tx] = b;
      }
    }
    __syncthreads();
    {
      const long long c = c0 + tx;
#pragma unroll
      for (int ly = 0; ly < 32; ly += 8) {
        const int l = l0 + ty + ly;
        if (c < C && l < L) {
          const long long o = (long long)l * C + c;
          const float d = td[tx][ty + ly];
          const float gv = d * te[tx][ty + ly] * 0.5f / sqrtf(Sigma[o]);
          g[o] = gv;
          dmeanT[o] = d;
          gsum += gv;
        }
      }
    }
  }
  colacc[ty][tx] = gsum;
  __syncthreads();
  if (ty == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += colacc[i][tx];
    if (c0 + tx < C) qbar[c0 + tx] = -s;
  }
  double tot = block_sum((double)gsum, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// ---- warp GP (fp64 inside) ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
warp_sample_fwd_kernel(const double* __restrict__ meanT, const double* __restrict__ v,
                       const double* __restrict__ q, const float* __restrict__ var_u,
                       const float* __restrict__ X, const float* __restrict__ slopes,
                       const float* __restrict__ intercept, const float* __restrict__ eps,
                       long long n, int D, int S, float* __restrict__ Gmean, float* __restrict__ Gs,
                       double* __restrict__ Gs64, int* __restrict__ bad) {
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  int flag = 0;
  if (c < n) {
    const double var0 = exp((double)var_u[0]), qc = q[c];
    double x[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) x[d] = d < D ? (double)X[c * D + d] : 0.0;
    for (int j = 0; j < D; ++j) {
      const long long o = (long long)j * n + c;
      const double var = var0 - qc + v[o] + TWO_JITTER;
      double mu = (double)intercept[j] + meanT[o];  // linear mean function of the warp GP at x_c
#pragma unroll
      for (int d = 0; d < MAXD; ++d)
        if (d < D) mu += x[d] * (double)slopes[d * D + j];
      if (!(var > 0.0)) flag = 1;
      Gmean[c * D + j] = (float)mu;
      for (int s = 0; s < S; ++s) {
        const long long e = ((long long)s * n + c) * D + j;
        const double gv = mu + var * (double)eps[e];  // variance used as the std
        Gs[e] = (float)gv;
        if (Gs64 != nullptr) Gs64[e] = gv;  // unrounded copy: what the data GP's covariance is built from
      }
    }
  }
  flag = __syncthreads_or(flag);
  if (threadIdx.x == 0) bad[blockIdx.x] = flag;
}

// partials per block: [0] = sum g (for d var_u), [1 + d*D + j] = sum_c x[c,d] dmu[c,j], [1 + D*D + j] = sum_c dmu[c,j]
constexpr int WS_NPART = 1 + MAXD * MAXD + MAXD;

__global__ void __launch_bounds__(256)
warp_sample_bwd_kernel(const float* __restrict__ dGmean, const float* __restrict__ dGs,
                       const double* __restrict__ dGs64, const float* __restrict__ eps,
                       const float* __restrict__ X, long long n,
                       int D, int S, double* __restrict__ dmeanT, double* __restrict__ g,
                       double* __restrict__ qbar, double* __restrict__ part) {
  __shared__ double red[4];
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  double gtot = 0.0, dmu[MAXD], x[MAXD];
#pragma unroll
  for (int d = 0; d < MAXD; ++d) {
    dmu[d] = 0.0;
    x[d] = (c < n && d < D) ? (double)X[c * D + d] : 0.0;
  }
  if (c < n) {
#pragma unroll
    for (int j = 0; j < MAXD; ++j)
      if (j < D) {
        double dm = dGmean ? (double)dGmean[c * D + j] : 0.0, gj = 0.0;
        for (int s = 0; s < S; ++s) {
          const long long e = ((long long)s * n + c) * D + j;
          // the data GP's gradient arrives along the fp64 copy of the draws, a caller's own along the fp32 one
          const double d = (dGs ? (double)dGs[e] : 0.0) + (dGs64 ? dGs64[e] : 0.0);
          dm += d;
          gj += d * (double)eps[e];
        }
        dmeanT[(long long)j * n + c] = dm;
        g[(long long)j * n + c] = gj;
        gtot += gj;
        dmu[j] = dm;
      }
    qbar[c] = -gtot;
  }
  double* pb = part + (long long)blockIdx.x * WS_NPART;
  double t = block_sum(gtot, red);
  if (threadIdx.x == 0) pb[0] = t;
#pragma unroll
  for (int d = 0; d < MAXD; ++d)
#pragma unroll
    for (int j = 0; j < MAXD; ++j)
      if (d < D && j < D) {
        t = block_sum(x[d] * dmu[j], red);
        if (threadIdx.x == 0) pb[1 + d * D + j] = t;
      }
#pragma unroll
  for (int j = 0; j < MAXD; ++j)
    if (j < D) {
      t = block_sum(dmu[j], red);
      if (threadIdx.x == 0) pb[1 + D * D + j] = t;
    }
}

// sums the block partials in block order: dvar_u = exp(var_u) * sum g ; dslopes ; dintercept
__global__ void __launch_bounds__(64)
warp_sample_bwd_finish_kernel(const double* __restrict__ part, long long nb, int D,
                              const float* __restrict__ var_u, float* __restrict__ dvar_u,
                              float* __restrict__ dslopes, float* __restrict__ dintercept) {
  const int k = threadIdx.x, nk = 1 + D * D + D;
  if (k >= nk) return;
  double s = 0.0;
  for (long long b = 0; b < nb; ++b) s += part[b * WS_NPART + k];
  if (k == 0) dvar_u[0] = (float)(s * exp((double)var_u[0]));
  else if (k < 1 + D * D) dslopes[k - 1] = (float)s;
  else dintercept[k - 1 - D * D] = (float)s;
}

// ---- linear mean function at the inducing points and the variational residual -------------------
// mu_z = scale * (Z A + b) (fp32, kept as an attribute like the reference's self.mu_z_G);
// resid = delta - mu_z formed in fp64 from the fp32 parameters.  One block: M x D is tiny.
__global__ void __launch_bounds__(256)
mean_resid_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ slopes,
                      const float* __restrict__ intercept, const float* __restrict__ delta, int M,
                      int D, double scale, float* __restrict__ mu_z, double* __restrict__ resid) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < M * D; i += gridDim.x * 256) {
    const int m = i / D, j = i - m * D;
    double mu = (double)intercept[j];
    for (int d = 0; d < D; ++d) mu += (double)Z[m * D + d] * (double)slopes[d * D + j];
    mu *= scale;
    mu_z[i] = (float)mu;
    resid[i] = (double)delta[i] - mu;
  }
}

// r = dLoss/d resid [M,D] (fp64): ddelta = r ; dZ = -scale r A^T ; dA = -scale Z^T r ; db = -scale sum_m r
__global__ void __launch_bounds__(256)
mean_resid_bwd_kernel(const double* __restrict__ r, const float* __restrict__ Z,
                      const float* __restrict__ slopes, int M, int D, double scale,
                      float* __restrict__ ddelta, float* __restrict__ dZ,
                      float* __restrict__ dslopes, float* __restrict__ dintercept) {
  __shared__ double red[4];
  double accA[MAXD][MAXD], accb[MAXD];
#pragma unroll
  for (int d = 0; d < MAXD; ++d) {
    accb[d] = 0.0;
#pragma unroll
    for (int j = 0; j < MAXD; ++j) accA[d][j] = 0.0;
  }
  for (int m = threadIdx.x; m < M; m += 256) {
    double rr[MAXD], zz[MAXD];
#pragma unroll
    for (int j = 0; j < MAXD; ++j) {
      rr[j] = j < D ? r[m * D + j] : 0.0;
      zz[j] = j < D ? (double)Z[m * D + j] : 0.0;
    }
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
      if (d < D) {
        ddelta[m * D + d] = (float)rr[d];
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < MAXD; ++j)
          if (j < D) {
            s += rr[j] * (double)slopes[d * D + j];
            accA[d][j] += zz[d] * rr[j];
          }
        dZ[m * D + d] = (float)(-scale * s);
        accb[d] += rr[d];
      }
  }
#pragma unroll
  for (int d = 0; d < MAXD; ++d)
    if (d < D) {
#pragma unroll
      for (int j = 0; j < MAXD; ++j)
        if (j < D) {
          const double t = block_sum(accA[d][j], red);
          if (threadIdx.x == 0) dslopes[d * D + j] = (float)(-scale * t);
        }
      const double t = block_sum(accb[d], red);
      if (threadIdx.x == 0) dintercept[d] = (float)(-scale * t);
    }
}

// ---- Gaussian likelihood --------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
loglik_fwd_kernel(const float* __restrict__ F, const float* __restrict__ Y,
                  const float* __restrict__ noise_u, long long tot, long long NP,
                  double* __restrict__ part) {
  __shared__ double red[4];
  const double s = exp((double)noise_u[0]) + 1e-5;  // "variance" used as std
  const float inv = (float)(1.0 / s);
  const double cst = -log(s) - 0.9189385332046727;  // -log(s) - 0.5*log(2*pi)
  double acc = 0.0;
  for (long long i0 = blockIdx.x * 256LL * 4; i0 < tot; i0 += (long long)gridDim.x * 256 * 4) {
    float a = 0.f;
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * 256 + threadIdx.x;
      if (i < tot) {
        const float z = (Y[i % NP] - F[i]) * inv;
        a += z * z;
        ++cnt;
      }
    }
    acc += -0.5 * (double)a + cst * cnt;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256)
loglik_bwd_kernel(const float* __restrict__ F, const float* __restrict__ Y,
                  const float* __restrict__ noise_u, const double* __restrict__ gout, int S,
                  long long tot, long long NP, float* __restrict__ dF, double* __restrict__ part,
                  const float* __restrict__ gloss = nullptr) {
  __shared__ double red[4];
  const double s = exp((double)noise_u[0]) + 1e-5;
  const float inv = (float)(1.0 / s);
  // upstream gradient of this log-likelihood: gout[0], or -gloss[0] when the ELBO glue is fused in
  const double up = gloss != nullptr ? -(double)gloss[0] : gout[0];
  const float coef = (float)(up / (s * s * (double)S));
  double acc = 0.0;  // sum z^2 - 1 ; dLL/ds = acc / s / S
  for (long long i0 = blockIdx.x * 256LL * 4; i0 < tot; i0 += (long long)gridDim.x * 256 * 4) {
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * 256 + threadIdx.x;
      if (i < tot) {
        const float r = Y[i % NP] - F[i];
        dF[i] = coef * r;
        const float z = r * inv;
        a += z * z - 1.f;
      }
    }
    acc += (double)a;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// dnoise_u = gout * (sum(z^2-1)/s/S) * exp(noise_u)
__global__ void loglik_bwd_finish_kernel(const double* __restrict__ part, int n,
                                         const float* __restrict__ noise_u,
                                         const double* __restrict__ gout, int S,
                                         float* __restrict__ dnoise_u, const float* __restrict__ gloss = nullptr,
                                         double* __restrict__ dkl = nullptr, int n_kl = 0,
                                         double kl_scale = 0.0, float* __restrict__ zero_base = nullptr,
                                         int zero_n = 0, double sub = 0.0) {  // sub: part sums z^2, not z^2 - 1
  __shared__ double red[4];
  if (zero_base != nullptr) {  // the whole noise-gradient vector starts at zero (entries no term names stay so)
    for (int t = threadIdx.x; t < zero_n; t += blockDim.x) zero_base[t] = 0.f;
    __syncthreads();
  }
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const double e = exp((double)noise_u[0]), sc = e + 1e-5;
    const double up = gloss != nullptr ? -(double)gloss[0] : gout[0];
    dnoise_u[0] = (float)(up * (s - sub) / sc / (double)S * e);
  }
  if (dkl != nullptr)  // fused ELBO glue: dkl[t] = kl_scale * gloss
    for (int t = threadIdx.x; t < n_kl; t += blockDim.x) dkl[t] = kl_scale * (double)gloss[0];
}

// fused forward finish: ll[i] = sum(part_i) / S_i for every likelihood term, then
// loss = kl_scale * sum(kl) - sum_i ll[i]   (vgpsa.py:540); one block
struct ElboFinishArgs {
  const double* part[GPSA_MAX_MODS];
  int nb[GPSA_MAX_MODS], S[GPSA_MAX_MODS];
  // fused terms (gpsa_quadform_elbo_f32): part sums z^2 only; ll = (-0.5 sum z^2 + (-log s - log(2 pi)/2) tot) / S
  const float* z2_noise[GPSA_MAX_MODS];  // non-null: this term's log "variance"
  double tot[GPSA_MAX_MODS];
  int n_ll, n_kl;
  const double* kl;
  double kl_scale;
  double* ll;
  float* loss;
};
__global__ void __launch_bounds__(256) elbo_loss_finish_kernel(ElboFinishArgs a) {
  __shared__ double red[4];
  double lsum = 0.0;
  for (int i = 0; i < a.n_ll; ++i) {
    double s = 0.0;
    for (int k = threadIdx.x; k < a.nb[i]; k += 256) s += a.part[i][k];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
      if (a.z2_noise[i] != nullptr) {
        const double sd = exp((double)a.z2_noise[i][0]) + 1e-5;
        s = -0.5 * s + (-log(sd) - 0.9189385332046727) * a.tot[i];
      }
      const double v = s / (double)a.S[i];
      a.ll[i] = v;
      lsum += v;
    }
  }
  double k = 0.0;
  for (int t = threadIdx.x; t < a.n_kl; t += 256) k += a.kl[t];
  k = block_sum(k, red);
  if (threadIdx.x == 0) a.loss[0] = (float)(a.kl_scale * k - lsum);
}

// ---- fused ELBO (gpsa_quadform_elbo_f32) -> the backward's conventions ------------------------------------------
// per column: qbar[c] = -gl sum_l g[l,c]; part[block] = gl sum g (for d var);  with an upstream gradient gl != 1
// g and dmeanT are scaled in place (they were formed at gl = 1)
// tick != nullptr (a zeroed word; the step engine): the LAST block to arrive closes dvar = exp(var_u) sum(part) itself, in
// the order sum_scale_kernel adds them - the finishing launch is not needed (round 6)
template <typename TO>
__global__ void __launch_bounds__(256)
elbo_post_kernel(float* __restrict__ g, float* __restrict__ dmeanT, long long C, int L, const float* __restrict__ gloss,
                 float* __restrict__ qbar, double* __restrict__ part, float* __restrict__ abar, int M,
                 int* __restrict__ tick, const float* __restrict__ var_u, TO* __restrict__ dvar) {
  __shared__ double red[4];
  __shared__ int last_s;
  const float gl = gloss[0];
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  float s = 0.f;
  if (c < C) {
    if (gl == 1.f) {  // (uniform) the usual case: a pure column sum, ten loads in flight per thread
      int l = 0;
      for (; l + 10 <= L; l += 10) {
        float v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = g[(long long)(l + u) * C + c];
#pragma unroll
        for (int u = 0; u < 10; ++u) s += v[u];
      }
      for (; l < L; ++l) s += g[(long long)l * C + c];
    } else {
      for (int l = 0; l < L; ++l) {
        const long long o = (long long)l * C + c;
        const float v = g[o] * gl;
        s += v;
        g[o] = v;
        dmeanT[o] *= gl;
      }
      for (int m = 0; m < M; ++m) abar[(long long)m * C + c] *= gl;  // (was a launch of its own that exits when gl == 1)
    }
    qbar[c] = -s;
  }
  const double t = block_sum((double)s, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = t;
    last_s = 0;
    if (tick != nullptr) {
      __threadfence();
      last_s = atomicAdd(tick, 1) == (int)gridDim.x - 1;
    }
  }
  __syncthreads();
  if (last_s) {  // (block-uniform)
    __threadfence();
    double a = 0.0;
    for (long long i = threadIdx.x; i < (long long)gridDim.x; i += 256)
      a += __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = block_sum(a, red);
    if (threadIdx.x == 0) {
      dvar[0] = (TO)(a * exp((double)var_u[0]));
      *tick = 0;
    }
  }
}
__global__ void __launch_bounds__(256) scale_unless_one_kernel(float* __restrict__ x, long long n,
                                                               const float* __restrict__ gloss) {
  const float gl = gloss[0];
  if (gl == 1.f) return;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] *= gl;
}

static inline int loglik_blocks(long long tot) {
  long long b = cdiv(tot, 1024);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- ELBO scalar glue ------------------------------------------------------------------------------
// loss = -(sum_i ll[i]) + kl_scale * sum_t kl[t]   (vgpsa.py:540), fp64 sums in a fixed order, fp32 result;
// one launch instead of the sum / neg / mul / add / cast chain, and one for its adjoint.
__global__ void __launch_bounds__(256)
elbo_fwd_kernel(const double* __restrict__ ll, int n_ll, const double* __restrict__ kl, int n_kl,
                double kl_scale, float* __restrict__ loss) {
  __shared__ double red[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < n_kl; i += 256) a += kl[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n_ll; ++i) s += ll[i];
    loss[0] = (float)(kl_scale * a - s);
  }
}

__global__ void __launch_bounds__(256)
elbo_bwd_kernel(const float* __restrict__ gloss, int n_ll, int n_kl, double kl_scale,
                double* __restrict__ dll, double* __restrict__ dkl) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const double g = (double)gloss[0];
  if (i < n_ll) dll[i] = -g;
  if (i < n_kl) dkl[i] = kl_scale * g;
}

}  // namespace gpsa

/* ---- LMC likelihood without F_obs (vgpsa.py:428-432, 532-538; round 4) ---------------------------------------------
 * An LMC modality's observed draws are  F_obs[s,n,:] = F_latent[s,n,:] W  ([S,N,L] x [L,P]); the reference
 * materialises them ([S,N,P]: 400 MB at BASELINE config 3) and the likelihood, its gradient dF_obs and the two LMC
 * gradient products each stream that size again.  Here ONE pass per tile of 16 columns forms, for every (column c, p):
 *     fobs = F[c,:] . W[:,p];   r = Y[c mod N, p] - fobs;   z^2 += (r / s)^2;   dfo = -r / (s^2 S)
 *     dW[l,p] += F[c,l] dfo          (a thread owns its p: register accumulators over all the block's columns)
 *     dF[c,l]  = sum_p dfo W[l,p]    (the tile's dfo through LDS, one thread per (c, l))
 * - nothing of size [S,N,P] exists.  Everything at upstream gradient 1 (the caller scales; linear).
 * zpart: nparts doubles (the block partials of sum z^2, tail zeroed) for gpsa_elbo_loss_fused_fwd / _bwd. */
namespace gpsa {
constexpr int LMC_TC = 16, LMC_PC = 512, LMC_PS = LMC_PC + 4;  // columns per tile, p per chunk, LDS row stride
// (rows stay 16-byte aligned for the ds_read_b128 of the dF pass; 516 = 4 mod 32: lanes on rows l, l + 8 share banks)

template <int LB>
__global__ void __launch_bounds__(256)
lmc_loglik_kernel(const float* __restrict__ F, const float* __restrict__ W, const float* __restrict__ Y,
                  const float* __restrict__ noise_u, int S, long long N, int L, int P, long long C,
                  double* __restrict__ zpart, int nparts, float* __restrict__ dF, float* __restrict__ dWpart) {
  extern __shared__ __attribute__((aligned(16))) float lmc_smem[];
  float* sF = lmc_smem;                    // [16][LB]
  float* sW = sF + LMC_TC * LB;            // [LB][513]
  float* sD = sW + LB * LMC_PS;            // [16][513]
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const double sN = exp((double)noise_u[0]) + 1e-5;  // "variance" used as std
  const float inv = (float)(1.0 / sN);
  const float coef = (float)(-1.0 / (sN * sN * (double)S));
  const long long ntiles = (C + LMC_TC - 1) / LMC_TC;
  double z2 = 0.0;
  for (int p0 = 0; p0 < P; p0 += LMC_PC) {
    float w[2][LB], dw[2][LB];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int p = p0 + tid + 256 * k;
#pragma unroll
      for (int l = 0; l < LB; ++l) {
        w[k][l] = (p < P && l < L) ? W[(long long)l * P + p] : 0.f;
        dw[k][l] = 0.f;
        sW[l * LMC_PS + tid + 256 * k] = w[k][l];
      }
    }
    __syncthreads();
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const long long c0 = t * LMC_TC;
      for (int e = tid; e < LMC_TC * LB; e += 256) {
        const int cc = e / LB, l = e - cc * LB;
        const long long c = c0 + cc;
        sF[e] = (c < C && l < L) ? F[c * L + l] : 0.f;
      }
      __syncthreads();
      float z2l = 0.f;
#pragma unroll 4
      for (int cc = 0; cc < LMC_TC; ++cc) {
        const long long c = c0 + cc;
        const bool okc = c < C;
        const float* yrow = Y + (okc ? (c % N) : 0) * P;
        float f[LB];
#pragma unroll
        for (int l = 0; l < LB; ++l) f[l] = sF[cc * LB + l];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int p = p0 + tid + 256 * k;
          float fobs = 0.f;
#pragma unroll
          for (int l = 0; l < LB; ++l) fobs = fmaf(f[l], w[k][l], fobs);
          const bool ok = okc && p < P;
          const float r = ok ? yrow[p] - fobs : 0.f;
          const float z = r * inv;
          z2l = fmaf(z, z, z2l);
          const float dfo = coef * r;
          sD[cc * LMC_PS + tid + 256 * k] = dfo;
#pragma unroll
          for (int l = 0; l < LB; ++l) dw[k][l] = fmaf(f[l], dfo, dw[k][l]);
        }
      }
      z2 += (double)z2l;
      __syncthreads();
      for (int e = tid; e < LMC_TC * LB; e += 256) {
        const int cc = e / LB, l = e - cc * LB;
        const float* dr = sD + cc * LMC_PS;
        const float* wr = sW + l * LMC_PS;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
        for (int p = 0; p < LMC_PC; p += 4) {
          const float4 d4 = *reinterpret_cast<const float4*>(dr + p), w4 = *reinterpret_cast<const float4*>(wr + p);
          s0 = fmaf(d4.x, w4.x, s0);
          s1 = fmaf(d4.y, w4.y, s1);
          s2 = fmaf(d4.z, w4.z, s2);
          s3 = fmaf(d4.w, w4.w, s3);
        }
        const long long c = c0 + cc;
        if (c < C && l < L) {
          const long long o = c * L + l;
          const float sum = (s0 + s1) + (s2 + s3);
          dF[o] = p0 == 0 ? sum : dF[o] + sum;  // (this block owns the column in every chunk of p)
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int p = p0 + tid + 256 * k;
      if (p < P)
#pragma unroll
        for (int l = 0; l < LB; ++l)
          if (l < L) dWpart[((long long)blockIdx.x * L + l) * P + p] = dw[k][l];
    }
    __syncthreads();
  }
  z2 = block_sum(z2, red);
  if (tid == 0) zpart[blockIdx.x] = z2;
  if (blockIdx.x == 0)
    for (int i = (int)gridDim.x + tid; i < nparts; i += 256) zpart[i] = 0.0;
}

static inline int lmc_blocks(long long C, int nparts) {
  long long g = (C + LMC_TC - 1) / LMC_TC;
  const long long cap = 2LL * num_cus();
  if (g > cap) g = cap;
  if (g > nparts) g = nparts;
  return (int)(g < 1 ? 1 : g);
}
}  // namespace gpsa

extern "C" {

int gpsa_data_sample_fwd(const float* meanT, const float* v, const double* q, const float* var_u,
                         const float* eps, long long C, int L, float* F, float* Sigma, void* stream) {
  if (C < 1 || L < 1) return GPSA_EINVAL;
  dim3 grid((unsigned)cdiv(C, 32), (unsigned)cdiv(L, 32));
  gpsa::data_sample_fwd_kernel<<<grid, 256, 0, as_stream(stream)>>>(meanT, v, q, var_u, eps, C, L, F, Sigma);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_data_sample_bwd(const float* dF, const float* eps, const float* Sigma, const float* var_u,
                         long long C, int L, float* g, float* dmeanT, float* qbar, int dvar_dtype, void* dvar_u,
                         void* workspace, long long workspace_bytes, void* stream) {
  if (C < 1 || L < 1 || (dvar_dtype != GPSA_F32 && dvar_dtype != GPSA_F64)) return GPSA_EINVAL;
  const long long nb = cdiv(C, 32);
  if (workspace_bytes < nb * 8) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* part = (double*)workspace;
  gpsa::data_sample_bwd_kernel<<<(unsigned)nb, 256, 0, st>>>(dF, eps, Sigma, C, L, g, dmeanT, qbar, part);
  if (dvar_dtype == GPSA_F64)
    gpsa::sum_scale_kernel<float, double><<<1, 256, 0, st>>>(part, nb, var_u, nullptr, 1.0, (double*)dvar_u);
  else
    gpsa::sum_scale_kernel<float, float><<<1, 256, 0, st>>>(part, nb, var_u, nullptr, 1.0, (float*)dvar_u);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_warp_sample_fwd(const double* meanT, const double* v, const double* q, const float* var_u,
                         const float* X, const float* slopes, const float* intercept,
                         const float* eps, long long n, int D, int S, float* Gmean, float* Gs,
                         double* Gs64, int* bad, void* stream) {
  if (n < 1 || D < 1 || D > gpsa::MAXD || S < 0) return GPSA_EINVAL;
  gpsa::warp_sample_fwd_kernel<<<(unsigned)cdiv(n, 256), 256, 0, as_stream(stream)>>>(
      meanT, v, q, var_u, X, slopes, intercept, eps, n, D, S, Gmean, Gs, Gs64, bad);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_warp_sample_bwd(const float* dGmean, const float* dGs, const double* dGs64, const float* eps,
                         const float* var_u, const float* X, long long n, int D, int S, double* dmeanT, double* g,
                         double* qbar, float* dvar_u, float* dslopes, float* dintercept,
                         void* workspace, long long workspace_bytes, void* stream) {
  if (n < 1 || D < 1 || D > gpsa::MAXD || S < 0) return GPSA_EINVAL;
  const long long nb = cdiv(n, 256);
  if (workspace_bytes < nb * gpsa::WS_NPART * 8) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* part = (double*)workspace;
  gpsa::warp_sample_bwd_kernel<<<(unsigned)nb, 256, 0, st>>>(dGmean, dGs, dGs64, eps, X, n, D, S, dmeanT,
                                                             g, qbar, part);
  gpsa::warp_sample_bwd_finish_kernel<<<1, 64, 0, st>>>(part, nb, D, var_u, dvar_u, dslopes, dintercept);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_mean_resid_fwd(const float* Z, const float* slopes, const float* intercept,
                        const float* delta, int M, int D, double scale, float* mu_z, double* resid,
                        void* stream) {
  if (M < 1 || D < 1 || D > gpsa::MAXD) return GPSA_EINVAL;
  gpsa::mean_resid_fwd_kernel<<<(unsigned)cdiv((long long)M * D, 256), 256, 0, as_stream(stream)>>>(
      Z, slopes, intercept, delta, M, D, scale, mu_z, resid);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_mean_resid_bwd(const double* dresid, const float* Z, const float* slopes, int M, int D,
                        double scale, float* ddelta, float* dZ, float* dslopes, float* dintercept,
                        void* stream) {
  if (M < 1 || D < 1 || D > gpsa::MAXD) return GPSA_EINVAL;
  gpsa::mean_resid_bwd_kernel<<<1, 256, 0, as_stream(stream)>>>(dresid, Z, slopes, M, D, scale, ddelta,
                                                                dZ, dslopes, dintercept);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_loglik_fwd(const float* F, const float* Y, const float* noise_u, int S, long long N, int P,
                    double* out, void* workspace, long long workspace_bytes, void* stream) {
  if (S < 1 || N < 1 || P < 1) return GPSA_EINVAL;
  const long long NP = N * P, tot = NP * S;
  const int nb = gpsa::loglik_blocks(tot);
  if (workspace_bytes < (long long)nb * 8) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* part = (double*)workspace;
  gpsa::loglik_fwd_kernel<<<nb, 256, 0, st>>>(F, Y, noise_u, tot, NP, part);
  gpsa::sum_scale_kernel<float, double><<<1, 256, 0, st>>>(part, nb, nullptr, nullptr, 1.0 / S, out);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_loglik_bwd(const float* F, const float* Y, const float* noise_u, const double* gout, int S,
                    long long N, int P, float* dF, float* dnoise_u, void* workspace,
                    long long workspace_bytes, void* stream) {
  if (S < 1 || N < 1 || P < 1) return GPSA_EINVAL;
  const long long NP = N * P, tot = NP * S;
  const int nb = gpsa::loglik_blocks(tot);
  if (workspace_bytes < (long long)nb * 8) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* part = (double*)workspace;
  gpsa::loglik_bwd_kernel<<<nb, 256, 0, st>>>(F, Y, noise_u, gout, S, tot, NP, dF, part);
  gpsa::loglik_bwd_finish_kernel<<<1, 256, 0, st>>>(part, nb, noise_u, gout, S, dnoise_u);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_elbo_fwd(const double* ll, int n_ll, const double* kl, int n_kl, double kl_scale, float* loss,
                  void* stream) {
  if (n_ll < 1 || n_kl < 0) return GPSA_EINVAL;
  gpsa::elbo_fwd_kernel<<<1, 256, 0, as_stream(stream)>>>(ll, n_ll, kl, n_kl, kl_scale, loss);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_elbo_bwd(const float* gloss, int n_ll, int n_kl, double kl_scale, double* dll, double* dkl,
                  void* stream) {
  if (n_ll < 1 || n_kl < 0) return GPSA_EINVAL;
  const int n = n_ll > n_kl ? n_ll : n_kl;
  gpsa::elbo_bwd_kernel<<<(n + 255) / 256, 256, 0, as_stream(stream)>>>(gloss, n_ll, n_kl, kl_scale, dll, dkl);
  GPSA_LAUNCH_CHECK();
  return 0;
}

/* ---- likelihood + ELBO fused: one host call each way, one finishing launch for all of it ------------------ */
int gpsa_elbo_loss_fwd(int n_ll, const float* const* F, const float* const* Y, const float* const* noise_u,
                       const int* S, const long long* N, const int* P, const double* kl, int n_kl, double kl_scale,
                       float* loss, double* ll_out, void* workspace, long long workspace_bytes, void* stream) {
  if (n_ll < 1 || n_ll > GPSA_MAX_MODS || !F || !Y || !noise_u || !S || !N || !P || !loss || !ll_out) return GPSA_EINVAL;
  if (workspace_bytes < 8LL * 4100 * n_ll) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  gpsa::ElboFinishArgs a;
  a.n_ll = n_ll;
  a.n_kl = kl ? n_kl : 0;
  a.kl = kl;
  a.kl_scale = kl_scale;
  a.ll = ll_out;
  a.loss = loss;
  for (int i = 0; i < n_ll; ++i) {
    if (S[i] < 1 || N[i] < 1 || P[i] < 1) return GPSA_EINVAL;
    const long long NP = N[i] * P[i], tot = NP * S[i];
    const int nb = gpsa::loglik_blocks(tot);
    double* part = reinterpret_cast<double*>(workspace) + 4100LL * i;
    gpsa::loglik_fwd_kernel<<<nb, 256, 0, st>>>(F[i], Y[i], noise_u[i], tot, NP, part);
    a.part[i] = part;
    a.nb[i] = nb;
    a.S[i] = S[i];
    a.z2_noise[i] = nullptr;
    a.tot[i] = (double)tot;
  }
  gpsa::elbo_loss_finish_kernel<<<1, 256, 0, st>>>(a);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_elbo_loss_bwd(int n_ll, const float* const* F, const float* const* Y, const float* const* noise_u,
                       const int* S, const long long* N, const int* P, const float* gloss, int n_kl, double kl_scale,
                       float* const* dF, float* const* dnoise, float* dnoise_all, int n_noise, double* dkl,
                       void* workspace, long long workspace_bytes, void* stream) {
  if (n_ll < 1 || n_ll > GPSA_MAX_MODS || !F || !Y || !noise_u || !S || !N || !P || !gloss || !dF || !dnoise)
    return GPSA_EINVAL;
  if (workspace_bytes < 8LL * 4100 * n_ll) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  for (int i = 0; i < n_ll; ++i) {
    if (S[i] < 1 || N[i] < 1 || P[i] < 1) return GPSA_EINVAL;
    const long long NP = N[i] * P[i], tot = NP * S[i];
    const int nb = gpsa::loglik_blocks(tot);
    double* part = reinterpret_cast<double*>(workspace) + 4100LL * i;
    gpsa::loglik_bwd_kernel<<<nb, 256, 0, st>>>(F[i], Y[i], noise_u[i], nullptr, S[i], tot, NP, dF[i], part, gloss);
    // the first term's finishing launch also writes dkl
    gpsa::loglik_bwd_finish_kernel<<<1, 256, 0, st>>>(part, nb, noise_u[i], nullptr, S[i], dnoise[i], gloss,
                                                      i == 0 ? dkl : nullptr, n_kl, kl_scale,
                                                      i == 0 ? dnoise_all : nullptr, n_noise);
  }
  GPSA_LAUNCH_CHECK();
  return 0;
}

/* the fused forward's outputs (formed at upstream gradient 1) as the backward wants them: g, dmeanT, abar scaled by the
 * loss's upstream gradient (nothing is touched when that is 1), qbar = -sum_l g, dvar_u = exp(var_u) sum g.
 * g_ext: [L+1][C], row L receives qbar.  abar: [M][C].  workspace: 8 * ceil(C/256) bytes */
int gpsa_elbo_fused_post(float* g_ext, float* dmeanT, float* abar, int M, long long C, int L, const float* gloss,
                         const float* var_u, int dvar_dtype, void* dvar_u, void* workspace, long long workspace_bytes,
                         void* stream) {
  if (M < 1 || C < 1 || L < 1 || !g_ext || !dmeanT || !abar || !gloss || !var_u || !dvar_u) return GPSA_EINVAL;
  if (dvar_dtype != GPSA_F32 && dvar_dtype != GPSA_F64) return GPSA_EINVAL;
  const long long nb = cdiv(C, 256);
  if (workspace_bytes < nb * 8) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* part = (double*)workspace;
  gpsa::elbo_post_kernel<double><<<(unsigned)nb, 256, 0, st>>>(g_ext, dmeanT, C, L, gloss, g_ext + (long long)L * C, part,
                                                               abar, M, nullptr, nullptr, nullptr);
  if (dvar_dtype == GPSA_F64)
    gpsa::sum_scale_kernel<float, double><<<1, 256, 0, st>>>(part, nb, var_u, nullptr, 1.0, (double*)dvar_u);
  else
    gpsa::sum_scale_kernel<float, float><<<1, 256, 0, st>>>(part, nb, var_u, nullptr, 1.0, (float*)dvar_u);
  GPSA_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

namespace gpsa {
// Zero fill as a KERNEL of this library (round 6).  hipMemsetAsync inside a stream capture becomes a memset node, and
// in a replayed graph with parallel branches that node was seen to land AFTER the kernel node behind it on the same
// stream (the step's backward: the word the next kernel read still held the forward's scratch; found when that kernel
// became the region's first reader - tests/test_hip_parity.py::test_eager_forward_between_graph_replays...).  A kernel
// node is ordered like every other launch.  GPSA_ZERO_KERNEL=0: the runtime's memset.
__global__ void __launch_bounds__(256) zero_fill_kernel(uint4* __restrict__ p, long long n16, int tail) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i < n16) p[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) reinterpret_cast<unsigned char*>(p + n16)[threadIdx.x] = 0;
}
int zero_fill_async(void* ptr, size_t bytes, hipStream_t st) {
  static const bool off = [] { const char* e = getenv("GPSA_ZERO_KERNEL"); return e && e[0] == '0'; }();
  if (bytes == 0) return 0;
  if (off || (reinterpret_cast<uintptr_t>(ptr) & 15)) return (int)hipMemsetAsync(ptr, 0, bytes, st);
  const long long n16 = (long long)(bytes / 16);
  const long long blocks = n16 > 0 ? cdiv(n16, 256) : 1;
  zero_fill_kernel<<<(unsigned)blocks, 256, 0, st>>>(reinterpret_cast<uint4*>(ptr), n16, (int)(bytes % 16));
  GPSA_LAUNCH_CHECK();
  return 0;
}
// ... and device-to-device copies likewise (a memcpy NODE is the same kind of node; none was seen to misbehave, but the
// consumers of the step's copies - the KL terms in front of the loss kernel, a slice's accumulator - follow them at once)
__global__ void __launch_bounds__(256) copy_kernel(uint4* __restrict__ d, const uint4* __restrict__ s, long long n16,
                                                   int tail) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i < n16) d[i] = s[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < tail)
    reinterpret_cast<unsigned char*>(d + n16)[threadIdx.x] = reinterpret_cast<const unsigned char*>(s + n16)[threadIdx.x];
}
int copy_async(void* dst, const void* src, size_t bytes, hipStream_t st) {
  static const bool off = [] { const char* e = getenv("GPSA_ZERO_KERNEL"); return e && e[0] == '0'; }();
  if (bytes == 0) return 0;
  if (off || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15))
    return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
  const long long n16 = (long long)(bytes / 16);
  const long long blocks = n16 > 0 ? cdiv(n16, 256) : 1;
  copy_kernel<<<(unsigned)blocks, 256, 0, st>>>(reinterpret_cast<uint4*>(dst), reinterpret_cast<const uint4*>(src), n16,
                                                (int)(bytes % 16));
  GPSA_LAUNCH_CHECK();
  return 0;
}
// gpsa_elbo_fused_post in ONE launch (dvar_u fp64): the post kernel's last block closes dvar through ``tick``, a device
// word that is zero at entry and zero again afterwards (the step engine's backward keeps a few in its zero-filled region)
int elbo_fused_post_ticket(float* g_ext, float* dmeanT, float* abar, int M, long long C, int L, const float* gloss,
                           const float* var_u, double* dvar_u, double* part, int* tick, hipStream_t st) {
  const long long nb = cdiv(C, 256);
  elbo_post_kernel<double><<<(unsigned)nb, 256, 0, st>>>(g_ext, dmeanT, C, L, gloss, g_ext + (long long)L * C, part, abar,
                                                         M, tick, var_u, dvar_u);
  GPSA_LAUNCH_CHECK();
  return 0;
}
}  // namespace gpsa

extern "C" {

/* gpsa_elbo_loss_fwd / _bwd with some likelihood terms FUSED into the step (gpsa_step_io.fuse_elbo): zpart[i] non-null
 * = term i's partial sums of z^2 (nparts doubles, gpsa_step_io.ll_part); F[i] / dF[i] are then ignored. */
int gpsa_elbo_loss_fused_fwd(int n_ll, const float* const* F, const float* const* Y, const float* const* noise_u,
                             const int* S, const long long* N, const int* P, const double* const* zpart, int nparts,
                             const double* kl, int n_kl, double kl_scale, float* loss, double* ll_out, void* workspace,
                             long long workspace_bytes, void* stream) {
  if (n_ll < 1 || n_ll > GPSA_MAX_MODS || !F || !Y || !noise_u || !S || !N || !P || !loss || !ll_out || !zpart)
    return GPSA_EINVAL;
  if (workspace_bytes < 8LL * 4100 * n_ll) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  gpsa::ElboFinishArgs a;
  a.n_ll = n_ll;
  a.n_kl = kl ? n_kl : 0;
  a.kl = kl;
  a.kl_scale = kl_scale;
  a.ll = ll_out;
  a.loss = loss;
  for (int i = 0; i < n_ll; ++i) {
    if (S[i] < 1 || N[i] < 1 || P[i] < 1) return GPSA_EINVAL;
    const long long NP = N[i] * P[i], tot = NP * S[i];
    a.S[i] = S[i];
    a.z2_noise[i] = nullptr;
    a.tot[i] = (double)tot;
    if (zpart[i] != nullptr) {
      if (nparts < 1) return GPSA_EINVAL;
      a.part[i] = zpart[i];
      a.nb[i] = nparts;
      a.z2_noise[i] = noise_u[i];
      continue;
    }
    const int nb = gpsa::loglik_blocks(tot);
    double* part = reinterpret_cast<double*>(workspace) + 4100LL * i;
    gpsa::loglik_fwd_kernel<<<nb, 256, 0, st>>>(F[i], Y[i], noise_u[i], tot, NP, part);
    a.part[i] = part;
    a.nb[i] = nb;
  }
  gpsa::elbo_loss_finish_kernel<<<1, 256, 0, st>>>(a);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_elbo_loss_fused_bwd(int n_ll, const float* const* F, const float* const* Y, const float* const* noise_u,
                             const int* S, const long long* N, const int* P, const double* const* zpart, int nparts,
                             const float* gloss, int n_kl, double kl_scale, float* const* dF, float* const* dnoise,
                             float* dnoise_all, int n_noise, double* dkl, void* workspace, long long workspace_bytes,
                             void* stream) {
  if (n_ll < 1 || n_ll > GPSA_MAX_MODS || !F || !Y || !noise_u || !S || !N || !P || !gloss || !dF || !dnoise || !zpart)
    return GPSA_EINVAL;
  if (workspace_bytes < 8LL * 4100 * n_ll) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  for (int i = 0; i < n_ll; ++i) {
    if (S[i] < 1 || N[i] < 1 || P[i] < 1) return GPSA_EINVAL;
    const long long NP = N[i] * P[i], tot = NP * S[i];
    if (zpart[i] != nullptr) {
      gpsa::loglik_bwd_finish_kernel<<<1, 256, 0, st>>>(zpart[i], nparts, noise_u[i], nullptr, S[i], dnoise[i], gloss,
                                                        i == 0 ? dkl : nullptr, n_kl, kl_scale,
                                                        i == 0 ? dnoise_all : nullptr, n_noise, (double)tot);
      continue;
    }
    const int nb = gpsa::loglik_blocks(tot);
    double* part = reinterpret_cast<double*>(workspace) + 4100LL * i;
    gpsa::loglik_bwd_kernel<<<nb, 256, 0, st>>>(F[i], Y[i], noise_u[i], nullptr, S[i], tot, NP, dF[i], part, gloss);
    gpsa::loglik_bwd_finish_kernel<<<1, 256, 0, st>>>(part, nb, noise_u[i], nullptr, S[i], dnoise[i], gloss,
                                                      i == 0 ? dkl : nullptr, n_kl, kl_scale,
                                                      i == 0 ? dnoise_all : nullptr, n_noise);
  }
  GPSA_LAUNCH_CHECK();
  return 0;
}

long long gpsa_lmc_loglik_workspace(long long C, int L, int P, int nparts) {
  if (C < 1 || L < 1 || P < 1 || nparts < 1) return 0;
  return (long long)gpsa::lmc_blocks(C, nparts) * L * P * 4 + 256;
}

int gpsa_lmc_loglik_fused_f32(const float* F, const float* W, const float* Y, const float* noise_u, int S, long long N,
                              int L, int P, double* zpart, int nparts, float* dF, float* dW, void* workspace,
                              long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (!F || !W || !Y || !noise_u || !zpart || !dF || !dW || S < 1 || N < 1 || L < 1 || P < 1 || nparts < 1)
    return GPSA_EINVAL;
  if (L > 64) return GPSA_EUNSUPPORTED;
  const long long C = (long long)S * N;
  if (workspace_bytes < gpsa_lmc_loglik_workspace(C, L, P, nparts)) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const int G = lmc_blocks(C, nparts);
  float* part = (float*)workspace;
  // round 5: the three products on the matrix cores, the observations read once per spot tile (csrc/lmc.hip);
  // GPSA_LMC_MFMA=0: round 4's vector-pipe kernel below (L <= 32)
  static const bool mfma_off = [] { const char* e = getenv("GPSA_LMC_MFMA"); return e && e[0] == '0'; }();
  if (!mfma_off) {
    // (its tiles are 16 spots: at most one workgroup per tile)
    const long long nt = cdiv(N, 16);
    const int Gm = (int)(nt < G ? nt : G);
    int rc = lmc_mfma_launch(F, W, Y, noise_u, S, N, L, P, zpart, nparts, dF, part, Gm, st);
    if (rc != GPSA_EUNSUPPORTED) {
      if (rc) return rc;
      const long long n = (long long)L * P;
      reduce_rows_kernel<float, float><<<(unsigned)cdiv(n, 64), 256, 0, st>>>(part, Gm, n, n, dW, 1.0);
      GPSA_LAUNCH_CHECK();
      return 0;
    }
  }
  if (L > 32) return GPSA_EUNSUPPORTED;
#define GPSA_LMC_CASE(LBV)                                                                                 \
  {                                                                                                        \
    const size_t sm = (size_t)(LMC_TC * LBV + LBV * LMC_PS + LMC_TC * LMC_PS) * 4;                         \
    if (sm > 65536) {                                                                                      \
      hipError_t e = hipFuncSetAttribute((const void*)lmc_loglik_kernel<LBV>,                              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);             \
      if (e != hipSuccess) return (int)e;                                                                  \
    }                                                                                                      \
    lmc_loglik_kernel<LBV><<<G, 256, sm, st>>>(F, W, Y, noise_u, S, N, L, P, C, zpart, nparts, dF, part);  \
  }
  switch ((L + 3) / 4) {  // register arrays of L rounded up to a multiple of 4
    case 1: GPSA_LMC_CASE(4) break;
    case 2: GPSA_LMC_CASE(8) break;
    case 3: GPSA_LMC_CASE(12) break;
    case 4: GPSA_LMC_CASE(16) break;
    case 5: GPSA_LMC_CASE(20) break;
    case 6: GPSA_LMC_CASE(24) break;
    case 7: GPSA_LMC_CASE(28) break;
    default: GPSA_LMC_CASE(32) break;
  }
#undef GPSA_LMC_CASE
  GPSA_LAUNCH_CHECK();
  // dW[l,p] = sum over the blocks' partials, in block order (deterministic)
  const long long n = (long long)L * P;
  reduce_rows_kernel<float, float><<<(unsigned)cdiv(n, 64), 256, 0, st>>>(part, G, n, n, dW, 1.0);
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_version(void) { return 100; }
#ifndef GPSA_SOURCE_HASH
#define GPSA_SOURCE_HASH "unstamped"
#endif
const char* gpsa_source_hash(void) { return "GPSA_SOURCE_HASH=" GPSA_SOURCE_HASH; }
const char* gpsa_build_arch(void) { return "gfx950"; }

}  // extern "C"
