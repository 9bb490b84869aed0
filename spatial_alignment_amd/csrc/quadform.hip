// The dominant contraction of the GPSA step: the variational variance term
//     v[l,c] = alpha_c^T Omega_l alpha_c            (forward,  gpsa/models/vgpsa.py:192-196)
//     dalpha_c = 2 sum_l g[l,c] Omega_l alpha_c     (backward wrt alpha)
//     dOmega_l = sum_c g[l,c] alpha_c alpha_c^T     (backward wrt Omega)
// plus the whitening products Y = P X (vgpsa.py:177-180).  The reference materialises the
// [S,L,N,M] tensor a_t_Omega_tril; here nothing of that size ever reaches HBM.
//
// fp32, M <= 256:  MFMA path (v_mfma_f32_16x16x4_f32, exact fp32).  One wave owns a slab of 16*NCT
//   columns of alpha, kept in registers for the whole kernel as the MFMA B operand; the M x M left
//   operand streams through LDS in 16-deep K chunks (double buffered, register-staged prefetch).
//   The K index is permuted so that lane quarter q owns k = 16t+4q+r: the accumulator rows a lane
//   holds are then exactly the alpha rows it holds, and the quadratic form closes in registers
//   (2 shuffles per column tile, no LDS round trip).
// otherwise: generic tiled path built from gemm.hip + small fused elementwise kernels.
#include <stdlib.h>

#include "qf_common.hpp"

namespace gpsa {

template <typename T>
int gemm_launch(int transA, int transB, int m, int n, long long k, double alpha, const T* A,
                long long lda, long long sA, const T* B, long long ldb, long long sB, double beta,
                T* C, long long ldc, long long sC, int batch, int splitk, void* ws,
                long long ws_bytes, hipStream_t st);
// the same with a triangle mode (gemm.hip): 1 = op(A) upper triangular, 2 = lower-triangle blocks of C only
template <typename T>
int gemm_launch_tri(int transA, int transB, int m, int n, long long k, double alpha, const T* A,
                    long long lda, long long sA, const T* B, long long ldb, long long sB, double beta,
                    T* C, long long ldc, long long sC, int batch, int splitk, void* ws,
                    long long ws_bytes, hipStream_t st, int tri);

// + kscale [batch][k]: left operand A[m][k] * kscale[k] (non-transposed A); cscale [batch][n]: result columns
template <typename T>
int gemm_launch_scaled(int transA, int transB, int m, int n, long long k, double alpha, const T* A,
                       long long lda, long long sA, const T* B, long long ldb, long long sB, double beta,
                       T* C, long long ldc, long long sC, int batch, int splitk, void* ws,
                       long long ws_bytes, hipStream_t st, int tri, const T* kscale, long long sKs,
                       const T* cscale, long long sCs);

// U[l] = diag(Omega_l) + 2 * strict upper triangle of Omega_l (zeros below), stored as TD:
// a^T Omega a = a^T U a for symmetric Omega, and U a costs half the products of Omega a
template <typename TS, typename TD>
__global__ void tri_upper_kernel(const TS* __restrict__ src, int M, long long n, TD* __restrict__ dst) {
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= n) return;
  const long long e = idx % ((long long)M * M);
  const int i = (int)(e / M), k = (int)(e % M);
  dst[idx] = k > i ? (TD)(2.0 * (double)src[idx]) : (k == i ? (TD)src[idx] : TD(0));
}

// C[b][i][k] = C[b][k][i] for k > i (mirror the lower triangle of each M x M block)
template <typename T>
__global__ void mirror_lower_kernel(T* __restrict__ Cm, int M, long long n) {
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= n) return;
  const long long mm = (long long)M * M, b = idx / mm, e = idx % mm;
  const int i = (int)(e / M), k = (int)(e % M);
  if (k > i) Cm[idx] = Cm[b * mm + (long long)k * M + i];
}

// ------------------------------------------------------------------------------------------------
// generic helpers
// ------------------------------------------------------------------------------------------------
// out[m,c] = X[m,c] * g[c]
template <typename T>
__global__ void colscale_kernel(const T* __restrict__ X, const T* __restrict__ g, int M, long long C,
                                T* __restrict__ out) {
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  if (c >= C) return;
  const T gv = g[c];
  for (int m = blockIdx.y; m < M; m += gridDim.y) out[(long long)m * C + c] = X[(long long)m * C + c] * gv;
}

// out[b][m,c] = X[b / per][m,c] * g[b][c]   (grid.z = b; per = outputs that share one X panel: all of them
// for a single layer, L per view for a batch of views' layers)
template <typename T>
__global__ void colscale_batched_kernel(const T* __restrict__ X, const T* __restrict__ g, int M, long long C,
                                        T* __restrict__ out, int per = 0x7fffffff) {
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  if (c >= C) return;
  const T gv = g[(long long)blockIdx.z * C + c];
  T* o = out + (long long)blockIdx.z * M * C;
  const T* x = X + (long long)(blockIdx.z / per) * M * C;
  for (int m = blockIdx.y; m < M; m += gridDim.y) o[(long long)m * C + c] = x[(long long)m * C + c] * gv;
}

// out[m,c] = Y[m,c] + s * d[c] * X[m,c]      (blockIdx.z = problem of a batch of contiguous [M,C] panels)
template <typename T>
__global__ void col_axpy_kernel(const T* __restrict__ Y, const T* __restrict__ X,
                                const T* __restrict__ d, T s, int M, long long C,
                                T* __restrict__ out) {
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  if (c >= C) return;
  const long long pb = (long long)blockIdx.z * M * C;
  const T dv = s * d[(long long)blockIdx.z * C + c];
  for (int m = blockIdx.y; m < M; m += gridDim.y) {
    const long long o = pb + (long long)m * C + c;
    out[o] = Y[o] + dv * X[o];
  }
}

// v[b,c] = sum_m X[m,c] * Tm[b,m,c]; block = 64 columns x 4 row quarters (summed through LDS)
template <typename T>
__global__ void __launch_bounds__(256)
coldot_kernel(const T* __restrict__ X, const T* __restrict__ Tm, int M, long long C,
              T* __restrict__ v, long long vstride) {
  __shared__ T red[4][64];
  const int lane = threadIdx.x & 63, qr = threadIdx.x >> 6;
  const long long c = blockIdx.x * 64LL + lane;
  const T* t = Tm + (long long)blockIdx.y * M * C;
  T s = T(0);
  if (c < C) {  // four independent partial sums: the loads of one row do not wait for the previous row
    T s1 = T(0), s2 = T(0), s3 = T(0);
    int m = qr;
    for (; m + 12 < M; m += 16) {
      s += X[(long long)m * C + c] * t[(long long)m * C + c];
      s1 += X[(long long)(m + 4) * C + c] * t[(long long)(m + 4) * C + c];
      s2 += X[(long long)(m + 8) * C + c] * t[(long long)(m + 8) * C + c];
      s3 += X[(long long)(m + 12) * C + c] * t[(long long)(m + 12) * C + c];
    }
    for (; m < M; m += 4) s += X[(long long)m * C + c] * t[(long long)m * C + c];
    s = (s + s1) + (s2 + s3);
  }
  red[qr][lane] = s;
  __syncthreads();
  if (qr == 0 && c < C) v[(long long)blockIdx.y * vstride + c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// out[m,c] = 2 sum_l g[l,c] W[l][m,c]  (+ sum_l A[m,l] dm[l,c] when A is given: the mean term's share)
// blockIdx.z = problem of a batch of contiguous operands
template <typename T>
__global__ void col_wsum_kernel(const T* __restrict__ W, const T* __restrict__ g, int M, long long C,
                                int L, const T* __restrict__ A, const T* __restrict__ dm,
                                T* __restrict__ out) {
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  if (c >= C) return;
  {
    const long long b = blockIdx.z;
    W += b * L * M * C;
    g += b * L * C;
    if (A != nullptr) {
      A += b * M * L;
      dm += b * L * C;
    }
    out += b * M * C;
  }
  for (int m = blockIdx.y; m < M; m += gridDim.y) {
    T s = T(0), t = T(0);
    for (int l = 0; l < L; ++l) {
      s += g[(long long)l * C + c] * W[((long long)l * M + m) * C + c];
      if (A != nullptr) t += A[(long long)m * L + l] * dm[(long long)l * C + c];
    }
    out[(long long)m * C + c] = T(2) * s + t;
  }
}

// The same sum over products kept ROW-MAJOR [L][M][C] (the large-M data GP, M > 256), fp32: one thread = 4 columns x
// RB rows, so that g / dm are read once per RB rows and RB independent 16-byte loads of W are in flight per
// output; W is read exactly once (nontemporal: it does not come back).
template <int RB>
__global__ void __launch_bounds__(256)
col_wsum_rows_kernel(const float* __restrict__ W, const float* __restrict__ g, int M, long long C, int L,
                     const float* __restrict__ A, const float* __restrict__ dm, float* __restrict__ out) {
  const long long c = (blockIdx.x * 256LL + threadIdx.x) * 4;
  if (c >= C) return;
  const int m0 = blockIdx.y * RB;
  f32x4_t acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bool full = c + 4 <= C && (C & 3) == 0;  // aligned 16-byte accesses
  for (int l = 0; l < L; ++l) {
    f32x4_t gv, dv = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (full) {
      gv = *reinterpret_cast<const f32x4_t*>(g + (long long)l * C + c);
      if (A != nullptr) dv = *reinterpret_cast<const f32x4_t*>(dm + (long long)l * C + c);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        gv[i] = (c + i < C) ? g[(long long)l * C + c + i] : 0.f;
        dv[i] = (A != nullptr && c + i < C) ? dm[(long long)l * C + c + i] : 0.f;
      }
    }
    gv *= 2.f;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int m = m0 + r;
      if (m < M) {  // block-uniform
        const float* wp = W + ((long long)l * M + m) * C + c;
        f32x4_t w;
        if (full) {
          w = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(wp));
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = (c + i < C) ? wp[i] : 0.f;
        }
        const float a = (A != nullptr) ? A[(long long)m * L + l] : 0.f;
        acc[r] += gv * w + a * dv;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int m = m0 + r;
    if (m < M) {
      float* op = out + (long long)m * C + c;
      if (full) {
        *reinterpret_cast<f32x4_t*>(op) = acc[r];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (c + i < C) op[i] = acc[r][i];
      }
    }
  }
}

// coldot for output b plus, in the same pass over X, mean[b,c] = sum_m A[m,b] X[m,c]   (A [M,L])
template <typename T>
__global__ void __launch_bounds__(256)
coldot_mean_kernel(const T* __restrict__ X, const T* __restrict__ Tm, const T* __restrict__ A, int M,
                   long long C, int L, T* __restrict__ v, T* __restrict__ mean) {
  __shared__ T red[2][4][64];
  const int lane = threadIdx.x & 63, qr = threadIdx.x >> 6, b = blockIdx.y;
  const long long c = blockIdx.x * 64LL + lane;
  {  // blockIdx.z = problem of a batch of contiguous operands
    const long long pz = blockIdx.z;
    X += pz * M * C;
    Tm += pz * L * M * C;
    A += pz * M * L;
    v += pz * L * C;
    mean += pz * L * C;
  }
  const T* t = Tm + (long long)b * M * C;
  T s0 = T(0), s1 = T(0), m0 = T(0), m1 = T(0);
  if (c < C) {
    int m = qr;
    for (; m + 4 < M; m += 8) {
      const T x0 = X[(long long)m * C + c], x1 = X[(long long)(m + 4) * C + c];
      s0 += x0 * t[(long long)m * C + c];
      s1 += x1 * t[(long long)(m + 4) * C + c];
      m0 += x0 * A[(long long)m * L + b];
      m1 += x1 * A[(long long)(m + 4) * L + b];
    }
    for (; m < M; m += 4) {
      const T x0 = X[(long long)m * C + c];
      s0 += x0 * t[(long long)m * C + c];
      m0 += x0 * A[(long long)m * L + b];
    }
  }
  red[0][qr][lane] = s0 + s1;
  red[1][qr][lane] = m0 + m1;
  __syncthreads();
  if (qr == 0 && c < C) {
    v[(long long)b * C + c] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
    mean[(long long)b * C + c] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
  }
}

// batch > 1: ``batch`` layers with contiguous operands (alpha [batch][M,C], Omega [batch][L,M,M], W
// [batch][L,M,C], v / meanT [batch][L,C], dcT [batch][M,L]): the views' warp GPs in ONE launch sequence.
// The product W_b = Omega_b alpha_b is ONE strided-batched product with the L outputs stacked as rows.
template <typename T>
int quadform_fwd_keep(const T* alpha, const T* Omega, int M, long long C, int L, T* v, T* W,
                      const T* dcT, T* meanT, hipStream_t st, int batch = 1) {
  int rc;
  if (batch == 1)
    rc = gemm_launch<T>(0, 0, M, (int)C, M, 1.0, Omega, M, (long long)M * M, alpha, C, 0, 0.0, W, C,
                        (long long)M * C, L, 1, nullptr, 0, st);
  else
    rc = gemm_launch<T>(0, 0, L * M, (int)C, M, 1.0, Omega, M, (long long)L * M * M, alpha, C,
                        (long long)M * C, 0.0, W, C, (long long)L * M * C, batch, 1, nullptr, 0, st);
  if (rc) return rc;
  dim3 grid((unsigned)cdiv(C, 64), (unsigned)L, (unsigned)batch);
  if (dcT != nullptr)
    coldot_mean_kernel<T><<<grid, 256, 0, st>>>(alpha, W, dcT, M, C, L, v, meanT);
  else if (batch == 1)
    coldot_kernel<T><<<grid, 256, 0, st>>>(alpha, W, M, C, v, C);
  else
    return GPSA_EINVAL;
  GPSA_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int quadform_bwd_alpha_kept(const T* W, const T* g, int M, long long C, int L, const T* dcT, const T* dmeanT,
                            T* dalpha, hipStream_t st, int batch = 1) {
  dim3 grid((unsigned)cdiv(C, 256), (unsigned)((M < 64) ? M : 64), (unsigned)batch);
  col_wsum_kernel<T><<<grid, 256, 0, st>>>(W, g, M, C, L, dcT, dmeanT, dalpha);
  GPSA_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int generic_quadform_fwd(const T* alpha, const T* Omega, int M, long long C, int L, T* v, void* ws,
                         long long ws_bytes, hipStream_t st) {
  const long long per = (long long)M * C * (long long)sizeof(T);
  int lc = (int)(ws_bytes / per);
  if (lc < 1) return GPSA_EWORKSPACE;
  if (lc > L) lc = L;
  T* Tm = reinterpret_cast<T*>(ws);
  for (int l0 = 0; l0 < L; l0 += lc) {
    const int nb = (L - l0 < lc) ? L - l0 : lc;
    // Omega holds U_l = diag + 2 strict-upper (tri_upper_kernel): block row m0 contracts k >= m0 only
    int rc = gemm_launch_tri<T>(0, 0, M, (int)C, M, 1.0, Omega + (long long)l0 * M * M, M,
                                (long long)M * M, alpha, C, 0, 0.0, Tm, C, (long long)M * C, nb, 1,
                                nullptr, 0, st, 1);
    if (rc) return rc;
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)nb);
    coldot_kernel<T><<<grid, 256, 0, st>>>(alpha, Tm, M, C, v + (long long)l0 * C, C);
    GPSA_LAUNCH_CHECK();
  }
  return 0;
}

template <typename T>
int generic_quadform_bwd_alpha(const T* alpha, const T* Omega, const T* g, int M, long long C, int L,
                               T* dalpha, void* ws, long long ws_bytes, hipStream_t st) {
  // Omega_l (alpha o g_l) = (Omega_l alpha) o g_l: the column scale rides in the product's epilogue, nothing
  // of size M x C is materialised per output
  (void)ws;
  (void)ws_bytes;
  for (int l = 0; l < L; ++l) {
    int rc = gemm_launch_scaled<T>(0, 0, M, (int)C, M, 2.0, Omega + (long long)l * M * M, M, 0, alpha, C, 0,
                                   l == 0 ? 0.0 : 1.0, dalpha, C, 0, 1, 1, nullptr, 0, st, 0, nullptr, 0,
                                   g + (long long)l * C, 0);
    if (rc) return rc;
  }
  return 0;
}

static inline int gram_splitk(long long C, int M) {
  // enough K-splits that a single M x M product still fills the chip (tiles of 64 x 64)
  const long long tiles = cdiv(M, 64) * cdiv(M, 64);
  static const long long target = [] { const char* e = getenv("GPSA_GRAM_SK_TARGET"); return e ? atoll(e) : 512LL; }();
  long long s = cdiv(target, tiles);
  const long long cap = C / 256 > 1 ? C / 256 : 1;
  if (s > cap) s = cap;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return (int)s;
}


template <typename T>
int generic_quadform_bwd_omega(const T* alpha, const T* g, int M, long long C, int L, T* dOmega,
                               void* ws, long long ws_bytes, hipStream_t st) {
  // up to 4 outputs per pass: one scaling launch and one batched split-K product for the group.  (Scaling the
  // left operand by g along the contracted index INSIDE the product - gemm_launch_scaled's kscale - was
  // measured slower than this materialised copy: 26.5 vs 25.1 ms at M = 500, 55 vs 43 ms at M = 1000.)
  if constexpr (sizeof(T) == 4) {
    // fp32: the LDS-DMA Gram kernel (16-byte aligned rows; its partial slabs must fit the workspace)
    const int nblk = (int)cdiv(M, 128), pairs = nblk * (nblk + 1) / 2;
    long long ns = cdiv(1024, (long long)pairs * L);
    const long long nch = cdiv(C, 16);
    if (ns > nch / 8) ns = nch / 8 > 0 ? nch / 8 : 1;
    if (ns > 16) ns = 16;
    if (ns < 1) ns = 1;
    const long long Cpad = nch * 16, part_b = (long long)L * ns * M * M * 4;
    const long long need = part_b + (long long)L * Cpad * 4;
    if (!gram_big_off() && (C & 3) == 0 && C >= 16 && (reinterpret_cast<uintptr_t>(alpha) & 15) == 0 && need <= ws_bytes &&
        L <= 65535) {
      float* gpad = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + part_b);
      pad_rows_kernel<<<(unsigned)cdiv((long long)L * Cpad, 256), 256, 0, st>>>(g, L, C, Cpad, gpad);
      GPSA_LAUNCH_CHECK();
      GramBigArgs a{alpha, gpad, reinterpret_cast<float*>(ws), M, L, (int)ns, nblk, C, Cpad, 0, big_phase()};
      const int lb = big_remap_lb();
      const long long combos = (long long)pairs * ns * cdiv(L, lb > 0 ? lb : 1);
      if (lb > 0 && 8 * lb * cdiv(combos, 8) < 0x7fffffffLL) {
        a.lb = lb;
        gram_big_launch(dim3((unsigned)(8 * lb * cdiv(combos, 8))), st, a);
      } else {
        gram_big_launch(dim3((unsigned)pairs, (unsigned)ns, (unsigned)L), st, a);
      }
      GPSA_LAUNCH_CHECK();
      gram_big_reduce_kernel<float><<<dim3((unsigned)cdiv((long long)M * M, 256), (unsigned)L), 256, 0, st>>>(
          a.part, M, (int)ns, dOmega);
      GPSA_LAUNCH_CHECK();
      return 0;
    }
  }
  const int sk = gram_splitk(C, M);
  const long long tmp_b = (long long)M * C * (long long)sizeof(T);
  const long long part_b = (sk > 1) ? (long long)sk * M * M * (long long)sizeof(T) : 0;
  long long nbmax = ws_bytes / (tmp_b + part_b);
  if (nbmax < 1) return GPSA_EWORKSPACE;
  if (nbmax > 4) nbmax = 4;
  for (int l0 = 0; l0 < L; l0 += (int)nbmax) {
    const int nb = (int)((L - l0 < nbmax) ? L - l0 : nbmax);
    T* tmp = reinterpret_cast<T*>(ws);
    void* part = reinterpret_cast<char*>(ws) + (long long)nb * tmp_b;
    dim3 grid((unsigned)cdiv(C, 256), (unsigned)((M < 64) ? M : 64), (unsigned)nb);
    colscale_batched_kernel<T><<<grid, 256, 0, st>>>(alpha, g + (long long)l0 * C, M, C, tmp);
    GPSA_LAUNCH_CHECK();
    // symmetric result: only the blocks touching the lower triangle are computed, then mirrored
    int rc = gemm_launch_tri<T>(0, 1, M, M, C, 1.0, tmp, C, (long long)M * C, alpha, C, 0, 0.0,
                                dOmega + (long long)l0 * M * M, M, (long long)M * M, nb, sk, part,
                                (long long)nb * part_b, st, 2);
    if (rc) return rc;
    const long long nn = (long long)nb * M * M;
    mirror_lower_kernel<T><<<(unsigned)cdiv(nn, 256), 256, 0, st>>>(dOmega + (long long)l0 * M * M, M, nn);
    GPSA_LAUNCH_CHECK();
  }
  return 0;
}

// q[c] = sum_m Y[m,c]^2; block = 64 columns x 4 row quarters
template <typename T>
__global__ void __launch_bounds__(256)
colsq_kernel(const T* __restrict__ Y, int M, long long C, T* __restrict__ q) {
  __shared__ T red[4][64];
  const int lane = threadIdx.x & 63, qr = threadIdx.x >> 6;
  const long long c = blockIdx.x * 64LL + lane;
  T s = T(0);
  if (c < C)
    for (int m = qr; m < M; m += 4) {
      const T y = Y[(long long)m * C + c];
      s += y * y;
    }
  red[qr][lane] = s;
  __syncthreads();
  if (qr == 0 && c < C) q[c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// ------------------------------------------------------------------------------------------------
// MFMA panel kernels (fp32)
// ------------------------------------------------------------------------------------------------

template <typename TS>
__global__ void pack_panels_kernel(const TS* __restrict__ src, int M, int MB, int L, int transpose,
                                   float* __restrict__ dst, int layout, const float* __restrict__ drow) {
  const int MP = MB * 16;
  const long long per = (long long)MP * MP;
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= per * L) return;
  const int l = (int)(idx / per);
  const int e = (int)(idx % per);
  const int kc = e / (MP * 16);
  const int rem = e % (MP * 16);
  const int rt = rem / 256, lane = (rem % 256) / 4, r = rem % 4;
  const int i = rt * 16 + (lane & 15);
  const bool kstep = (layout & PACK_KSTEP) || ((layout & PACK_KSTEP_LAST) && kc == MB - 1);
  const int k = kc * 16 + (kstep ? r * 4 + (lane >> 4) : (lane >> 4) * 4 + r);
  float v = 0.f;
  if (i < M && k < M) {
    const TS* sp = src + (long long)l * M * M;
    v = (float)(transpose ? sp[(long long)k * M + i] : sp[(long long)i * M + k]);
  } else if (drow != nullptr && i == M && k < M) {
    // row M of operand l (the first padding row) carries delta[:, l] ([M][L] row-major): the product's row M is then
    // delta_l^T alpha - the data GP's mean, out of MFMAs the padding executes anyway (panel_elbo_kernel)
    v = drow[(long long)k * L + l];
  }
  // symmetric quadratic form: only tiles kc >= rt are used; off-diagonal ones count twice
  if (layout & PACK_SYM_UPPER) v = (kc > rt) ? 2.f * v : (rt == kc ? v : 0.f);
  dst[idx] = v;
}

// row tiles of the panel / quadratic-form kernels' instantiations (0: no MFMA variant, generic path).
// 24 and 32 (M <= 512) run one column tile per wave: the accumulators + the B slab of a wave are then
// 2 x 4 MB registers, the whole 512-register budget at MB = 32 (it spills there).  Measured against the
// generic tiled path at C = 50k, L = 8: accumulate 1.5x (M = 380) and 1.33x (M = 500) faster, quadratic
// form 2.7x faster at 380 but 2.4x SLOWER at 500, store (L = 1) slower at both: hence the caps below.
constexpr int MB_MAX_ACCUM = 32, MB_MAX_QUAD = 24, MB_MAX_STORE = 16;
static inline int mfma_mb_for(int M) {
  const int mb = (M + 15) / 16;
  if (mb <= 2) return 2;
  if (mb <= 4) return 4;
  if (mb <= 7) return 7;
  if (mb <= 13) return 13;
  if (mb <= 16) return 16;
  if (mb <= 24) return 24;
  if (mb <= 32) return 32;
  return 0;
}

// the Gram kernel keeps ALL lower tiles of an output in one workgroup's accumulators: M <= 256 only
static inline int gram_mb_for(int M) {
  const int mb = mfma_mb_for(M);
  return mb <= 16 ? mb : 0;
}

static inline int panel_nct_for(int MB) { return MB >= 24 ? 1 : (MB == 16 ? 2 : (MB == 13 ? 3 : 4)); }

static inline bool force_generic() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GPSA_FORCE_GENERIC");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

// Adds, in workgroup order, the partial-tile slabs that panel_mfma_kernel<ACCUM> wrote: tile t is
// shared by the workgroups whose item ranges [b*T/G, (b+1)*T/G) cut into [t*L, (t+1)*L).
__global__ void __launch_bounds__(256)
panel_slab_reduce_kernel(const float* __restrict__ slab, int M, int MP, int wgcols, long long C, int L,
                         long long ntiles, int G, float* __restrict__ out) {
  const long long t = blockIdx.x;
  const long long T = ntiles * L, lo = t * L, hi = lo + L;
  long long b0 = lo * G / T;
  while (b0 > 0 && (b0 * T / G) > lo) --b0;
  while (((b0 + 1) * T / G) <= lo) ++b0;
  long long b1 = b0;
  while (b1 + 1 < G && ((b1 + 1) * T / G) < hi) ++b1;
  if (b0 == b1) return;  // a single workgroup owned the whole tile: it stored straight to `out`
  // slab offsets of the contributors, worked out once per block (not per element: 64-bit divisions)
  __shared__ long long soff[512];
  __shared__ int scount;
  if (threadIdx.x == 0) {
    int n = 0;
    for (long long b = b0; b <= b1 && n < 512; ++b) {
      const long long i0 = b * T / G, i1 = (b + 1) * T / G;
      if (i1 <= i0) continue;
      const int which = ((i0 / L) == t) ? 0 : 1;  // the tile is this workgroup's first tile, else its last
      soff[n++] = (b * 2 + which) * (long long)MP * wgcols;
    }
    scount = n;
  }
  __syncthreads();
  const int n = scount;
  const long long cbase = t * wgcols;
  for (int e = threadIdx.x + blockIdx.y * 256; e < M * wgcols; e += 256 * gridDim.y) {
    const int row = e / wgcols, cl = e % wgcols;
    const long long c = cbase + cl;
    if (c >= C) continue;
    const float* p = slab + (long long)row * wgcols + cl;
    float s = 0.f;
    for (int k = 0; k < n; k += 4) {  // four contributors' loads in flight; added in workgroup order: reproducible
      const float v0 = p[soff[k]];
      const float v1 = (k + 1 < n) ? p[soff[k + 1]] : 0.f;
      const float v2 = (k + 2 < n) ? p[soff[k + 2]] : 0.f;
      const float v3 = (k + 3 < n) ? p[soff[k + 3]] : 0.f;
      s += v0;
      if (k + 1 < n) s += v1;
      if (k + 2 < n) s += v2;
      if (k + 3 < n) s += v3;
    }
    out[(long long)row * C + c] = s;
  }
}

// Backward of the kept form: out[m,c] = scale * sum_l g[l,c] * (Omega_l alpha)[m,c], streaming the products the
// forward kept (panel_mfma_kernel<QUAD>'s ``keep``, in its fragment order) exactly once: memory-bound, 13 independent
// 16-byte loads per thread and output.  Block = the four waves' slots of one (column tile, ct).
template <int MB, int NCT, int RTB>
__global__ void __launch_bounds__(256)
kept_wsum_kernel(const float* __restrict__ keep, const float* __restrict__ g, int M, long long C, int L,
                 long long ntiles, float scale, const float* __restrict__ A, const float* __restrict__ dm,
                 float* __restrict__ out) {
  // blockIdx.z: a group of RTB row tiles (more blocks and fewer registers than one thread per column: a short
  // column range otherwise leaves the chip with one block per CU and a chain of L load latencies each)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  const long long tile = blockIdx.x;
  const int ct = blockIdx.y, rt0 = blockIdx.z * RTB;
  const long long c = tile * (64 * NCT) + (long long)w * (16 * NCT) + ct * 16 + j;
  f32x4 acc[RTB];
#pragma unroll
  for (int i = 0; i < RTB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4* kp = reinterpret_cast<const f32x4*>(keep) + (((tile * 4 + w) * NCT + ct) * MB + rt0) * 64 + lane;
  const long long lstride = ntiles * 4 * NCT * MB * 64;  // f32x4 elements between consecutive l
  int l = 0;
  for (; l + 1 < L; l += 2) {  // two outputs' loads in flight
    const float g0 = (c < C) ? g[(long long)l * C + c] : 0.f;
    const float g1 = (c < C) ? g[(long long)(l + 1) * C + c] : 0.f;
    const f32x4* p0 = kp + (long long)l * lstride;
    const f32x4* p1 = p0 + lstride;
    f32x4 u0[RTB], u1[RTB];
#pragma unroll
    for (int i = 0; i < RTB; ++i)
      if (rt0 + i < MB) {
        u0[i] = __builtin_nontemporal_load(&p0[i * 64]);
        u1[i] = __builtin_nontemporal_load(&p1[i * 64]);
      }
#pragma unroll
    for (int i = 0; i < RTB; ++i)
      if (rt0 + i < MB) {
        acc[i] += g0 * u0[i];
        acc[i] += g1 * u1[i];
      }
  }
  if (l < L) {
    const float g0 = (c < C) ? g[(long long)l * C + c] : 0.f;
    const f32x4* p0 = kp + (long long)l * lstride;
#pragma unroll
    for (int i = 0; i < RTB; ++i)
      if (rt0 + i < MB) acc[i] += g0 * __builtin_nontemporal_load(&p0[i * 64]);
  }
  // the mean term's share  sum_l A[m,l] dm[l,c]  (A = delta [M,L], dm = d mean [L,C]) on the matrix cores: the
  // thread's accumulators are already MFMA C-layout tiles (row 4 kq + r, column j), so each 16 x 16 tile takes
  // ceil(L / 4) MFMA steps with A[16 rt + j][4 s + kq] and dm[4 s + kq][c_j] as operands - no separate product
  // and no second pass over the [M,C] panel
  f32x4 tacc[RTB];
#pragma unroll
  for (int i = 0; i < RTB; ++i) tacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (A != nullptr) {  // uniform
    for (int l0 = 0; l0 < L; l0 += 4) {
      const int ll = l0 + kq;
      const float b = (ll < L && c < C) ? dm[(long long)ll * C + c] : 0.f;
#pragma unroll
      for (int i = 0; i < RTB; ++i) {
        const int row = (rt0 + i) * 16 + j;
        const float a = (rt0 + i < MB && ll < L && row < M) ? A[(long long)row * L + ll] : 0.f;
        tacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, tacc[i], 0, 0, 0);
      }
    }
  }
  if (c < C) {
#pragma unroll
    for (int i = 0; i < RTB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (rt0 + i) * 16 + kq * 4 + r;
        if (rt0 + i < MB && row < M) out[(long long)row * C + c] = fmaf(scale, acc[i][r], tacc[i][r]);
      }
  }
}

template <int MODE>
int panel_mfma_launch(int MBsel, const float* Ppk, const float* X, const float* g, int M,
                      long long C, int L, float* out, float* colsq, float scale, float* slab,
                      hipStream_t st, float* keep = nullptr) {
#define GPSA_PANEL_CASE(MBV, NCTV)                                                              \
  case MBV:                                                                                     \
    if constexpr (MBV <= 16 || MODE == MODE_ACCUM) {  /* 24 / 32 row tiles: accumulate only */  \
    const long long ntiles = cdiv(C, 64 * NCTV), T = ntiles * L;                                \
    const int wgs_per_cu = (MBV * NCTV >= 24) ? 1 : 2;                                          \
    long long grid = (long long)num_cus() * wgs_per_cu;                                         \
    if (MODE == MODE_STORE) grid = T;              /* L == 1: one item per tile */              \
    if (grid > T) grid = T;                                                                     \
    constexpr int RLV = 2;  /* QUAD: the caller packs the last chunk in K-step order (PACK_KSTEP_LAST) */ \
    if (RLV == 2 && M - 16 * (MBV - 1) <= 8)                                                    \
      panel_mfma_kernel<MBV, NCTV, MODE, RLV><<<(unsigned)grid, 256, 0, st>>>(                  \
          Ppk, X, g, M, C, L, out, colsq, scale, slab, keep);                                   \
    else                                                                                        \
      panel_mfma_kernel<MBV, NCTV, MODE, 4><<<(unsigned)grid, 256, 0, st>>>(                    \
          Ppk, X, g, M, C, L, out, colsq, scale, slab, keep);                                   \
    if (MODE == MODE_ACCUM) {                                                                   \
      /* few tiles (a short column range): more blocks per tile, the reduce is latency-bound */ \
      dim3 rg((unsigned)ntiles, ntiles >= 512 ? 8 : (ntiles >= 128 ? 16 : 32));               \
      panel_slab_reduce_kernel<<<rg, 256, 0, st>>>(slab, M, MBV * 16, 64 * NCTV, C, L, ntiles,  \
                                                   (int)grid, out);                             \
    }                                                                                           \
    } else {                                                                                    \
      return GPSA_EUNSUPPORTED;                                                                 \
    }                                                                                           \
    break;
  switch (MBsel) {
    GPSA_PANEL_CASE(2, 4)
    GPSA_PANEL_CASE(4, 4)
    GPSA_PANEL_CASE(7, 4)
    GPSA_PANEL_CASE(13, 3)
    GPSA_PANEL_CASE(16, 2)
    GPSA_PANEL_CASE(24, 1)
    GPSA_PANEL_CASE(32, 1)
    default:
      return GPSA_EUNSUPPORTED;
  }
#undef GPSA_PANEL_CASE
  GPSA_LAUNCH_CHECK();
  return 0;
}

// floats of slab space behind the packed operand: 2 partial tiles per workgroup of the persistent grid
static inline long long accum_slab_floats(int MB) {
  const int nct = panel_nct_for(MB);
  const long long G = (long long)num_cus() * ((MB * nct >= 24) ? 1 : 2);
  return G * 2 * (long long)MB * 16 * 64 * nct;
}

static int quad_sym_launch(int MBsel, const float* Ppk, const float* X, int M, long long C, int L,
                           float* out, hipStream_t st) {
#define GPSA_QS_CASE(MBV, NCTV)                                                                  \
  case MBV: {                                                                                    \
    const long long T = cdiv(C, 64 * NCTV) * L;                                                  \
    long long grid = (long long)num_cus() * ((MBV * NCTV >= 24) ? 1 : 2);                        \
    if (grid > T) grid = T;                                                                      \
    if (M - 16 * (MBV - 1) <= 8)                                                                 \
      quad_sym_mfma_kernel<MBV, NCTV, 2><<<(unsigned)grid, 256, 0, st>>>(Ppk, X, M, C, L, out);  \
    else                                                                                         \
      quad_sym_mfma_kernel<MBV, NCTV, 4><<<(unsigned)grid, 256, 0, st>>>(Ppk, X, M, C, L, out);  \
  } break;
  switch (MBsel) {
    GPSA_QS_CASE(2, 4)
    GPSA_QS_CASE(4, 4)
    GPSA_QS_CASE(7, 4)
    GPSA_QS_CASE(13, 3)
    GPSA_QS_CASE(16, 2)
    GPSA_QS_CASE(24, 1)
    default:
      return GPSA_EUNSUPPORTED;
  }
#undef GPSA_QS_CASE
  GPSA_LAUNCH_CHECK();
  return 0;
}

static inline int gram_nsplit(long long C, int L) {
  // grid = L x nsplit workgroups, one per CU: ONE round of workgroups that (nearly) fills the chip beat
  // 2-4 rounds at every size measured (2.44 vs 2.52 ms at C = 100k, 0.34 vs 0.43 ms at C = 12.5k; L = 50):
  // fewer prologues, and a third of the partial sums for gram_reduce_kernel to add
  const long long nch = cdiv(C, GR_KC);
  const int cus = num_cus();
  static const int forced = [] { const char* e = getenv("GPSA_GRAM_NSPLIT"); return e ? atoi(e) : 0; }();
  if (forced > 0) return (int)(forced < nch ? forced : nch);
  // W workgroups per output group fill the chip once; each takes c = ceil(nch / W) chunks.  (Few outputs -
  // the warp GPs' L = 2 - used to be held to >= 4 chunks per workgroup, i.e. 39 workgroups on 256 CUs at
  // C = 10k: 75 us; one chunk each, 157 workgroups: 44 us including the larger reduce.)
  long long W = cus / (L > 0 ? L : 1);
  if (W < 1) W = 1;
  const long long c = cdiv(nch, W);
  long long ns = cdiv(nch, c);
  if (ns > 256) ns = 256;
  if (ns < 1) ns = 1;
  return (int)ns;
}

// gpad[l][c] = g[l][c] for c < C, 0 for C <= c < Cpad
__global__ void pad_rows_kernel(const float* __restrict__ g, int L, long long C, long long Cpad,
                                float* __restrict__ gpad) {
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= (long long)L * Cpad) return;
  const long long l = idx / Cpad, c = idx % Cpad;
  gpad[idx] = c < C ? g[l * C + c] : 0.f;
}

static inline long long gram_gpad_floats(long long C, int L) { return (long long)L * cdiv(C, GR_KC) * GR_KC; }

// outputs per workgroup: two where the accumulators of both fit (MB <= 13), there are at least two, and
// the column range is long enough for the saved staging to outweigh the doubled number of partial slabs
// the reduce kernel adds (2.05 vs 2.28 ms at C = 100k, 0.28 vs 0.30 ms at C = 12.5k; L = 50)
static inline int gram_nl(int MB, int L, long long C) {
  static const int forced = [] { const char* e = getenv("GPSA_GRAM_NL"); return e ? atoi(e) : 0; }();
  const bool can = L >= 2 && MB <= 13;
  if (forced == 1 || forced == 2) return (forced == 2 && can) ? 2 : 1;
  return (can && C >= 8192) ? 2 : 1;
}

// dmean / ddelta (both or neither; the d-delta option of gram_mfma_kernel): ddelta [M][L] = dbeta ddelta + alpha dmean^T
// out of the first padding row of the LAST tile row; the caller has checked gram_delta_ok
static int gram_mfma_launch(int MBsel, const float* alpha, const float* g, int M, long long C, int L,
                            void* dOmega, int out_dtype, float* ws, hipStream_t st, const float* dmean = nullptr,
                            float* ddelta = nullptr, float dbeta = 0.f) {
  const int nl = gram_nl(MBsel, L, C);
  const int ns = gram_nsplit(C, (L + nl - 1) / nl);
  const long long Cpad = cdiv(C, GR_KC) * GR_KC;
  const float* gpad = ws;
  long long gstride = Cpad;
  float* part = ws + ((gram_gpad_floats(C, L) + 63) / 64) * 64;
  const bool al = (C % 4 == 0) && (C >= 8) && ((reinterpret_cast<uintptr_t>(alpha) & 15) == 0);
  if (dmean != nullptr && !(al && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dmean)) & 15) == 0))
    return GPSA_EUNSUPPORTED;
  if (al && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    // the kernel reads g where it lies (its column groups past C come from a block of zeros): no padded copy
    gpad = g;
    gstride = C;
  } else {
    pad_rows_kernel<<<(unsigned)cdiv((long long)L * Cpad, 256), 256, 0, st>>>(g, L, C, Cpad, ws);
    GPSA_LAUNCH_CHECK();
  }
  dim3 grid((unsigned)((L + nl - 1) / nl), (unsigned)ns);
#define GPSA_GRAM_CASE(MBV)                                                                        \
  case MBV:                                                                                        \
    if (nl == 2 && MBV <= 13) {                                                                    \
      if (al) gram_mfma_kernel<MBV, true, (MBV <= 13 ? 2 : 1)><<<grid, 256, 0, st>>>(alpha, gpad, gstride, M, C, L, ns, part, dmean);  \
      else gram_mfma_kernel<MBV, false, (MBV <= 13 ? 2 : 1)><<<grid, 256, 0, st>>>(alpha, gpad, gstride, M, C, L, ns, part, nullptr);    \
    } else {                                                                                       \
      if (al) gram_mfma_kernel<MBV, true, 1><<<grid, 256, 0, st>>>(alpha, gpad, gstride, M, C, L, ns, part, dmean); \
      else gram_mfma_kernel<MBV, false, 1><<<grid, 256, 0, st>>>(alpha, gpad, gstride, M, C, L, ns, part, nullptr);   \
    }                                                                                              \
    break;
  switch (MBsel) {
    GPSA_GRAM_CASE(2)
    GPSA_GRAM_CASE(4)
    GPSA_GRAM_CASE(7)
    GPSA_GRAM_CASE(13)
    GPSA_GRAM_CASE(16)
    default:
      return GPSA_EUNSUPPORTED;
  }
#undef GPSA_GRAM_CASE
  GPSA_LAUNCH_CHECK();
  dim3 rgrid((unsigned)cdiv(M, 32), (unsigned)cdiv(M + (ddelta != nullptr ? 1 : 0), 8), (unsigned)L);
  if (out_dtype == GPSA_F64)
    gram_reduce_kernel<double><<<rgrid, 256, 0, st>>>(part, M, MBsel * 16, L, ns, (double*)dOmega, ddelta, dbeta);
  else
    gram_reduce_kernel<float><<<rgrid, 256, 0, st>>>(part, M, MBsel * 16, L, ns, (float*)dOmega, ddelta, dbeta);
  GPSA_LAUNCH_CHECK();
  return 0;
}

static inline long long gram_ws_bytes(int MB, long long C, int L) {
  const int nl = gram_nl(MB, L, C);
  return (((gram_gpad_floats(C, L) + 63) / 64) * 64 +
          (long long)L * gram_nsplit(C, (L + nl - 1) / nl) * MB * 16 * MB * 16) * 4;
}

// The M x M operands (Omega_l, L^-1) may arrive in either precision: the MFMA paths convert while
// packing, the generic paths take a converted copy from the head of the workspace.
static int pack_any(int p_dtype, const void* src, int M, int MB, int L, int transpose, float* dst,
                    hipStream_t st, int layout = 0, const float* drow = nullptr) {
  const long long tot = (long long)L * MB * 16 * MB * 16;
  const unsigned grid = (unsigned)cdiv(tot, 256);
  if (p_dtype == GPSA_F32)
    pack_panels_kernel<float><<<grid, 256, 0, st>>>((const float*)src, M, MB, L, transpose, dst, layout, drow);
  else if (p_dtype == GPSA_F64)
    pack_panels_kernel<double><<<grid, 256, 0, st>>>((const double*)src, M, MB, L, transpose, dst, layout, drow);
  else
    return GPSA_EINVAL;
  GPSA_LAUNCH_CHECK();
  return 0;
}

template <typename TS, typename TD>
__global__ void convert_kernel(const TS* __restrict__ src, long long n, TD* __restrict__ dst) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i < n) dst[i] = (TD)src[i];
}

// *out = P viewed as T: P itself when it already is, else a converted copy carved off the workspace
template <typename T>
static int operand_as(int p_dtype, const void* P, long long n, const T** out, void** ws,
                      long long* ws_bytes, hipStream_t st) {
  const int want = sizeof(T) == 8 ? GPSA_F64 : GPSA_F32;
  if (p_dtype == want) {
    *out = (const T*)P;
    return 0;
  }
  if (p_dtype != GPSA_F32 && p_dtype != GPSA_F64) return GPSA_EINVAL;
  const long long need = ((n * (long long)sizeof(T) + 255) / 256) * 256;
  if (*ws_bytes < need) return GPSA_EWORKSPACE;
  T* dst = (T*)*ws;
  if (p_dtype == GPSA_F32)
    convert_kernel<float, T><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float*)P, n, dst);
  else
    convert_kernel<double, T><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const double*)P, n, dst);
  GPSA_LAUNCH_CHECK();
  *out = dst;
  *ws = (char*)*ws + need;
  *ws_bytes -= need;
  return 0;
}

}  // namespace gpsa

namespace gpsa {
// the triangular operand U of the generic quadratic form, always a copy at the head of the workspace
template <typename T>
static int tri_operand(int p_dtype, const void* P, int M, int L, const T** out, void** ws, long long* ws_bytes,
                       hipStream_t st) {
  const long long n = (long long)L * M * M;
  const long long need = (n * (long long)sizeof(T) + 255) / 256 * 256;
  if (*ws_bytes < need) return GPSA_EWORKSPACE;
  T* dst = reinterpret_cast<T*>(*ws);
  if (p_dtype == GPSA_F32)
    tri_upper_kernel<float, T><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float*)P, M, n, dst);
  else if (p_dtype == GPSA_F64)
    tri_upper_kernel<double, T><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const double*)P, M, n, dst);
  else
    return GPSA_EINVAL;
  GPSA_LAUNCH_CHECK();
  *ws = reinterpret_cast<char*>(*ws) + need;
  *ws_bytes -= need;
  *out = dst;
  return 0;
}
// [R][C] -> [R][Cp] (zero beyond C) and back: the LDS-DMA kernels of the M > 256 data GP want 16-byte aligned rows of
// the [M, C] / [L, C] panels, i.e. a column count that is a multiple of 4; any other C (S * N is whatever the data
// has) runs them on padded copies - two extra panels next to L of them
__global__ void __launch_bounds__(256) pad_cols_kernel(const float* __restrict__ src, long long R, long long C, long long Cp,
                                                       float* __restrict__ dst) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= R * Cp) return;
  const long long r = i / Cp, c = i - r * Cp;
  dst[i] = c < C ? src[r * C + c] : 0.f;
}
__global__ void __launch_bounds__(256) unpad_cols_kernel(const float* __restrict__ src, long long R, long long C, long long Cp,
                                                         float* __restrict__ dst) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= R * C) return;
  const long long r = i / C, c = i - r * C;
  dst[i] = src[r * Cp + c];
}
static inline bool big_wants_pad(int M, long long C) {
  static const bool off = [] { const char* e = getenv("GPSA_BIG_PANEL"); return e && e[0] == '0'; }();
  return !off && M > 256 && C >= 128 && (C & 3) != 0;
}
static inline long long pad4(long long C) { return (C + 3) & ~3LL; }
// carve ``floats`` (rounded to 64) off the END of a workspace; nullptr when it does not fit
static inline float* ws_tail(void* ws, long long* bytes, long long floats) {
  const long long need = ((floats + 63) & ~63LL) * 4;
  if (*bytes < need) return nullptr;
  *bytes -= need;
  *bytes &= ~255LL;
  return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + *bytes);
}

// dalpha = 2 sum_l Omega_l (g_l o alpha) through big_accum_kernel; GPSA_EUNSUPPORTED: shape / workspace not covered
static int big_accum_launch(int omega_dtype, const float* alpha, const void* Omega, const float* g, int M, long long C,
                            int L, float* dalpha, void* workspace, long long workspace_bytes, hipStream_t st) {
  if (!big_panel_ok(M, C, L, alpha) || (omega_dtype != GPSA_F64 && omega_dtype != GPSA_F32)) return GPSA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(dalpha) & 15) != 0 || (reinterpret_cast<uintptr_t>(g) & 15) != 0) return GPSA_EUNSUPPORTED;
  const int Mp = (M + 15) / 16 * 16, nrb = (int)cdiv(M, 128), ns = big_accum_nsplit(M, C, L);
  const long long n = (long long)L * M * Mp, pb = (n * 4 + 255) & ~255LL;
  if (workspace_bytes < pb + big_accum_ws_bytes(M, C, L)) return GPSA_EUNSUPPORTED;
  float* Pp = (float*)workspace;
  float* part = ns > 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pb) : dalpha;
  if (omega_dtype == GPSA_F64)
    pad_k_kernel<double><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const double*)Omega, M, Mp, L, Pp);
  else
    pad_k_kernel<float><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float*)Omega, M, Mp, L, Pp);
  GPSA_LAUNCH_CHECK();
  const long long ctiles = cdiv(C, 128), ct8 = cdiv(ctiles, 8);
  BigAccumArgs aa{Pp, alpha, g, part, M, Mp, L, nrb, ns, C, ctiles, 2.f, big_phase()};
  // (measured, no faster: 3 workgroups per CU; a 4-slot ring with three stages in flight)
  big_accum_kernel<3><<<(unsigned)(8 * nrb * ct8 * ns), 256, 0, st>>>(aa);
  GPSA_LAUNCH_CHECK();
  if (ns > 1) {
    const long long n4 = (long long)M * C / 4;
    big_accum_reduce_kernel<<<(unsigned)cdiv(n4, 256), 256, 0, st>>>(part, ns, n4, dalpha);
    GPSA_LAUNCH_CHECK();
  }
  return 0;
}
}  // namespace gpsa

extern "C" {

long long gpsa_quadform_workspace(int dtype, int M, long long C, int L) {
  const long long sz = (dtype == GPSA_F64) ? 8 : 4;
  const int MB = gpsa::mfma_mb_for(M);
  long long mfma = 0;
  if (dtype == GPSA_F32 && MB) {
    mfma = ((long long)L * MB * 16 * MB * 16 + gpsa::accum_slab_floats(MB)) * 4;
    const long long gw = gpsa::gram_mb_for(M) ? gpsa::gram_ws_bytes(MB, C, L) : 0;
    if (gw > mfma) mfma = gw;
  }
  int lc = L < 4 ? L : 4;
  long long generic = (long long)M * C * sz * lc;                         // fwd: lc slabs of T
  long long bo = ((long long)M * C * sz + (long long)gpsa::gram_splitk(C, M) * M * M * sz) * lc;  // bwd_omega
  long long r = generic > bo ? generic : bo;
  r += (long long)L * M * M * sz + 256;  // converted copy of Omega (generic paths, other precision)
  if (dtype == GPSA_F32 && M > 128) {     // big_quad / big_accum: padded fp32 operand + the splits' partial results
    const long long big = (long long)L * M * ((M + 15) / 16 * 16) * 4 + gpsa::big_accum_ws_bytes(M, C, L) + 512;
    if (big > r) r = big;
  }
  r = (r > mfma ? r : mfma) + 256;
  if (dtype == GPSA_F32 && gpsa::big_wants_pad(M, C)) {  // padded copies of alpha, g and of the result panel
    const long long Cp = gpsa::pad4(C);
    r = gpsa_quadform_workspace(dtype, M, Cp, L) + ((long long)M * Cp + (long long)L * Cp + (long long)(L > M ? L : M) * Cp) * 4 +
        4096;
  }
  return r;
}

int gpsa_quadform_fwd(int dtype, int omega_dtype, const void* alpha, const void* Omega, int M,
                      long long C, int L, void* v, void* workspace, long long workspace_bytes,
                      void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32) {
    const int MB = mfma_mb_for(M);
    if (MB && MB <= MB_MAX_QUAD && !force_generic()) {
      if (workspace_bytes < (long long)L * MB * 16 * MB * 16 * 4) return GPSA_EWORKSPACE;
      float* Ppk = (float*)workspace;
      static const bool full_env = [] { const char* e = getenv("GPSA_QUAD_FULL"); return e && e[0] == '1'; }();
      const bool full = full_env && MB <= 16;  // (the full-product kernel holds M <= 256)
      // (the last chunk goes in K-step order exactly when quad_sym_launch picks the step-skipping kernel)
      const int sym = PACK_SYM_UPPER | ((M - 16 * (MB - 1) <= 8) ? PACK_KSTEP_LAST : 0);
      const int klast = (M - 16 * (MB - 1) <= 8) ? PACK_KSTEP_LAST : 0;  // matches panel_mfma_launch's RL choice
      int rc = pack_any(omega_dtype, Omega, M, MB, L, 0, Ppk, st, full ? klast : sym);
      if (rc) return rc;
      if (!full) return quad_sym_launch(MB, Ppk, (const float*)alpha, M, C, L, (float*)v, st);
      return panel_mfma_launch<MODE_QUAD>(MB, Ppk, (const float*)alpha, nullptr, M, C, L, (float*)v,
                                          nullptr, 1.f, nullptr, st);
    }
    if (big_wants_pad(M, C)) {  // unaligned column count: the same kernels on zero-padded copies
      const long long Cp = pad4(C);
      long long rest = workspace_bytes;
      float* ap = ws_tail(workspace, &rest, (long long)M * Cp);
      float* vp = ap ? ws_tail(workspace, &rest, (long long)L * Cp) : nullptr;
      if (vp) {
        pad_cols_kernel<<<(unsigned)cdiv((long long)M * Cp, 256), 256, 0, st>>>((const float*)alpha, M, C, Cp, ap);
        GPSA_LAUNCH_CHECK();
        const int rc = gpsa_quadform_fwd(dtype, omega_dtype, ap, Omega, M, Cp, L, vp, workspace, rest, stream);
        if (rc) return rc;
        unpad_cols_kernel<<<(unsigned)cdiv((long long)L * C, 256), 256, 0, st>>>(vp, L, C, Cp, (float*)v);
        GPSA_LAUNCH_CHECK();
        return 0;
      }
    }
    if (big_panel_ok(M, C, L, alpha) && (omega_dtype == GPSA_F64 || omega_dtype == GPSA_F32)) {
      // block-triangular LDS-DMA form, closed in the kernel: nothing of size M x C per output is written
      const int Mp = (M + 15) / 16 * 16;
      const long long n = (long long)L * M * Mp;
      if (workspace_bytes >= n * 4) {
        float* Pp = (float*)workspace;
        if (omega_dtype == GPSA_F64)
          pad_k_tri_kernel<double><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const double*)Omega, M, Mp, L, Pp);
        else
          pad_k_tri_kernel<float><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float*)Omega, M, Mp, L, Pp);
        GPSA_LAUNCH_CHECK();
        BigQuadArgs qa{Pp, (const float*)alpha, (float*)v, nullptr, M, Mp, L, C, 0, big_phase()};
        const int lb = big_remap_lb();
        const long long combos = cdiv(C, 128) * cdiv(L, lb > 0 ? lb : 1);
        if (lb > 0 && 8 * lb * cdiv(combos, 8) < 0x7fffffffLL) {
          qa.lb = lb;
          big_quad_kernel<true, false, 3><<<(unsigned)(8 * lb * cdiv(combos, 8)), 256, 0, st>>>(qa);
        } else {
          big_quad_kernel<true, false, 3><<<dim3((unsigned)cdiv(C, 128), (unsigned)L), 256, 0, st>>>(qa);
        }
        GPSA_LAUNCH_CHECK();
        return 0;
      }
    }
    const float* Om;
    int rc = tri_operand<float>(omega_dtype, Omega, M, L, &Om, &workspace, &workspace_bytes, st);
    if (rc) return rc;
    return generic_quadform_fwd<float>((const float*)alpha, Om, M, C, L, (float*)v, workspace,
                                       workspace_bytes, st);
  }
  if (dtype == GPSA_F64) {
    const double* Om;
    int rc = tri_operand<double>(omega_dtype, Omega, M, L, &Om, &workspace, &workspace_bytes, st);
    if (rc) return rc;
    return generic_quadform_fwd<double>((const double*)alpha, Om, M, C, L, (double*)v, workspace,
                                        workspace_bytes, st);
  }
  return GPSA_EINVAL;
}

// M <= 256: the register-resident full-product kernel, products in its accumulator order;
// beyond: one tiled product per output into a row-major [L][M][C] buffer (the generic MFMA product)
static inline bool keep_mfma_path(int M) {
  const int MB = gpsa::mfma_mb_for(M);
  return MB && MB <= 16 && !gpsa::force_generic();
}

long long gpsa_quadform_keep_f32_workspace(int M, int L) {
  if (M < 1 || L < 1) return 0;
  if (keep_mfma_path(M)) {
    const int MB = gpsa::mfma_mb_for(M);
    // + four chunks of slack behind the packed operand (round 6): the staging ring of the panel kernels keeps
    // requesting chunks behind a workgroup's last one until its loop ends (values it never multiplies); for the
    // workgroup that owns the LAST chunk those requests used to leave the workspace - a device fault when the
    // workspace happened to end at a mapping boundary (found by tools/fuzz_kernels.py through the raw C ABI; inside the
    // step engine the bytes behind it were the arena's own).  The sibling workspaces have their slabs there.
    return (long long)L * MB * 16 * MB * 16 * 4 + 4LL * MB * 16 * 16 * 4;
  }
  return (long long)L * M * ((M + 15) / 16 * 16) * 4 + 256;  // fp32 copy of Omega, contraction index padded to 16
}

long long gpsa_quadform_keep_f32_bytes(int M, long long C, int L) {
  if (M < 1 || C < 1 || L < 1) return 0;
  // (M > 256 with a column count that is not a multiple of 4: the kept buffer's rows would not be 16-byte aligned;
  //  the caller recomputes instead - 0 = "cannot keep" - through the kernels' padded-copy path)
  if (!keep_mfma_path(M) && gpsa::big_wants_pad(M, C)) return 0;
  if (keep_mfma_path(M)) {
    const int MB = gpsa::mfma_mb_for(M);
    const long long wgcols = 64LL * gpsa::panel_nct_for(MB);
    return (long long)L * cdiv(C, wgcols) * wgcols * MB * 16 * 4;
  }
  return (long long)L * M * C * 4;
}

int gpsa_quadform_fwd_keep_f32(int omega_dtype, const float* alpha, const void* Omega, int M, long long C, int L,
                               float* v, float* W, void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1 || !alpha || !Omega || !v || !W) return GPSA_EINVAL;
  if (workspace_bytes < gpsa_quadform_keep_f32_workspace(M, L)) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  if (keep_mfma_path(M)) {
    const int MB = mfma_mb_for(M);
    float* Ppk = (float*)workspace;
    const int klast = (M - 16 * (MB - 1) <= 8) ? PACK_KSTEP_LAST : 0;  // matches panel_mfma_launch's RL choice
    int rc = pack_any(omega_dtype, Omega, M, MB, L, 0, Ppk, st, klast);
    if (rc) return rc;
    return panel_mfma_launch<MODE_QUAD>(MB, Ppk, alpha, nullptr, M, C, L, v, nullptr, 1.f, nullptr, st, W);
  }
  if (C > 0x7fffffffLL) return GPSA_EINVAL;
  {
    // the LDS-DMA product (16-byte aligned rows of alpha; zero-padded fp32 copy of Omega in the workspace)
    static const bool off = [] { const char* e = getenv("GPSA_PROD_BIG"); return e && e[0] == '0'; }();
    const int Mp = (M + 15) / 16 * 16;
    const long long need = (long long)L * M * Mp * 4;
    const long long ctiles = cdiv(C, 128);
    if (!off && (C & 3) == 0 && C >= 16 && (reinterpret_cast<uintptr_t>(alpha) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(W) & 15) == 0 && need <= workspace_bytes && L <= 65535 && ctiles <= 65535) {
      float* Pp = (float*)workspace;
      const long long n = (long long)L * M * Mp;
      if (omega_dtype == GPSA_F64)
        pad_k_kernel<double><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const double*)Omega, M, Mp, L, Pp);
      else if (omega_dtype == GPSA_F32)
        pad_k_kernel<float><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float*)Omega, M, Mp, L, Pp);
      else
        return GPSA_EINVAL;
      GPSA_LAUNCH_CHECK();
      static const bool old_pb = [] { const char* e = getenv("GPSA_PROD_BIG"); return e && e[0] == '1'; }();
      if (!old_pb && big_panel_ok(M, C, L, alpha)) {  // product, kept copy and the closing column sums in one kernel
        BigQuadArgs qa{Pp, alpha, v, W, M, Mp, L, C, 0, big_phase()};
        const int lb = big_remap_lb();
        const long long combos = ctiles * cdiv(L, lb > 0 ? lb : 1);
        if (lb > 0 && 8 * lb * cdiv(combos, 8) < 0x7fffffffLL) {
          qa.lb = lb;
          big_quad_kernel<false, true, 3><<<(unsigned)(8 * lb * cdiv(combos, 8)), 256, 0, st>>>(qa);
        } else {
          big_quad_kernel<false, true, 3><<<dim3((unsigned)ctiles, (unsigned)L), 256, 0, st>>>(qa);
        }
        GPSA_LAUNCH_CHECK();
        return 0;
      }
      ProdBigArgs pa{Pp, alpha, W, M, Mp, L, C};
      prod_big_kernel<<<dim3((unsigned)cdiv(M, 128), (unsigned)L, (unsigned)ctiles), 256, 0, st>>>(pa);
      GPSA_LAUNCH_CHECK();
      for (int l0 = 0; l0 < L; l0 += 16384) {
        const int nb = (L - l0 < 16384) ? L - l0 : 16384;
        coldot_kernel<float><<<dim3((unsigned)cdiv(C, 64), (unsigned)nb), 256, 0, st>>>(
            alpha, W + (long long)l0 * M * C, M, C, v + (long long)l0 * C, C);
        GPSA_LAUNCH_CHECK();
      }
      return 0;
    }
  }
  const float* Om;
  int rc = operand_as<float>(omega_dtype, Omega, (long long)L * M * M, &Om, &workspace, &workspace_bytes, st);
  if (rc) return rc;
  // W[l] = Omega[l] alpha: batches of outputs (grid.z <= 65535)
  for (int l0 = 0; l0 < L; l0 += 16384) {
    const int nb = (L - l0 < 16384) ? L - l0 : 16384;
    rc = gemm_launch<float>(0, 0, M, (int)C, M, 1.0, Om + (long long)l0 * M * M, M, (long long)M * M, alpha, C, 0, 0.0,
                            W + (long long)l0 * M * C, C, (long long)M * C, nb, 1, nullptr, 0, st);
    if (rc) return rc;
    coldot_kernel<float><<<dim3((unsigned)cdiv(C, 64), (unsigned)nb), 256, 0, st>>>(
        alpha, W + (long long)l0 * M * C, M, C, v + (long long)l0 * C, C);
    GPSA_LAUNCH_CHECK();
  }
  return 0;
}

/* ---- forward + likelihood + abar in one pass (panel_elbo_kernel) ------------------------------------------ */
// MB = 13: one wave per SIMD with 32 columns (NCT = 2), or - GPSA_ELBO_NCT=1 - two workgroups per CU with 16 columns per
// wave and half the register file each (the other wave's MFMAs fill the matrix pipe during a wave's closing and ring issues)
static inline int elbo_nct_for(int MB) {
  static const int nct13 = [] { const char* e = getenv("GPSA_ELBO_NCT"); return (e && e[0] == '1') ? 1 : 2; }();
  return MB == 13 ? nct13 : (MB == 16 ? 2 : 4);  // (16 row tiles, M <= 256: 3 x 128 registers + the working set)
}
static inline bool elbo_path(int M) {
  const int MB = gpsa::mfma_mb_for(M);
  return MB && MB <= 16 && !gpsa::force_generic();
}

int gpsa_quadform_elbo_parts(void) { return gpsa::num_cus() * 2; }

long long gpsa_quadform_elbo_f32_workspace(int M, long long C, int L) {
  if (M < 1 || C < 1 || L < 1 || !elbo_path(M) || C > GPSA_PANEL_MAX_C) return 0;
  const int MB = gpsa::mfma_mb_for(M), nct = elbo_nct_for(MB);
  const long long G = gpsa_quadform_elbo_parts();
  return ((long long)L * MB * 16 * MB * 16 + G * 2 * (long long)MB * 16 * 64 * nct) * 4;
}

// the mean can ride in the product's first padding row (gpsa_quadform_elbo_delta_f32) when row M lies in the LAST row tile
static inline bool elbo_delta_ok(int M) {
  const int MB = gpsa::mfma_mb_for(M);
  static const bool off = [] { const char* e = getenv("GPSA_ELBO_DELTA"); return e && e[0] == '0'; }();
  return !off && elbo_path(M) && M > 16 * (MB - 1) && M < 16 * MB;
}

int gpsa_quadform_elbo_takes_delta(int M) { return M >= 1 && elbo_delta_ok(M) ? 1 : 0; }

static int elbo_launch(int omega_dtype, const float* alpha, const void* Omega, int M, long long C, int L,
                       const float* meanT, const float* delta, const double* q, const float* var_u, const float* eps,
                       const float* Y, long long N, int S, const float* noise_u, float* g, float* dmeanT, float* abar,
                       double* part, float* FT, void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1 || N < 1 || S < 1 || !alpha || !Omega || (!meanT && !delta) || !q || !var_u || !eps || !Y ||
      !noise_u || !g || !dmeanT || !abar || !part)
    return GPSA_EINVAL;
  if (delta != nullptr && !elbo_delta_ok(M)) return GPSA_EUNSUPPORTED;
  if (!elbo_path(M) || C > GPSA_PANEL_MAX_C) return GPSA_EUNSUPPORTED;
  if (workspace_bytes < gpsa_quadform_elbo_f32_workspace(M, C, L)) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const int MB = mfma_mb_for(M);
  float* Ppk = (float*)workspace;
  float* slab = Ppk + (long long)L * MB * 16 * MB * 16;
  const int klast = (M - 16 * (MB - 1) <= 8) ? PACK_KSTEP_LAST : 0;
  int rc = pack_any(omega_dtype, Omega, M, MB, L, 0, Ppk, st, klast, delta);
  if (rc) return rc;
  if (delta != nullptr) meanT = nullptr;  // the kernel reads the mean from row M of its own product
  const int gmax = gpsa_quadform_elbo_parts();
  ElboArgs a{Ppk, alpha, M, C, L, meanT, q, var_u, eps, Y, noise_u, N, S, g, dmeanT, FT, abar, slab, part, gmax};
  long long grid = 0;
#define GPSA_ELBO_CASE(MBV, NCTV)                                                                       \
  case MBV * 8 + NCTV: {                                                                                        \
    const long long ntiles = cdiv(C, 64 * NCTV), T = ntiles * L;                                        \
    grid = (long long)num_cus() * ((MBV * NCTV >= 14) ? 1 : 2);                                         \
    if (grid > T) grid = T;                                                                             \
    constexpr bool HEAD = MBV == 13 && NCTV == 2;  /* M > 16 (MB - 1): the instantiation without row clamps */ \
    static const bool pair = [] { const char* e = getenv("GPSA_ELBO_PAIR"); return !(e && e[0] == '0'); }();  \
    if (HEAD && pair && M > 16 * (MBV - 1) && M - 16 * (MBV - 1) <= 8)                                  \
      panel_elbo_kernel<MBV, NCTV, 2, HEAD, HEAD><<<(unsigned)grid, 256, 0, st>>>(a);                   \
    else if (HEAD && pair && M > 16 * (MBV - 1))                                                        \
      panel_elbo_kernel<MBV, NCTV, 4, HEAD, HEAD><<<(unsigned)grid, 256, 0, st>>>(a);                   \
    else if (HEAD && M > 16 * (MBV - 1) && M - 16 * (MBV - 1) <= 8)                                     \
      panel_elbo_kernel<MBV, NCTV, 2, HEAD><<<(unsigned)grid, 256, 0, st>>>(a);                         \
    else if (HEAD && M > 16 * (MBV - 1))                                                                \
      panel_elbo_kernel<MBV, NCTV, 4, HEAD><<<(unsigned)grid, 256, 0, st>>>(a);                         \
    else if (M - 16 * (MBV - 1) <= 8)                                                                   \
      panel_elbo_kernel<MBV, NCTV, 2><<<(unsigned)grid, 256, 0, st>>>(a);                               \
    else                                                                                                \
      panel_elbo_kernel<MBV, NCTV, 4><<<(unsigned)grid, 256, 0, st>>>(a);                               \
    dim3 rg((unsigned)ntiles, ntiles >= 512 ? 8 : (ntiles >= 128 ? 16 : 32));                           \
    panel_slab_reduce_kernel<<<rg, 256, 0, st>>>(slab, M, MBV * 16, 64 * NCTV, C, L, ntiles, (int)grid, abar); \
  } break;
  switch (MB * 8 + elbo_nct_for(MB)) {
    GPSA_ELBO_CASE(2, 4)
    GPSA_ELBO_CASE(4, 4)
    GPSA_ELBO_CASE(7, 4)
    GPSA_ELBO_CASE(13, 2)
    GPSA_ELBO_CASE(13, 1)
    GPSA_ELBO_CASE(16, 2)
    default:
      return GPSA_EUNSUPPORTED;
  }
#undef GPSA_ELBO_CASE
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_quadform_elbo_f32(int omega_dtype, const float* alpha, const void* Omega, int M, long long C, int L,
                           const float* meanT, const double* q, const float* var_u, const float* eps, const float* Y,
                           long long N, int S, const float* noise_u, float* g, float* dmeanT, float* abar, double* part,
                           float* FT, void* workspace, long long workspace_bytes, void* stream) {
  if (!meanT) return GPSA_EINVAL;
  return elbo_launch(omega_dtype, alpha, Omega, M, C, L, meanT, nullptr, q, var_u, eps, Y, N, S, noise_u, g, dmeanT, abar,
                     part, FT, workspace, workspace_bytes, stream);
}

int gpsa_quadform_elbo_delta_f32(int omega_dtype, const float* alpha, const void* Omega, int M, long long C, int L,
                                 const float* delta, const double* q, const float* var_u, const float* eps,
                                 const float* Y, long long N, int S, const float* noise_u, float* g, float* dmeanT,
                                 float* abar, double* part, float* FT, void* workspace, long long workspace_bytes,
                                 void* stream) {
  if (!delta) return GPSA_EINVAL;
  return elbo_launch(omega_dtype, alpha, Omega, M, C, L, nullptr, delta, q, var_u, eps, Y, N, S, noise_u, g, dmeanT, abar,
                     part, FT, workspace, workspace_bytes, stream);
}

int gpsa_quadform_bwd_alpha_kept_f32(const float* W, const float* g, int M, long long C, int L, const float* dcT,
                                     const float* dmeanT, float* dalpha, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1 || !W || !g || !dalpha) return GPSA_EINVAL;
  if ((dcT == nullptr) != (dmeanT == nullptr)) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (!keep_mfma_path(M)) {  // row-major products
    constexpr int RB = 8;
    col_wsum_rows_kernel<RB><<<dim3((unsigned)cdiv(C, 1024), (unsigned)cdiv(M, RB)), 256, 0, st>>>(
        W, g, M, C, L, dcT, dmeanT, dalpha);
    GPSA_LAUNCH_CHECK();
    return 0;
  }
  const int MB = mfma_mb_for(M);
#define GPSA_KEPT_CASE(MBV, NCTV)                                                                        \
  case MBV: {                                                                                            \
    const long long ntiles = cdiv(C, 64 * NCTV);                                                         \
    constexpr int RTB = 4;                                                                                \
    kept_wsum_kernel<MBV, NCTV, RTB><<<dim3((unsigned)ntiles, NCTV, (MBV + RTB - 1) / RTB), 256, 0, st>>>( \
        W, g, M, C, L, ntiles, 2.f, dcT, dmeanT, dalpha);                                                \
    break;                                                                                               \
  }
  switch (MB) {
    GPSA_KEPT_CASE(2, 4)
    GPSA_KEPT_CASE(4, 4)
    GPSA_KEPT_CASE(7, 4)
    GPSA_KEPT_CASE(13, 3)
    GPSA_KEPT_CASE(16, 2)
    default:
      return GPSA_EUNSUPPORTED;
  }
#undef GPSA_KEPT_CASE
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_quadform_bwd_alpha(int dtype, int omega_dtype, const void* alpha, const void* Omega,
                            const void* g, int M, long long C, int L, void* dalpha, void* workspace,
                            long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32) {
    const int MB = mfma_mb_for(M);
    if (big_wants_pad(M, C) && !(MB && MB <= MB_MAX_ACCUM && !force_generic())) {
      const long long Cp = pad4(C);
      long long rest = workspace_bytes;
      float* ap = ws_tail(workspace, &rest, (long long)M * Cp);
      float* gp = ap ? ws_tail(workspace, &rest, (long long)L * Cp) : nullptr;
      float* dp = gp ? ws_tail(workspace, &rest, (long long)M * Cp) : nullptr;
      if (dp) {
        pad_cols_kernel<<<(unsigned)cdiv((long long)M * Cp, 256), 256, 0, st>>>((const float*)alpha, M, C, Cp, ap);
        pad_cols_kernel<<<(unsigned)cdiv((long long)L * Cp, 256), 256, 0, st>>>((const float*)g, L, C, Cp, gp);
        GPSA_LAUNCH_CHECK();
        const int rc = gpsa_quadform_bwd_alpha(dtype, omega_dtype, ap, Omega, gp, M, Cp, L, dp, workspace, rest, stream);
        if (rc) return rc;
        unpad_cols_kernel<<<(unsigned)cdiv((long long)M * C, 256), 256, 0, st>>>(dp, M, C, Cp, (float*)dalpha);
        GPSA_LAUNCH_CHECK();
        return 0;
      }
    }
    // 256 < M <= 512: the register-resident kernel stays ahead of the LDS-DMA one (BASELINE config 4: 319 vs 342 ms);
    // GPSA_ACCUM_PANEL=0 takes the LDS-DMA kernel there too (tests, A/B)
    static const bool panel_off = [] { const char* e = getenv("GPSA_ACCUM_PANEL"); return e && e[0] == '0'; }();
    if (MB > 16 && panel_off && !force_generic()) {
      const int rc = big_accum_launch(omega_dtype, (const float*)alpha, Omega, (const float*)g, M, C, L, (float*)dalpha,
                                      workspace, workspace_bytes, st);
      if (rc != GPSA_EUNSUPPORTED) return rc;
    }
    if (MB && MB <= MB_MAX_ACCUM && !force_generic()) {
      const long long pk = (long long)L * MB * 16 * MB * 16;
      if (workspace_bytes < (pk + accum_slab_floats(MB)) * 4) return GPSA_EWORKSPACE;
      float* Ppk = (float*)workspace;
      int rc = pack_any(omega_dtype, Omega, M, MB, L, 0, Ppk, st, PACK_KSTEP);
      if (rc) return rc;
      return panel_mfma_launch<MODE_ACCUM>(MB, Ppk, (const float*)alpha, (const float*)g, M, C, L,
                                           (float*)dalpha, nullptr, 2.f, Ppk + pk, st);
    }
    {
      const int rc = big_accum_launch(omega_dtype, (const float*)alpha, Omega, (const float*)g, M, C, L, (float*)dalpha,
                                      workspace, workspace_bytes, st);
      if (rc != GPSA_EUNSUPPORTED) return rc;
    }
    const float* Om;
    int rc = operand_as<float>(omega_dtype, Omega, (long long)L * M * M, &Om, &workspace, &workspace_bytes, st);
    if (rc) return rc;
    return generic_quadform_bwd_alpha<float>((const float*)alpha, Om, (const float*)g, M, C, L,
                                             (float*)dalpha, workspace, workspace_bytes, st);
  }
  if (dtype == GPSA_F64) {
    const double* Om;
    int rc = operand_as<double>(omega_dtype, Omega, (long long)L * M * M, &Om, &workspace, &workspace_bytes, st);
    if (rc) return rc;
    return generic_quadform_bwd_alpha<double>((const double*)alpha, Om, (const double*)g, M, C, L,
                                              (double*)dalpha, workspace, workspace_bytes, st);
  }
  return GPSA_EINVAL;
}

int gpsa_quadform_fwd_keep(int dtype, const void* alpha, const void* Omega, int M, long long C, int L,
                           void* v, void* W, const void* dcT, void* meanT, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1 || C > 0x7fffffffLL) return GPSA_EINVAL;
  if ((dcT == nullptr) != (meanT == nullptr)) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32)
    return quadform_fwd_keep<float>((const float*)alpha, (const float*)Omega, M, C, L, (float*)v,
                                    (float*)W, (const float*)dcT, (float*)meanT, st);
  if (dtype == GPSA_F64)
    return quadform_fwd_keep<double>((const double*)alpha, (const double*)Omega, M, C, L, (double*)v,
                                     (double*)W, (const double*)dcT, (double*)meanT, st);
  return GPSA_EINVAL;
}

int gpsa_quadform_bwd_alpha_kept(int dtype, const void* W, const void* g, int M, long long C, int L,
                                 const void* dcT, const void* dmeanT, void* dalpha, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1) return GPSA_EINVAL;
  if ((dcT == nullptr) != (dmeanT == nullptr)) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32)
    return quadform_bwd_alpha_kept<float>((const float*)W, (const float*)g, M, C, L, (const float*)dcT,
                                          (const float*)dmeanT, (float*)dalpha, st);
  if (dtype == GPSA_F64)
    return quadform_bwd_alpha_kept<double>((const double*)W, (const double*)g, M, C, L, (const double*)dcT,
                                           (const double*)dmeanT, (double*)dalpha, st);
  return GPSA_EINVAL;
}

int gpsa_quadform_bwd_omega(int dtype, int out_dtype, const void* alpha, const void* g, int M, long long C,
                            int L, void* dOmega, void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1) return GPSA_EINVAL;
  if (out_dtype != GPSA_F32 && out_dtype != GPSA_F64) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32) {
    const int MB = gram_mb_for(M);
    if (MB && !force_generic()) {
      if (workspace_bytes < gram_ws_bytes(MB, C, L)) return GPSA_EWORKSPACE;
      return gram_mfma_launch(MB, (const float*)alpha, (const float*)g, M, C, L, dOmega, out_dtype,
                              (float*)workspace, st);
    }
    if (out_dtype != dtype) return GPSA_EUNSUPPORTED;
    if (big_wants_pad(M, C)) {  // unaligned column count: the LDS-DMA Gram kernel on zero-padded copies
      const long long Cp = pad4(C);
      long long rest = workspace_bytes;
      float* ap = ws_tail(workspace, &rest, (long long)M * Cp);
      float* gp = ap ? ws_tail(workspace, &rest, (long long)L * Cp) : nullptr;
      if (gp) {
        pad_cols_kernel<<<(unsigned)cdiv((long long)M * Cp, 256), 256, 0, st>>>((const float*)alpha, M, C, Cp, ap);
        pad_cols_kernel<<<(unsigned)cdiv((long long)L * Cp, 256), 256, 0, st>>>((const float*)g, L, C, Cp, gp);
        GPSA_LAUNCH_CHECK();
        return generic_quadform_bwd_omega<float>(ap, gp, M, Cp, L, (float*)dOmega, workspace, rest, st);
      }
    }
    return generic_quadform_bwd_omega<float>((const float*)alpha, (const float*)g, M, C, L,
                                             (float*)dOmega, workspace, workspace_bytes, st);
  }
  if (dtype == GPSA_F64) {
    if (out_dtype != dtype) return GPSA_EUNSUPPORTED;
    return generic_quadform_bwd_omega<double>((const double*)alpha, (const double*)g, M, C, L,
                                              (double*)dOmega, workspace, workspace_bytes, st);
  }
  return GPSA_EINVAL;
}

// the d-delta option of the Gram kernel: M in the last row tile with a padding row behind it, the MFMA path, 4-column alignment
static inline bool gram_delta_ok(int M, long long C) {
  static const bool off = [] { const char* e = getenv("GPSA_GRAM_DELTA"); return e && e[0] == '0'; }();
  const int MB = gpsa::gram_mb_for(M);
  return !off && MB != 0 && !gpsa::force_generic() && M > 16 * (MB - 1) && M < 16 * MB && (C % 4 == 0) && C >= 8;
}

int gpsa_quadform_bwd_omega_takes_delta(int M, long long C) { return M >= 1 && C >= 1 && gram_delta_ok(M, C) ? 1 : 0; }

int gpsa_quadform_bwd_omega_delta_f32(int out_dtype, const float* alpha, const float* g, const float* dmeanT, int M,
                                      long long C, int L, void* dOmega, float* ddelta, double dbeta, void* workspace,
                                      long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1 || !alpha || !g || !dmeanT || !dOmega || !ddelta) return GPSA_EINVAL;
  if (out_dtype != GPSA_F32 && out_dtype != GPSA_F64) return GPSA_EINVAL;
  if (!gram_delta_ok(M, C)) return GPSA_EUNSUPPORTED;
  const int MB = gram_mb_for(M);
  if (workspace_bytes < gram_ws_bytes(MB, C, L)) return GPSA_EWORKSPACE;
  return gram_mfma_launch(MB, alpha, g, M, C, L, dOmega, out_dtype, (float*)workspace, as_stream(stream), dmeanT, ddelta,
                          (float)dbeta);
}

/* batched fp64 forms of the two calls above and of gpsa_col_axpy: ``batch`` layers with contiguous operands
 * (see quadform_fwd_keep) - the warp GPs of several views in one launch sequence. */
int gpsa_quadform_fwd_keep_batched_f64(const double* alpha, const double* Omega, int M, long long C, int L,
                                       double* v, double* W, const double* dcT, double* meanT, int batch,
                                       void* stream) {
  if (M < 1 || C < 1 || L < 1 || batch < 1 || C > 0x7fffffffLL || dcT == nullptr || meanT == nullptr)
    return GPSA_EINVAL;
  return gpsa::quadform_fwd_keep<double>(alpha, Omega, M, C, L, v, W, dcT, meanT, as_stream(stream), batch);
}

int gpsa_quadform_bwd_alpha_kept_batched_f64(const double* W, const double* g, int M, long long C, int L,
                                             const double* dcT, const double* dmeanT, double* dalpha,
                                             int batch, void* stream) {
  if (M < 1 || C < 1 || L < 1 || batch < 1) return GPSA_EINVAL;
  if ((dcT == nullptr) != (dmeanT == nullptr)) return GPSA_EINVAL;
  return gpsa::quadform_bwd_alpha_kept<double>(W, g, M, C, L, dcT, dmeanT, dalpha, as_stream(stream), batch);
}

int gpsa_col_axpy_batched_f64(const double* Y, const double* X, const double* d, double s, int M, long long C,
                              double* out, int batch, void* stream) {
  if (M < 1 || C < 1 || batch < 1) return GPSA_EINVAL;
  dim3 grid((unsigned)cdiv(C, 256), (unsigned)((M < 64) ? M : 64), (unsigned)batch);
  gpsa::col_axpy_kernel<double><<<grid, 256, 0, as_stream(stream)>>>(Y, X, d, s, M, C, out);
  GPSA_LAUNCH_CHECK();
  return 0;
}

/* dOmega[b][l] = sum_c g[b][l,c] alpha[b][:,c] alpha[b][:,c]^T for ``batch`` fp64 layers with contiguous
 * operands (alpha [batch][M,C], g [batch][L,C], dOmega [batch][L,M,M]): one scaling launch and ONE
 * strided-batched split-K product with the L outputs of a layer stacked as rows.
 * workspace >= gpsa_gram_batched_workspace(M, C, L, batch). */
long long gpsa_gram_batched_workspace(int M, long long C, int L, int batch) {
  const int sk = gpsa::gram_splitk(C, M);
  return ((long long)L * M * C + (sk > 1 ? (long long)sk * L * M * M : 0)) * 8LL * batch + 256;
}

int gpsa_gram_batched_f64(const double* alpha, const double* g, int M, long long C, int L, double* dOmega,
                          int batch, void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1 || batch < 1) return GPSA_EINVAL;
  if (workspace_bytes < gpsa_gram_batched_workspace(M, C, L, batch)) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const int sk = gram_splitk(C, M);
  double* tmp = (double*)workspace;
  double* part = tmp + (long long)batch * L * M * C;
  dim3 grid((unsigned)cdiv(C, 256), (unsigned)((M < 64) ? M : 64), (unsigned)(batch * L));
  colscale_batched_kernel<double><<<grid, 256, 0, st>>>(alpha, g, M, C, tmp, L);
  GPSA_LAUNCH_CHECK();
  return gemm_launch<double>(0, 1, L * M, M, C, 1.0, tmp, C, (long long)L * M * C, alpha, C, (long long)M * C,
                             0.0, dOmega, M, (long long)L * M * M, batch, sk, part,
                             (long long)batch * sk * L * M * M * 8, st);
}

int gpsa_col_axpy(int dtype, const void* Y, const void* X, const void* d, double s, int M,
                  long long C, void* out, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)cdiv(C, 256), (unsigned)((M < 64) ? M : 64));
  if (dtype == GPSA_F32)
    col_axpy_kernel<float><<<grid, 256, 0, st>>>((const float*)Y, (const float*)X, (const float*)d,
                                                 (float)s, M, C, (float*)out);
  else if (dtype == GPSA_F64)
    col_axpy_kernel<double><<<grid, 256, 0, st>>>((const double*)Y, (const double*)X,
                                                  (const double*)d, s, M, C, (double*)out);
  else
    return GPSA_EINVAL;
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_panel_mm(int dtype, int p_dtype, int transP, const void* P, const void* X, int M, long long C,
                  void* Y, void* colsq, void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  const int tp = transP ? 1 : 0;
  if (dtype == GPSA_F32) {
    const int MB = mfma_mb_for(M);
    if (MB && MB <= MB_MAX_STORE && !force_generic()) {
      if (workspace_bytes < (long long)MB * 16 * MB * 16 * 4) return GPSA_EWORKSPACE;
      float* Ppk = (float*)workspace;
      int rc = pack_any(p_dtype, P, M, MB, 1, tp, Ppk, st, PACK_KSTEP);
      if (rc) return rc;
      return panel_mfma_launch<MODE_STORE>(MB, Ppk, (const float*)X, nullptr, M, C, 1, (float*)Y,
                                           (float*)colsq, 1.f, nullptr, st);
    }
    const float* Pc;
    int rc = operand_as<float>(p_dtype, P, (long long)M * M, &Pc, &workspace, &workspace_bytes, st);
    if (rc) return rc;
    rc = gemm_launch<float>(tp, 0, M, (int)C, M, 1.0, Pc, M, 0, (const float*)X, C, 0, 0.0, (float*)Y,
                            C, 0, 1, 1, nullptr, 0, st);
    if (rc) return rc;
    if (colsq) {
      colsq_kernel<float><<<(unsigned)cdiv(C, 64), 256, 0, st>>>((const float*)Y, M, C, (float*)colsq);
      GPSA_LAUNCH_CHECK();
    }
    return 0;
  }
  if (dtype == GPSA_F64) {
    const double* Pc;
    int rc = operand_as<double>(p_dtype, P, (long long)M * M, &Pc, &workspace, &workspace_bytes, st);
    if (rc) return rc;
    rc = gemm_launch<double>(tp, 0, M, (int)C, M, 1.0, Pc, M, 0, (const double*)X, C, 0, 0.0,
                             (double*)Y, C, 0, 1, 1, nullptr, 0, st);
    if (rc) return rc;
    if (colsq) {
      colsq_kernel<double><<<(unsigned)cdiv(C, 64), 256, 0, st>>>((const double*)Y, M, C, (double*)colsq);
      GPSA_LAUNCH_CHECK();
    }
    return 0;
  }
  return GPSA_EINVAL;
}

}  // extern "C"
