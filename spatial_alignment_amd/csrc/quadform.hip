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

#include "common.hpp"

namespace gpsa {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

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

// ------------------------------------------------------------------------------------------------
// Gram sums beyond the register-resident kernel (M > 256: BASELINE configs 4 / 5), fp32 matrix cores:
//     P[l][i][j] = sum_c g[l,c] alpha[i,c] alpha[j,c]       for the 128 x 128 blocks touching the lower triangle
// One workgroup = one block pair (bi >= bj), one output l, one slice of the columns.  Both operands are rows of
// alpha, contiguous along the contracted index c: a chunk of 16 columns of the 128 + 128 rows moves to LDS by
// LDS-DMA as sixteen 1-KiB pieces in MFMA-fragment order (lane j + 16 kq holds alpha[16 p + j][c0 + 4 kq .. +3]),
// so a fragment read is one conflict-free ds_read_b128 and nothing is staged through registers or transposed
// through ds_write (the generic tiled product spends 45 % of its LDS cycles in bank conflicts on exactly that,
// and ran this shape at 0.35 matrix-pipe utilisation).  g scales the left fragment as it is read (no [M, C]
// scaled copy of alpha per output).  Three-slot ring, two stages in flight, one barrier per 64 MFMAs per wave.
// The forward declarations of glds16 / lds_addr / GPSA_DMA_* are below (panel kernels); this kernel is
// instantiated after them.
struct GramBigArgs {
  const float* alpha;  // [M][C]
  const float* g;      // [L][Cpad], zero beyond C (Cpad = a multiple of 16)
  float* part;         // [L][nsplit][M][M]   (lower blocks written)
  int M, L, nsplit, nblk;
  long long C, Cpad;
  int lb;  // > 0: 1-D grid, workgroups that share an XCD (ids equal mod 8) come in runs of ``lb`` outputs of ONE
           // (block pair, column split): they read the same rows of alpha at about the same time, from that XCD's L2
};
__global__ void pad_rows_kernel(const float* __restrict__ g, int L, long long C, long long Cpad,
                                float* __restrict__ gpad);
__global__ void gram_big_kernel(GramBigArgs a);
// W[l] = P[l] X for large M (see prod_big_kernel)
struct ProdBigArgs {
  const float* P;  // [L][M][Mp], zero for k >= M (Mp = a multiple of 16)
  const float* X;  // [M][C]
  float* W;        // [L][M][C]
  int M, Mp, L;
  long long C;
};
__global__ void prod_big_kernel(ProdBigArgs a);
template <typename TO>
__global__ void gram_big_reduce_kernel(const float* __restrict__ part, int M, int nsplit, TO* __restrict__ out);
// outputs per run of same-XCD workgroups in the large-M kernels (GPSA_BIG_LB; 0 = the plain 3-D / 2-D grids)
static inline int big_remap_lb() {
  static const int v = [] { const char* e = getenv("GPSA_BIG_LB"); return e ? atoi(e) : 16; }();
  return v;
}
static inline bool gram_big_off() {
  static const bool v = [] { const char* e = getenv("GPSA_GRAM_BIG"); return e && e[0] == '0'; }();
  return v;
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
      GramBigArgs a{alpha, gpad, reinterpret_cast<float*>(ws), M, L, (int)ns, nblk, C, Cpad, 0};
      const int lb = big_remap_lb();
      const long long combos = (long long)pairs * ns * cdiv(L, lb > 0 ? lb : 1);
      if (lb > 0 && 8 * lb * cdiv(combos, 8) < 0x7fffffffLL) {
        a.lb = lb;
        gram_big_kernel<<<(unsigned)(8 * lb * cdiv(combos, 8)), 256, 0, st>>>(a);
      } else {
        gram_big_kernel<<<dim3((unsigned)pairs, (unsigned)ns, (unsigned)L), 256, 0, st>>>(a);
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
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MODE_QUAD = 0, MODE_ACCUM = 1, MODE_STORE = 2 };

// src [L][M][M] (row-major) -> dst fp32, zero padded, in MFMA-fragment order:
//   dst[l][kc][rt][kq][j][r] = P_l[16 rt + j][16 kc + 4 kq + r]      (PACK_KSTEP: ... + 4 r + kq)
// so that K chunk kc of matrix l is one contiguous MP*64-byte block made of MB 1-KiB pieces, and
// piece rt, copied lane-linearly into LDS (global_load_lds, lane = j + 16 kq), is read back as the
// A fragment of row tile rt by one conflict-free ds_read_b128 at lane*16 bytes.
// PACK_KSTEP orders the 16 K values of a chunk so that MFMA step r contracts k = 4 r .. 4 r + 3 (instead
// of r, r+4, r+8, r+12): with M % 16 != 0 the trailing steps of the last chunk are then all padding and
// the ACCUM / STORE kernels skip them.  (QUAD keeps the interleaved order: there the B slab doubles as
// the C-layout operand that closes the form in registers.)
// PACK_SYM_UPPER: the symmetric quadratic form's operand (tiles kc >= rt only, off-diagonal ones doubled);
// PACK_KSTEP_LAST: K-step order for the last chunk only.
enum { PACK_SYM_UPPER = 1, PACK_KSTEP = 2, PACK_KSTEP_LAST = 4 };

template <typename TS>
__global__ void pack_panels_kernel(const TS* __restrict__ src, int M, int MB, int L, int transpose,
                                   float* __restrict__ dst, int layout) {
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
  }
  // symmetric quadratic form: only tiles kc >= rt are used; off-diagonal ones count twice
  if (layout & PACK_SYM_UPPER) v = (kc > rt) ? 2.f * v : (rt == kc ? v : 0.f);
  dst[idx] = v;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// 16-byte-per-lane LDS-DMA: lane i copies 16 B from its own global address to LDS byte address
// lds_base + 16 i (lds_base wave-uniform).  Issued from inline asm on purpose: hipcc then neither
// counts it in its vmcnt bookkeeping nor orders later ds_reads of the OTHER buffer behind it (with the
// builtin it drains vmcnt(0) before every fragment read, serialising the prefetch).  The issuing
// code waits with GPSA_DMA_DRAIN() before the barrier that publishes the buffer.
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
#define GPSA_DMA_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// wait until at most N of this wave's vector-memory operations are outstanding (N = the LDS-DMA
// operations of the newest stage: everything older, i.e. the stage about to be read, has landed)
#define GPSA_DMA_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(lds_ptr_t)(p);
}

__global__ void __launch_bounds__(256, 2) gram_big_kernel(GramBigArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[3][16 * 256];
  __shared__ __attribute__((aligned(16))) float sg[3][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4, wr = w >> 1, wc = w & 1;
  // workgroup -> (block pair, column split, output)
  int t, l, sp;
  if (a.lb > 0) {
    const long long id = blockIdx.x, slot = id >> 3;
    const int npair = a.nblk * (a.nblk + 1) / 2, units = npair * a.nsplit, nlb = (a.L + a.lb - 1) / a.lb;
    const long long q = (slot / a.lb) * 8 + (id & 7);  // (unit, block of outputs), dealt round-robin to the XCDs
    if (q >= (long long)units * nlb) return;
    const int unit = (int)(q % units);
    l = (int)(q / units) * a.lb + (int)(slot % a.lb);
    if (l >= a.L) return;
    t = unit % npair;
    sp = unit / npair;
  } else {
    t = blockIdx.x;
    l = blockIdx.z;
    sp = blockIdx.y;
  }
  // block pair t -> (bi, bj), bj <= bi, row-major over the lower triangle
  int bi = 0;
  while (t > bi) {
    t -= bi + 1;
    ++bi;
  }
  const int bj = t;
  const int M = a.M;
  const long long C = a.C;
  const long long nch = (C + 15) / 16;
  const long long ch0 = (long long)sp * nch / a.nsplit, ch1 = (long long)(sp + 1) * nch / a.nsplit;
  const float* gl = a.g + (long long)l * a.Cpad;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // stage chunk CH into ring slot BUF: wave w moves pieces w, w+4, w+8, w+12 (0..7: rows of block bi, 8..15: bj);
  // rows >= M are clamped (they only feed outputs >= M, never stored), columns beyond C to the last aligned group
  // (they meet g == 0: g is zero-padded to whole chunks).  The chunk's 16 values of g ride along as a fifth
  // operation of every wave (all four write the same 64 bytes): a counted vmcnt(5) then means "everything but the
  // newest stage has landed".
  // (row pointers are fixed per piece: only the column offset moves with the chunk - recomputing row * C per stage
  //  was a dozen 64-bit multiply-adds per iteration, issued while the matrix pipe stood still)
  const float* rowp[4];
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) {
    const int piece = pc * 4 + w;
    int row = ((piece < 8) ? bi * 128 + piece * 16 : bj * 128 + (piece - 8) * 16) + j;
    row = row < M ? row : M - 1;
    rowp[pc] = a.alpha + (long long)row * C;
  }
  const unsigned glds0 = __builtin_amdgcn_readfirstlane(lds_addr(&lds[0][0]));
  const unsigned gsg0 = __builtin_amdgcn_readfirstlane(lds_addr(&sg[0][0]));
#define GPSA_GB_STAGE(CH, BUF)                                                                \
  {                                                                                           \
    long long col__ = (long long)(CH) * 16 + kq * 4;                                          \
    col__ = col__ < C - 4 ? col__ : C - 4;                                                    \
    const unsigned d__ = glds0 + (unsigned)(BUF) * (16 * 256 * 4) + (unsigned)w * 1024;       \
    glds16(rowp[0] + col__, d__);                                                             \
    glds16(rowp[1] + col__, d__ + 4 * 1024);                                                  \
    glds16(rowp[2] + col__, d__ + 8 * 1024);                                                  \
    glds16(rowp[3] + col__, d__ + 12 * 1024);                                                 \
    if (lane < 4) glds16(gl + (long long)(CH) * 16 + lane * 4, gsg0 + (unsigned)(BUF) * 64);  \
  }
  if (ch0 < ch1) {
    GPSA_GB_STAGE(ch0, 0)
    GPSA_GB_STAGE(ch0 + 1 < ch1 ? ch0 + 1 : ch0, 1)
  }
  GPSA_DMA_WAIT(5);
  __syncthreads();
  int buf = 0;
  for (long long ch = ch0; ch < ch1; ++ch) {
    // slot (buf + 2) % 3 held chunk ch - 1: everyone left it before the barrier that ended that iteration
    GPSA_GB_STAGE(ch + 2 < ch1 ? ch + 2 : ch1 - 1, buf == 0 ? 2 : buf - 1)
    const float* base = &lds[buf][lane * 4];
    const float4 gk = *reinterpret_cast<const float4*>(&sg[buf][kq * 4]);
    float4 av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 x = *reinterpret_cast<const float4*>(base + (wr * 4 + i) * 256);
      av[i] = make_float4(x.x * gk.x, x.y * gk.y, x.z * gk.z, x.w * gk.w);
      bv[i] = *reinterpret_cast<const float4*>(base + (8 + wc * 4 + i) * 256);
    }
#define GPSA_GB_MMA(F)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                             \
      acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].F, bv[k].F, acc[i][k], 0, 0, 0);
    GPSA_GB_MMA(x)
    GPSA_GB_MMA(y)
    GPSA_GB_MMA(z)
    GPSA_GB_MMA(w)
#undef GPSA_GB_MMA
    GPSA_DMA_WAIT(5);
    __syncthreads();
    buf = (buf == 2) ? 0 : buf + 1;
  }
  GPSA_DMA_DRAIN();
#undef GPSA_GB_STAGE
  float* P = a.part + ((long long)l * a.nsplit + sp) * M * M;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bi * 128 + wr * 64 + i * 16 + kq * 4 + r, col = bj * 128 + wc * 64 + k * 16 + j;
        if (row < M && col < M) P[(long long)row * M + col] = acc[i][k][r];
      }
}

// The large-M full product W[l] = P[l] X  ([M,M] x [M,C], fp32 matrix cores) with both operands staged by LDS-DMA
// in MFMA-fragment order, like gram_big_kernel.  P rows are contiguous along the contracted index: a piece is
// 16 rows x 16 k, lane j + 16 kq holding P[16 p + j][k0 + 4 kq .. +3] (component F = MFMA step F, which contracts
// k0 + {F, 4+F, 8+F, 12+F}).  X rows are contiguous along the OUTPUT index: piece F of a 64-column group is the
// four rows k0 + 4 kq + F with lane j holding columns 4 j .. 4 j + 3, so component G feeds the MFMA tile of the
// columns {4 j + G} - and the four tiles' results of a lane are four CONSECUTIVE columns: one 16-byte store.
// One workgroup = a 128 x 128 tile of one output; grid (row blocks, outputs, column tiles): the workgroups that
// run together share the column tile of X.  P is zero-padded along k (garbage rows of X beyond M meet zeros).
__global__ void __launch_bounds__(256, 2) prod_big_kernel(ProdBigArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[3][16 * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4, wr = w >> 1, wc = w & 1;
  const int M = a.M, Mp = a.Mp, l = blockIdx.y;
  const long long C = a.C;
  const int m0 = blockIdx.x * 128;
  const long long c0 = (long long)blockIdx.z * 128;
  const float* Pl = a.P + (long long)l * M * Mp;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // pieces 0..7: rows m0 + 16 p of P; 8..15: X piece (group = (p - 8) >> 2, F = (p - 8) & 3)
#define GPSA_PB_STAGE(CH, BUF)                                                                \
  {                                                                                           \
    const int k0__ = (CH) * 16;                                                               \
    _Pragma("unroll") for (int pc = 0; pc < 4; ++pc) {                                        \
      const int piece = pc * 4 + w;                                                           \
      const float* src__;                                                                     \
      if (piece < 8) {                                                                        \
        int row__ = m0 + piece * 16 + j;                                                      \
        row__ = row__ < M ? row__ : M - 1;                                                    \
        src__ = Pl + (long long)row__ * Mp + k0__ + kq * 4;                                   \
      } else {                                                                                \
        const int grp__ = (piece - 8) >> 2, F__ = (piece - 8) & 3;                            \
        int krow__ = k0__ + kq * 4 + F__;                                                     \
        krow__ = krow__ < M ? krow__ : M - 1;                                                 \
        long long col__ = c0 + grp__ * 64 + j * 4;                                            \
        col__ = col__ < C - 4 ? col__ : C - 4;                                                \
        src__ = a.X + (long long)krow__ * C + col__;                                          \
      }                                                                                       \
      glds16(src__, __builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][piece * 256])));       \
    }                                                                                         \
  }
  const int nch = Mp / 16;
  GPSA_PB_STAGE(0, 0)
  GPSA_PB_STAGE(nch > 1 ? 1 : 0, 1)
  GPSA_DMA_WAIT(4);
  __syncthreads();
  int buf = 0;
  for (int ch = 0; ch < nch; ++ch) {
    GPSA_PB_STAGE(ch + 2 < nch ? ch + 2 : nch - 1, buf == 0 ? 2 : buf - 1)
    const float* base = &lds[buf][lane * 4];
    float4 av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      av[i] = *reinterpret_cast<const float4*>(base + (wr * 4 + i) * 256);
      bv[i] = *reinterpret_cast<const float4*>(base + (8 + wc * 4 + i) * 256);
    }
    // step F: A = av[rt].F ; B tile G = bv[F].G
#define GPSA_PB_MMA(F, BF)                                                                    \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].F, BF.x, acc[i][0], 0, 0, 0);      \
    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].F, BF.y, acc[i][1], 0, 0, 0);      \
    acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].F, BF.z, acc[i][2], 0, 0, 0);      \
    acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].F, BF.w, acc[i][3], 0, 0, 0);      \
  }
    GPSA_PB_MMA(x, bv[0])
    GPSA_PB_MMA(y, bv[1])
    GPSA_PB_MMA(z, bv[2])
    GPSA_PB_MMA(w, bv[3])
#undef GPSA_PB_MMA
    GPSA_DMA_WAIT(4);
    __syncthreads();
    buf = (buf == 2) ? 0 : buf + 1;
  }
  GPSA_DMA_DRAIN();
#undef GPSA_PB_STAGE
  float* Wl = a.W + (long long)l * M * C;
  const long long col = c0 + wc * 64 + j * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wr * 64 + i * 16 + kq * 4 + r;
      if (row < M && col < C) {  // C % 4 == 0: the four columns are in or out together
        const f32x4 o = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(Wl + (long long)row * C + col));
      }
    }
}

// P [n][M][M] (TS) -> fp32 [n][M][Mp], zero for k >= M
template <typename TS>
__global__ void pad_k_kernel(const TS* __restrict__ src, int M, int Mp, long long n, float* __restrict__ dst) {
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= n * M * Mp) return;
  const int k = (int)(idx % Mp);
  const long long rowi = idx / Mp;
  dst[idx] = k < M ? (float)src[rowi * M + k] : 0.f;
}

// P [n][M][M] (TS) -> fp32 [n][M][Mp]: U = diag + 2 strict-upper (zero below the diagonal and for k >= M):
// a^T P a = a^T U a for symmetric P, and row block m0 of U a contracts k >= m0 only
template <typename TS>
__global__ void pad_k_tri_kernel(const TS* __restrict__ src, int M, int Mp, long long n, float* __restrict__ dst) {
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= n * M * Mp) return;
  const int k = (int)(idx % Mp);
  const long long rowi = idx / Mp;
  const int i = (int)(rowi % M);
  dst[idx] = (k < M && k >= i) ? (float)(k > i ? 2.0 * (double)src[rowi * M + k] : (double)src[rowi * M + k]) : 0.f;
}

// ------------------------------------------------------------------------------------------------
// M > 256 with many outputs (BASELINE configs 4 / 5 at their stated size: L = 2000 / 1000): the quadratic form
// and its alpha-gradient WITHOUT materialising the products Omega_l alpha (L M C floats: 160 / 800 GB there).
// Both kernels are prod_big_kernel's 128 x 128 tile with the same LDS-DMA staging in MFMA-fragment order, run as
// ONE software pipeline over a flattened sequence of tiles so that the accumulators (and the ring) stay live:
//
//   big_quad_kernel<TRI, STORE>: one workgroup = (output l, 128 columns), walking the row blocks rb = 0 .. nrb-1.
//     After the last K chunk of a row block the accumulators hold W[rows of rb][cols]; they are multiplied by alpha
//     read in the SAME (C-layout) positions and summed into four per-lane column sums; the workgroup closes
//     v[l, cols] in fixed order (deterministic).  TRI: the operand is U_l = diag + 2 strict-upper(Omega_l) and
//     row block rb starts at K chunk 8 rb (block-triangular: 10 of 16 / 36 of 64 block products at M = 500 / 1000).
//     STORE (training with kept products): the full product, each accumulator block also leaving for W[l]
//     as 16-byte nontemporal stores - prod_big_kernel + the closing column-dot pass in one kernel.
//   big_accum_kernel: one workgroup = (row block rb, 128 columns), walking l = l0 .. l1-1:
//     out[rows, cols] = scale * sum_l Omega_l[rows, :] (g[l, cols] o alpha[:, cols]); g scales the B fragments as
//     they are read, so one accumulator set runs over (l, k).  Workgroups are numbered so that the ones that run
//     together on an XCD (ids equal mod 8 under the observed round-robin placement; speed only) cover all row blocks
//     of a few column tiles: an XCD's L2 then streams Omega_l once per l for every column tile it is working on.
// ------------------------------------------------------------------------------------------------
struct BigQuadArgs {
  const float* P;  // [L][M][Mp]  (TRI: U_l, else Omega_l), zero for k >= M
  const float* X;  // alpha [M][C]
  float* v;        // [L][C]
  float* W;        // STORE: [L][M][C]
  int M, Mp, L;
  long long C;
  int lb;  // > 0: 1-D grid; same-XCD workgroups come in runs of ``lb`` outputs of ONE column tile (they share its
           // alpha tile in that XCD's L2; each U_l / Omega_l is then shared by the few column tiles the XCD works on)
};

#define GPSA_BIG_MMA(F, BF)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].F, BF.x, acc[i][0], 0, 0, 0);      \
    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].F, BF.y, acc[i][1], 0, 0, 0);      \
    acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].F, BF.z, acc[i][2], 0, 0, 0);      \
    acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].F, BF.w, acc[i][3], 0, 0, 0);      \
  }

template <bool TRI, bool STORE, int NS>
__global__ void __launch_bounds__(256, 2) big_quad_kernel(BigQuadArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[NS][16 * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4, wr = w >> 1, wc = w & 1;
  const int M = a.M, Mp = a.Mp;
  const long long C = a.C;
  int l;
  long long ctile;
  if (a.lb > 0) {
    const long long id = blockIdx.x, slot = id >> 3, ctiles = (C + 127) / 128;
    const int nlb = (a.L + a.lb - 1) / a.lb;
    const long long q = (slot / a.lb) * 8 + (id & 7);  // (column tile, block of outputs)
    if (q >= ctiles * nlb) return;
    ctile = q % ctiles;
    l = (int)(q / ctiles) * a.lb + (int)(slot % a.lb);
    if (l >= a.L) return;
  } else {
    l = blockIdx.y;
    ctile = blockIdx.x;
  }
  const long long c0 = ctile * 128;
  const float* Pl = a.P + (long long)l * M * Mp;
  const int nch = Mp / 16, nrb = (M + 127) / 128;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 vs = (f32x4){0.f, 0.f, 0.f, 0.f};
  // alpha at the accumulators' own (row, column) positions, for the closing sum: K chunk 8 rb + 4 wr + i of row
  // block rb stages exactly the rows 16 i + 4 kq + F of this wave's 64 as its B fragments (bv[F] = alpha[k0 + 4 kq
  // + F][the lane's four columns]) - captured as they pass, no second read of alpha
  float4 aC[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) aC[i][r] = make_float4(0.f, 0.f, 0.f, 0.f);
  // the lane's four output columns (clamped: columns beyond C are computed on repeated data and never stored)
  long long colc = c0 + wc * 64 + j * 4;
  const bool col_ok = colc < C;  // C % 4 == 0: the four columns are in or out together
  colc = colc < C - 4 ? colc : C - 4;
  // stage cursor: two chunks ahead of the compute cursor; past the end it keeps re-staging the last chunk
  // ... as pointers advanced by constants (see big_accum_kernel): the wave's two row groups of P (+16 floats per
  // chunk; a new row block: recomputed, once per ~8-63 chunks) and the X rows k0 + 4 kq + w of the two column groups
  // (+16 C floats per chunk; a new row block restarts them at its first chunk)
  int s_rb = 0, s_ch = 0;
  long long xc0 = c0 + j * 4, xc1 = c0 + 64 + j * 4;
  xc0 = xc0 < C - 4 ? xc0 : C - 4;
  xc1 = xc1 < C - 4 ? xc1 : C - 4;
  const float* const xtop0 = a.X + (long long)(kq * 4 + w) * C + xc0;  // chunk 0
  const float* const xtop1 = a.X + (long long)(kq * 4 + w) * C + xc1;
  const bool last_oob = (nch - 1) * 16 + kq * 4 + w >= M;  // the last chunk may reach beyond row M - 1 of X
  const float* const xclamp0 = a.X + (long long)(M - 1) * C + xc0;
  const float* const xclamp1 = a.X + (long long)(M - 1) * C + xc1;
  const long long xstep = 16 * C;
  const float* sx0 = xtop0;
  const float* sx1 = xtop1;
  const float *sp0, *sp1;
#define GPSA_BQ_ROWS()                                                              \
  {                                                                                 \
    int r0__ = s_rb * 128 + w * 16 + j, r1__ = s_rb * 128 + (4 + w) * 16 + j;       \
    r0__ = r0__ < M ? r0__ : M - 1;                                                 \
    r1__ = r1__ < M ? r1__ : M - 1;                                                 \
    sp0 = Pl + (long long)r0__ * Mp + s_ch * 16 + kq * 4;                           \
    sp1 = Pl + (long long)r1__ * Mp + s_ch * 16 + kq * 4;                           \
  }
  GPSA_BQ_ROWS()
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(&lds[0][0]));
#define GPSA_BQ_ADVANCE()                                \
  {                                                      \
    if (s_ch + 1 < nch) {                                \
      ++s_ch;                                            \
      sp0 += 16;                                         \
      sp1 += 16;                                         \
      sx0 += xstep;                                      \
      sx1 += xstep;                                      \
    } else if (s_rb + 1 < nrb) {                         \
      ++s_rb;                                            \
      s_ch = TRI ? s_rb * 8 : 0;                         \
      GPSA_BQ_ROWS()                                     \
      sx0 = xtop0 + (long long)s_ch * xstep;             \
      sx1 = xtop1 + (long long)s_ch * xstep;             \
    }                                                    \
  }
#define GPSA_BQ_STAGE(BUF)                                                                    \
  {                                                                                           \
    const unsigned d__ = lds0 + (unsigned)(BUF) * (16 * 256 * 4) + (unsigned)w * 1024;        \
    const bool oob__ = last_oob && s_ch == nch - 1;                                           \
    glds16(sp0, d__);                                                                         \
    glds16(sp1, d__ + 4 * 1024);                                                              \
    glds16(oob__ ? xclamp0 : sx0, d__ + 8 * 1024);                                            \
    glds16(oob__ ? xclamp1 : sx1, d__ + 12 * 1024);                                           \
  }
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0) {
    GPSA_BQ_STAGE(s0)
    GPSA_BQ_ADVANCE()
  }
  GPSA_DMA_WAIT(4 * (NS - 2));
  __syncthreads();
  int buf = 0;
  for (int rb = 0; rb < nrb; ++rb) {
    for (int ch = TRI ? rb * 8 : 0; ch < nch; ++ch) {
      GPSA_BQ_STAGE(buf == 0 ? NS - 1 : buf - 1)
      GPSA_BQ_ADVANCE()
      const float* base = &lds[buf][lane * 4];
      float4 av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av[i] = *reinterpret_cast<const float4*>(base + (wr * 4 + i) * 256);
        bv[i] = *reinterpret_cast<const float4*>(base + (8 + wc * 4 + i) * 256);
      }
      {
        const int cc = ch - rb * 8 - wr * 4;  // wave-uniform
        if (cc == 0) { aC[0][0] = bv[0]; aC[0][1] = bv[1]; aC[0][2] = bv[2]; aC[0][3] = bv[3]; }
        else if (cc == 1) { aC[1][0] = bv[0]; aC[1][1] = bv[1]; aC[1][2] = bv[2]; aC[1][3] = bv[3]; }
        else if (cc == 2) { aC[2][0] = bv[0]; aC[2][1] = bv[1]; aC[2][2] = bv[2]; aC[2][3] = bv[3]; }
        else if (cc == 3) { aC[3][0] = bv[0]; aC[3][1] = bv[1]; aC[3][2] = bv[2]; aC[3][3] = bv[3]; }
      }
      GPSA_BIG_MMA(x, bv[0])
      GPSA_BIG_MMA(y, bv[1])
      GPSA_BIG_MMA(z, bv[2])
      GPSA_BIG_MMA(w, bv[3])
      GPSA_DMA_WAIT(4 * (NS - 2));
      __syncthreads();
      buf = (buf == NS - 1) ? 0 : buf + 1;
    }
    // close row block rb: v += sum_rows alpha[row, col] W[row, col]  (accumulator (i, G, r) = row 16 i + 4 kq + r of
    // the wave's 64, column 4 j + G of its 64)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb * 128 + wr * 64 + i * 16 + kq * 4 + r;
        const f32x4 o = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        const f32x4 xa = (f32x4){aC[i][r].x, aC[i][r].y, aC[i][r].z, aC[i][r].w};
        if (row < M) {
          vs += xa * o;
          if (STORE) {
            if (col_ok)
              __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(a.W + ((long long)l * M + row) * C + colc));
          }
        }
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  GPSA_DMA_DRAIN();
#undef GPSA_BQ_STAGE
#undef GPSA_BQ_ROWS
#undef GPSA_BQ_ADVANCE
  // column sums: over the four lane quarters (rows), then over the two waves that share the columns
#pragma unroll
  for (int G = 0; G < 4; ++G) {
    vs[G] += __shfl_xor(vs[G], 16);
    vs[G] += __shfl_xor(vs[G], 32);
  }
  __syncthreads();  // every wave has left the ring
  float* red = &lds[0][0];
  if (kq == 0) *reinterpret_cast<f32x4*>(red + wr * 128 + wc * 64 + j * 4) = vs;
  __syncthreads();
  if (tid < 128 && c0 + tid < C) a.v[(long long)l * C + c0 + tid] = red[tid] + red[128 + tid];
}

struct BigAccumArgs {
  const float* P;  // [L][M][Mp] Omega_l, zero for k >= M
  const float* X;  // alpha [M][C]
  const float* g;  // [L][C]
  float* out;      // [nsplit][M][C]
  int M, Mp, L, nrb, nsplit;
  long long C, ctiles;
  float scale;
};

template <int NS>
__global__ void __launch_bounds__(256, 2) big_accum_kernel(BigAccumArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[NS][16 * 256];
  __shared__ __attribute__((aligned(16))) float sg[NS][128];  // g[l, the 128 columns] of each stage's output l
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4, wr = w >> 1, wc = w & 1;
  const int M = a.M, Mp = a.Mp;
  const long long C = a.C;
  // workgroup id -> (row block, column tile, split of the outputs): ids equal mod 8 share an XCD
  // same-XCD workgroups (slot order): row block fastest, then the split of the outputs, then the column tile: the
  // 64 that run together on an XCD cover all row blocks x all splits of 64 / (nrb nsplit) column tiles - few alpha
  // tiles (they stay in that XCD's L2 across the outputs), every Omega_l[row block] shared by those column tiles
  const long long id = blockIdx.x, slot = id >> 3, ct8 = (a.ctiles + 7) / 8;
  const int rb = (int)(slot % a.nrb);
  const long long t = slot / a.nrb;
  const int sp = (int)(t % a.nsplit);
  const long long ct = (t / a.nsplit) * 8 + (id & 7);
  if (ct >= a.ctiles) return;
  (void)ct8;
  const long long c0 = ct * 128;
  const int l0 = (int)((long long)sp * a.L / a.nsplit), l1 = (int)((long long)(sp + 1) * a.L / a.nsplit);
  const int nch = Mp / 16;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  long long colc = c0 + wc * 64 + j * 4;
  const bool col_ok = colc < C;
  colc = colc < C - 4 ? colc : C - 4;
  // Stage cursor (two chunks ahead of the compute cursor) as POINTERS advanced by constants: the wave's four pieces
  // are two row groups of Omega_l[row block] (16 rows each, k contiguous: +16 floats per chunk, +M Mp per output)
  // and the X rows k0 + 4 kq + w of the two 64-column groups (+16 C floats per chunk, back to the top per output).
  // Recomputing them from (l, chunk) cost ~60 64-bit multiply-adds per iteration, issued while the matrix pipe of
  // BOTH resident waves stood still (the two workgroups of a CU run this loop in phase).
  int s_l = l0, s_ch = 0;
  int prow0 = rb * 128 + w * 16 + j, prow1 = rb * 128 + (4 + w) * 16 + j;
  prow0 = prow0 < M ? prow0 : M - 1;
  prow1 = prow1 < M ? prow1 : M - 1;
  const float* sp0 = a.P + (long long)l0 * M * Mp + (long long)prow0 * Mp + kq * 4;
  const float* sp1 = a.P + (long long)l0 * M * Mp + (long long)prow1 * Mp + kq * 4;
  long long xc0 = c0 + j * 4, xc1 = c0 + 64 + j * 4;
  xc0 = xc0 < C - 4 ? xc0 : C - 4;
  xc1 = xc1 < C - 4 ? xc1 : C - 4;
  const float* const xtop0 = a.X + (long long)(kq * 4 + w) * C + xc0;  // chunk 0
  const float* const xtop1 = a.X + (long long)(kq * 4 + w) * C + xc1;
  // the last chunk may reach beyond row M - 1 of X (Omega is zero there): those lanes read row M - 1 instead
  const bool last_oob = (nch - 1) * 16 + kq * 4 + w >= M;
  const float* const xclamp0 = a.X + (long long)(M - 1) * C + xc0;
  const float* const xclamp1 = a.X + (long long)(M - 1) * C + xc1;
  const float* sx0 = xtop0;
  const float* sx1 = xtop1;
  const long long xstep = 16 * C, pnext = (long long)M * Mp - (long long)(nch - 1) * 16;
  // the stage's g rides along as a fifth operation of every wave (all four write the same 512 bytes; a load the
  // compiler sees would make it drain the ring - vmcnt(0) - in every iteration): vmcnt(5) = "all but the newest stage"
  long long gcol = c0 + (lane & 31) * 4;
  gcol = gcol < C - 4 ? gcol : C - 4;
  const float* sgp = a.g + (long long)l0 * C + gcol;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(&lds[0][0]));
  const unsigned sg0 = __builtin_amdgcn_readfirstlane(lds_addr(&sg[0][0]));
#define GPSA_BA_ADVANCE()                                \
  {                                                      \
    if (s_ch + 1 < nch) {                                \
      ++s_ch;                                            \
      sp0 += 16;                                         \
      sp1 += 16;                                         \
      sx0 += xstep;                                      \
      sx1 += xstep;                                      \
    } else if (s_l + 1 < l1) {                           \
      ++s_l;                                             \
      s_ch = 0;                                          \
      sp0 += pnext;                                      \
      sp1 += pnext;                                      \
      sx0 = xtop0;                                       \
      sx1 = xtop1;                                       \
      sgp += C;                                          \
    }                                                    \
  }
#define GPSA_BA_STAGE(BUF)                                                                    \
  {                                                                                           \
    const unsigned d__ = lds0 + (unsigned)(BUF) * (16 * 256 * 4) + (unsigned)w * 1024;        \
    const bool oob__ = last_oob && s_ch == nch - 1;                                           \
    glds16(sp0, d__);                                                                         \
    glds16(sp1, d__ + 4 * 1024);                                                              \
    glds16(oob__ ? xclamp0 : sx0, d__ + 8 * 1024);                                            \
    glds16(oob__ ? xclamp1 : sx1, d__ + 12 * 1024);                                           \
    if (lane < 32) glds16(sgp, sg0 + (unsigned)(BUF) * 512);                                  \
  }
  if (l0 < l1) {
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) {
      GPSA_BA_STAGE(s0)
      GPSA_BA_ADVANCE()
    }
  }
  GPSA_DMA_WAIT(5 * (NS - 2));
  __syncthreads();
  int buf = 0;
  for (int l = l0; l < l1; ++l) {
    for (int ch = 0; ch < nch; ++ch) {
      GPSA_BA_STAGE(buf == 0 ? NS - 1 : buf - 1)
      GPSA_BA_ADVANCE()
      const float* base = &lds[buf][lane * 4];
      const float4 gl = *reinterpret_cast<const float4*>(&sg[buf][wc * 64 + j * 4]);
      float4 av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av[i] = *reinterpret_cast<const float4*>(base + (wr * 4 + i) * 256);
        const float4 x = *reinterpret_cast<const float4*>(base + (8 + wc * 4 + i) * 256);
        bv[i] = make_float4(x.x * gl.x, x.y * gl.y, x.z * gl.z, x.w * gl.w);
      }
      GPSA_BIG_MMA(x, bv[0])
      GPSA_BIG_MMA(y, bv[1])
      GPSA_BIG_MMA(z, bv[2])
      GPSA_BIG_MMA(w, bv[3])
      GPSA_DMA_WAIT(5 * (NS - 2));
      __syncthreads();
      buf = (buf == NS - 1) ? 0 : buf + 1;
    }
  }
  GPSA_DMA_DRAIN();
#undef GPSA_BA_STAGE
#undef GPSA_BA_ADVANCE
  float* O = a.out + (long long)sp * M * C;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rb * 128 + wr * 64 + i * 16 + kq * 4 + r;
      if (row < M && col_ok) {
        const f32x4 o = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]} * a.scale;
        *reinterpret_cast<f32x4*>(O + (long long)row * C + colc) = o;
      }
    }
}
#undef GPSA_BIG_MMA

// out[e] = sum_s part[s][e]  (fixed order), four floats per thread
__global__ void __launch_bounds__(256) big_accum_reduce_kernel(const float* __restrict__ part, int nsplit, long long n4,
                                                              float* __restrict__ out) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
  for (int p = 1; p < nsplit; ++p) s += reinterpret_cast<const f32x4*>(part)[(long long)p * n4 + i];
  reinterpret_cast<f32x4*>(out)[i] = s;
}

// shapes the two kernels cover (everything else stays on the generic tiled product)
static inline bool big_panel_ok(int M, long long C, int L, const void* alpha) {
  static const bool off = [] { const char* e = getenv("GPSA_BIG_PANEL"); return e && e[0] == '0'; }();
  return !off && M > 128 && (C & 3) == 0 && C >= 128 && cdiv(C, 128) * cdiv(M, 128) * 32 < 0x7fffffffLL && L <= 65535 &&
         (reinterpret_cast<uintptr_t>(alpha) & 15) == 0;
}
// splits of the outputs for big_accum_kernel: the fewest (<= 4) that fill the rounds of workgroups (2 per CU) to
// >= 90 %, else the fullest
static inline int big_accum_nsplit(int M, long long C, int L) {
  static const int forced = [] { const char* e = getenv("GPSA_BA_NSPLIT"); return e ? atoi(e) : 0; }();
  if (forced > 0) return (forced <= L) ? forced : 1;
  const long long wgs = cdiv(M, 128) * cdiv(C, 128), slots = 2LL * num_cus();
  int best = 1;
  double beff = 0.0;
  for (int s = 1; s <= 4 && (s == 1 || L / s >= 8); ++s) {
    const double eff = (double)(wgs * s) / (double)(cdiv(wgs * s, slots) * slots);
    if (eff > beff) { beff = eff; best = s; }
    if (eff >= 0.9) break;
  }
  return best;
}
static inline long long big_accum_ws_bytes(int M, long long C, int L) {
  const int ns = big_accum_nsplit(M, C, L);
  return ns > 1 ? (long long)ns * M * C * 4 : 0;
}

// Omega[b] = A[b] A[b]^T + jitter I for a batch of small fp32 parameter matrices (M = 200: the 54 variational
// covariances of a step), fp64 matrix cores.  One workgroup = a 64 x 64 block pair (bi >= bj) of one matrix; both
// operands are rows of A, contiguous along the contracted index: 16-k chunks of the 64 + 64 rows move to LDS by
// LDS-DMA as eight 1-KiB pieces in fragment order (a lane's float4 = four consecutive k = the four MFMA steps of
// the chunk), widened to fp64 as they are read.  The generic product staged the same operands through registers
// and transposing LDS stores: 44 us for the 54 matrices against 29 us here (a chunk is only 16 MFMAs per wave, so
// the loop overhead shows; a six-slot ring with five stages in flight was SLOWER, 33 us: not a DMA-latency bound).
struct OmegaDmaArgs {
  const float* A0;
  const float* A1;
  double* O0;
  double* O1;
  int n0, M;
  double jitter;
};
__global__ void __launch_bounds__(256, 4) omega_fwd_dma_kernel(OmegaDmaArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[3][8 * 256];
  typedef double f64x4_ __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4, wr = w >> 1, wc = w & 1;
  int bi = 0, t = blockIdx.x;
  while (t > bi) {
    t -= bi + 1;
    ++bi;
  }
  const int bj = t, M = a.M;
  int b = blockIdx.z;
  const float* A = a.A0;
  double* O = a.O0;
  if (b >= a.n0) {
    b -= a.n0;
    A = a.A1;
    O = a.O1;
  }
  A += (long long)b * M * M;
  O += (long long)b * M * M;
  f64x4_ acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k) acc[i][k] = (f64x4_){0.0, 0.0, 0.0, 0.0};
  // pieces 0..3: rows of block bi, 4..7: rows of block bj; wave w moves pieces w and w + 4
#define GPSA_OM_STAGE(CH, BUF)                                                                \
  {                                                                                           \
    int col__ = (CH) * 16 + kq * 4;                                                           \
    col__ = col__ < M - 4 ? col__ : M - 4;                                                    \
    _Pragma("unroll") for (int pc = 0; pc < 2; ++pc) {                                        \
      const int piece = pc * 4 + w;                                                           \
      int row__ = ((piece < 4) ? bi * 64 + piece * 16 : bj * 64 + (piece - 4) * 16) + j;      \
      row__ = row__ < M ? row__ : M - 1;                                                      \
      glds16(A + (long long)row__ * M + col__,                                                \
             __builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][piece * 256])));               \
    }                                                                                         \
  }
  const int nch = (M + 15) / 16;
  GPSA_OM_STAGE(0, 0)
  GPSA_OM_STAGE(nch > 1 ? 1 : 0, 1)
  GPSA_DMA_WAIT(2);
  __syncthreads();
  int buf = 0;
  for (int ch = 0; ch < nch; ++ch) {
    GPSA_OM_STAGE(ch + 2 < nch ? ch + 2 : nch - 1, buf == 0 ? 2 : buf - 1)
    const float* base = &lds[buf][lane * 4];
    float4 av[2], bv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      av[i] = *reinterpret_cast<const float4*>(base + (wr * 2 + i) * 256);
      bv[i] = *reinterpret_cast<const float4*>(base + (4 + wc * 2 + i) * 256);
    }
    // the chunk's columns beyond M were clamped onto real ones: they must not count (left operand zeroed)
    const int kb = ch * 16 + kq * 4;
    if (kb + 3 >= M) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (kb + 0 >= M) av[i].x = 0.f;
        if (kb + 1 >= M) av[i].y = 0.f;
        if (kb + 2 >= M) av[i].z = 0.f;
        if (kb + 3 >= M) av[i].w = 0.f;
      }
    }
#define GPSA_OM_MMA(F)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                               \
    _Pragma("unroll") for (int k = 0; k < 2; ++k)                                             \
      acc[i][k] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[i].F, (double)bv[k].F, acc[i][k], 0, 0, 0);
    GPSA_OM_MMA(x)
    GPSA_OM_MMA(y)
    GPSA_OM_MMA(z)
    GPSA_OM_MMA(w)
#undef GPSA_OM_MMA
    GPSA_DMA_WAIT(2);
    __syncthreads();
    buf = (buf == 2) ? 0 : buf + 1;
  }
  GPSA_DMA_DRAIN();
#undef GPSA_OM_STAGE
  // fp64 C layout: row = kq + 4 r, column = j.  Both halves are written (the product is symmetric bit for bit:
  // the mirrored entry is the same sum of the same products in the same order)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bi * 64 + wr * 32 + i * 16 + kq + 4 * r, col = bj * 64 + wc * 32 + k * 16 + j;
        if (row < M && col < M) {
          const double v = acc[i][k][r] + (row == col ? a.jitter : 0.0);
          O[(long long)row * M + col] = v;
          if (bi != bj) O[(long long)col * M + row] = v;
        }
      }
}

int omega_fwd_dma_launch(const float* A0, int n0, double* O0, const float* A1, int n1, double* O1, int M, double jitter,
                         hipStream_t st) {
  static const bool off = [] { const char* e = getenv("GPSA_OMEGA_DMA"); return e && e[0] == '0'; }();
  if (off || (M & 3) != 0 || M < 16 || (reinterpret_cast<uintptr_t>(A0) & 15) != 0 ||
      (n1 > 0 && (reinterpret_cast<uintptr_t>(A1) & 15) != 0) || n0 + n1 > 65535)
    return GPSA_EUNSUPPORTED;
  const int nb = (int)cdiv(M, 64);
  OmegaDmaArgs a{A0, A1, O0, O1, n1 > 0 ? n0 : 0x7fffffff, M, jitter};
  omega_fwd_dma_kernel<<<dim3((unsigned)(nb * (nb + 1) / 2), 1, (unsigned)(n0 + n1)), 256, 0, st>>>(a);
  GPSA_LAUNCH_CHECK();
  return 0;
}

// dA[b] = 2 G[b] A[b] (G fp64 symmetric gradient of Omega, A the fp32 parameter, dA fp32): the adjoint of the
// kernel above for the same batch, same staging.  G's rows are contiguous along the contracted index: a piece is
// 16 rows x 8 k, a lane's double2 = two consecutive k (component F = one of the two MFMA steps of the piece).
// A's rows are contiguous along the output index: its piece for a step holds the four rows k of that step with lane
// j on the columns 4 j .. 4 j + 3, so component G feeds the MFMA tile of the columns {4 j + G} and a lane's four
// results are one 16-byte store.  One workgroup = 64 x 64 outputs, one wave = 16 rows x 64 columns.
struct OmegaBwdArgs {
  const double* G0;
  const double* G1;
  const float* A0;
  const float* A1;
  float* D0;
  float* D1;
  int n0, M;
};
__global__ void __launch_bounds__(256, 4) omega_bwd_dma_kernel(OmegaBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[3][12 * 256];
  typedef double f64x4_ __attribute__((ext_vector_type(4)));
  typedef double f64x2_ __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int M = a.M, m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  int b = blockIdx.z;
  const double* Gm = a.G0;
  const float* A = a.A0;
  float* D = a.D0;
  if (b >= a.n0) {
    b -= a.n0;
    Gm = a.G1;
    A = a.A1;
    D = a.D1;
  }
  Gm += (long long)b * M * M;
  A += (long long)b * M * M;
  D += (long long)b * M * M;
  f64x4_ acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f64x4_){0.0, 0.0, 0.0, 0.0};
  // pieces 0..7: G, row tile p >> 1, k half p & 1;  8..11: A, step s = p - 8 = 2 h + F (rows k0 + 8 h + 2 kq + F)
#define GPSA_OB_STAGE(CH, BUF)                                                                \
  {                                                                                           \
    const int k0__ = (CH) * 16;                                                               \
    _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) {                                        \
      const int piece = pc * 4 + w;                                                           \
      const void* src__;                                                                      \
      if (piece < 8) {                                                                        \
        int row__ = m0 + (piece >> 1) * 16 + j;                                               \
        row__ = row__ < M ? row__ : M - 1;                                                    \
        int col__ = k0__ + (piece & 1) * 8 + kq * 2;                                          \
        col__ = col__ < M - 2 ? col__ : M - 2;                                                \
        src__ = Gm + (long long)row__ * M + col__;                                            \
      } else {                                                                                \
        const int s__ = piece - 8;                                                            \
        int krow__ = k0__ + (s__ >> 1) * 8 + kq * 2 + (s__ & 1);                              \
        krow__ = krow__ < M ? krow__ : M - 1;                                                 \
        int col__ = n0 + j * 4;                                                               \
        col__ = col__ < M - 4 ? col__ : M - 4;                                                \
        src__ = A + (long long)krow__ * M + col__;                                            \
      }                                                                                       \
      glds16(reinterpret_cast<const float*>(src__),                                           \
             __builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][piece * 256])));               \
    }                                                                                         \
  }
  const int nch = (M + 15) / 16;
  GPSA_OB_STAGE(0, 0)
  GPSA_OB_STAGE(nch > 1 ? 1 : 0, 1)
  GPSA_DMA_WAIT(3);
  __syncthreads();
  int buf = 0;
  for (int ch = 0; ch < nch; ++ch) {
    GPSA_OB_STAGE(ch + 2 < nch ? ch + 2 : nch - 1, buf == 0 ? 2 : buf - 1)
    const float* base = &lds[buf][lane * 4];
    f64x2_ g2[2];
    float4 bv[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      g2[h] = *reinterpret_cast<const f64x2_*>(base + (w * 2 + h) * 256);
      // columns of G beyond M were clamped onto real ones: they must not count
      const int kb = ch * 16 + h * 8 + kq * 2;
      if (kb >= M) g2[h].x = 0.0;
      if (kb + 1 >= M) g2[h].y = 0.0;
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) bv[s_] = *reinterpret_cast<const float4*>(base + (8 + s_) * 256);
#define GPSA_OB_MMA(GA, BV)                                                                   \
  acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(GA, (double)BV.x, acc[0], 0, 0, 0);           \
  acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(GA, (double)BV.y, acc[1], 0, 0, 0);           \
  acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(GA, (double)BV.z, acc[2], 0, 0, 0);           \
  acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(GA, (double)BV.w, acc[3], 0, 0, 0);
    GPSA_OB_MMA(g2[0].x, bv[0])
    GPSA_OB_MMA(g2[0].y, bv[1])
    GPSA_OB_MMA(g2[1].x, bv[2])
    GPSA_OB_MMA(g2[1].y, bv[3])
#undef GPSA_OB_MMA
    GPSA_DMA_WAIT(3);
    __syncthreads();
    buf = (buf == 2) ? 0 : buf + 1;
  }
  GPSA_DMA_DRAIN();
#undef GPSA_OB_STAGE
  const int col = n0 + j * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + w * 16 + kq + 4 * r;
    if (row < M && col < M)  // M % 4 == 0: the four columns are in or out together
      *reinterpret_cast<f32x4*>(D + (long long)row * M + col) =
          (f32x4){(float)(2.0 * acc[0][r]), (float)(2.0 * acc[1][r]), (float)(2.0 * acc[2][r]), (float)(2.0 * acc[3][r])};
  }
}

int omega_bwd_dma_launch(const double* G0, const float* A0, float* D0, int n0, const double* G1, const float* A1,
                         float* D1, int n1, int M, hipStream_t st) {
  static const bool off = [] { const char* e = getenv("GPSA_OMEGA_DMA"); return e && e[0] == '0'; }();
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (off || (M & 3) != 0 || M < 16 || !al(G0) || !al(A0) || !al(D0) || (n1 > 0 && (!al(G1) || !al(A1) || !al(D1))) ||
      n0 + n1 > 65535)
    return GPSA_EUNSUPPORTED;
  const unsigned nb = (unsigned)cdiv(M, 64);
  OmegaBwdArgs a{G0, G1, A0, A1, D0, D1, n1 > 0 ? n0 : 0x7fffffff, M};
  omega_bwd_dma_kernel<<<dim3(nb, nb, (unsigned)(n0 + n1)), 256, 0, st>>>(a);
  GPSA_LAUNCH_CHECK();
  return 0;
}

// out[l][i][j] = out[l][j][i] = sum_s part[l][s][max(i,j)][min(i,j)]  (the lower blocks hold every i >= j)
template <typename TO>
__global__ void __launch_bounds__(256) gram_big_reduce_kernel(const float* __restrict__ part, int M, int nsplit,
                                                              TO* __restrict__ out) {
  const long long mm = (long long)M * M, e = blockIdx.x * 256LL + threadIdx.x;
  if (e >= mm) return;
  const int l = blockIdx.y, i = (int)(e / M), jj = (int)(e % M);
  const long long src = (jj <= i) ? e : (long long)jj * M + i;
  const float* p = part + (long long)l * nsplit * mm + src;
  float s = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) s += p[(long long)sp * mm];
  out[(long long)l * mm + e] = (TO)s;
}

// Visiting order of a workgroup's column tiles.  The item range [it0, it1) covers tiles tile0..tile1;
// the first and the last may be partial in l.  Full tiles are visited first, so that every workgroup
// sweeps l = 0..L-1 in step with all the others (they start together and run at the MFMA rate): the 32
// workgroups of an XCD then stream the SAME packed Omega_l chunk within microseconds of each other
// and share it through their L2 instead of each pulling it over the fabric.  Then the partial last
// tile (l from 0, still in phase) and the partial first tile.
struct TileOrder {
  long long tile0, tile1, nfull, n;
  int lo0, hi1, L, fp, lp;
  __device__ TileOrder(long long it0, long long it1, int L_) {
    L = L_;
    tile0 = it0 / L;
    tile1 = (it1 - 1) / L;
    lo0 = (int)(it0 - tile0 * L);
    hi1 = (int)(it1 - 1 - tile1 * L);
    if (tile0 == tile1) {
      fp = lp = 0;
      nfull = 0;
      n = 1;
    } else {
      fp = lo0 != 0;
      lp = hi1 != L - 1;
      nfull = (tile1 - lp) - (tile0 + fp) + 1;
      n = nfull + fp + lp;
    }
  }
  __device__ void get(long long step, long long& tile, int& a, int& b) const {
    if (tile0 == tile1) {
      tile = tile0; a = lo0; b = hi1;
    } else if (step < nfull) {
      tile = tile0 + fp + step; a = 0; b = L - 1;
    } else if (lp && step == nfull) {
      tile = tile1; a = 0; b = hi1;
    } else {
      tile = tile0; a = lo0; b = L - 1;
    }
  }
};

// Persistent, balanced schedule: the work is the list of items (column tile, l) in column-tile-major
// order; workgroup b of G processes the contiguous item range [b*T/G, (b+1)*T/G).  The wave's slab of
// X is (re)loaded only when the column tile changes (at most ~T/G/L + 2 times).  In ACCUM mode a column
// tile whose l-range is split between two workgroups is combined with float atomics into a
// pre-zeroed output (at most two contributors per element => order-independent result).
// RL: MFMA steps of the last K chunk that are issued (4, or 2 when M % 16 <= 8 leaves the rest padding;
// ACCUM / STORE only, see PACK_KSTEP).
template <int MB, int NCT, int MODE, int RL>
__global__ void __launch_bounds__(256, (MB * NCT >= 24) ? 1 : 2)
panel_mfma_kernel(const float* __restrict__ Ppk,  // [L][MB][MP][16]
                  const float* __restrict__ X,    // [M][C]
                  const float* __restrict__ g,    // [L][C]   (ACCUM)
                  int M, long long C, int L,
                  float* __restrict__ out,        // QUAD: v [L][C]; ACCUM/STORE: Y [M][C]
                  float* __restrict__ colsq,      // STORE: optional [C]
                  float out_scale,
                  float* __restrict__ slab,       // ACCUM: [gridDim.x][2][MP][WGCOLS] partial tiles
                  float* __restrict__ keep = nullptr) {  // QUAD: optional, the products P_l X in fragment order
  constexpr int MP = MB * 16;
  constexpr int WGCOLS = 64 * NCT;
  constexpr int CHUNK = MP * 16;                    // floats per K chunk (MB pieces of 256 floats)
  constexpr int NPW = (MB + 3) / 4;                 // LDS-DMA pieces per wave per stage (uniform)
  constexpr int BUFF = NPW * 4 * 256;               // floats per LDS buffer (incl. dummy slots)
  __shared__ __attribute__((aligned(16))) float lds[3][BUFF];  // 3-deep ring, 2 stages in flight

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;

  const long long ntiles = (C + WGCOLS - 1) / WGCOLS;
  const long long T = ntiles * L;
  const long long it0 = (long long)blockIdx.x * T / gridDim.x;
  const long long it1 = (long long)(blockIdx.x + 1) * T / gridDim.x;
  if (it0 >= it1) return;

  float xb[NCT][MB][4];
  float xl[NCT][4];  // QUAD with RL < 4: the last chunk's B operand in K-step order (see PACK_KSTEP_LAST)
  f32x4 acc[MB][NCT];
  // K chunk Q of the packed left operand -> LDS buffer BUF by LDS-DMA (no VGPR staging, no ds_write):
  // wave w moves pieces w, w+4, ... (1 KiB each, lane-linear); every wave issues exactly NPW
  // operations per stage (the surplus ones re-load the last piece into an unused slot) so that a
  // counted vmcnt(NPW) means "everything but the newest stage has landed".
#define GPSA_STAGE(Q, BUF)                                                                     \
  {                                                                                            \
    const float* src__ = Ppk + (long long)(Q) * CHUNK + lane * 4;                              \
    _Pragma("unroll") for (int pc = 0; pc < NPW; ++pc) {                                       \
      const int piece = pc * 4 + w;                                                            \
      glds16(src__ + (piece < MB ? piece : MB - 1) * 256,                                      \
             __builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][piece * 256])));                \
    }                                                                                          \
  }
  // the chunk stream of this workgroup follows the tile visiting order; staged two chunks ahead
  const TileOrder ord(it0, it1, L);
  long long sstep = 0, stile_;
  int sa_, sb_, skc = 0;
  ord.get(0, stile_, sa_, sb_);
  int sl = sa_;
  bool sdone = false;
#define GPSA_STAGE_PIECE(BUF, PC)                                                              \
  {                                                                                            \
    const float* src__ = Ppk + ((long long)sl * MB + skc) * CHUNK + lane * 4;                  \
    const int piece = (PC) * 4 + w;                                                            \
    glds16(src__ + (piece < MB ? piece : MB - 1) * 256,                                        \
           __builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][piece * 256])));                  \
  }
#define GPSA_STAGE_ADVANCE()                                                                   \
  {                                                                                            \
    if (!sdone) {                                                                              \
      if (skc + 1 < MB) ++skc;                                                                 \
      else if (sl < sb_) { skc = 0; ++sl; }                                                    \
      else if (sstep + 1 < ord.n) { ++sstep; ord.get(sstep, stile_, sa_, sb_); sl = sa_; skc = 0; } \
      else sdone = true;                                                                       \
    }                                                                                          \
  }
#define GPSA_STAGE_NEXT(BUF)                                                                   \
  {                                                                                            \
    GPSA_STAGE((long long)sl * MB + skc, BUF)                                                  \
    if (!sdone) {                                                                              \
      if (skc + 1 < MB) ++skc;                                                                 \
      else if (sl < sb_) { skc = 0; ++sl; }                                                    \
      else if (sstep + 1 < ord.n) { ++sstep; ord.get(sstep, stile_, sa_, sb_); sl = sa_; skc = 0; } \
      else sdone = true;                                                                       \
    }                                                                                          \
  }
  // flush the accumulators of column tile TILE (ACCUM / STORE).  PLAIN: this workgroup covered all l of
  // the tile -> straight to the output.  Otherwise the partial sum goes to one of this workgroup's two
  // slabs (WHICH = 0: its first tile, 1: its last tile) and panel_slab_reduce_kernel adds the slabs of
  // a tile in workgroup order: no atomics, any number of contributors, bitwise reproducible.
#define GPSA_FLUSH(TILE, PLAIN, WHICH)                                                      \
  {                                                                                         \
    const long long cw__ = (TILE) * WGCOLS + (long long)w * (16 * NCT);                     \
    const bool pl__ = (PLAIN);                                                              \
    float* dst__ = pl__ ? out : slab + ((long long)blockIdx.x * 2 + (WHICH)) * MP * WGCOLS; \
    const long long rs__ = pl__ ? C : (long long)WGCOLS;                                    \
    const int mlim__ = pl__ ? M : MP;                                                       \
    _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct) {                                    \
      const long long c = cw__ + ct * 16 + j;                                               \
      const long long col__ = pl__ ? c : (long long)(w * (16 * NCT) + ct * 16 + j);         \
      const bool okc__ = pl__ ? (c < C) : true;                                             \
      float s = 0.f;                                                                        \
      _Pragma("unroll") for (int rt = 0; rt < MB; ++rt)                                     \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                     \
          const int row = rt * 16 + kq * 4 + r;                                             \
          const float y = acc[rt][ct][r] * out_scale;                                       \
          s += y * y;                                                                       \
          if (okc__ && row < mlim__) dst__[(long long)row * rs__ + col__] = y;              \
        }                                                                                   \
      if (MODE == MODE_STORE && colsq != nullptr) {                                         \
        s += __shfl_xor(s, 16, 64);                                                         \
        s += __shfl_xor(s, 32, 64);                                                         \
        if (kq == 0 && c < C) colsq[c] = s;                                                 \
      }                                                                                     \
    }                                                                                       \
  }

  int buf = 0;  // ring slot being computed; slot (buf+2)%3 receives the stage issued now
  GPSA_STAGE_NEXT(0)
  GPSA_STAGE_NEXT(1)
  GPSA_DMA_WAIT(NPW);
  __syncthreads();

  for (long long step = 0; step < ord.n; ++step) {
    long long tile;
    int l_lo, l_hi;  // inclusive
    ord.get(step, tile, l_lo, l_hi);
    const long long cw = tile * WGCOLS + (long long)w * (16 * NCT);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const long long c = cw + ct * 16 + j;
#pragma unroll
      for (int t = 0; t < MB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // B operand of MFMA step r (the packed operand's K order)
          const int row = t * 16 + ((MODE == MODE_QUAD) ? kq * 4 + r : r * 4 + kq);
          xb[ct][t][r] = (c < C && row < M) ? X[(long long)row * C + c] : 0.f;
        }
      if (MODE == MODE_QUAD && RL < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (MB - 1) * 16 + r * 4 + kq;
          xl[ct][r] = (r < RL && c < C && row < M) ? X[(long long)row * C + c] : 0.f;
        }
      }
    }
#pragma unroll
    for (int rt = 0; rt < MB; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int l = l_lo; l <= l_hi; ++l) {
      float gv[NCT];
      if (MODE == MODE_ACCUM) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const long long c = cw + ct * 16 + j;
          gv[ct] = (c < C) ? g[(long long)l * C + c] : 0.f;
        }
      }
#pragma unroll
      for (int kc = 0; kc < MB; ++kc) {
        float bv[NCT][4];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            bv[ct][r] = (MODE == MODE_ACCUM) ? xb[ct][kc][r] * gv[ct]
                        : ((MODE == MODE_QUAD && RL < 4 && kc == MB - 1) ? xl[ct][r] : xb[ct][kc][r]);
        const float* base = &lds[buf][lane * 4];
        // A fragments are read one row tile ahead of the MFMAs that consume them (LDS latency
        // hides under the previous tile's 4*NCT MFMAs instead of stalling the matrix pipe)
        float4 a_nxt = *reinterpret_cast<const float4*>(base);
#pragma unroll
        for (int rt = 0; rt < MB; ++rt) {
          const float4 a4 = a_nxt;
          const float av[4] = {a4.x, a4.y, a4.z, a4.w};
          // The non-matrix work of a row tile - the next fragment's LDS read, and the staging of the next-but-one
          // chunk (its slot was free since the barrier that ended the previous chunk: cursor arithmetic and LDS-DMA
          // issues) - is pinned BETWEEN the K steps of the tile, a few instructions behind each group of NCT MFMAs:
          // a 16x16x4 fp32 MFMA occupies the pipe for 32 cycles and the wave (alone on its SIMD) can issue ~6 other
          // instructions in that shadow, but a dozen of them in one clump in front of a tile overrun it and leave
          // the pipe idle.
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (!(kc == MB - 1 && r >= RL)) {  // all-padding K steps are skipped (compile time)
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct)
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[ct][r], acc[rt][ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (r == 0) {
              if (rt + 1 < MB) a_nxt = *reinterpret_cast<const float4*>(base + (rt + 1) * 256);
            } else if (r == 1) {
              if (MB >= NPW + 3) {
                if (rt < NPW) GPSA_STAGE_PIECE(buf == 0 ? 2 : buf - 1, rt)
                if (rt == NPW) GPSA_STAGE_ADVANCE()
              } else if (rt == 0) {
                GPSA_STAGE_NEXT(buf == 0 ? 2 : buf - 1)
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        GPSA_DMA_WAIT(NPW);
        __syncthreads();
        buf = (buf == 2) ? 0 : buf + 1;
      }
      if (MODE == MODE_QUAD) {
        // v[l,c] = sum over the rows this lane holds of acc * alpha, then across the 4 lane quarters
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          float s = 0.f;
          const long long c = cw + ct * 16 + j;
          // keep: the product Omega_l alpha leaves through HBM once, for the backward's streaming pass, in the
          // accumulators' own order (one 16-byte store per lane and 16 x 16 block, 1 KiB contiguous per wave;
          // row-major [M][C] stores of 64-byte segments cost 1.7 ms per 4 GB here):
          //   keep[l][tile][wave][ct][rt][lane][r]  =  (Omega_l alpha)[16 rt + 4 kq + r][column of (tile, wave, ct, j)]
          f32x4* kp = nullptr;
          if (keep != nullptr)  // block-uniform
            kp = reinterpret_cast<f32x4*>(keep) +
                 (((((long long)l * ntiles + tile) * 4 + w) * NCT + ct) * MB) * 64 + lane;
#pragma unroll
          for (int rt = 0; rt < MB; ++rt) {
            if (kp != nullptr) __builtin_nontemporal_store(acc[rt][ct], &kp[rt * 64]);  // written once, read once, much later
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              s += acc[rt][ct][r] * xb[ct][rt][r];
              acc[rt][ct][r] = 0.f;
            }
          }
          s += __shfl_xor(s, 16, 64);
          s += __shfl_xor(s, 32, 64);
          if (kq == 0 && c < C) out[(long long)l * C + c] = s;
        }
      }
    }
    if (MODE != MODE_QUAD) {
      const bool plain = (l_lo == 0) && (l_hi == L - 1);
      GPSA_FLUSH(tile, plain, (tile == ord.tile0) ? 0 : 1)
    }
  }
  GPSA_DMA_DRAIN();  // nothing may still be writing this workgroup's LDS when it exits
#undef GPSA_STAGE
#undef GPSA_STAGE_PIECE
#undef GPSA_STAGE_ADVANCE
#undef GPSA_STAGE_NEXT
#undef GPSA_FLUSH
}

// ------------------------------------------------------------------------------------------------
// The data GP's forward, its Gaussian likelihood and the backward's  abar = 2 sum_l g_l Omega_l alpha  in ONE
// pass over the products  W_l = Omega_l alpha  (vgpsa.py:186-204 variance, :334-351 draw, :532-538 likelihood).
// The gradient of the ELBO wrt the draw's variance,
//     g[l,c] = dLoss/dF[c,l] * eps[c,l] / (2 sqrt(var[l,c])),   dLoss/dF = -(Y - F) / (s^2 S)   (loss = ... - LL),
// is elementwise in (l, c) once v[l,c] = alpha_c . W_l[:,c] is known: the workgroup that has just closed output l
// of a column tile holds W_l for those columns in its accumulators, so g_l W_l joins a second accumulator set there
// and the products never leave the chip - no 4 GB kept copy written by the forward and streamed back by the
// backward (0.7-0.8 ms per step at the headline size).  Everything is formed at upstream gradient 1: the
// backward scales by the loss's actual upstream gradient (linear).
// Same schedule as panel_mfma_kernel<QUAD> (persistent balanced items, LDS-DMA ring, register-resident alpha
// slab); the second accumulator set leaves like ACCUM's (plain store, or slabs for a column tile whose outputs
// are split between workgroups).  The per-(l, column) inputs mean / eps / Y reach the closing through LDS-DMA
// too (4-byte gathers issued under the output's first chunk): a compiler-visible load there would make hipcc wait
// for vmcnt(0), i.e. for the two ring stages in flight.
struct ElboArgs {
  const float* Ppk;      // packed Omega, as for panel_mfma_kernel<QUAD>
  const float* X;        // alpha [M][C]
  int M;
  long long C;
  int L;
  const float* meanT;    // [L][C]
  const double* q;       // [C]  k_uf^T K^-1 k_uf
  const float* var_u;    // log of the data kernel's variance
  const float* eps;      // [C][L] standard-normal draws
  const float* Y;        // [N][L] observations, column c belongs to row c % N
  const float* noise_u;  // log of the likelihood's "variance" (used as a standard deviation: SURVEY quirk 5)
  long long N;
  int S;
  float* g;              // [L][C]
  float* dmeanT;         // [L][C]  dLoss/dmean = dLoss/dF
  float* abar;           // [M][C]
  float* slab;           // 2 partial tiles per workgroup (accum_slab layout with this kernel's NCT)
  double* part;          // [nparts] sum of z^2 = ((Y - F) / s)^2 over the workgroup's items; entries >= gridDim.x: 0
  int nparts;
};

template <int MB, int NCT, int RL>
__global__ void __launch_bounds__(256, (MB * NCT >= 14) ? 1 : 2) panel_elbo_kernel(ElboArgs a) {
  constexpr int MP = MB * 16;
  constexpr int WGCOLS = 64 * NCT;
  constexpr int CHUNK = MP * 16;
  constexpr int NPW = (MB + 3) / 4;
  constexpr int BUFF = NPW * 4 * 256;
  constexpr int NGATHER = (3 * NCT * 16 + 63) / 64;  // 4-byte LDS-DMA operations per wave and output
  __shared__ __attribute__((aligned(16))) float lds[3][BUFF];
  __shared__ __attribute__((aligned(16))) float sgat[4][NGATHER * 64];  // [wave][(ct*3 + kind)*16 + j]: mean, eps, Y
  __shared__ double red[4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const float* __restrict__ Ppk = a.Ppk;
  const float* __restrict__ X = a.X;
  const int M = a.M, L = a.L;
  const long long C = a.C;

  const long long ntiles = (C + WGCOLS - 1) / WGCOLS;
  const long long T = ntiles * L;
  const long long it0 = (long long)blockIdx.x * T / gridDim.x;
  const long long it1 = (long long)(blockIdx.x + 1) * T / gridDim.x;
  if (blockIdx.x == 0)
    for (int i = (int)gridDim.x + tid; i < a.nparts; i += 256) a.part[i] = 0.0;
  if (it0 >= it1) {
    if (tid == 0) a.part[blockIdx.x] = 0.0;
    return;
  }

  float xb[NCT][MB][4];
  float xl[NCT][4];
  f32x4 acc[MB][NCT], ab[MB][NCT];
#define GPSA_STAGE(Q, BUF)                                                                     \
  {                                                                                            \
    const float* src__ = Ppk + (long long)(Q) * CHUNK + lane * 4;                              \
    _Pragma("unroll") for (int pc = 0; pc < NPW; ++pc) {                                       \
      const int piece = pc * 4 + w;                                                            \
      glds16(src__ + (piece < MB ? piece : MB - 1) * 256,                                      \
             __builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][piece * 256])));                \
    }                                                                                          \
  }
  const TileOrder ord(it0, it1, L);
  long long sstep = 0, stile_;
  int sa_, sb_, skc = 0;
  ord.get(0, stile_, sa_, sb_);
  int sl = sa_;
  bool sdone = false;
#define GPSA_STAGE_PIECE(BUF, PC)                                                              \
  {                                                                                            \
    const float* src__ = Ppk + ((long long)sl * MB + skc) * CHUNK + lane * 4;                  \
    const int piece = (PC) * 4 + w;                                                            \
    glds16(src__ + (piece < MB ? piece : MB - 1) * 256,                                        \
           __builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][piece * 256])));                  \
  }
#define GPSA_STAGE_ADVANCE()                                                                   \
  {                                                                                            \
    if (!sdone) {                                                                              \
      if (skc + 1 < MB) ++skc;                                                                 \
      else if (sl < sb_) { skc = 0; ++sl; }                                                    \
      else if (sstep + 1 < ord.n) { ++sstep; ord.get(sstep, stile_, sa_, sb_); sl = sa_; skc = 0; } \
      else sdone = true;                                                                       \
    }                                                                                          \
  }
#define GPSA_STAGE_NEXT(BUF)                                                                   \
  {                                                                                            \
    GPSA_STAGE((long long)sl * MB + skc, BUF)                                                  \
    GPSA_STAGE_ADVANCE()                                                                       \
  }

  // likelihood constants (elementwise.hip: loglik_*_kernel)
  const double sN = exp((double)a.noise_u[0]) + 1e-5;
  const float inv = (float)(1.0 / sN);
  const float coef = (float)(-1.0 / (sN * sN * (double)a.S));  // dLoss/dF = coef (Y - F) at upstream gradient 1
  const double var0 = exp((double)a.var_u[0]);
  double z2 = 0.0;

  int buf = 0;
  GPSA_STAGE_NEXT(0)
  GPSA_STAGE_NEXT(1)
  GPSA_DMA_WAIT(NPW);
  __syncthreads();

  for (long long step = 0; step < ord.n; ++step) {
    long long tile;
    int l_lo, l_hi;
    ord.get(step, tile, l_lo, l_hi);
    const long long cw = tile * WGCOLS + (long long)w * (16 * NCT);
    float resid[NCT];
    bool okc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const long long c = cw + ct * 16 + j;
      okc[ct] = c < C;
      // sigma^2 - q formed in fp64 before rounding (data_sample_fwd_kernel)
      resid[ct] = okc[ct] ? (float)(var0 - a.q[c]) : 1.f;
#pragma unroll
      for (int t = 0; t < MB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = t * 16 + kq * 4 + r;
          xb[ct][t][r] = (c < C && row < M) ? X[(long long)row * C + c] : 0.f;
        }
      if (RL < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (MB - 1) * 16 + r * 4 + kq;
          xl[ct][r] = (r < RL && c < C && row < M) ? X[(long long)row * C + c] : 0.f;
        }
      }
    }
    // gather addresses of this lane for output l_lo: operation o moves element (o*64 + lane) of the wave's
    // [(ct*3 + kind)*16 + j] table; kind 0: mean[l][c] (next output: + C), 1: eps[c][l] (+ 1), 2: Y[c % N][l] (+ 1)
    const float* gp[NGATHER];
    long long gstep[NGATHER];
#pragma unroll
    for (int o = 0; o < NGATHER; ++o) {
      int e = o * 64 + lane;
      if (e >= 3 * NCT * 16) e = 0;  // surplus lanes re-load element 0 (never read)
      const int ct = e / 48, kind = (e % 48) / 16, jj = e % 16;
      long long c = cw + ct * 16 + jj;
      c = c < C ? c : C - 1;
      gp[o] = kind == 0 ? a.meanT + (long long)l_lo * C + c
                        : (kind == 1 ? a.eps + c * L + l_lo : a.Y + (c % a.N) * L + l_lo);
      gstep[o] = kind == 0 ? C : 1;
    }
#pragma unroll
    for (int rt = 0; rt < MB; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) asm("v_accvgpr_write_b32 %0, 0" : "=a"(ab[rt][ct][r]));
      }

    for (int l = l_lo; l <= l_hi; ++l) {
#pragma unroll
      for (int kc = 0; kc < MB; ++kc) {
        float bv[NCT][4];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[ct][r] = (RL < 4 && kc == MB - 1) ? xl[ct][r] : xb[ct][kc][r];
        const float* base = &lds[buf][lane * 4];
        float4 a_nxt = *reinterpret_cast<const float4*>(base);
        if (kc == 0) {
          // this output's mean / eps / Y: BEFORE the chunk's ring stage is issued, so that the counted wait at the
          // end of the chunk (all but the newest NPW operations) covers them
#pragma unroll
          for (int o = 0; o < NGATHER; ++o) {
            unsigned keep__;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                : "=&s"(keep__)
                : "v"(gp[o]), "s"(__builtin_amdgcn_readfirstlane(lds_addr(&sgat[w][o * 64])))
                : "memory");
            gp[o] += gstep[o];
          }
        }
#pragma unroll
        for (int rt = 0; rt < MB; ++rt) {
          const float4 a4 = a_nxt;
          const float av[4] = {a4.x, a4.y, a4.z, a4.w};
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (!(kc == MB - 1 && r >= RL)) {
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct)
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                    av[r], bv[ct][r], (kc == 0 && r == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[rt][ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (r == 0) {
              if (rt + 1 < MB) a_nxt = *reinterpret_cast<const float4*>(base + (rt + 1) * 256);
            } else if (r == 1) {
              if (MB >= NPW + 3) {
                if (rt < NPW) GPSA_STAGE_PIECE(buf == 0 ? 2 : buf - 1, rt)
                if (rt == NPW) GPSA_STAGE_ADVANCE()
              } else if (rt == 0) {
                GPSA_STAGE_NEXT(buf == 0 ? 2 : buf - 1)
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        GPSA_DMA_WAIT(NPW);
        __syncthreads();
        buf = (buf == 2) ? 0 : buf + 1;
      }
      // closing of output l: v, the draw, its likelihood term and gradient, and g_l W_l into the second set
      float z2l = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        float s = 0.f;
#pragma unroll
        for (int rt = 0; rt < MB; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s += acc[rt][ct][r] * xb[ct][rt][r];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mean = sgat[w][(ct * 3 + 0) * 16 + j];
        const float e = sgat[w][(ct * 3 + 1) * 16 + j];
        const float y = sgat[w][(ct * 3 + 2) * 16 + j];
        const float var = resid[ct] + s + 2e-5f;  // TWO_JITTER (elementwise.hip)
        const float sd = sqrtf(var);
        const float rres = y - (mean + sd * e);
        const float dF = coef * rres;
        const float gv = okc[ct] ? dF * e * 0.5f / sd : 0.f;
        if (okc[ct] && kq == 0) {
          const long long o = (long long)l * C + cw + ct * 16 + j;
          a.g[o] = gv;
          a.dmeanT[o] = dF;
          const float z = rres * inv;
          z2l += z * z;
        }
        // g_l W_l joins the second accumulator set, which lives in the AGPR file like the first (every access through
        // an "a"-constrained operand): left to itself the allocator homes it in VGPRs - the VALU cannot address AGPRs -
        // and evicts the alpha slab to AGPRs instead, 150 register copies in front of the MFMAs of every output.
#pragma unroll
        for (int rt = 0; rt < MB; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t;
            asm("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(ab[rt][ct][r]));
            t = fmaf(gv, acc[rt][ct][r], t);
            asm("v_accvgpr_write_b32 %0, %1" : "=a"(ab[rt][ct][r]) : "v"(t));
          }
      }
      z2 += (double)z2l;
    }
    // the column tile's abar: straight to the output when this workgroup covered all its outputs, else a slab
    {
      const bool pl = (l_lo == 0) && (l_hi == L - 1);
      const int which = (tile == ord.tile0) ? 0 : 1;
      float* dst = pl ? a.abar : a.slab + ((long long)blockIdx.x * 2 + which) * MP * WGCOLS;
      const long long rs = pl ? C : (long long)WGCOLS;
      const int mlim = pl ? M : MP;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const long long c = cw + ct * 16 + j;
        const long long col = pl ? c : (long long)(w * (16 * NCT) + ct * 16 + j);
        const bool ok = pl ? (c < C) : true;
#pragma unroll
        for (int rt = 0; rt < MB; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + kq * 4 + r;
            float t;
            asm("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(ab[rt][ct][r]));
            if (ok && row < mlim) dst[(long long)row * rs + col] = 2.f * t;
          }
      }
    }
  }
  GPSA_DMA_DRAIN();
  z2 = block_sum(z2, red);
  if (tid == 0) a.part[blockIdx.x] = z2;
#undef GPSA_STAGE
#undef GPSA_STAGE_PIECE
#undef GPSA_STAGE_ADVANCE
#undef GPSA_STAGE_NEXT
}

// ------------------------------------------------------------------------------------------------
// Symmetric quadratic form:  v[l,c] = alpha_c^T Omega_l alpha_c  using only the upper-triangle tiles
//   v = sum_rt alpha_rt . ( Omega[rt,rt] alpha_rt + 2 sum_{kc>rt} Omega[rt,kc] alpha_kc )
// (the factor 2 and the zero lower tiles are baked into the packed operand, PACK_SYM_UPPER).
// 91 instead of 169 tile products at M = 200.  K chunks are processed in pairs (MB-1-p, p) so that
// every step between two barriers has the same MB+1 row-tile products.  The upper triangle puts the
// padding of M % 16 != 0 into the K direction of the last chunk's MB products, where whole MFMA steps
// are skipped (RL of 4 issued, as in panel_mfma_kernel; that chunk is packed in K-step order and
// contracts against the small slab xk), instead of into the rows of a tile row that cannot be.  Same register-resident alpha
// slab, LDS-DMA staging, persistent balanced items and in-register closing as panel_mfma_kernel.
// ------------------------------------------------------------------------------------------------
template <int MB, int NCT, int RL>
__global__ void __launch_bounds__(256, (MB * NCT >= 24) ? 1 : 2)
quad_sym_mfma_kernel(const float* __restrict__ Ppk,  // [L][MB][MB][256] PACK_SYM_UPPER-packed
                     const float* __restrict__ X, int M, long long C, int L,
                     float* __restrict__ out) {
  constexpr int WGCOLS = 64 * NCT;
  constexpr int PER_L = MB * MB * 256;              // floats per packed matrix
  constexpr int NSTEP = (MB + 1) / 2;               // chunk pairs (the middle chunk stands alone)
  constexpr int NPW = (MB + 1 + 3) / 4;             // LDS-DMA pieces per wave per step (uniform)
  constexpr int BUFP = NPW * 4;                     // piece slots per LDS buffer
  __shared__ __attribute__((aligned(16))) float lds[3][BUFP * 256];  // ring, 2 stages in flight

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const long long ntiles = (C + WGCOLS - 1) / WGCOLS;
  const long long T = ntiles * L;
  const long long it0 = (long long)blockIdx.x * T / gridDim.x;
  const long long it1 = (long long)(blockIdx.x + 1) * T / gridDim.x;
  if (it0 >= it1) return;

  float xb[NCT][MB][4];
  float xk[NCT][RL < 4 ? RL : 1];
  f32x4 acc[MB][NCT];
  // step P of matrix LL: pieces (rt, kc=MB-1-P) for rt = 0..MB-1-P go to LDS slots 0..MB-1-P, then
  // pieces (rt, kc=P) for rt = 0..P to slots MB-P..MB   (second group absent when 2P == MB-1).
  // Every wave issues exactly NPW operations (surplus slots re-load the step's first piece).
#define GPSA_QS_STAGE(LL, P, BUF)                                                              \
  {                                                                                            \
    const float* m__ = Ppk + (long long)(LL) * PER_L + lane * 4;                               \
    const int p__ = (P), q__ = MB - 1 - p__;                                                   \
    const int n1__ = MB - p__, n2__ = (q__ != p__) ? p__ + 1 : 0;                              \
    _Pragma("unroll") for (int pc = 0; pc < NPW; ++pc) {                                       \
      const int sl = pc * 4 + w;                                                               \
      const int se = sl < n1__ + n2__ ? sl : 0;                                                \
      const int kc__ = se < n1__ ? q__ : p__;                                                  \
      const int rt__ = se < n1__ ? se : se - n1__;                                             \
      glds16(m__ + (kc__ * MB + rt__) * 256,                                                   \
             __builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][sl * 256])));                   \
    }                                                                                          \
  }
  // step stream of this workgroup (follows the tile visiting order), staged two steps ahead
  const TileOrder ord(it0, it1, L);
  long long sstep = 0, stile_;
  int sa_, sb_, sp_ = 0;
  ord.get(0, stile_, sa_, sb_);
  int sl_ = sa_;
  bool sdone = false;
#define GPSA_QS_STAGE_NEXT(BUF)                                                                \
  {                                                                                            \
    GPSA_QS_STAGE(sl_, sp_, BUF)                                                               \
    if (!sdone) {                                                                              \
      if (sp_ + 1 < NSTEP) ++sp_;                                                              \
      else if (sl_ < sb_) { sp_ = 0; ++sl_; }                                                  \
      else if (sstep + 1 < ord.n) { ++sstep; ord.get(sstep, stile_, sa_, sb_); sl_ = sa_; sp_ = 0; } \
      else sdone = true;                                                                       \
    }                                                                                          \
  }

  int buf = 0;
  GPSA_QS_STAGE_NEXT(0)
  GPSA_QS_STAGE_NEXT(1)
  GPSA_DMA_WAIT(NPW);
  __syncthreads();

  for (long long step = 0; step < ord.n; ++step) {
    long long tile;
    int l_lo, l_hi;
    ord.get(step, tile, l_lo, l_hi);
    const long long cw = tile * WGCOLS + (long long)w * (16 * NCT);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const long long c = cw + ct * 16 + j;
#pragma unroll
      for (int t = 0; t < MB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = t * 16 + kq * 4 + r;
          xb[ct][t][r] = (c < C && row < M) ? X[(long long)row * C + c] : 0.f;
        }
      if (RL < 4) {
#pragma unroll
        for (int r = 0; r < RL; ++r) {  // last chunk, K-step order: MFMA step r contracts rows 4 r .. 4 r + 3
          const int row = (MB - 1) * 16 + r * 4 + kq;
          xk[ct][r] = (c < C && row < M) ? X[(long long)row * C + c] : 0.f;
        }
      }
    }
#pragma unroll
    for (int rt = 0; rt < MB; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int l = l_lo; l <= l_hi; ++l) {
#pragma unroll
      for (int p = 0; p < NSTEP; ++p) {
        GPSA_QS_STAGE_NEXT(buf == 0 ? 2 : buf - 1)
        const float* base = &lds[buf][lane * 4];
        const int q = MB - 1 - p;
        const int n1 = MB - p, n2 = (q != p) ? p + 1 : 0;
        float4 a_nxt = *reinterpret_cast<const float4*>(base);
#pragma unroll
        for (int sl = 0; sl < MB + 1; ++sl) {
          if (sl < n1 + n2) {
            const int kc = sl < n1 ? q : p;
            const int rt = sl < n1 ? sl : sl - n1;
            const float4 a4 = a_nxt;
            if (sl + 1 < n1 + n2) a_nxt = *reinterpret_cast<const float4*>(base + (sl + 1) * 256);
            __builtin_amdgcn_sched_barrier(0);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (kc == MB - 1 && r >= RL) continue;  // all-padding K steps (compile time)
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct) {
                const float b = (RL < 4 && kc == MB - 1) ? xk[ct][r < RL ? r : 0] : xb[ct][kc][r];
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], b, acc[rt][ct], 0, 0, 0);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        GPSA_DMA_WAIT(NPW);
        __syncthreads();
        buf = (buf == 2) ? 0 : buf + 1;
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        float sacc = 0.f;
#pragma unroll
        for (int rt = 0; rt < MB; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sacc += acc[rt][ct][r] * xb[ct][rt][r];
            acc[rt][ct][r] = 0.f;
          }
        sacc += __shfl_xor(sacc, 16, 64);
        sacc += __shfl_xor(sacc, 32, 64);
        const long long c = cw + ct * 16 + j;
        if (kq == 0 && c < C) out[(long long)l * C + c] = sacc;
      }
    }
  }
  GPSA_DMA_DRAIN();
#undef GPSA_QS_STAGE
#undef GPSA_QS_STAGE_NEXT
}

// ------------------------------------------------------------------------------------------------
// MFMA Gram kernel:  dOmega_l = sum_c g[l,c] alpha_c alpha_c^T   (lower-triangle 16x16 tiles)
// grid (L, nsplit): workgroup (l, s) sweeps its share of the columns in 32-column chunks staged in
// LDS; the NT = MB(MB+1)/2 lower tiles are dealt to the 4 waves by whole tile rows (GramPlan).
// Both MFMA operands of a tile are rows of the same LDS image (A: rows of tile-row, scaled by g in
// registers once per row and K block; B: rows of tile-column); the K index (columns c) is permuted as
// in the panel kernel so that one ds_read_b128 feeds four MFMAs.  Partials [L][nsplit][MP][MP] are summed + mirrored by a second
// kernel (deterministic).
// ------------------------------------------------------------------------------------------------
constexpr int GR_KC = 64;  // columns per staged chunk (four 16-deep MFMA K blocks)

// Ownership of the lower-triangle tiles: whole tile ROWS are dealt to the 4 waves (longest row first,
// first fit: 23 / 23 / 23 / 22 tiles at MB = 13), so that a wave loads and g-scales the A fragment of a row once per K block
// and then only streams the B fragments of that row's columns: half the LDS fragment reads of a
// tile-by-tile deal, and no separate scaling pass over the chunk.
template <int MB>
struct GramPlan {
  static constexpr int NT = MB * (MB + 1) / 2;
  int cnt[4];
  int rr[4][NT], cc[4][NT];
  constexpr GramPlan() : cnt{}, rr{}, cc{} {
    const int cap = (NT + 3) / 4;  // first fit, longest row first, into waves of ceil(NT / 4) tiles
    for (int r = MB - 1; r >= 0; --r) {
      int best = -1;
      for (int w = 0; w < 4 && best < 0; ++w)
        if (cnt[w] + r + 1 <= cap) best = w;
      if (best < 0) {
        best = 0;
        for (int w = 1; w < 4; ++w)
          if (cnt[w] < cnt[best]) best = w;
      }
      for (int c = 0; c <= r; ++c) {
        rr[best][cnt[best]] = r;
        cc[best][cnt[best]] = c;
        ++cnt[best];
      }
    }
  }
  constexpr int max_cnt() const {
    int m = 0;
    for (int w = 0; w < 4; ++w) m = cnt[w] > m ? cnt[w] : m;
    return m;
  }
};

// one staged chunk (NKB K blocks of 16 columns) of wave W's tiles:
//     acc[s] += (g-scaled row fragment) x (column fragment)
// Tiles go in groups of GR_G with their MFMAs interleaved, so that an accumulator is touched again only
// every GR_G-th MFMA: with two chains the kernel ran at 2/3 of the MFMA issue rate (the back-to-back
// dependent latency of v_mfma_f32_16x16x4_f32 is well above two issue intervals), four chains hide it.
// The fragments of the next group - of the next K block after the last group - are fetched while this
// group computes, so the matrix pipe only sees a cold start once per chunk.  Every wave runs the same
// EVEN number of groups per block (the two fragment register sets then keep their roles from one trip
// of the K-block loop to the next); surplus slots repeat the wave's last tile into scratch accumulators
// acc[NS .. NS+GR_G-1] (never stored).
constexpr int GR_G = 4;

// NL outputs l per workgroup share every staged byte and every fragment read (their row fragments differ
// only by the g row they are scaled with): the non-MFMA instructions of a chunk are amortised over NL x
// the MFMAs.
template <int MB, int NKB, int W, int NS, int NL>
__device__ __forceinline__ void gram_wave_chunk(const float* __restrict__ img,
                                                const float* __restrict__ gvec, int kq,
                                                f32x4 (&acc)[NL][NS + GR_G]) {
  constexpr GramPlan<MB> P{};
  constexpr int N = P.cnt[W];
  constexpr int NGRP = (((NS + GR_G - 1) / GR_G) + 1) & ~1;
  auto frag = [&](int kb, int tile_row) {
    return *reinterpret_cast<const float4*>(img + kb * 256 + tile_row * (NKB * 256));
  };
  // slot s of the wave's schedule: tile (rr, cc), or a repeat of the last tile into scratch
  auto tile_of = [](int s) { return s < N ? s : N - 1; };
  auto new_row = [&](int s) { return s < N && (s == 0 || P.rr[W][s] != P.rr[W][s - 1]); };
  // The raw row fragment is fetched with the group's column fragments, one group ahead; it is scaled by
  // g when the group is CONSUMED (4 multiplies per row, output and K block, next to MFMAs that do not
  // depend on them) - scaling at fetch time would wait out the LDS round trip of a read issued a moment ago.
  float4 araw[2][GR_G], fb[2][GR_G];
  float4 gk[NL], gn[NL], arow[NL];
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    gn[q] = gk[q] = *reinterpret_cast<const float4*>(gvec + q * GR_KC + kq * 4);
    arow[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#define GPSA_GR_FETCH(SLOT, KB, GP)                                                           \
  _Pragma("unroll") for (int u = 0; u < GR_G; ++u) {                                          \
    const int t__ = tile_of(GR_G * (GP) + u);                                                 \
    if (new_row(GR_G * (GP) + u)) araw[SLOT][u] = frag(KB, P.rr[W][t__]);                     \
    fb[SLOT][u] = frag(KB, P.cc[W][t__]);                                                     \
  }
  GPSA_GR_FETCH(0, 0, 0)
#pragma unroll 1
  for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
    for (int q = 0; q < NL; ++q) gk[q] = gn[q];
#pragma unroll
    for (int gp = 0; gp < NGRP; ++gp) {
      const int cur = gp & 1, nxt = cur ^ 1;
      if (gp + 1 < NGRP) {
        GPSA_GR_FETCH(nxt, kb, gp + 1)
      } else if (kb + 1 < NKB) {  // first group of the next K block (and its g rows)
#pragma unroll
        for (int q = 0; q < NL; ++q)
          gn[q] = *reinterpret_cast<const float4*>(gvec + q * GR_KC + (kb + 1) * 16 + kq * 4);
        GPSA_GR_FETCH(nxt, kb + 1, 0)
      }
      __builtin_amdgcn_sched_barrier(0);
      float4 a[NL][GR_G];
      int sl[GR_G];
#pragma unroll
      for (int u = 0; u < GR_G; ++u) {
        const int s_ = GR_G * gp + u;
        sl[u] = s_ < N ? s_ : NS + u;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
          if (new_row(s_)) {
            const float4 r_ = araw[cur][u];
            arow[q] = make_float4(r_.x * gk[q].x, r_.y * gk[q].y, r_.z * gk[q].z, r_.w * gk[q].w);
          }
          a[q][u] = arow[q];
        }
      }
#define GPSA_GR_MMA(F)                                                                          \
  _Pragma("unroll") for (int q = 0; q < NL; ++q)                                                \
    _Pragma("unroll") for (int u = 0; u < GR_G; ++u)                                            \
      acc[q][sl[u]] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][u].F, fb[cur][u].F, acc[q][sl[u]], 0, 0, 0);
      GPSA_GR_MMA(x)
      GPSA_GR_MMA(y)
      GPSA_GR_MMA(z)
      GPSA_GR_MMA(w)
#undef GPSA_GR_MMA
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef GPSA_GR_FETCH
}

template <int MB, int W, int NS>
__device__ __forceinline__ void gram_wave_store(const f32x4 (&acc)[NS + GR_G], float* __restrict__ P_, int j,
                                                int kq) {
  constexpr GramPlan<MB> P{};
  constexpr int MP = MB * 16;
#pragma unroll
  for (int s = 0; s < P.cnt[W]; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      P_[(long long)(P.rr[W][s] * 16 + kq * 4 + r) * MP + P.cc[W][s] * 16 + j] = acc[s][r];
}

// The body of gram_mfma_kernel for wave W of its workgroup.  The four waves run DIFFERENT tile schedules
// (gram_wave_chunk<W>), each a straight-line instantiation; with one ``switch (w)`` per chunk inside a common loop
// the accumulators (2 x 22 tiles = 176 registers) crossed a control-flow join every iteration and the register
// allocator moved ALL of them between the VGPR and AGPR files there - 178 v_accvgpr_write per chunk of 182 MFMAs,
// issued with the matrix pipe idle (one wave per SIMD): the kernel sat at 0.76 pipe utilisation.  With the whole
// loop inside the per-wave instantiation the accumulators have one home.  (Barriers are counted per workgroup,
// not per program counter: the four waves meet at theirs from four different loops.)
template <int MB, bool ALIGNED, int NL, int W>
__device__ __forceinline__ void gram_wave_run(const float* __restrict__ alpha, const float* __restrict__ g, int M,
                                              long long C, int L, int nsplit, float* __restrict__ part,
                                              float* __restrict__ sA_, float* __restrict__ sG_) {
  constexpr int MP = MB * 16;
  constexpr GramPlan<MB> PLAN{};
  constexpr int NS = PLAN.max_cnt();
  constexpr int NKB = GR_KC / 16, NPIECE = MB * NKB;
  constexpr int NPW = (NPIECE + 3) / 4;
  constexpr int SA_STRIDE = NPW * 4 * 256, SG_STRIDE = NL * GR_KC;
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int w = W;
  const int j = lane & 15, kq = lane >> 4;
  const int l0 = blockIdx.x * NL, sp = blockIdx.y;  // outputs l0 .. l0+NL-1 (clamped: a surplus one is not stored)
  const long long nch = (C + GR_KC - 1) / GR_KC;
  const long long ch0 = (long long)sp * nch / nsplit, ch1 = (long long)(sp + 1) * nch / nsplit;

  f32x4 acc[NL][NS + GR_G];
#pragma unroll
  for (int q = 0; q < NL; ++q)
#pragma unroll
    for (int s = 0; s < NS + GR_G; ++s) acc[q][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging (LDS-DMA, ALIGNED): rows >= M are clamped to row M-1 and columns >= C to the last aligned
  // group; the clamped rows only feed output rows/cols >= M (never read back) and the clamped columns
  // meet g == 0 (g is zero-padded to a multiple of GR_KC columns by the launcher: gpad, row stride
  // Cpad).  Every wave issues exactly NPW + 1 operations per stage (surplus pieces re-load piece 0 into
  // an unused slot; all four waves DMA the same 128 bytes of g) so that a counted vmcnt(NPW+1) means
  // "everything but the newest stage has landed".
  const long long Cpad = nch * GR_KC;
#define GPSA_GR_STAGE(CH, BUF)                                                               \
  {                                                                                          \
    const long long cb__ = (long long)(CH) * GR_KC;                                          \
    if (ALIGNED) {                                                                           \
      _Pragma("unroll") for (int pc = 0; pc < NPW; ++pc) {                                   \
        const int slot = pc * 4 + w;                                                         \
        const int piece = slot < NPIECE ? slot : 0;                                          \
        const int rb = piece / NKB, kb = piece % NKB;                                        \
        int row = rb * 16 + j;                                                               \
        row = row < M ? row : M - 1;                                                         \
        long long col = cb__ + kb * 16 + kq * 4;                                             \
        col = col < C - 4 ? col : C - 4;                                                     \
        glds16(alpha + (long long)row * C + col,                                             \
               __builtin_amdgcn_readfirstlane(lds_addr(sA_ + (BUF) * SA_STRIDE + slot * 256))); \
      }                                                                                      \
      if (lane < NL * (GR_KC / 4)) {                                                         \
        const int lq__ = min(l0 + lane / (GR_KC / 4), L - 1);                                \
        glds16(g + (long long)lq__ * Cpad + cb__ + (lane % (GR_KC / 4)) * 4,                 \
               __builtin_amdgcn_readfirstlane(lds_addr(sG_ + (BUF) * SG_STRIDE)));           \
      }                                                                                      \
    } else {                                                                                 \
      for (int e = tid; e < NPIECE * 256; e += 256) {                                        \
        const int piece = e >> 8, ln = (e >> 2) & 63, r = e & 3;                             \
        const int rb = piece / NKB, kb = piece % NKB;                                        \
        const int row = rb * 16 + (ln & 15);                                                 \
        const long long col = cb__ + kb * 16 + (ln >> 4) * 4 + r;                            \
        sA_[(BUF) * SA_STRIDE + e] = (row < M && col < C) ? alpha[(long long)row * C + col] : 0.f; \
      }                                                                                      \
      if (tid < NL * GR_KC)                                                                  \
        sG_[(BUF) * SG_STRIDE + tid] = g[(long long)min(l0 + tid / GR_KC, L - 1) * Cpad + cb__ + tid % GR_KC]; \
    }                                                                                        \
  }

  if (ch1 > ch0) GPSA_GR_STAGE(ch0, 0)
  GPSA_DMA_DRAIN();
  __syncthreads();
  int buf = 0;
  for (long long ch = ch0; ch < ch1; ++ch) {
    // the other slot held chunk ch-1: everyone left it before the barrier that ended that iteration
    if (ch + 1 < ch1) GPSA_GR_STAGE(ch + 1, buf ^ 1)
    const float* img = sA_ + buf * SA_STRIDE + lane * 4;
    gram_wave_chunk<MB, NKB, W, NS, NL>(img, sG_ + buf * SG_STRIDE, kq, acc);
    GPSA_DMA_DRAIN();  // chunk ch+1 (issued a whole chunk of MFMAs ago) has landed
    __syncthreads();
    buf ^= 1;
  }
  GPSA_DMA_DRAIN();
#undef GPSA_GR_STAGE
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    if (l0 + q >= L) break;
    float* P = part + ((long long)(l0 + q) * nsplit + sp) * MP * MP;
    gram_wave_store<MB, W, NS>(acc[q], P, j, kq);
  }
}

template <int MB, bool ALIGNED, int NL>
__global__ void __launch_bounds__(256, (MB >= 13 || NL > 1) ? 1 : 2)
gram_mfma_kernel(const float* __restrict__ alpha, const float* __restrict__ g, int M, long long C,
                 int L, int nsplit, float* __restrict__ part) {
  constexpr int NKB = GR_KC / 16, NPIECE = MB * NKB;  // 1-KiB pieces (16 rows x 16 cols) per chunk
  // LDS image of a chunk: piece (rb, kb) at float offset (rb*NKB + kb)*256, stored in MFMA-fragment
  // order: lane j + 16 kq holds alpha[16 rb + j][cb + 16 kb + 4 kq .. +3]  => a fragment read is one
  // conflict-free ds_read_b128 at lane*16 bytes.
  constexpr int NPW = (NPIECE + 3) / 4;  // LDS-DMA pieces per wave per stage (uniform; + 1 for g)
  // two slots: the chunk being multiplied and the next one in flight (a chunk is ~12k MFMA cycles per
  // wave, far longer than the DMA latency, so one stage ahead is enough and the chunks can be big)
  __shared__ __attribute__((aligned(16))) float sA[2][NPW * 4 * 256];
  __shared__ __attribute__((aligned(16))) float sG[2][NL * GR_KC];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  switch (w) {
    case 0: gram_wave_run<MB, ALIGNED, NL, 0>(alpha, g, M, C, L, nsplit, part, &sA[0][0], &sG[0][0]); break;
    case 1: gram_wave_run<MB, ALIGNED, NL, 1>(alpha, g, M, C, L, nsplit, part, &sA[0][0], &sG[0][0]); break;
    case 2: gram_wave_run<MB, ALIGNED, NL, 2>(alpha, g, M, C, L, nsplit, part, &sA[0][0], &sG[0][0]); break;
    default: gram_wave_run<MB, ALIGNED, NL, 3>(alpha, g, M, C, L, nsplit, part, &sA[0][0], &sG[0][0]); break;
  }
}

template <typename TO>
__global__ void gram_reduce_kernel(const float* __restrict__ part, int M, int MP, int L, int nsplit,
                                   TO* __restrict__ out) {
  // grid: (ceil(M / 32) column blocks, M rows, L); threads 32 x 8 (8 rows per block in y)
  const int jj = blockIdx.x * 32 + (threadIdx.x & 31);
  const int i = blockIdx.y * 8 + (threadIdx.x >> 5);
  const int l = blockIdx.z;
  if (i >= M || jj > i) return;
  const float* p = part + (long long)l * nsplit * MP * MP + (long long)i * MP + jj;
  // fp64 output: the partials are widened before they are added; four independent running sums (the
  // partials of an element are MP*MP apart: one chain is one load latency per partial)
  const long long mm = (long long)MP * MP;
  TO s = TO(0), s1 = TO(0), s2 = TO(0), s3 = TO(0);
  int sp = 0;
  for (; sp + 3 < nsplit; sp += 4) {
    s += (TO)p[sp * mm];
    s1 += (TO)p[(sp + 1) * mm];
    s2 += (TO)p[(sp + 2) * mm];
    s3 += (TO)p[(sp + 3) * mm];
  }
  for (; sp < nsplit; ++sp) s += (TO)p[sp * mm];
  s = (s + s1) + (s2 + s3);
  TO* o = out + (long long)l * M * M;
  o[(long long)i * M + jj] = s;
  if (jj != i) o[(long long)jj * M + i] = s;
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

static int gram_mfma_launch(int MBsel, const float* alpha, const float* g, int M, long long C, int L,
                            void* dOmega, int out_dtype, float* ws, hipStream_t st) {
  const int nl = gram_nl(MBsel, L, C);
  const int ns = gram_nsplit(C, (L + nl - 1) / nl);
  const long long Cpad = cdiv(C, GR_KC) * GR_KC;
  float* gpad = ws;
  float* part = ws + ((gram_gpad_floats(C, L) + 63) / 64) * 64;
  pad_rows_kernel<<<(unsigned)cdiv((long long)L * Cpad, 256), 256, 0, st>>>(g, L, C, Cpad, gpad);
  GPSA_LAUNCH_CHECK();
  dim3 grid((unsigned)((L + nl - 1) / nl), (unsigned)ns);
  const bool al = (C % 4 == 0) && (C >= 8) && ((reinterpret_cast<uintptr_t>(alpha) & 15) == 0);
#define GPSA_GRAM_CASE(MBV)                                                                        \
  case MBV:                                                                                        \
    if (nl == 2 && MBV <= 13) {                                                                    \
      if (al) gram_mfma_kernel<MBV, true, (MBV <= 13 ? 2 : 1)><<<grid, 256, 0, st>>>(alpha, gpad, M, C, L, ns, part);  \
      else gram_mfma_kernel<MBV, false, (MBV <= 13 ? 2 : 1)><<<grid, 256, 0, st>>>(alpha, gpad, M, C, L, ns, part);    \
    } else {                                                                                       \
      if (al) gram_mfma_kernel<MBV, true, 1><<<grid, 256, 0, st>>>(alpha, gpad, M, C, L, ns, part); \
      else gram_mfma_kernel<MBV, false, 1><<<grid, 256, 0, st>>>(alpha, gpad, M, C, L, ns, part);   \
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
  dim3 rgrid((unsigned)cdiv(M, 32), (unsigned)cdiv(M, 8), (unsigned)L);
  if (out_dtype == GPSA_F64)
    gram_reduce_kernel<double><<<rgrid, 256, 0, st>>>(part, M, MBsel * 16, L, ns, (double*)dOmega);
  else
    gram_reduce_kernel<float><<<rgrid, 256, 0, st>>>(part, M, MBsel * 16, L, ns, (float*)dOmega);
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
                    hipStream_t st, int layout = 0) {
  const long long tot = (long long)L * MB * 16 * MB * 16;
  const unsigned grid = (unsigned)cdiv(tot, 256);
  if (p_dtype == GPSA_F32)
    pack_panels_kernel<float><<<grid, 256, 0, st>>>((const float*)src, M, MB, L, transpose, dst, layout);
  else if (p_dtype == GPSA_F64)
    pack_panels_kernel<double><<<grid, 256, 0, st>>>((const double*)src, M, MB, L, transpose, dst, layout);
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
  BigAccumArgs aa{Pp, alpha, g, part, M, Mp, L, nrb, ns, C, ctiles, 2.f};
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
        BigQuadArgs qa{Pp, (const float*)alpha, (float*)v, nullptr, M, Mp, L, C, 0};
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
    return (long long)L * MB * 16 * MB * 16 * 4;
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
        BigQuadArgs qa{Pp, alpha, v, W, M, Mp, L, C, 0};
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
static inline int elbo_nct_for(int MB) { return MB == 13 ? 2 : 4; }
static inline bool elbo_path(int M) {
  const int MB = gpsa::mfma_mb_for(M);
  return MB && MB <= 13 && !gpsa::force_generic();
}

int gpsa_quadform_elbo_parts(void) { return gpsa::num_cus() * 2; }

long long gpsa_quadform_elbo_f32_workspace(int M, long long C, int L) {
  if (M < 1 || C < 1 || L < 1 || !elbo_path(M)) return 0;
  const int MB = gpsa::mfma_mb_for(M), nct = elbo_nct_for(MB);
  const long long G = gpsa_quadform_elbo_parts();
  return ((long long)L * MB * 16 * MB * 16 + G * 2 * (long long)MB * 16 * 64 * nct) * 4;
}

int gpsa_quadform_elbo_f32(int omega_dtype, const float* alpha, const void* Omega, int M, long long C, int L,
                           const float* meanT, const double* q, const float* var_u, const float* eps, const float* Y,
                           long long N, int S, const float* noise_u, float* g, float* dmeanT, float* abar, double* part,
                           void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1 || N < 1 || S < 1 || !alpha || !Omega || !meanT || !q || !var_u || !eps || !Y ||
      !noise_u || !g || !dmeanT || !abar || !part)
    return GPSA_EINVAL;
  if (!elbo_path(M)) return GPSA_EUNSUPPORTED;
  if (workspace_bytes < gpsa_quadform_elbo_f32_workspace(M, C, L)) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const int MB = mfma_mb_for(M);
  float* Ppk = (float*)workspace;
  float* slab = Ppk + (long long)L * MB * 16 * MB * 16;
  const int klast = (M - 16 * (MB - 1) <= 8) ? PACK_KSTEP_LAST : 0;
  int rc = pack_any(omega_dtype, Omega, M, MB, L, 0, Ppk, st, klast);
  if (rc) return rc;
  const int gmax = gpsa_quadform_elbo_parts();
  ElboArgs a{Ppk, alpha, M, C, L, meanT, q, var_u, eps, Y, noise_u, N, S, g, dmeanT, abar, slab, part, gmax};
  long long grid = 0;
#define GPSA_ELBO_CASE(MBV, NCTV)                                                                       \
  case MBV: {                                                                                           \
    const long long ntiles = cdiv(C, 64 * NCTV), T = ntiles * L;                                        \
    grid = (long long)num_cus() * ((MBV * NCTV >= 14) ? 1 : 2);                                         \
    if (grid > T) grid = T;                                                                             \
    if (M - 16 * (MBV - 1) <= 8)                                                                        \
      panel_elbo_kernel<MBV, NCTV, 2><<<(unsigned)grid, 256, 0, st>>>(a);                               \
    else                                                                                                \
      panel_elbo_kernel<MBV, NCTV, 4><<<(unsigned)grid, 256, 0, st>>>(a);                               \
    dim3 rg((unsigned)ntiles, ntiles >= 512 ? 8 : (ntiles >= 128 ? 16 : 32));                           \
    panel_slab_reduce_kernel<<<rg, 256, 0, st>>>(slab, M, MBV * 16, 64 * NCTV, C, L, ntiles, (int)grid, abar); \
  } break;
  switch (MB) {
    GPSA_ELBO_CASE(2, 4)
    GPSA_ELBO_CASE(4, 4)
    GPSA_ELBO_CASE(7, 4)
    GPSA_ELBO_CASE(13, 2)
    default:
      return GPSA_EUNSUPPORTED;
  }
#undef GPSA_ELBO_CASE
  GPSA_LAUNCH_CHECK();
  return 0;
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
