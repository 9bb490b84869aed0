// Generic strided-batched dense product (fp32 / fp64), LDS-tiled, optional deterministic split-K.
// Used for the small M x M inducing-point algebra (fp64) and as the shape-generic path next to the
// MFMA panel kernels in quadform.hip.  Replaces torch.matmul / torch.mm calls of
// gpsa/models/vgpsa.py:179-196, 207-210, 227, 302, 430.
#include "internal.hpp"

namespace gpsa {

constexpr int GB_M = 64, GB_N = 64, GB_K = 16;

// second segment of a batch: problems b >= nb0 take their operands / result from these bases (index b - nb0,
// same strides and leading dimensions): one launch over matrices that live in two separate parameter tensors
template <typename TIA, typename TIB, typename TO>
struct GemmSeg2 {
  int nb0;
  const TIA* A;
  const TIB* B;
  TO* C;
};

template <typename T, bool TA, bool TB>
__global__ void __launch_bounds__(256, 4)
gemm_kernel(int m, int n, long long k, T alpha, const T* __restrict__ A, long long lda,
            long long sA, const T* __restrict__ B, long long ldb, long long sB, T beta,
            T* __restrict__ C, long long ldc, long long sC, int splitk, T* __restrict__ part) {
  __shared__ T As[GB_K][GB_M + 4];
  __shared__ T Bs[GB_K][GB_N + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int b = blockIdx.z / splitk, sp = blockIdx.z % splitk;
  const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
  long long kchunk = (k + splitk - 1) / splitk;
  kchunk = (kchunk + GB_K - 1) / GB_K * GB_K;
  const long long kbeg = (long long)sp * kchunk;
  const long long kend = (kbeg + kchunk < k) ? kbeg + kchunk : k;
  const T* Ab = A + (long long)b * sA;
  const T* Bb = B + (long long)b * sB;
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = T(0);

  // register-staged software pipeline: the global loads of K tile t+1 are in flight while tile t is
  // multiplied out of LDS (the small M x M products of the step are latency-, not throughput-bound)
  T ra[4], rb[4];
  auto fetch = [&](long long k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      int r, kk;
      if (TA) { r = e % GB_M; kk = e / GB_M; } else { r = e / GB_K; kk = e % GB_K; }
      const long long gr = m0 + r, gk = k0 + kk;
      T v = T(0);
      if (gr < m && gk < kend) v = TA ? Ab[gk * lda + gr] : Ab[gr * lda + gk];
      ra[i] = v;
      int c, kb;
      if (TB) { kb = e % GB_K; c = e / GB_K; } else { c = e % GB_N; kb = e / GB_N; }
      const long long gc = n0 + c, gk2 = k0 + kb;
      T u = T(0);
      if (gc < n && gk2 < kend) u = TB ? Bb[gc * ldb + gk2] : Bb[gk2 * ldb + gc];
      rb[i] = u;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      int r, kk;
      if (TA) { r = e % GB_M; kk = e / GB_M; } else { r = e / GB_K; kk = e % GB_K; }
      As[kk][r] = ra[i];
      int c, kb;
      if (TB) { kb = e % GB_K; c = e / GB_K; } else { c = e % GB_N; kb = e / GB_N; }
      Bs[kb][c] = rb[i];
    }
  };
  if (kbeg < kend) {
    fetch(kbeg);
    stash();
  }
  __syncthreads();
  for (long long k0 = kbeg; k0 < kend; k0 += GB_K) {
    const bool more = k0 + GB_K < kend;
    if (more) fetch(k0 + GB_K);
#pragma unroll 4
    for (int kk = 0; kk < GB_K; ++kk) {
      T a[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * bb[j];
    }
    __syncthreads();
    if (more) stash();
    __syncthreads();
  }
  if (splitk == 1) {
    T* Cb = C + (long long)b * sC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = m0 + ty * 4 + i;
      if (r >= m) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = n0 + tx * 4 + j;
        if (c >= n) continue;
        T* p = Cb + (long long)r * ldc + c;
        *p = (beta == T(0)) ? alpha * acc[i][j] : alpha * acc[i][j] + beta * (*p);
      }
    }
  } else {
    T* P = part + ((long long)blockIdx.z) * m * n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = m0 + ty * 4 + i;
      if (r >= m) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = n0 + tx * 4 + j;
        if (c < n) P[(long long)r * n + c] = acc[i][j];
      }
    }
  }
}

// fp64 variant on the matrix cores (v_mfma_f64_16x16x4_f64): same 64 x 64 x 16 workgroup tile and the
// same register-staged prefetch, but each wave owns a 32 x 32 quadrant as 2 x 2 MFMA tiles, so one K
// step of 4 costs 4 LDS fragment reads per 4 MFMAs instead of 4 ds_read_b128 per 16 vector FMAs (the
// vector kernel above is LDS-bandwidth-bound at twice its FMA time in fp64).

// TIA / TIB: storage types of the operands (fp32 parameters are read as stored and widened), TO: type of
// C; SYMA: the left operand is A + A^T (square A); diag is added to C[i][i] (split-K == 1 only).  These
// cover Omega = A A^T + 1e-5 I straight from the fp32 parameter and its adjoint dA = (G + G^T) A.
// TC: arithmetic type (double: v_mfma_f64_16x16x4_f64, float: v_mfma_f32_16x16x4_f32 - both take one
// element per lane and K step, lane = (row or column) + 16 * k; only the C/D row layouts differ).
// out[i] = p[i] for i < lim (the leading in-range elements of a contiguous group of 4), else 0
template <typename TI, typename TC>
__device__ __forceinline__ void load4(const TI* __restrict__ p, bool vec_ok, long long lim, TC (&out)[4]) {
  if (vec_ok && lim >= 4) {
    if (sizeof(TI) == 4) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      out[0] = (TC)v.x; out[1] = (TC)v.y; out[2] = (TC)v.z; out[3] = (TC)v.w;
    } else {
      const double2 v0 = *reinterpret_cast<const double2*>(p);
      const double2 v1 = *reinterpret_cast<const double2*>(p + 2);
      out[0] = (TC)v0.x; out[1] = (TC)v0.y; out[2] = (TC)v1.x; out[3] = (TC)v1.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = (i < lim) ? (TC)p[i] : TC(0);
  }
}

template <typename TC> struct MfmaTile;
template <> struct MfmaTile<double> {
  typedef double vec __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ vec mma(double a, double b, vec c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int kq, int r) { return kq + 4 * r; }
};
template <> struct MfmaTile<float> {
  typedef float vec __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ vec mma(float a, float b, vec c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int kq, int r) { return 4 * kq + r; }
};

template <typename TC, bool TA, bool TB, typename TIA = TC, typename TIB = TC, typename TO = TC,
          bool SYMA = false>
__global__ void __launch_bounds__(256, 4)
gemm_mfma_kernel(int m, int n, long long k, TC alpha, const TIA* __restrict__ A, long long lda,
                 long long sA, const TIB* __restrict__ B, long long ldb, long long sB, TC beta,
                 TO* __restrict__ C, long long ldc, long long sC, int splitk, TC* __restrict__ part,
                 TC diag = TC(0), int tri = 0, const TIB* __restrict__ kscale = nullptr, long long sKs = 0,
                 const TIB* __restrict__ cscale = nullptr, long long sCs = 0,
                 GemmSeg2<TIA, TIB, TO> seg2 = GemmSeg2<TIA, TIB, TO>{0x7fffffff, nullptr, nullptr, nullptr},
                 const TIA* __restrict__ A2 = nullptr, const float* __restrict__ kscale32 = nullptr) {
  // A2 + kscale32 (with !TA, batch 1): the left operand is A[m][k] + kscale32[k] * A2[m][k] (same layout as A), formed
  //   as the tile is staged - the exact inducing-point gradient's W = gamma + qbar o alpha, never written out;
  // kscale (with !TA): the left operand is A[m][k] * kscale[b][k] (a column-scaled panel alpha o g contracted
  //   along its columns without materialising it: the Gram product sum_c g a a^T);
  // cscale: the result's column n is scaled by cscale[b][n] (Omega (alpha o g) = (Omega alpha) o g).
  // tri = GEMM_TRI_UPPER_A: op(A) is upper triangular (square, m == k): block row m0 contracts k >= m0 only
  //       (the symmetric quadratic form a^T Omega a = a^T (diag + 2 strict-upper) a at half the flops);
  // tri = GEMM_TRI_LOWER_C: only the blocks of C that touch its lower triangle are computed (a symmetric
  //       product such as sum_c g a a^T; the caller mirrors them) - the others are left untouched.
  // tri = GEMM_TRI_LTL (TA, !TB, A == B square lower triangular, split-K 1): C = A^T A contracts k >= max(i, j)
  //       only; the blocks touching the lower triangle are computed and written to both halves
  //       (K^-1 = L^-T L^-1 at a third of the block products).
  if ((tri == GEMM_TRI_LOWER_C || tri == GEMM_TRI_LTL) &&
      (int)(blockIdx.x * GB_N) > (int)(blockIdx.y * GB_M) + GB_M - 1)
    return;
  typedef typename MfmaTile<TC>::vec acc_t;
  // K tile of 16 (32 was tried: fewer resident workgroups, 20-40 % slower on every shape used here; fetching TWO
  // tiles ahead through a second register set was tried too: 10 % slower at M = 1000, no gain on the small batches
  // (round 4, again: nor on the long-K fp64 products of the warp GPs' backward, which run the matrix pipe 39 % of the
  // time, nor on the skinny fp32 products of the data GP's mean term; K tiles of 32 / 64 are 10 - 50 % slower there;
  // one straight-line K loop per live-tile shape instead of a test in front of every MFMA: fp64 unchanged, the fp32
  // products 12 - 20 % SLOWER - tools/microbench/gemm64_time.py, gemm32_skinny_time.py);
  // a 128 x 128 workgroup tile (4 x 4 MFMA tiles per wave, half the staged bytes per flop) was 2 % slower at
  // M = 500 and 8 % slower at M = 1000 end to end: the 64-tile is not bound by the operand traffic; an XCD-aware
  // block order (blocks with equal dispatch index mod 8 on adjacent tiles, short grid edge fastest) gained 7 % on
  // the alpha-gradient at M = 1000 and lost 8 % on the variance form and more on the fp64 products: not kept)
  constexpr int GK = 16, NG = GK / 16;
  __shared__ TC As[2][GK][GB_M + 4];  // double-buffered: one barrier per K tile
  __shared__ TC Bs[2][GK][GB_N + 4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 32, wn = (w & 1) * 32, j = lane & 15, kq = lane >> 4;
  int b = blockIdx.z / splitk;
  const int sp = blockIdx.z % splitk;
  if (b >= seg2.nb0) {  // block-uniform
    b -= seg2.nb0;
    A = seg2.A;
    B = seg2.B;
    C = seg2.C;
  }
  const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
  long long kchunk = (k + splitk - 1) / splitk;
  kchunk = (kchunk + GK - 1) / GK * GK;
  long long kbeg = (long long)sp * kchunk;
  const long long kend = (kbeg + kchunk < k) ? kbeg + kchunk : k;
  if ((tri == GEMM_TRI_UPPER_A || tri == GEMM_TRI_LTL) && kbeg < m0) kbeg = m0;  // m0: a multiple of the K tile
  const TIA* Ab = A + (long long)b * sA;
  const TIB* Bb = B + (long long)b * sB;
  acc_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) acc[i][jj] = (acc_t){TC(0), TC(0), TC(0), TC(0)};
  const bool lm0 = m0 + wm < m, lm1 = m0 + wm + 16 < m, ln0 = n0 + wn < n, ln1 = n0 + wn + 16 < n;

  // Staging: every thread owns 4 elements of each operand tile that are CONTIGUOUS in memory (along K
  // for a row-major left / transposed right operand, along the row / column index otherwise) and
  // fetches them with 16-byte loads when the whole group is in range and aligned; edge groups and
  // unaligned operands take guarded scalar loads of the same elements.
  TC ra[NG][4], rb[NG][4];
  const bool va_ok = !SYMA && (lda % (16 / (int)sizeof(TIA)) == 0) && (sA % (16 / (int)sizeof(TIA)) == 0) &&
                     ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool vb_ok = (ldb % (16 / (int)sizeof(TIB)) == 0) && (sB % (16 / (int)sizeof(TIB)) == 0) &&
                     ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  // A: (row ar + i*ari, k ak + i*aki), i = 0..3     B: (k bk + i*bki, col bc + i*bci)
  const int ar = TA ? (tid % 16) * 4 : tid / 4, ak = TA ? tid / 16 : (tid % 4) * 4;
  const int bc = TB ? tid / 4 : (tid % 16) * 4, bk = TB ? (tid % 4) * 4 : tid / 16;
#define GPSA_G64_FETCH(K0)                                                                \
  _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) {                                     \
    const long long gr = m0 + ar, gk = (K0) + ak + 16 * g_;                               \
    const TIA* pa = TA ? Ab + gk * lda + gr : Ab + gr * lda + gk;                         \
    const long long lim_a = TA ? ((gk < kend) ? (long long)m - gr : 0)                    \
                               : ((gr < m) ? kend - gk : 0);                              \
    load4<TIA, TC>(pa, va_ok, lim_a, ra[g_]);                                             \
    if (!TA && kscale != nullptr) {                                                       \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                       \
        if (i < lim_a) ra[g_][i] *= (TC)kscale[(long long)b * sKs + gk + i];              \
    }                                                                                     \
    if (!TA && A2 != nullptr) {                                                           \
      TC r2[4];                                                                           \
      load4<TIA, TC>(A2 + gr * lda + gk, va_ok, lim_a, r2);                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                       \
        if (i < lim_a) ra[g_][i] += (TC)kscale32[gk + i] * r2[i];                         \
    }                                                                                     \
    if (SYMA) {                                                                           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                       \
        if (i < lim_a) ra[g_][i] += (TC)(TA ? Ab[(gr + i) * lda + gk] : Ab[(gk + i) * lda + gr]); \
    }                                                                                     \
    const long long gc = n0 + bc, gk2 = (K0) + bk + 16 * g_;                              \
    const TIB* pb = TB ? Bb + gc * ldb + gk2 : Bb + gk2 * ldb + gc;                       \
    const long long lim_b = TB ? ((gc < n) ? kend - gk2 : 0)                              \
                               : ((gk2 < kend) ? (long long)n - gc : 0);                  \
    load4<TIB, TC>(pb, vb_ok, lim_b, rb[g_]);                                             \
  }
#define GPSA_G64_STASH(BUF)                                                               \
  _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) {                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                       \
      if (TA) As[BUF][ak + 16 * g_][ar + i] = ra[g_][i];                                  \
      else As[BUF][ak + 16 * g_ + i][ar] = ra[g_][i];                                     \
      if (TB) Bs[BUF][bk + 16 * g_ + i][bc] = rb[g_][i];                                  \
      else Bs[BUF][bk + 16 * g_][bc + i] = rb[g_][i];                                     \
    }                                                                                     \
  }
  if (kbeg < kend) {
    GPSA_G64_FETCH(kbeg)
    GPSA_G64_STASH(0)
  }
  __syncthreads();
  int cur = 0;
  for (long long k0 = kbeg; k0 < kend; k0 += GK) {
    const bool more = k0 + GK < kend;
    if (more) GPSA_G64_FETCH(k0 + GK)
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
      const int kk = ks * 4 + kq;
      const TC a0 = As[cur][kk][wm + j], a1 = As[cur][kk][wm + 16 + j];
      const TC b0 = Bs[cur][kk][wn + j], b1 = Bs[cur][kk][wn + 16 + j];
      // MFMA tiles that lie wholly outside the matrix are skipped (wave-uniform): M = 200 costs
      // 3.5 x 3.5 tile units instead of 4 x 4
      if (lm0 && ln0) acc[0][0] = MfmaTile<TC>::mma(a0, b0, acc[0][0]);
      if (lm0 && ln1) acc[0][1] = MfmaTile<TC>::mma(a0, b1, acc[0][1]);
      if (lm1 && ln0) acc[1][0] = MfmaTile<TC>::mma(a1, b0, acc[1][0]);
      if (lm1 && ln1) acc[1][1] = MfmaTile<TC>::mma(a1, b1, acc[1][1]);
    }
    // the other buffer was last read one tile ago, before the previous barrier
    if (more) GPSA_G64_STASH(cur ^ 1)
    __syncthreads();
    cur ^= 1;
  }
#undef GPSA_G64_FETCH
#undef GPSA_G64_STASH
  // C/D layout: col = lane & 15; row = (lane >> 4) + 4 * reg (fp64) or 4 * (lane >> 4) + reg (fp32)
  // beta != 0: all sixteen old values of this thread are requested first (clamped addresses, one uniform branch) -
  // read inside the store loop, each load sat behind the previous store: sixteen memory round trips in a row at the
  // end of a workgroup that lives for four K tiles (the data GP's abar += delta dmean^T: 66 -> 57 us for 100 MB;
  // requesting ALL of a short K range's operand tiles up front as well changed nothing: 49 vs 51 us without beta)
  TC cold[2][2][4];
  const bool rmw = splitk == 1 && beta != TC(0);
  if (rmw) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = m0 + wm + tm * 16 + MfmaTile<TC>::row(kq, r), col = n0 + wn + tn * 16 + j;
          row = row < m ? row : m - 1;
          col = col < n ? col : n - 1;
          cold[tm][tn][r] = (TC)C[(long long)b * sC + (long long)row * ldc + col];
        }
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + tm * 16 + MfmaTile<TC>::row(kq, r), col = n0 + wn + tn * 16 + j;
        if (row < m && col < n) {
          TC y = acc[tm][tn][r];
          if (cscale != nullptr) y *= (TC)cscale[(long long)b * sCs + col];
          if (splitk == 1) {
            TO* p = C + (long long)b * sC + (long long)row * ldc + col;
            TC v = rmw ? alpha * y + beta * cold[tm][tn][r] : alpha * y;
            if (row == col) v += diag;
            *p = (TO)v;
            if (tri == GEMM_TRI_LTL) C[(long long)b * sC + (long long)col * ldc + row] = (TO)v;
          } else {
            part[((long long)blockIdx.z * m + row) * n + col] = y;
          }
        }
      }
}

// 64 outputs per block, four waves each summing every fourth partial with four independent accumulators
// (the partials of one output are L*m*n apart: a single running sum is one load latency per partial)
template <typename T>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const T* __restrict__ part, int batch, int splitk, int m, int n, T alpha, T beta,
                     T* __restrict__ C, long long ldc, long long sC) {
  __shared__ T red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long long mn = (long long)m * n;
  const long long idx = blockIdx.x * 64LL + lane;
  const bool ok = idx < mn * batch;
  const int b = ok ? (int)(idx / mn) : 0;
  const long long e = ok ? idx % mn : 0;
  T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
  if (ok) {
    const T* p = part + (long long)b * splitk * mn + e;
    int sp = grp;
    for (; sp + 12 < splitk; sp += 16) {
      a0 += p[sp * mn];
      a1 += p[(sp + 4) * mn];
      a2 += p[(sp + 8) * mn];
      a3 += p[(sp + 12) * mn];
    }
    for (; sp < splitk; sp += 4) a0 += p[sp * mn];
  }
  red[grp][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && ok) {
    const T s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int r = (int)(e / n), c = (int)(e % n);
    T* q = C + (long long)b * sC + (long long)r * ldc + c;
    *q = (beta == T(0)) ? alpha * s : alpha * s + beta * (*q);
  }
}

static inline bool gemm_force_vector() {
  static const bool v = [] { const char* e = getenv("GPSA_GEMM_VECTOR"); return e && e[0] == '1'; }();
  return v;
}

template <typename T>
int gemm_launch_tri(int transA, int transB, int m, int n, long long k, double alpha, const T* A,
                    long long lda, long long sA, const T* B, long long ldb, long long sB, double beta,
                    T* C, long long ldc, long long sC, int batch, int splitk, void* ws,
                    long long ws_bytes, hipStream_t st, int tri);
template <typename T>
int gemm_launch_scaled(int transA, int transB, int m, int n, long long k, double alpha, const T* A,
                       long long lda, long long sA, const T* B, long long ldb, long long sB, double beta,
                       T* C, long long ldc, long long sC, int batch, int splitk, void* ws,
                       long long ws_bytes, hipStream_t st, int tri, const T* kscale, long long sKs,
                       const T* cscale, long long sCs);

template <typename T>
int gemm_launch(int transA, int transB, int m, int n, long long k, double alpha, const T* A,
                long long lda, long long sA, const T* B, long long ldb, long long sB, double beta,
                T* C, long long ldc, long long sC, int batch, int splitk, void* ws,
                long long ws_bytes, hipStream_t st) {
  return gemm_launch_tri<T>(transA, transB, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, batch, splitk,
                            ws, ws_bytes, st, GEMM_TRI_NONE);
}

template <typename T>
int gemm_launch_tri(int transA, int transB, int m, int n, long long k, double alpha, const T* A,
                    long long lda, long long sA, const T* B, long long ldb, long long sB, double beta,
                    T* C, long long ldc, long long sC, int batch, int splitk, void* ws,
                    long long ws_bytes, hipStream_t st, int tri) {
  return gemm_launch_scaled<T>(transA, transB, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, batch,
                               splitk, ws, ws_bytes, st, tri, nullptr, 0, nullptr, 0);
}

// + kscale [batch][k] on the (non-transposed) left operand's K index, cscale [batch][n] on the result's columns
template <typename T>
int gemm_launch_scaled(int transA, int transB, int m, int n, long long k, double alpha, const T* A,
                       long long lda, long long sA, const T* B, long long ldb, long long sB, double beta,
                       T* C, long long ldc, long long sC, int batch, int splitk, void* ws,
                       long long ws_bytes, hipStream_t st, int tri, const T* kscale, long long sKs,
                       const T* cscale, long long sCs) {
  if (m < 1 || n < 1 || k < 1 || batch < 1 || splitk < 1) return GPSA_EINVAL;
  if (kscale != nullptr && transA) return GPSA_EINVAL;
  if (splitk != 1 && (tri == GEMM_TRI_UPPER_A || tri == GEMM_TRI_LTL)) return GPSA_EINVAL;
  if (tri == GEMM_TRI_LTL && (!transA || transB || m != n || m != k || A != B)) return GPSA_EINVAL;
  if ((long long)batch * splitk > 65535) return GPSA_EINVAL;
  T* part = nullptr;
  if (splitk > 1) {
    if (ws_bytes < (long long)batch * splitk * m * n * (long long)sizeof(T)) return GPSA_EWORKSPACE;
    part = reinterpret_cast<T*>(ws);
  }
  dim3 grid((unsigned)cdiv(n, GB_N), (unsigned)cdiv(m, GB_M), (unsigned)(batch * splitk));
#define GPSA_GEMM_CASE(TA, TB)                                                               \
  gemm_kernel<T, TA, TB><<<grid, 256, 0, st>>>(m, n, k, (T)alpha, A, lda, sA, B, ldb, sB,     \
                                               (T)beta, C, ldc, sC, splitk, part)
#define GPSA_GEMMX_CASE(TA, TB)                                                                     \
  gemm_mfma_kernel<T, TA, TB><<<grid, 256, 0, st>>>(m, n, k, (T)alpha, A, lda, sA, B, ldb, sB, (T)beta, \
                                                    C, ldc, sC, splitk, part, T(0), tri, kscale, sKs, cscale, sCs)
  // products with at least one MFMA tile in each direction run on the matrix cores (always with a triangle
  // mode or a scale vector: only that kernel knows them)
  // (round 4: also products with fewer than 16 rows or columns - the matrix-core kernel pads them inside a tile and
  //  is 1.1 - 2.3 times faster than the vector kernel on every such product of a step: the ten-latent-GP mean term
  //  96 -> 42 us, its adjoint 122 -> 61, the LMC products 209 -> 114 and 255 -> 126, the warp GPs' [M, 2] gradient
  //  14.0 -> 12.6; GPSA_GEMM_MFMA_MIN=16 restores the old rule)
  static const int mfma_min = [] { const char* e = getenv("GPSA_GEMM_MFMA_MIN"); return e ? atoi(e) : 1; }();
  const bool mfma = (m >= mfma_min && n >= mfma_min && !gemm_force_vector()) || tri != GEMM_TRI_NONE ||
                    kscale != nullptr || cscale != nullptr;
  if (cscale != nullptr && splitk != 1) return GPSA_EINVAL;  // the split-K reduce does not know the scale
  if (mfma) {
    if (!transA && !transB) GPSA_GEMMX_CASE(false, false);
    else if (transA && !transB) GPSA_GEMMX_CASE(true, false);
    else if (!transA && transB) GPSA_GEMMX_CASE(false, true);
    else GPSA_GEMMX_CASE(true, true);
  } else if (!transA && !transB) GPSA_GEMM_CASE(false, false);
  else if (transA && !transB) GPSA_GEMM_CASE(true, false);
  else if (!transA && transB) GPSA_GEMM_CASE(false, true);
  else GPSA_GEMM_CASE(true, true);
#undef GPSA_GEMM_CASE
#undef GPSA_GEMMX_CASE
  GPSA_LAUNCH_CHECK();
  if (splitk > 1) {
    const long long tot = (long long)m * n * batch;
    splitk_reduce_kernel<T><<<(unsigned)cdiv(tot, 64), 256, 0, st>>>(part, batch, splitk, m, n,
                                                                     (T)alpha, (T)beta, C, ldc, sC);
    GPSA_LAUNCH_CHECK();
  }
  return 0;
}

// Products accumulated in fp64 on the matrix cores from operands stored in either precision, result
// stored in either precision (split-K partials in fp64).  The step engine's few mixed products: K^-1 times
// an fp32 gradient panel, the C-long product of two fp32 panels that must be ADDED in fp64.
template <typename TO>
__global__ void __launch_bounds__(256)
splitk_reduce64_kernel(const double* __restrict__ part, int batch, int splitk, int m, int n, double alpha,
                       double beta, TO* __restrict__ C, long long ldc, long long sC) {
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long long mn = (long long)m * n;
  const long long idx = blockIdx.x * 64LL + lane;
  const bool ok = idx < mn * batch;
  const int b = ok ? (int)(idx / mn) : 0;
  const long long e = ok ? idx % mn : 0;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (ok) {
    const double* p = part + (long long)b * splitk * mn + e;
    int sp = grp;
    for (; sp + 12 < splitk; sp += 16) {
      a0 += p[sp * mn];
      a1 += p[(sp + 4) * mn];
      a2 += p[(sp + 8) * mn];
      a3 += p[(sp + 12) * mn];
    }
    for (; sp < splitk; sp += 4) a0 += p[sp * mn];
  }
  red[grp][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && ok) {
    const double s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int r = (int)(e / n), c = (int)(e % n);
    TO* q = C + (long long)b * sC + (long long)r * ldc + c;
    *q = (TO)((beta == 0.0) ? alpha * s : alpha * s + beta * (double)(*q));
  }
}

template <typename TIA, typename TIB, typename TO>
int gemm64_launch(int transA, int transB, int m, int n, long long k, double alpha, const TIA* A,
                  long long lda, long long sA, const TIB* B, long long ldb, long long sB, double beta,
                  TO* C, long long ldc, long long sC, int batch, int splitk, void* ws, long long ws_bytes,
                  hipStream_t st) {
  if (m < 1 || n < 1 || k < 1 || batch < 1 || splitk < 1) return GPSA_EINVAL;
  if ((long long)batch * splitk > 65535) return GPSA_EINVAL;
  double* part = nullptr;
  if (splitk > 1) {
    if (ws_bytes < (long long)batch * splitk * m * n * 8) return GPSA_EWORKSPACE;
    part = reinterpret_cast<double*>(ws);
  }
  dim3 grid((unsigned)cdiv(n, GB_N), (unsigned)cdiv(m, GB_M), (unsigned)(batch * splitk));
#define GPSA_G64X(TA, TB)                                                                              \
  gemm_mfma_kernel<double, TA, TB, TIA, TIB, TO, false><<<grid, 256, 0, st>>>(                          \
      m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, splitk, part, 0.0)
  if (!transA && !transB) GPSA_G64X(false, false);
  else if (transA && !transB) GPSA_G64X(true, false);
  else if (!transA && transB) GPSA_G64X(false, true);
  else GPSA_G64X(true, true);
#undef GPSA_G64X
  GPSA_LAUNCH_CHECK();
  if (splitk > 1) {
    const long long tot = (long long)m * n * batch;
    splitk_reduce64_kernel<TO><<<(unsigned)cdiv(tot, 64), 256, 0, st>>>(part, batch, splitk, m, n, alpha, beta,
                                                                        C, ldc, sC);
    GPSA_LAUNCH_CHECK();
  }
  return 0;
}
#define GPSA_G64_INST(TIA, TIB, TO)                                                                          \
  template int gemm64_launch<TIA, TIB, TO>(int, int, int, int, long long, double, const TIA*, long long,      \
                                           long long, const TIB*, long long, long long, double, TO*, long long, \
                                           long long, int, int, void*, long long, hipStream_t);
GPSA_G64_INST(double, float, float)   // gamma = K^-1 abar beyond the projection kernel's size
GPSA_G64_INST(float, float, double)   // dK_uu = -W alpha^T on few columns; dc ddc^T
GPSA_G64_INST(double, double, float)  // alpha = K^-1 K_uf stored fp32 beyond the projection kernel's size
#undef GPSA_G64_INST

// C += alpha (A + d o A2) B^T, [m, k] x [n, k]^T in fp64 with deterministic split-K (exact_dkuu below)
static int gemm_sum2_launch(int m, int n, long long k, double alpha, const double* A, const double* A2, const float* d,
                            const double* B, long long ld, double* C, long long ldc, int splitk, void* ws,
                            long long ws_bytes, hipStream_t st) {
  if (m < 1 || n < 1 || k < 1 || splitk < 1 || splitk > 65535) return GPSA_EINVAL;
  double* part = nullptr;
  if (splitk > 1) {
    if (ws_bytes < (long long)splitk * m * n * 8) return GPSA_EWORKSPACE;
    part = reinterpret_cast<double*>(ws);
  }
  dim3 grid((unsigned)cdiv(n, GB_N), (unsigned)cdiv(m, GB_M), (unsigned)splitk);
  gemm_mfma_kernel<double, false, true, double, double, double, false><<<grid, 256, 0, st>>>(
      m, n, k, alpha, A, ld, 0, B, ld, 0, 1.0, C, ldc, 0, splitk, part, 0.0, 0, nullptr, 0, nullptr, 0,
      GemmSeg2<double, double, double>{0x7fffffff, nullptr, nullptr, nullptr}, A2, d);
  GPSA_LAUNCH_CHECK();
  if (splitk > 1) {
    splitk_reduce_kernel<double><<<(unsigned)cdiv((long long)m * n, 64), 256, 0, st>>>(part, 1, splitk, m, n, alpha, 1.0,
                                                                                       C, ldc, 0);
    GPSA_LAUNCH_CHECK();
  }
  return 0;
}

// explicit instantiations used from other translation units
template int gemm_launch_scaled<float>(int, int, int, int, long long, double, const float*, long long, long long,
                                       const float*, long long, long long, double, float*, long long, long long,
                                       int, int, void*, long long, hipStream_t, int, const float*, long long,
                                       const float*, long long);
template int gemm_launch_scaled<double>(int, int, int, int, long long, double, const double*, long long, long long,
                                        const double*, long long, long long, double, double*, long long, long long,
                                        int, int, void*, long long, hipStream_t, int, const double*, long long,
                                        const double*, long long);
template int gemm_launch_tri<float>(int, int, int, int, long long, double, const float*, long long,
                                    long long, const float*, long long, long long, double, float*,
                                    long long, long long, int, int, void*, long long, hipStream_t, int);
template int gemm_launch_tri<double>(int, int, int, int, long long, double, const double*, long long,
                                     long long, const double*, long long, long long, double, double*,
                                     long long, long long, int, int, void*, long long, hipStream_t, int);
template int gemm_launch<float>(int, int, int, int, long long, double, const float*, long long,
                                long long, const float*, long long, long long, double, float*,
                                long long, long long, int, int, void*, long long, hipStream_t);
template int gemm_launch<double>(int, int, int, int, long long, double, const double*, long long,
                                 long long, const double*, long long, long long, double, double*,
                                 long long, long long, int, int, void*, long long, hipStream_t);

// ---- thin update: out[M,C] += A[M,L] B[L,C], L small, C long (round 6) ------------------------------------------
// The data GP's "abar += delta_F dmean^T" (M = 200, L = 50, C = 100k): 180 MB that have to move once.  The tiled
// product above takes it as 4 x 1563 tiles of 64 x 64 and reads B once per row tile (240 MB, 59 us).  Here a
// workgroup owns 64 columns over ALL rows: A sits in LDS (row stride = 4 mod 8 words: the fragment reads are
// conflict-free), a wave keeps its 16 columns of B as MFMA fragments and walks the row tiles.  The product is formed
// TRANSPOSED (out^T tile = B^T A^T: the operands of the instruction swapped), so a lane's four accumulator values are
// four CONSECUTIVE COLUMNS of one row: one 16-byte load and store per lane and tile, and the next four tiles' loads
// are in flight while the current four are multiplied (a first version with 4-byte accesses and two tiles at a time
// ran at the tiled product's speed: 56 us, 3.2 TB/s - bytes in flight, not arithmetic).
template <int KS>
__global__ void __launch_bounds__(256, 3) thin_update_kernel(const float* __restrict__ A, int M, int L,
                                                          const float* __restrict__ B, long long C,
                                                          float* __restrict__ out, long long ntiles) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int ST = (KS % 2) ? 4 * KS : 4 * KS + 4;
  constexpr int G = 4;  // row tiles per group
  extern __shared__ float As[];  // [MB * 16][ST]
  const int MB = (M + 15) / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  for (int i = tid; i < MB * 16 * ST; i += 256) As[i] = 0.f;
  __syncthreads();
  // (the fill is a chain of cache latencies per thread if its loads are requested one at a time: 40 of them made the
  //  kernel no faster than the tiled product - 16-byte loads, several in flight)
  if ((M * L) % 4 == 0 && ((uintptr_t)A & 15) == 0) {
    const int n4 = M * L / 4;
#pragma unroll 4
    for (int i = tid; i < n4; i += 256) {
      const f32x4 v = reinterpret_cast<const f32x4*>(A)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) As[((4 * i + e) / L) * ST + ((4 * i + e) % L)] = v[e];
    }
  } else {
#pragma unroll 8
    for (int i = tid; i < M * L; i += 256) As[(i / L) * ST + (i % L)] = A[i];
  }
  __syncthreads();
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long cb = t * 64 + w * 16;        // the wave's 16 columns
    const long long c4 = cb + 4 * kq;            // this lane's four columns of out (C % 4 == 0: all or none exist)
    const bool ok4 = c4 < C;
    const long long cj = cb + j;                 // this lane's column of B
    const bool okj = cj < C;
    float b[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + kq;
      const float v = B[(long long)(k < L ? k : L - 1) * C + (okj ? cj : 0)];  // (clamped, unconditional)
      b[ks] = (k < L && okj) ? v : 0.f;
    }
    // (no branches in the group: a tile past the last one repeats the last one - its loads hit the cache, its products
    //  are dropped; a branch around each tile cut the MFMA chains into blocks of one: 135 us)
    f32x4 cur[G], nxt[G];
#define GPSA_TU_LOAD(DST, RT0)                                                                                 \
  _Pragma("unroll") for (int u = 0; u < G; ++u) {                                                              \
    const int r = 16 * min((RT0) + u, MB - 1) + j;                                                             \
    DST[u] = *reinterpret_cast<const f32x4*>(out + (long long)(r < M ? r : M - 1) * C + (ok4 ? c4 : 0));       \
  }
    GPSA_TU_LOAD(cur, 0)
    for (int rt0 = 0; rt0 < MB; rt0 += G) {
      GPSA_TU_LOAD(nxt, rt0 + G)  // (past the end: the last tile again)
      const float* ap[G];
#pragma unroll
      for (int u = 0; u < G; ++u) ap[u] = As + (16 * min(rt0 + u, MB - 1) + j) * ST + kq;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int u = 0; u < G; ++u)  // out^T tile: rows = the wave's columns (B^T fragment), columns = the tile's rows (A^T)
          cur[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ks], ap[u][4 * ks], cur[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int r = 16 * (rt0 + u) + j;
        if (rt0 + u < MB && r < M && ok4) *reinterpret_cast<f32x4*>(out + (long long)r * C + c4) = cur[u];
      }
#pragma unroll
      for (int u = 0; u < G; ++u) cur[u] = nxt[u];
    }
#undef GPSA_TU_LOAD
  }
}

static inline bool thin_update_ok(int M, int L, long long C) {
  static const bool off = [] { const char* e = getenv("GPSA_THIN_UPDATE"); return e && e[0] == '0'; }();
  return !off && M >= 1 && M <= 256 && L >= 1 && L <= 64 && C >= 4096 && C % 4 == 0;
}

}  // namespace gpsa

extern "C" {

long long gpsa_gemm_workspace(int dtype, int m, int n, int batch, int splitk) {
  if (splitk <= 1) return 0;
  return (long long)batch * splitk * m * n * (dtype == GPSA_F64 ? 8 : 4);
}

/* split of the C-long contraction: many short K slices (measured at the headline size, round 3: 32 slices 301 us,
 * 64: 200 us, 128 / 256 the same) */
static int exact_dkuu_splitk(int M, long long C) {
  const long long tiles = cdiv(M, gpsa::GB_M) * cdiv(M, gpsa::GB_N);
  long long s = cdiv(512, tiles > 0 ? tiles : 1);
  if (s > 256) s = 256;
  const long long want = C / 64 < 64 ? C / 64 : 64;
  if (s < want) s = want;
  if (s > C / 64) s = C / 64;
  return (int)(s < 1 ? 1 : s);
}
long long gpsa_exact_dkuu_workspace(int M, long long C) {
  const int sk = exact_dkuu_splitk(M, C);
  return sk > 1 ? (long long)sk * M * M * 8 : 0;
}
int gpsa_exact_dkuu_f64(const double* G, const double* A, const float* d, int M, long long C, double* dK,
                        void* workspace, long long workspace_bytes, void* stream) {
  if (!G || !A || !d || !dK || M < 1 || C < 1) return GPSA_EINVAL;
  return gpsa::gemm_sum2_launch(M, M, C, -1.0, G, A, d, A, C, dK, M, exact_dkuu_splitk(M, C), workspace,
                                workspace_bytes, as_stream(stream));
}

int gpsa_gemm(int dtype, int transA, int transB, int m, int n, long long k, double alpha,
              const void* A, long long lda, long long strideA, const void* B, long long ldb,
              long long strideB, double beta, void* C, long long ldc, long long strideC, int batch,
              int splitk, void* workspace, long long workspace_bytes, void* stream) {
  if (dtype == GPSA_F32)
    return gpsa::gemm_launch<float>(transA, transB, m, n, k, alpha, (const float*)A, lda, strideA,
                                    (const float*)B, ldb, strideB, beta, (float*)C, ldc, strideC,
                                    batch, splitk, workspace, workspace_bytes, as_stream(stream));
  if (dtype == GPSA_F64)
    return gpsa::gemm_launch<double>(transA, transB, m, n, k, alpha, (const double*)A, lda,
                                     strideA, (const double*)B, ldb, strideB, beta, (double*)C,
                                     ldc, strideC, batch, splitk, workspace, workspace_bytes,
                                     as_stream(stream));
  return GPSA_EINVAL;
}

/* Omega[b] = A[b] A[b]^T + jitter I in fp64 from the fp32 parameter, one launch */
int gpsa_omega_fwd(const float* A, int M, int batch, double jitter, double* Omega, void* stream) {
  return gpsa_omega_fwd2(A, batch, Omega, nullptr, 0, nullptr, M, jitter, stream);
}

/* the same over two parameter tensors in ONE launch (the warp GPs' and a modality's variational factors):
 * Omega0[b] from A0[b], b < n0; Omega1[b] from A1[b], b < n1 (n1 == 0: one segment) */
int gpsa_omega_fwd2(const float* A0, int n0, double* Omega0, const float* A1, int n1, double* Omega1, int M,
                    double jitter, void* stream) {
  if (M < 1 || n0 < 1 || n1 < 0) return GPSA_EINVAL;
  {
    const int rc = gpsa::omega_fwd_dma_launch(A0, n0, Omega0, A1, n1, Omega1, M, jitter, as_stream(stream));
    if (rc != GPSA_EUNSUPPORTED) return rc;
  }
  const long long mm = (long long)M * M;
  dim3 grid((unsigned)cdiv(M, gpsa::GB_N), (unsigned)cdiv(M, gpsa::GB_M), (unsigned)(n0 + n1));
  gpsa::GemmSeg2<float, float, double> seg{n1 > 0 ? n0 : 0x7fffffff, A1, A1, Omega1};
  gpsa::gemm_mfma_kernel<double, false, true, float, float, double, false><<<grid, 256, 0, as_stream(stream)>>>(
      M, M, M, 1.0, A0, M, mm, A0, M, mm, 0.0, Omega0, M, mm, 1, nullptr, jitter, 0, nullptr, 0, nullptr, 0, seg);
  GPSA_LAUNCH_CHECK();
  return 0;
}

/* dA[b] = (G[b] + G[b]^T) A[b]  (fp64 G, fp32 A, fp32 dA): adjoint of gpsa_omega_fwd, one launch.
 * symmetric != 0: the caller guarantees G = G^T (every gradient this package produces for Omega is:
 * mirrored Gram sums, differences of inverses of symmetric matrices) and dA = 2 G A skips the strided
 * reads of G^T. */
int gpsa_omega_bwd(const double* G, const float* A, int M, int batch, int symmetric, float* dA,
                   void* stream) {
  return gpsa_omega_bwd2(G, A, dA, batch, nullptr, nullptr, nullptr, 0, M, symmetric, stream);
}

/* two segments in one launch, as gpsa_omega_fwd2 */
int gpsa_omega_bwd2(const double* G0, const float* A0, float* dA0, int n0, const double* G1, const float* A1,
                    float* dA1, int n1, int M, int symmetric, void* stream) {
  if (M < 1 || n0 < 1 || n1 < 0) return GPSA_EINVAL;
  const long long mm = (long long)M * M;
  dim3 grid((unsigned)cdiv(M, gpsa::GB_N), (unsigned)cdiv(M, gpsa::GB_M), (unsigned)(n0 + n1));
  if (symmetric) {
    const int rc = gpsa::omega_bwd_dma_launch(G0, A0, dA0, n0, G1, A1, dA1, n1, M, as_stream(stream));
    if (rc != GPSA_EUNSUPPORTED) return rc;
  }
  gpsa::GemmSeg2<double, float, float> seg{n1 > 0 ? n0 : 0x7fffffff, G1, A1, dA1};
  if (symmetric)
    gpsa::gemm_mfma_kernel<double, false, false, double, float, float, false><<<grid, 256, 0, as_stream(stream)>>>(
        M, M, M, 2.0, G0, M, mm, A0, M, mm, 0.0, dA0, M, mm, 1, nullptr, 0.0, 0, nullptr, 0, nullptr, 0, seg);
  else
    gpsa::gemm_mfma_kernel<double, false, false, double, float, float, true><<<grid, 256, 0, as_stream(stream)>>>(
        M, M, M, 1.0, G0, M, mm, A0, M, mm, 0.0, dA0, M, mm, 1, nullptr, 0.0, 0, nullptr, 0, nullptr, 0, seg);
  GPSA_LAUNCH_CHECK();
  return 0;
}


/* out[M,C] += A[M,L] B[L,C] (fp32, row-major, contiguous; matrix cores): the thin-inner-dimension update of a long
 * panel in ONE pass over it.  GPSA_EUNSUPPORTED outside M <= 256, L <= 64, C >= 4096 (use gpsa_gemm with beta = 1). */
int gpsa_thin_update_f32(const float* A, int M, int L, const float* B, long long C, float* out, void* stream) {
  using namespace gpsa;
  if (!A || !B || !out || M < 1 || L < 1 || C < 1) return GPSA_EINVAL;
  if (!thin_update_ok(M, L, C) || ((uintptr_t)out & 15)) return GPSA_EUNSUPPORTED;  // (16-byte accesses of out)
  hipStream_t st = as_stream(stream);
  const long long ntiles = cdiv(C, 64);
  const int KS = (L + 3) / 4, MB = (M + 15) / 16;
  const int ST = (KS % 2) ? 4 * KS : 4 * KS + 4;
  const size_t lds = (size_t)MB * 16 * ST * 4;
  // persistent: the A fill (M L scattered words) is paid once per workgroup; as many workgroups as stay resident
  long long per_cu = (long long)(160 * 1024) / (long long)(lds + 1024);
  per_cu = per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu);  // (launch bounds: three)
  long long grid = per_cu * num_cus();
  if (grid > ntiles) grid = ntiles;
#define GPSA_TU(KS_)                                                                                   \
  case KS_: {                                                                                          \
    static per_device_flag flag;                                                                       \
    bool& set = flag.here();                                                                           \
    if (!set && lds > 65536) {                                                                         \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_update_kernel<KS_>),                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)    \
        return GPSA_EUNSUPPORTED;                                                                      \
      set = true;                                                                                      \
    }                                                                                                  \
    thin_update_kernel<KS_><<<(unsigned)grid, 256, lds, st>>>(A, M, L, B, C, out, ntiles);             \
  } break;
  switch (KS) {
    GPSA_TU(1) GPSA_TU(2) GPSA_TU(3) GPSA_TU(4) GPSA_TU(5) GPSA_TU(6) GPSA_TU(7) GPSA_TU(8)
    GPSA_TU(9) GPSA_TU(10) GPSA_TU(11) GPSA_TU(12) GPSA_TU(13) GPSA_TU(14) GPSA_TU(15) GPSA_TU(16)
    default: return GPSA_EUNSUPPORTED;
  }
#undef GPSA_TU
  GPSA_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
