// fp64 inducing-point factorisations: batched Cholesky and lower-triangular inverse, one workgroup
// per M x M matrix.  For M <= 200 the lower triangle lives packed in LDS (M(M+1)/2 * 8 B = 160.8 KB
// of the CU's 160 KiB at M = 200); larger M runs the same algorithm on the L2-resident matrix.
// Replaces torch.cholesky (gpsa/models/vgpsa.py:257, 320, 394, 412) and the triangular solves of
// torch.cholesky_solve (vgpsa.py:177).  Also: batched dot product and diagonal shift for the KL terms.
#include "common.hpp"
#include "internal.hpp"

namespace gpsa {

constexpr int LA_THREADS = 1024;
constexpr int LA_PACKED_MAX = 200;  // packed lower triangle + one row of scratch fits in 160 KiB

struct PackedLower {  // LDS, lower triangle only
  double* p;
  __device__ __forceinline__ double& at(int i, int j) const { return p[(i * (i + 1) >> 1) + j]; }
};
struct DenseGlobal {  // global row-major [M][M]
  double* p;
  int M;
  __device__ __forceinline__ double& at(int i, int j) const { return p[(long long)i * M + j]; }
};

// Right-looking Cholesky on accessor `a`; sdiag[M] scratch receives the diagonal of L.
template <bool FAST, typename Acc>
__device__ void chol_body(Acc a, int M, double* sdiag, double* logdet_out, int* info_out) {
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;  // 32 x 32
  double ld = 0.0;
  int info = 0;
  for (int j = 0; j < M; ++j) {
    const double d = a.at(j, j);
    if (!(d > 0.0)) {  // uniform across the workgroup (everyone read the same value)
      info = j + 1;
      break;
    }
    const double s = sqrt(d), inv = 1.0 / s;
    ld += log(s);
    if (tid == 0) sdiag[j] = s;
    for (int i = j + 1 + tid; i < M; i += LA_THREADS) a.at(i, j) *= inv;
    __syncthreads();
    const int R = (M - j - 1 + 31) >> 5;  // 32 x 32 blocks of the trailing matrix (uniform)
    if (FAST && R <= 7) {
      // all operands of a thread's updates are fetched before the FMAs (independent elements): the
      // update is bound by LDS bandwidth instead of one LDS round trip per element
      double li[7], lk[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const int i = j + 1 + ty + 32 * q, k = j + 1 + tx + 32 * q;
        li[q] = (q < R && i < M) ? a.at(i, j) : 0.0;
        lk[q] = (q < R && k < M) ? a.at(k, j) : 0.0;
      }
#pragma unroll
      for (int qa = 0; qa < 7; ++qa) {
        if (qa < R) {
          const int i = j + 1 + ty + 32 * qa;
          double e[7];
#pragma unroll
          for (int qb = 0; qb <= qa; ++qb) {
            const int k = j + 1 + tx + 32 * qb;
            e[qb] = (i < M && k <= i) ? a.at(i, k) : 0.0;
          }
#pragma unroll
          for (int qb = 0; qb <= qa; ++qb) {
            const int k = j + 1 + tx + 32 * qb;
            if (i < M && k <= i) a.at(i, k) = e[qb] - li[qa] * lk[qb];
          }
        }
      }
    } else {
      for (int i = j + 1 + ty; i < M; i += 32) {
        const double lij = a.at(i, j);
        for (int k = j + 1 + tx; k <= i; k += 32) a.at(i, k) -= lij * a.at(k, j);
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    *logdet_out = (info == 0) ? 2.0 * ld : __builtin_nan("");
    *info_out = info;
  }
  __syncthreads();
  if (info == 0)
    for (int j = tid; j < M; j += LA_THREADS) a.at(j, j) = sdiag[j];
  __syncthreads();
}

__global__ void __launch_bounds__(LA_THREADS)
chol_packed_kernel(double* __restrict__ A, int M, double* __restrict__ logdet,
                   int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* G = A + (long long)blockIdx.x * M * M;
  PackedLower a{lds};
  double* sdiag = lds + (M * (M + 1) >> 1);
  for (int e = threadIdx.x; e < M * M; e += LA_THREADS) {
    const int i = e / M, j = e - i * M;
    if (j <= i) a.at(i, j) = G[e];
  }
  __syncthreads();
  chol_body<false>(a, M, sdiag, logdet + blockIdx.x, info + blockIdx.x);  // <true>: measured slower (0.56 vs 0.41 ms)
  for (int e = threadIdx.x; e < M * M; e += LA_THREADS) {
    const int i = e / M, j = e - i * M;
    G[e] = (j <= i) ? a.at(i, j) : 0.0;
  }
}

constexpr int LA_GLOBAL_MAX = 4096;
__global__ void __launch_bounds__(LA_THREADS)
chol_global_kernel(double* __restrict__ A, int M, double* __restrict__ logdet,
                   int* __restrict__ info) {
  __shared__ double sdiag[LA_GLOBAL_MAX];
  DenseGlobal a{A + (long long)blockIdx.x * M * M, M};
  chol_body<false>(a, M, sdiag, logdet + blockIdx.x, info + blockIdx.x);
  for (int e = threadIdx.x; e < M * M; e += LA_THREADS) {
    const int i = e / M, j = e - i * M;
    if (j > i) a.p[e] = 0.0;
  }
}

// Row-oriented in-place inverse of a lower-triangular matrix held in accessor `a`.
// Row i of the inverse needs rows < i of the inverse and row i of L only, so L is overwritten row by
// row.  256 columns x 4-way split of the inner sum per pass; the 4 partial sums meet by shuffles.
template <typename Acc>
__device__ void tri_inv_body(Acc a, int M, double* rowL) {
  // lane = (column within a 16-column group, k-split 0..3): the 16 lanes of one k-split read 16
  // CONSECUTIVE elements of one packed row (conflict-free); the 4 partial sums meet by shuffles
  const int tid = threadIdx.x;
  const int ks = (tid >> 4) & 3, cl = (tid & 15) + ((tid >> 6) << 4);
  for (int i = 0; i < M; ++i) {
    for (int k = tid; k <= i; k += LA_THREADS) rowL[k] = a.at(i, k);
    __syncthreads();
    const double inv = 1.0 / rowL[i];
    for (int c0 = 0; c0 < i; c0 += LA_THREADS / 4) {
      const int c = c0 + cl;
      double s0 = 0.0, s1 = 0.0;
      if (c < i) {
        int k = c + ks;
        for (; k + 4 < i; k += 8) {
          s0 += rowL[k] * a.at(k, c);
          s1 += rowL[k + 4] * a.at(k + 4, c);
        }
        if (k < i) s0 += rowL[k] * a.at(k, c);
      }
      double sum = s0 + s1;
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      if (c < i && ks == 0) a.at(i, c) = -sum * inv;
    }
    if (tid == 0) a.at(i, i) = inv;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(LA_THREADS)
tri_inv_packed_kernel(const double* __restrict__ L, double* __restrict__ Linv, int M) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const double* G = L + (long long)blockIdx.x * M * M;
  double* O = Linv + (long long)blockIdx.x * M * M;
  PackedLower a{lds};
  double* rowL = lds + (M * (M + 1) >> 1);
  for (int e = threadIdx.x; e < M * M; e += LA_THREADS) {
    const int i = e / M, j = e - i * M;
    if (j <= i) a.at(i, j) = G[e];
  }
  __syncthreads();
  tri_inv_body(a, M, rowL);
  for (int e = threadIdx.x; e < M * M; e += LA_THREADS) {
    const int i = e / M, j = e - i * M;
    O[e] = (j <= i) ? a.at(i, j) : 0.0;
  }
}

__global__ void __launch_bounds__(LA_THREADS)
tri_inv_global_kernel(const double* __restrict__ L, double* __restrict__ Linv, int M) {
  __shared__ double rowL[LA_GLOBAL_MAX];
  const double* G = L + (long long)blockIdx.x * M * M;
  DenseGlobal a{Linv + (long long)blockIdx.x * M * M, M};
  for (int e = threadIdx.x; e < M * M; e += LA_THREADS) {
    const int i = e / M, j = e - i * M;
    a.p[e] = (j <= i) ? G[e] : 0.0;
  }
  __syncthreads();
  tri_inv_body(a, M, rowL);
}

// Fused factorisation: Linv = chol(A)^-1, logdet, info in ONE pass with the matrix held in registers.
// TS x TS threads own the lower triangle 2-D cyclically: thread (ty, tx) holds t[qa][qb] = element
// (TS qa + ty, TS qb + tx), qb <= qa.  TS = 16 (one wave per SIMD, up to 136 elements per thread) runs
// the per-column bookkeeping once per SIMD instead of four times and is the faster shape up to M = 208; TS = 32 above (GPSA_CHOL_TS forces one).  Step j of the right-looking factorisation publishes column j
// (rows >= j) and, for the forward substitution that runs in the same sweep, row j of the inverse
// (columns < j) through a double-buffered LDS vector: ONE barrier per column, no LDS read-modify-
// write.  The register position (i,k) holds A(i,k) while k > j and X(i,k) = L^-1(i,k) once k <= j
// (L(:,k) itself is dead after step k), so both updates are the same rank-1 FMA
//     t(i,k) -= L(i,j) * w(k),   w(k) = L(k,j) for k > j,   X(j,k)/L(j,j) for k < j,   1/L(j,j) at k = j
// over the rows i > j.  The outer loop over the 32-column tile index is unrolled at compile time so
// that every register index is static and dead tiles cost nothing.
template <int NT, int TS>
__global__ void __launch_bounds__(TS * TS)
chol_inv_reg_kernel(const double* __restrict__ A, int M, double* __restrict__ Linv,
                    double* __restrict__ logdet, int* __restrict__ info, int lda, long long sA, int ldo,
                    long long sO, int col0, int accumulate, int n0, int gap) {
  // lda / sA, ldo / sO: row and batch strides of the input and output (a diagonal block of a larger
  // matrix in the blocked factorisation); col0: column offset reported in info; accumulate: add to
  // logdet and keep an earlier block's info; n0 / gap: workgroup x handles matrix x (x < n0) or x + gap (a selection
  // of the batch in one launch: gpsa_chol_inv_sel_f64)
  constexpr int NTH = TS * TS;
  const int bx = (int)blockIdx.x < n0 ? (int)blockIdx.x : (int)blockIdx.x + gap;
  __shared__ double col[2][NT * TS];
  __shared__ double xrow[2][NT * TS];
  __shared__ double sdiag[NT * TS];
  __shared__ double red[16];
  const int tid = threadIdx.x, tx = tid % TS, ty = tid / TS;
  const double* G = A + (long long)bx * sA;
  double* O = Linv + (long long)bx * sO;
  double t[NT][NT];
#pragma unroll
  for (int qa = 0; qa < NT; ++qa)
#pragma unroll
    for (int qb = 0; qb < NT; ++qb) {
      const int i = TS * qa + ty, k = TS * qb + tx;
      t[qa][qb] = (qb <= qa && i < M && k < M) ? G[(long long)i * lda + k] : 0.0;
    }
  int bad = 0;
#pragma unroll
  for (int q0 = 0; q0 < NT; ++q0) {
#pragma unroll 1
    for (int r0 = 0; r0 < TS; ++r0) {
      const int j = TS * q0 + r0;
      if (j >= M || bad) break;  // uniform
      const int b = j & 1;
      if (tx == r0) {
#pragma unroll
        for (int qa = q0; qa < NT; ++qa) {
          const int i = TS * qa + ty;
          if (i >= j) col[b][i] = t[qa][q0];
        }
      }
      if (ty == r0) {
#pragma unroll
        for (int qb = 0; qb <= q0; ++qb) {
          const int k = TS * qb + tx;
          if (k < j) xrow[b][k] = t[q0][qb];
        }
      }
      __syncthreads();
      const double d = col[b][j];
      if (!(d > 0.0)) {  // uniform: everyone read the same value
        bad = j + 1;
        break;
      }
      // 1/sqrt(d) from the hardware estimate plus one third-order correction (error ~ e^3 with
      // e ~ 2^-26): ~10 dependent fp64 operations on the critical path instead of sqrt + divide
      const double y0 = __builtin_amdgcn_rsq(d);
      const double e = fma(-(d * y0), y0, 1.0);
      const double inv = fma(y0 * e, fma(0.375, e, 0.5), y0);
      if (tid == 0) sdiag[j] = inv;
      double w[NT], li[NT];
#pragma unroll
      for (int qb = 0; qb < NT; ++qb) {
        const int k = TS * qb + tx;
        if (qb > q0) w[qb] = col[b][k] * inv;
        else if (qb < q0) w[qb] = xrow[b][k] * inv;
        else w[qb] = (k > j) ? col[b][k] * inv : ((k == j) ? inv : xrow[b][k] * inv);
      }
#pragma unroll
      for (int qa = q0; qa < NT; ++qa) {
        const int i = TS * qa + ty;
        li[qa] = (i > j) ? col[b][i] * inv : 0.0;
      }
#pragma unroll
      for (int qa = q0; qa < NT; ++qa)
#pragma unroll
        for (int qb = 0; qb <= qa; ++qb) {
          if (qb == q0) {
            const double base = (tx == r0) ? 0.0 : t[qa][qb];  // column j: L(i,j) leaves, X(i,j) enters
            t[qa][qb] = base - li[qa] * w[qb];
          } else {
            t[qa][qb] -= li[qa] * w[qb];
          }
        }
      if (ty == r0) {  // row j of the inverse is final: scale by 1/L(j,j), diagonal = 1/L(j,j)
#pragma unroll
        for (int qb = 0; qb <= q0; ++qb) {
          const int k = TS * qb + tx;
          if (k < j) t[q0][qb] *= inv;
          else if (k == j) t[q0][qb] = inv;
        }
      }
    }
  }
  __syncthreads();
  // logdet = 2 sum log L(j,j), reduced in a fixed order
  double lg = 0.0;
  if (!bad)
    for (int j = tid; j < M; j += NTH) lg -= log(sdiag[j]);  // sdiag = 1 / L(j,j)
  lg = block_sum(lg, red);
  if (tid == 0) {
    const double ld = bad ? __builtin_nan("") : 2.0 * lg;
    if (accumulate) {
      logdet[bx] += ld;
      if (info[bx] == 0 && bad) info[bx] = col0 + bad;
    } else {
      logdet[bx] = ld;
      info[bx] = bad ? col0 + bad : 0;
    }
  }
#pragma unroll
  for (int qa = 0; qa < NT; ++qa)
#pragma unroll
    for (int qb = 0; qb < NT; ++qb) {
      const int i = TS * qa + ty, k = TS * qb + tx;
      if (i < M && k < M) O[(long long)i * ldo + k] = (qb <= qa && k <= i) ? t[qa][qb] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Blocked form of the fused factorisation (M <= 208): the same register-resident 2-D cyclic layout, but the
// columns are processed in panels of 16 with TWO block barriers per panel instead of one per column.
//   (1) the panel's columns t(:, J) and the row block's partial inverse t(J, <J) go to LDS        [barrier]
//   (2) every wave factorises the 16 x 16 diagonal block for itself (one row per lane, cross-lane reads:
//       no barrier, no waiting for a designated wave);
//   (3) one thread per row below the block solves L(i,J) L_JJ^T = A(i,J) (forward substitution over 16
//       columns), one thread per column k < J_end solves L_JJ X(J,k) = T(J,k) (T = e_k inside the block):
//       rows + columns = 16 NT <= 256 threads, all independent                                    [barrier]
//   (4) the row block of t becomes final X(J, :); every row below gets the rank-16 update
//           t(i,k) (-)= sum_{j in J} L(i,j) w_j(k),   w_j(k) = L(k,j) for k > J,  X(j,k) for k <= J_end
//       (positions k in J start from zero: L(i,j) leaves, X(i,j) enters) - the same FMAs the per-column
//       kernel issues, without its 16 barriers and 16 dependent pivot chains.
// 167 -> (see DESIGN) us for the 57 matrices of a step; the dependent chain per matrix is what a step waits for.
constexpr int CB_LS = 18;  // LDS row stride (doubles) of the panel buffers: conflict-free 16-byte reads

// LDS-qualified pointers: the helpers below are real functions (one copy for all panels); with generic
// pointers every access would be a flat load
typedef __attribute__((address_space(3))) double lds_f64;

// value of ``v`` in lane ``src`` (compile-time constant) for every lane: two v_readlane_b32, no LDS round trip
template <int SRC>
__device__ __forceinline__ double lane_bcast(double v) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, SRC);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), SRC);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// 16 x 16 factorisation of the diagonal block by ONE wave (lane l holds row l & 15 in registers): D row-major, stride
// CB_LS.  The pivot and the entries of the two columns next to it cross the lanes as v_readlane broadcasts from the
// owning lane's register, the other columns' entries come back from a per-step LDS buffer the lanes publish to (their
// loads land under the chain): the 16-step dependency chain holds no LDS round trip.
// Round 6: a SHORTER chain per pivot.  Round 2's was  readlane -> 5 selects (padding / positivity) -> rsq -> 4
// refinement ops -> two scalings -> the update of the next pivot: ~14 dependent fp64 operations, 16 pivots per block,
// 13 blocks per matrix at M = 200 (43 us of the kernel's 155; 36 now).  The chain now carries only what the NEXT
// pivot needs:
//     d -> 1/d (rcp + one third-order correction: 4 operations) -> mi = column * (1/d) -> the update's FMA,
// i.e. the square-root-free (L D L^T) recurrence: P(i,c) -= P(i,r) P(c,r) / d_r.  The lanes keep the UNSCALED
// columns; 1/sqrt(d) of all 16 pivots is formed once, in parallel across the lanes, after the last step, and the
// factor's columns are scaled then.  Padding needs no select (the identity padding the kernel loads keeps d = 1
// through every update: the padded rows' off-diagonal entries are exact zeros), and a non-positive pivot is only
// RECORDED (nothing on the chain reads the flag; the caller drops the matrix).
template <int R, int C>
struct Ldl16Col {
  static __device__ __forceinline__ void run(double (&P)[16], double mi, double colr, const lds_f64* cb) {
    if (C <= R + 2) P[C] = fma(-mi, lane_bcast<C>(colr), P[C]);
    else P[C] = fma(-mi, cb[C], P[C]);
    Ldl16Col<R, C + 1>::run(P, mi, colr, cb);
  }
};
template <int R>
struct Ldl16Col<R, 16> {
  static __device__ __forceinline__ void run(double (&)[16], double, double, const lds_f64*) {}
};

template <int R>
struct Ldl16Step {
  static __device__ __forceinline__ void run(double (&P)[16], lds_f64* colbuf, int j0, int lane, int& bad) {
    const double colr = P[R];          // this row's entry of column R (unscaled)
    lds_f64* cb = colbuf + R * 16;
    if (R + 3 < 16) {
      if (lane < 16) cb[lane] = colr;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    const double d = lane_bcast<R>(colr);
    if (!(d > 0.0) && !bad) bad = j0 + R + 1;  // (off the chain)
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double rinv = fma(r0 * e, 1.0 + e, r0);  // 1/d to e^3 (e ~ 2^-23 from the hardware estimate)
    if (R + 1 < 16) {
      const double mi = colr * rinv;
      Ldl16Col<R, R + 1>::run(P, mi, colr, cb);
    }
    Ldl16Step<R + 1>::run(P, colbuf, j0, lane, bad);
  }
};
template <>
struct Ldl16Step<16> {
  static __device__ __forceinline__ void run(double (&)[16], lds_f64*, int, int, int&) {}
};

template <int C>
struct Ldl16Scale {
  static __device__ __forceinline__ void run(double (&P)[16], double rs) {
    P[C] *= lane_bcast<C>(rs);
    Ldl16Scale<C + 1>::run(P, rs);
  }
};
template <>
struct Ldl16Scale<16> {
  static __device__ __forceinline__ void run(double (&)[16], double) {}
};

__device__ __noinline__ int chol16_wave_ldl(const lds_f64* D, lds_f64* Ld, lds_f64* dinv, lds_f64* colbuf, int j0) {
  const int lane = threadIdx.x & 63, row = lane & 15;
  double P[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) P[c] = D[row * CB_LS + c];
  int bad = 0;
  Ldl16Step<0>::run(P, colbuf, j0, lane, bad);
  // lane i's pivot d_i sits in P[i] (its own column was never scaled): 1/sqrt for all 16 at once
  double dm = P[0];
#pragma unroll
  for (int c = 1; c < 16; ++c) dm = (row == c) ? P[c] : dm;
  if (!(dm > 0.0)) dm = 1.0;  // (a flagged block: keep the arithmetic finite)
  const double y0 = __builtin_amdgcn_rsq(dm);
  const double e = fma(-(dm * y0), y0, 1.0);
  const double rs = fma(y0 * e, fma(0.375, e, 0.5), y0);
  Ldl16Scale<0>::run(P, rs);  // column c of the factor = unscaled column * 1/sqrt(d_c)
  if (lane < 16) {  // COLUMN-major (LdT[m * 16 + c] = L(c, m)): panel_solves_col reads a column per step
    dinv[lane] = rs;
#pragma unroll
    for (int c = 0; c < 16; ++c) Ld[c * 16 + lane] = P[c];
  }
  return bad;
}

// step (3), round 6.  Both kinds of thread solve the SAME recurrence against the block's factor,
//     x[c] = (x[c] - sum_{m < c} x[m] L(c, m)) / L(c, c):
// thread t < R - 16 for row 16 + t of the panel (L(i,J) L_JJ^T = A(i,J)), thread R - 16 + k for column k of
// X(J, :) (L_JJ X(J,k) = T(J,k), T = e_k inside the block) - only where x comes from and goes to differs, so one
// instruction stream serves a wave that holds both kinds (the round-2 form ran its two branches one after the other
// there).  Column-oriented: once x[m] is final the 15 - m updates it feeds are independent FMAs (the round-2 form
// walked each x[c]'s sum serially: 136 dependent FMAs per thread with an LDS read in front of each - 2.6 us per panel,
// 34 us per matrix), and column m + 1 of the factor is requested from LDS while column m is applied.
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) dbl2* lds_d2;
__device__ __noinline__ void panel_solves_col(lds_f64* Lp, int R, const lds_f64* LdT, const lds_f64* dinv,
                                              const lds_f64* Tb, int MP, lds_f64* XpT, int i0) {
  const int t = threadIdx.x, na = R - 16;
  if (t >= na + i0 + 16) return;
  const bool isrow = t < na;
  const int k = t - na;
  const bool unit = !isrow && k >= i0;  // identity column k - i0 of the block itself
  // x[e] <- src[e * sstride]; rows: the panel row, columns: Tb[e][k] (identity columns read a valid dummy)
  const lds_f64* src = isrow ? (const lds_f64*)(Lp + (16 + t) * CB_LS) : (Tb + (unit ? 0 : k));
  const int sstride = isrow ? 1 : MP;
  lds_f64* dst = isrow ? Lp + (16 + t) * CB_LS : XpT + k * CB_LS;
  double x[16], di[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = src[e * sstride];
#pragma unroll
  for (int e = 0; e < 16; e += 2) {
    const dbl2 d2 = *reinterpret_cast<lds_d2>(&dinv[e]);
    di[e] = d2.x;
    di[e + 1] = d2.y;
  }
  if (unit) {
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = (k - i0 == e) ? 1.0 : 0.0;
  }
  dbl2 col[2][8];  // col[m & 1][p] = L(2p .. 2p+1, m)
#pragma unroll
  for (int p_ = 0; p_ < 8; ++p_) col[0][p_] = *reinterpret_cast<lds_d2>(&LdT[0 * 16 + 2 * p_]);
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    if (m + 1 < 16) {
#pragma unroll
      for (int p_ = (m + 2) / 2; p_ < 8; ++p_)
        col[(m + 1) & 1][p_] = *reinterpret_cast<lds_d2>(&LdT[(m + 1) * 16 + 2 * p_]);
    }
    // (left alone, the scheduler sinks every load to one instruction in front of its first use - an LDS round trip
    //  per pair of FMAs; the fence keeps column m + 1's requests in front of column m's arithmetic)
    __builtin_amdgcn_sched_barrier(0);
    x[m] *= di[m];
#pragma unroll
    for (int c = m + 1; c < 16; ++c) {
      const double lcm = (c & 1) ? col[m & 1][c >> 1].y : col[m & 1][c >> 1].x;
      x[c] = fma(-x[m], lcm, x[c]);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) dst[e] = x[e];
}

static inline long long chol_blk_lds_bytes(int NT) {
  const long long MP = NT * 16;
  return (3 * MP * CB_LS + 16 * MP + 4 * 256 + 4 * 16 + 4 * 256 + MP + 8) * 8;
}

template <int NT>
__global__ void __launch_bounds__(256)
chol_inv_blk_kernel(const double* __restrict__ A, int M, double* __restrict__ Linv,
                    double* __restrict__ logdet, int* __restrict__ info, int skip, int n0, int gap) {
  constexpr int MP = NT * 16;
  const int bx = (int)blockIdx.x < n0 ? (int)blockIdx.x : (int)blockIdx.x + gap;  // (chol_inv_reg_kernel: n0 / gap)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* Lp0 = sm;                       // [MP][CB_LS]  panel (rows >= i0 of the 16 columns), buffer 0
  double* Lp1 = Lp0 + MP * CB_LS;         //              buffer 1
  double* XpT = Lp1 + MP * CB_LS;         // [MP][CB_LS]  XpT[k][jj] = X(i0 + jj, k)
  double* Tb = XpT + MP * CB_LS;          // [16][MP]     partial inverse rows of the block
  double* Ldw = Tb + 16 * MP;             // [4][256]     per-wave copy of the diagonal block's factor
  double* dinvw = Ldw + 4 * 256;          // [4][16]
  double* colw = dinvw + 4 * 16;          // [4][16][16] per-wave column buffers of the block factorisation
  double* sdiag = colw + 4 * 256;         // [MP]
  __shared__ double red[16];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* G = A + (long long)bx * M * M;
  double* O = Linv + (long long)bx * M * M;
  double t[NT][NT];
#pragma unroll
  for (int qa = 0; qa < NT; ++qa)
#pragma unroll
    for (int qb = 0; qb < NT; ++qb) {
      const int i = 16 * qa + ty, k = 16 * qb + tx;
      t[qa][qb] = (qb <= qa) ? ((i < M && k < M) ? G[(long long)i * M + k] : (i == k ? 1.0 : 0.0)) : 0.0;
    }
  int bad = 0;
#pragma unroll
  for (int q0 = 0; q0 < NT; ++q0) {
    const int i0 = 16 * q0;
    if (i0 >= M || bad) break;  // uniform
    double* Lp = (q0 & 1) ? Lp1 : Lp0;
    const int R = MP - i0;
    // (1) publish the panel and the row block's partial inverse
#pragma unroll
    for (int qa = q0; qa < NT; ++qa) Lp[((qa - q0) * 16 + ty) * CB_LS + tx] = t[qa][q0];
#pragma unroll
    for (int qb = 0; qb < q0; ++qb) Tb[ty * MP + 16 * qb + tx] = t[q0][qb];
    __syncthreads();
    // (2) diagonal block, every wave for itself
    if (!(skip & 1))
      bad = chol16_wave_ldl((const lds_f64*)Lp, (lds_f64*)(Ldw + w * 256), (lds_f64*)(dinvw + w * 16),
                            (lds_f64*)(colw + w * 256), i0);
    if (tid < 16) sdiag[i0 + tid] = dinvw[tid];  // wave 0's copy (written by its lane 0 above; same wave)
    if (bad) break;  // uniform: every wave factorised the same block
    // (3) panel rows and X columns, one thread each
    if (!(skip & 2))
      panel_solves_col((lds_f64*)Lp, R, (const lds_f64*)(Ldw + w * 256), (const lds_f64*)(dinvw + w * 16),
                       (const lds_f64*)Tb, MP, (lds_f64*)XpT, i0);
    __syncthreads();
    // (4) the row block of t is final; rank-16 update of the rows below
#pragma unroll
    for (int qb = 0; qb <= q0; ++qb) t[q0][qb] = XpT[(16 * qb + tx) * CB_LS + ty];
    if (q0 + 1 < NT && !(skip & 4)) {
#pragma unroll
      for (int qa = q0 + 1; qa < NT; ++qa) t[qa][q0] = 0.0;  // L(i,J) left for LDS, X(i,J) accumulates from zero
#pragma unroll 1
      for (int jj = 0; jj < 16; jj += 2) {
        double2 li[NT], wv[NT];
#pragma unroll
        for (int qa = q0 + 1; qa < NT; ++qa)
          li[qa] = *reinterpret_cast<const double2*>(&Lp[((qa - q0) * 16 + ty) * CB_LS + jj]);
#pragma unroll
        for (int qb = 0; qb < NT; ++qb) {
          if (qb > q0) wv[qb] = *reinterpret_cast<const double2*>(&Lp[((qb - q0) * 16 + tx) * CB_LS + jj]);
          else wv[qb] = *reinterpret_cast<const double2*>(&XpT[(16 * qb + tx) * CB_LS + jj]);
        }
#pragma unroll
        for (int qa = q0 + 1; qa < NT; ++qa)
#pragma unroll
          for (int qb = 0; qb <= qa; ++qb)
            t[qa][qb] = fma(-li[qa].y, wv[qb].y, fma(-li[qa].x, wv[qb].x, t[qa][qb]));
      }
    }
  }
  __syncthreads();
  double lg = 0.0;
  if (!bad)
    for (int j = tid; j < M; j += 256) lg -= log(sdiag[j]);  // sdiag = 1 / L(j,j)
  lg = block_sum(lg, red);
  if (tid == 0) {
    logdet[bx] = bad ? __builtin_nan("") : 2.0 * lg;
    info[bx] = bad;
  }
#pragma unroll
  for (int qa = 0; qa < NT; ++qa)
#pragma unroll
    for (int qb = 0; qb < NT; ++qb) {
      const int i = 16 * qa + ty, k = 16 * qb + tx;
      if (i < M && k < M) O[(long long)i * M + k] = (qb <= qa && k <= i) ? t[qa][qb] : 0.0;
    }
}

template <int NT>
static int chol_inv_blk_launch_nt(const double* A, int M, double* Linv, double* logdet, int* info, int batch,
                                  hipStream_t st, int n0, int gap) {
  static per_device_flag attr_flag;
  bool& attr_set = attr_flag.here();
  const long long lds = chol_blk_lds_bytes(NT);
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_inv_blk_kernel<NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GPSA_EUNSUPPORTED;
    attr_set = true;
  }
  // timing-only experiments (results are then wrong): bit 0 skips the block factorisation, 1 the solves, 2 the update
  static const int skip = [] { const char* e = getenv("GPSA_CHOL_SKIP"); return e ? atoi(e) : 0; }();
  chol_inv_blk_kernel<NT><<<batch, 256, (size_t)lds, st>>>(A, M, Linv, logdet, info, skip, n0, gap);
  GPSA_LAUNCH_CHECK();
  return 0;
}

// contiguous batch, M <= 208: the blocked kernel (GPSA_CHOL_BLOCKED=0 falls back to one barrier per column)
static int chol_inv_blk_launch(const double* A, int M, double* Linv, double* logdet, int* info, int batch,
                               hipStream_t st, int n0 = 0x7fffffff, int gap = 0) {
  const int nt = (M + 15) / 16;
  if (nt <= 2) return chol_inv_blk_launch_nt<2>(A, M, Linv, logdet, info, batch, st, n0, gap);
  if (nt <= 4) return chol_inv_blk_launch_nt<4>(A, M, Linv, logdet, info, batch, st, n0, gap);
  if (nt <= 7) return chol_inv_blk_launch_nt<7>(A, M, Linv, logdet, info, batch, st, n0, gap);
  if (nt <= 10) return chol_inv_blk_launch_nt<10>(A, M, Linv, logdet, info, batch, st, n0, gap);
  if (nt <= 13) return chol_inv_blk_launch_nt<13>(A, M, Linv, logdet, info, batch, st, n0, gap);
  return GPSA_EUNSUPPORTED;
}

// out[b] = sum_i A[b,i]*B[b,i]; grid (batch, nsplit): partial sums per split in part[b*nsplit+s], then summed
template <typename T>
__global__ void bdot_kernel(const T* __restrict__ A, long long sA, const T* __restrict__ B,
                            long long sB, long long n, double* __restrict__ part) {
  __shared__ double red[4];
  const T* a = A + (long long)blockIdx.x * sA;
  const T* b = B + (long long)blockIdx.x * sB;
  double s = 0.0;
  for (long long i = blockIdx.y * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.y * blockDim.x)
    s += (double)a[i] * (double)b[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[(long long)blockIdx.x * gridDim.y + blockIdx.y] = s;
}

template <typename T>
__global__ void bdot_finish_kernel(const double* __restrict__ part, int nsplit, int batch, T* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  double s = 0.0;
  for (int i = 0; i < nsplit; ++i) s += part[(long long)b * nsplit + i];
  out[b] = (T)s;
}

template <typename T>
__global__ void add_diag_kernel(T* __restrict__ A, int M, int batch, T s) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= (long long)M * batch) return;
  const long long b = idx / M, i = idx % M;
  A[b * M * M + i * M + i] += s;
}

// one launch of the register-resident kernel on (a diagonal block of) a batch of matrices
static int chol_inv_reg_launch(const double* A, int M, double* Linv, double* logdet, int* info, int lda,
                               long long sA, int ldo, long long sO, int col0, int accumulate, int batch,
                               hipStream_t st, int n0 = 0x7fffffff, int gap = 0) {
#define GPSA_CI_CASE(V, TSV)                                                                        \
  chol_inv_reg_kernel<V, TSV><<<batch, TSV * TSV, 0, st>>>(A, M, Linv, logdet, info, lda, sA, ldo, sO, \
                                                            col0, accumulate, n0, gap)
  // 16 x 16 threads (one wave per SIMD) while a thread's share fits comfortably in registers: 166 vs
  // 185 us at M = 200; 32 x 32 above (252 vs 287 us at M = 256)
  static const int forced = [] { const char* e = getenv("GPSA_CHOL_TS"); return e ? atoi(e) : 0; }();
  const int ts = forced ? forced : (M <= 208 ? 16 : 32);
  if (ts == 32) {
    const int nt = (M + 31) / 32;
    if (nt <= 1) GPSA_CI_CASE(1, 32);
    else if (nt <= 2) GPSA_CI_CASE(2, 32);
    else if (nt <= 4) GPSA_CI_CASE(4, 32);
    else if (nt <= 7) GPSA_CI_CASE(7, 32);
    else if (nt <= 8) GPSA_CI_CASE(8, 32);
    else return GPSA_EUNSUPPORTED;
  } else {
    const int nt = (M + 15) / 16;
    if (nt <= 2) GPSA_CI_CASE(2, 16);
    else if (nt <= 4) GPSA_CI_CASE(4, 16);
    else if (nt <= 7) GPSA_CI_CASE(7, 16);
    else if (nt <= 13) GPSA_CI_CASE(13, 16);
    else if (nt <= 16) GPSA_CI_CASE(16, 16);
    else return GPSA_EUNSUPPORTED;
  }
#undef GPSA_CI_CASE
  GPSA_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int gemm_launch(int transA, int transB, int m, int n, long long k, double alpha, const T* A,
                long long lda, long long sA, const T* B, long long ldb, long long sB, double beta, T* C,
                long long ldc, long long sC, int batch, int splitk, void* part, long long part_bytes,
                hipStream_t st);  // gemm.hip

static inline int chol_block_size(int M) {
  const int nb = (M + 255) / 256;
  int bs = (((M + nb - 1) / nb + 15) / 16) * 16;
  return bs > 256 ? 256 : bs;
}

// Blocked factorisation for M beyond the register-resident kernel: right-looking over diagonal blocks of
// <= 256 columns.  Block k:  L_kk^-1 by the register kernel on the (updated) diagonal block;  panel
// L[r,k] = W[r,k] L_kk^-T and trailing update W[r,r] -= L[r,k] L[r,k]^T as two fp64-MFMA products;  row k
// of the inverse X[k,<k] = -L_kk^-1 (L[k,<k] X[<k,<k]) as two more.  ~5 launches per block instead of one
// latency-bound global-memory sweep over all M columns (M = 1000, 17 matrices: 120 + 54 ms before).
static int chol_inv_blocked(const double* A, double* Linv, int M, int batch, double* logdet, int* info,
                            double* ws, hipStream_t st) {
  const long long mm = (long long)M * M;
  const int bs = chol_block_size(M), nb = (M + bs - 1) / bs;
  double* W = ws;                    // working copy of A (lower part is read)
  double* Lf = W + (long long)batch * mm;   // the factor's off-diagonal blocks
  double* T = Lf + (long long)batch * mm;   // [batch][bs][M] scratch
  const long long sT = (long long)bs * M;
  hipError_t e = hipMemcpyAsync(W, A, (size_t)batch * mm * 8, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return (int)e;
  {
    const int ez = zero_fill_async(Linv, (size_t)batch * mm * 8, st);
    if (ez != 0) return ez;
  }
  for (int k = 0; k < nb; ++k) {
    const long long o = (long long)k * bs;
    const int b = (int)((M - o < bs) ? M - o : bs);
    const long long r = o + b;
    const int mr = (int)(M - r);
    double* Dinv = Linv + o * M + o;
    int rc = chol_inv_reg_launch(W + o * M + o, b, Dinv, logdet, info, M, mm, M, mm, (int)o, k > 0, batch, st);
    if (rc) return rc;
    if (mr > 0) {
      rc = gemm_launch<double>(0, 1, mr, b, b, 1.0, W + r * M + o, M, mm, Dinv, M, mm, 0.0, Lf + r * M + o, M,
                               mm, batch, 1, nullptr, 0, st);
      if (rc) return rc;
      rc = gemm_launch<double>(0, 1, mr, mr, b, -1.0, Lf + r * M + o, M, mm, Lf + r * M + o, M, mm, 1.0,
                               W + r * M + r, M, mm, batch, 1, nullptr, 0, st);
      if (rc) return rc;
    }
    if (k > 0) {
      rc = gemm_launch<double>(0, 0, b, (int)o, o, 1.0, Lf + o * M, M, mm, Linv, M, mm, 0.0, T, M, sT, batch,
                               1, nullptr, 0, st);
      if (rc) return rc;
      rc = gemm_launch<double>(0, 0, b, (int)o, b, -1.0, Dinv, M, mm, T, M, sT, 0.0, Linv + o * M, M, mm,
                               batch, 1, nullptr, 0, st);
      if (rc) return rc;
    }
  }
  return 0;
}

}  // namespace gpsa

extern "C" {

int gpsa_chol_f64(void* A, int M, int batch, void* logdet, int* info, void* stream) {
  using namespace gpsa;
  if (M < 1 || batch < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (M <= LA_PACKED_MAX) {
    const size_t lds = ((size_t)M * (M + 1) / 2 + M) * sizeof(double);
    static per_device_flag attr_flag;
    bool& attr_set = attr_flag.here();
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)chol_packed_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    chol_packed_kernel<<<batch, LA_THREADS, lds, st>>>((double*)A, M, (double*)logdet, info);
  } else {
    if (M > LA_GLOBAL_MAX) return GPSA_EUNSUPPORTED;
    chol_global_kernel<<<batch, LA_THREADS, 0, st>>>((double*)A, M, (double*)logdet, info);
  }
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_tri_inv_f64(const void* L, void* Linv, int M, int batch, void* stream) {
  using namespace gpsa;
  if (M < 1 || batch < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (M <= LA_PACKED_MAX) {
    const size_t lds = ((size_t)M * (M + 1) / 2 + M) * sizeof(double);
    static per_device_flag attr_flag;
    bool& attr_set = attr_flag.here();
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)tri_inv_packed_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    tri_inv_packed_kernel<<<batch, LA_THREADS, lds, st>>>((const double*)L, (double*)Linv, M);
  } else {
    if (M > LA_GLOBAL_MAX) return GPSA_EUNSUPPORTED;
    tri_inv_global_kernel<<<batch, LA_THREADS, 0, st>>>((const double*)L, (double*)Linv, M);
  }
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_chol_inv_f64(const void* A, void* Linv, int M, int batch, void* logdet, int* info,
                      void* stream) {
  using namespace gpsa;
  if (M < 1 || batch < 1) return GPSA_EINVAL;
  static const bool blocked = [] { const char* e = getenv("GPSA_CHOL_BLOCKED"); return !(e && e[0] == '0'); }();
  if (blocked && M <= 208) {
    int rc = chol_inv_blk_launch((const double*)A, M, (double*)Linv, (double*)logdet, info, batch, as_stream(stream));
    if (rc != GPSA_EUNSUPPORTED) return rc;
  }
  return chol_inv_reg_launch((const double*)A, M, (double*)Linv, (double*)logdet, info, M,
                             (long long)M * M, M, (long long)M * M, 0, 0, batch, as_stream(stream));
}

int gpsa_chol_inv_sel_f64(const void* A, void* Linv, int M, int batch, int n_always, int keep_lo, int keep_hi,
                          void* logdet, int* info, void* stream) {
  using namespace gpsa;
  if (M < 1 || batch < 1 || n_always < 0 || n_always > batch) return GPSA_EINVAL;
  if (M > 256) return GPSA_EUNSUPPORTED;
  if (keep_lo < n_always) keep_lo = n_always;
  if (keep_hi > batch) keep_hi = batch;
  if (keep_hi < keep_lo) keep_hi = keep_lo;
  const int n = n_always + (keep_hi - keep_lo), gap = keep_lo - n_always;
  if (n == 0) return 0;
  static const bool blocked = [] { const char* e = getenv("GPSA_CHOL_BLOCKED"); return !(e && e[0] == '0'); }();
  if (blocked && M <= 208) {
    int rc = chol_inv_blk_launch((const double*)A, M, (double*)Linv, (double*)logdet, info, n, as_stream(stream),
                                 n_always, gap);
    if (rc != GPSA_EUNSUPPORTED) return rc;
  }
  return chol_inv_reg_launch((const double*)A, M, (double*)Linv, (double*)logdet, info, M, (long long)M * M, M,
                             (long long)M * M, 0, 0, n, as_stream(stream), n_always, gap);
}

long long gpsa_chol_inv_blocked_workspace(int M, int batch) {
  if (M < 1 || batch < 1) return 0;
  const int bs = gpsa::chol_block_size(M);
  return ((long long)2 * M * M + (long long)bs * M) * batch * 8;
}

int gpsa_chol_inv_blocked_f64(const void* A, void* Linv, int M, int batch, void* logdet, int* info,
                              void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || batch < 1) return GPSA_EINVAL;
  if (workspace_bytes < gpsa_chol_inv_blocked_workspace(M, batch)) return GPSA_EWORKSPACE;
  return chol_inv_blocked((const double*)A, (double*)Linv, M, batch, (double*)logdet, info,
                          (double*)workspace, as_stream(stream));
}

int gpsa_bdot(int dtype, const void* A, long long strideA, const void* B, long long strideB,
              long long n, int batch, void* out, void* workspace, long long workspace_bytes,
              void* stream) {
  if (n < 1 || batch < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  int ns = (int)((n + 2047) / 2048);
  if (ns > 32) ns = 32;
  if (workspace_bytes < (long long)batch * ns * 8) return GPSA_EWORKSPACE;
  double* part = (double*)workspace;
  dim3 grid((unsigned)batch, (unsigned)ns);
  if (dtype == GPSA_F32) {
    gpsa::bdot_kernel<float><<<grid, 256, 0, st>>>((const float*)A, strideA, (const float*)B, strideB, n, part);
    gpsa::bdot_finish_kernel<float><<<(batch + 63) / 64, 64, 0, st>>>(part, ns, batch, (float*)out);
  } else if (dtype == GPSA_F64) {
    gpsa::bdot_kernel<double><<<grid, 256, 0, st>>>((const double*)A, strideA, (const double*)B, strideB, n, part);
    gpsa::bdot_finish_kernel<double><<<(batch + 63) / 64, 64, 0, st>>>(part, ns, batch, (double*)out);
  } else {
    return GPSA_EINVAL;
  }
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_add_diag(int dtype, void* A, int M, int batch, double s, void* stream) {
  if (M < 1 || batch < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  const unsigned nb = (unsigned)cdiv((long long)M * batch, 256);
  if (dtype == GPSA_F32)
    gpsa::add_diag_kernel<float><<<nb, 256, 0, st>>>((float*)A, M, batch, (float)s);
  else if (dtype == GPSA_F64)
    gpsa::add_diag_kernel<double><<<nb, 256, 0, st>>>((double*)A, M, batch, s);
  else
    return GPSA_EINVAL;
  GPSA_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
