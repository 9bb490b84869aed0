// Covariance-matrix generation and its backward (RBF / Matern-1/2 / Matern-3/2).
// Replaces gpsa/util/util.py:8-66 of the reference as called from gpsa/models/vgpsa.py:314-318,
// 390-392, 409.  HBM/VALU-bound: coalesced reads of the [C,D] coordinate block, Z staged in LDS,
// wavefront shuffle reductions for the per-inducing-point gradient sums.
#include "common.hpp"
#include "kmat_cov.hpp"

namespace gpsa {

constexpr int KM_ROWS = 16;

// A batch of covariance problems in ONE launch (blockIdx.z = problem b): the views' warp GPs of a step.
// Every problem has the same M, D and column stride C; its operands sit at fixed strides from the first
// problem's (elements): Z + b sZ, X + b sX, K / Kbar + b sK, the two hyper-parameters + b sP.  n[b] <= C is
// the number of LIVE columns of problem b: the forward writes zeros beyond it (the padding of the
// view-blocked layout: everything computed from a zero column is zero), the backward skips them.
constexpr int KM_MAXB = 16;
struct KmatBatch {
  long long sZ, sX, sK;
  int sP, ragged;  // ragged: use n[]; otherwise every problem has C live columns
  long long n[KM_MAXB];
  // backward only, optional: the gradient panel is Kbar[m,c] + axs * axd[c] * axX[m,c] (axX: a panel of Kbar's own
  // storage type - fp32 next to an fp32 Kbar, fp64 next to the exact mode's fp64 one; axd: fp32 vector) - the data
  // GP's dK_uf = gamma + 2 qbar o alpha formed while it is read, in the kernel's arithmetic type
  const void* axX;
  const float* axd;
  float axs;
};
static inline KmatBatch kmat_single() {
  KmatBatch kb;
  kb.sZ = kb.sX = kb.sK = 0;
  kb.sP = kb.ragged = 0;
  kb.axX = nullptr;
  kb.axd = nullptr;
  kb.axs = 0.f;
  for (int i = 0; i < KM_MAXB; ++i) kb.n[i] = 0;
  return kb;
}

// TI: storage type of the coordinates and hyper-parameters (the fp32 parameters are read as they are,
// no cast launches); T: type the covariance is computed and stored in.
// TX: storage type of X alone (the data GP reads the warp GP's unrounded fp64 draws next to fp32 parameters).
template <typename TI, typename TX, typename T, int KIND>
__global__ void __launch_bounds__(256)
kmat_fwd_kernel(const TI* __restrict__ Z, int M, const TX* __restrict__ X, long long C, int D,
                const TI* __restrict__ ls_u, const TI* __restrict__ var_u, T jitter,
                T* __restrict__ K, KmatBatch kb) {
  __shared__ T Zs[KM_ROWS][MAXD];
  const int b = blockIdx.z;
  Z += b * kb.sZ;
  X += b * kb.sX;
  K += b * kb.sK;
  ls_u += b * kb.sP;
  var_u += b * kb.sP;
  const long long live = kb.ragged ? kb.n[b] : C;
  const int m0 = blockIdx.y * KM_ROWS;
  if (threadIdx.x < KM_ROWS * MAXD) {
    int r = threadIdx.x / MAXD, d = threadIdx.x % MAXD;
    Zs[r][d] = (m0 + r < M && d < D) ? (T)Z[(long long)(m0 + r) * D + d] : T(0);
  }
  __syncthreads();
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  if (c >= C) return;
  const int mend = min(KM_ROWS, M - m0);
  if (c >= live) {  // padding column of a view block
    for (int r = 0; r < mend; ++r) K[(long long)(m0 + r) * C + c] = T(0);
    return;
  }
  const T ell = t_exp<T>((T)ls_u[0]), inv_ell = T(1) / ell, var = t_exp<T>((T)var_u[0]);
  T x[MAXD];
#pragma unroll
  for (int d = 0; d < MAXD; ++d) x[d] = (d < D) ? (T)X[c * D + d] : T(0);
  for (int r = 0; r < mend; ++r) {
    T k, cd, pl;
    cov_eval<T, KIND>(Zs[r], x, D, ell, inv_ell, var, k, cd, pl);
    if ((long long)(m0 + r) == c) k += jitter;
    K[(long long)(m0 + r) * C + c] = k;
  }
}

constexpr int KB_MCHUNK = 32;  // inducing rows per workgroup of the backward kernel ...
constexpr int KB_MCHUNK_SMALL = 8;  // ... and for problems that would otherwise leave most CUs idle
static inline int kb_rows(int M, long long C) {
  // K_uu (200 x 200) or one view's K_uf at a 1/8 shard are 7 - 35 workgroups with 32 rows each: a long
  // serial exp / reduce chain per thread on a nearly empty chip; 8 rows per workgroup quarter it
  return (cdiv(C, 256) * cdiv(M, KB_MCHUNK) < 1024) ? KB_MCHUNK_SMALL : KB_MCHUNK;
}
constexpr int KB_MAXBX = 1024;  // column-block workgroups per row chunk (beyond: each walks several blocks)
// the register-accumulating backward (kmat_bwd_d2_kernel): D = 2 and a problem big enough for KB_MCHUNK rows
constexpr int KB_D2_ROWS = 16;  // (24 rows per workgroup spill: 300 us)
static inline int kmat_bwd_d2_rows() { return KB_D2_ROWS; }
static inline bool kmat_bwd_d2(int M, long long Ctot, int D) {
  // (every size measured is faster this way: 117 -> 70 us at C = 100k, 41 -> 25 at 20k, 28 -> 19 at 12.5k,
  //  15 -> 11 at 2.5k; M = 200.  GPSA_KMAT_BWD_D2=0: the older kernel)
  static const bool off = [] { const char* e = getenv("GPSA_KMAT_BWD_D2"); return e && e[0] == '0'; }();
  (void)M;
  (void)Ctot;
  return !off && D == 2;
}

// grid (column blocks of 256, row chunks of KB_MCHUNK): one thread per column c, looping over the
// chunk's inducing rows.  Deterministic partials:
//   zpart[bx][m*D+d]      dZ contribution of column block bx (rows of chunk by only)
//   xpart[by][c*D+d]      dX contribution of row chunk by
//   spart[bx*ny+by][0..1] d ls_u, d var_u
// TK: storage type of Kbar (an fp32 gradient panel can be contracted in fp64: T = double)
template <typename TI, typename T, int KIND, int MCH, typename TK = T, typename TX = TI>
__global__ void __launch_bounds__(256)
kmat_bwd_kernel(const TI* __restrict__ Z, int M, const TX* __restrict__ X, long long C, int D,
                const TI* __restrict__ ls_u, const TI* __restrict__ var_u,
                const TK* __restrict__ Kbar, T* __restrict__ zpart, T* __restrict__ xpart,
                T* __restrict__ spart, KmatBatch bt) {
  {  // problem b of the batch: operands at fixed strides, partial sums in its own workspace region
    const long long b = blockIdx.z;
    Z += b * bt.sZ;
    X += b * bt.sX;
    Kbar += b * bt.sK;
    ls_u += b * bt.sP;
    var_u += b * bt.sP;
    zpart += b * (long long)gridDim.x * M * D;
    if (xpart != nullptr) xpart += b * (long long)gridDim.y * C * D;
    spart += b * (long long)gridDim.x * gridDim.y * 2;
  }
  const long long nlive = bt.ragged ? bt.n[blockIdx.z] : C;
  __shared__ T Zs[MCH][MAXD];
  __shared__ T acc[4][MCH][MAXD];
  __shared__ T red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m0 = blockIdx.y * MCH;
  const int mc = min(MCH, M - m0);
  const T ell = t_exp<T>((T)ls_u[0]), inv_ell = T(1) / ell, var = t_exp<T>((T)var_u[0]);
  if (threadIdx.x < MCH * MAXD) {
    const int r = threadIdx.x / MAXD, d = threadIdx.x % MAXD;
    Zs[r][d] = (r < mc && d < D) ? (T)Z[(long long)(m0 + r) * D + d] : T(0);
  }
  for (int i = threadIdx.x; i < 4 * MCH * MAXD; i += 256) (&acc[0][0][0])[i] = T(0);
  __syncthreads();
  T s_ls = T(0), s_var = T(0);
  // a workgroup walks column blocks bx, bx + gridDim.x, ...: the per-row sums of all of them meet in
  // LDS, so zpart has gridDim.x (<= KB_MAXBX) rows however many columns K has
  const long long ncb = (C + 255) / 256;
  for (long long cb = blockIdx.x; cb < ncb; cb += gridDim.x) {
    const long long c = cb * 256 + threadIdx.x;
    const bool live = c < nlive;
    T x[MAXD], dx[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
      x[d] = (live && d < D) ? (T)X[c * D + d] : T(0);
      dx[d] = T(0);
    }
    const T axc = (bt.axX != nullptr && live) ? (T)bt.axs * (T)bt.axd[c] : T(0);
    for (int r = 0; r < mc; ++r) {
      T k, cd, pl;
      cov_eval<T, KIND>(Zs[r], x, D, ell, inv_ell, var, k, cd, pl);
      T kb = live ? (T)Kbar[(long long)(m0 + r) * C + c] : T(0);
      if (bt.axX != nullptr && live)  // uniform branch
        kb += axc * (T)reinterpret_cast<const TK*>(bt.axX)[(long long)(m0 + r) * C + c];
      s_ls += kb * pl;
      s_var += kb * k;
      T wgt = kb * cd;
#pragma unroll
      for (int d = 0; d < MAXD; ++d)
        if (d < D) {
          T t = wgt * (Zs[r][d] - x[d]);  // dLoss/dz_d contribution ; dLoss/dx_d = -t
          dx[d] -= t;
          T tz = wave_sum(t);
          if (lane == 0) acc[w][r][d] += tz;
        }
    }
    if (xpart != nullptr && c < C) {
      T* xr = xpart + (long long)blockIdx.y * C * D;
      for (int d = 0; d < D; ++d) xr[c * D + d] = dx[d];
    }
  }
  __syncthreads();
  T* zrow = zpart + (long long)blockIdx.x * M * D;
  for (int i = threadIdx.x; i < mc * D; i += 256) {
    const int r = i / D, d = i % D;
    zrow[(long long)(m0 + r) * D + d] = acc[0][r][d] + acc[1][r][d] + acc[2][r][d] + acc[3][r][d];
  }
  T* sp = spart + ((long long)blockIdx.x * gridDim.y + blockIdx.y) * 2;
  T a = block_sum(s_ls, red);
  if (threadIdx.x == 0) sp[0] = a;
  T b2 = block_sum(s_var, red);
  if (threadIdx.x == 0) sp[1] = b2;
}

// The same for D = 2 and KB_MCHUNK rows per workgroup (round 4): a thread keeps the 2 KB_MCHUNK dZ sums of its
// columns in REGISTERS over all the column blocks its workgroup walks and the workgroup reduces them once, through
// LDS, at the end - the kernel above pays two fp64 cross-lane reductions (~50 instructions) per (row, column block)
// next to ~45 of covariance arithmetic.  Same partial arrays (zpart has gridDim.x rows: 6 times fewer at the
// headline size, so the finishing launch shrinks with it), same fixed summation order from run to run.
template <typename TI, typename T, int KIND, int MCH, typename TK = T, typename TX = TI>
__global__ void __launch_bounds__(256, 2)
kmat_bwd_d2_kernel(const TI* __restrict__ Z, int M, int rows, const TX* __restrict__ X, long long C,
                   const TI* __restrict__ ls_u, const TI* __restrict__ var_u,
                   const TK* __restrict__ Kbar, T* __restrict__ zpart, T* __restrict__ xpart,
                   T* __restrict__ spart, KmatBatch bt) {
  constexpr int D = 2;
  static_assert(MCH % 8 == 0, "the closing reduction takes 8 rows at a time");
  {
    const long long b = blockIdx.z;
    Z += b * bt.sZ;
    X += b * bt.sX;
    Kbar += b * bt.sK;
    ls_u += b * bt.sP;
    var_u += b * bt.sP;
    zpart += b * (long long)gridDim.x * M * D;
    if (xpart != nullptr) xpart += b * (long long)gridDim.y * C * D;
    spart += b * (long long)gridDim.x * gridDim.y * 2;
  }
  const long long nlive = bt.ragged ? bt.n[blockIdx.z] : C;
  constexpr int RSTR = 256 + 16;                // padded row of the reduction buffer (doubles)
  __shared__ T Zs[MCH][MAXD];
  __shared__ T rbuf[16 * RSTR];                 // 16 values x 256 threads at a time
  __shared__ T red[4];
  // rows <= MCH inducing rows per workgroup (the launcher spreads M evenly over the row chunks); the rows past mc
  // run with weight zero on a clamped address - no branch in the row loop, so its loads can be issued ahead
  const int m0 = blockIdx.y * rows;
  const int mc = min(rows, M - m0);
  const T ell = t_exp<T>((T)ls_u[0]), inv_ell = T(1) / ell, var = t_exp<T>((T)var_u[0]);
  if (threadIdx.x < MCH * MAXD) {
    const int r = threadIdx.x / MAXD, d = threadIdx.x % MAXD;
    Zs[r][d] = (r < mc && d < D) ? (T)Z[(long long)(m0 + r) * D + d] : T(0);
  }
  __syncthreads();
  T tz[MCH][D];
#pragma unroll
  for (int r = 0; r < MCH; ++r) tz[r][0] = tz[r][1] = T(0);
  T s_ls = T(0), s_var = T(0);
  const long long ncb = (C + 255) / 256;
  // A column block's operands first - its gradient values as stored (fp32 panels: 2 MCH registers), the thread's
  // point and column scale - then the arithmetic.
  struct Col {
    long long c;
    bool live;
    T x0, x1, axc;
  };
  const bool two = bt.axX != nullptr;  // (uniform)
  const TK* __restrict__ axp = reinterpret_cast<const TK*>(bt.axX);
  auto load = [&](long long cb, TK(&kraw)[MCH], TK(&araw)[MCH], Col& q) {
    q.c = cb * 256 + threadIdx.x;
    q.live = q.c < nlive;
    q.x0 = q.x1 = q.axc = T(0);
    const long long cc = q.live ? q.c : 0;
    if (q.live) {
      q.x0 = (T)X[q.c * D];
      q.x1 = (T)X[q.c * D + 1];
    }
    // (ONE uniform branch around the loads: a select per row made each second-panel load a branch of its own with a
    //  full wait behind it - sixteen memory latencies in a row)
    if (two) {
      if (q.live) q.axc = (T)bt.axs * (T)bt.axd[q.c];
#pragma unroll
      for (int r = 0; r < MCH; ++r) {
        const long long o = (long long)min(m0 + r, M - 1) * C + cc;
        kraw[r] = Kbar[o];
        araw[r] = axp[o];
      }
    } else {
#pragma unroll
      for (int r = 0; r < MCH; ++r) {
        kraw[r] = Kbar[(long long)min(m0 + r, M - 1) * C + cc];
        araw[r] = TK(0);
      }
    }
  };
  auto compute = [&](const TK(&kraw)[MCH], const TK(&araw)[MCH], const Col& q) {
    const T x[MAXD] = {q.x0, q.x1, T(0), T(0)};
    T dx0 = T(0), dx1 = T(0);
    // (the inducing rows are re-read from LDS in every column block: hoisted out of the loop they are 4 MCH more
    //  registers next to the 4 MCH of the sums; the opaque zero keeps them inside)
    int zo = 0;
    asm volatile("" : "+v"(zo));
    const T(*Zr)[MAXD] = Zs + zo;
#pragma unroll
    for (int r = 0; r < MCH; ++r) {
      T k, cd, pl;
      cov_eval<T, KIND>(Zr[r], x, D, ell, inv_ell, var, k, cd, pl);
      const T kb = (q.live && r < mc) ? (T)kraw[r] + q.axc * (T)araw[r] : T(0);
      s_ls += kb * pl;
      s_var += kb * k;
      const T wgt = kb * cd;
      const T t0 = wgt * (Zr[r][0] - x[0]), t1 = wgt * (Zr[r][1] - x[1]);
      dx0 -= t0;
      dx1 -= t1;
      tz[r][0] += t0;
      tz[r][1] += t1;
      // (four rows' worth of independent exp chains per scheduling region: all of them interleaved spill)
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (xpart != nullptr && q.c < C) {
      T* xr = xpart + (long long)blockIdx.y * C * D;
      xr[q.c * D] = dx0;
      xr[q.c * D + 1] = dx1;
    }
  };
  // (fetching the next block's operands while this one is worked on - a second register set - was slower:
  //  99 against 70 us at the headline size, one workgroup per CU instead of two)
  for (long long cb = blockIdx.x; cb < ncb; cb += gridDim.x) {
    TK k0[MCH];
    TK a0[MCH];
    Col q0;
    load(cb, k0, a0, q0);
    __builtin_amdgcn_sched_barrier(0);
    compute(k0, a0, q0);
  }
  // the workgroup's sums, 8 rows (16 values) at a time: value v of every thread -> rbuf[v][thread]; thread t then
  // adds the 16 entries (t & 15) + 16 i of value t >> 4 and the 16 lanes of a value meet through DPP
  T* zrow = zpart + (long long)blockIdx.x * M * D;
#pragma unroll
  for (int r0 = 0; r0 < MCH; r0 += 8) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      rbuf[(2 * q) * RSTR + threadIdx.x] = tz[r0 + q][0];
      rbuf[(2 * q + 1) * RSTR + threadIdx.x] = tz[r0 + q][1];
    }
    __syncthreads();
    const int v = threadIdx.x >> 4, sub = threadIdx.x & 15;
    T a = T(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) a += rbuf[v * RSTR + sub + 16 * i];
    a += dpp_move<0xB1>(a);
    a += dpp_move<0x4E>(a);
    a += dpp_move<0x141>(a);
    a += dpp_move<0x140>(a);
    const int r = r0 + (v >> 1);
    if (sub == 0 && r < mc) zrow[(long long)(m0 + r) * D + (v & 1)] = a;
  }
  T* sp = spart + ((long long)blockIdx.x * gridDim.y + blockIdx.y) * 2;
  T a = block_sum(s_ls, red);
  if (threadIdx.x == 0) sp[0] = a;
  T b2 = block_sum(s_var, red);
  if (threadIdx.x == 0) sp[1] = b2;
}

// Second pass of the backward: ONE launch sums all three partial arrays in a fixed order
//   dZ[i] = sum_bx zpart[bx][i]  (+ sum_by xpart[by][i] when Z and X are the same points: K_uu)
//   dX[i] = sum_by xpart[by][i] ;  dparams[0..1] = sum spart
// A block serves consecutive outputs of ONE of the three arrays: 16 outputs x 16 row groups for the tall,
// narrow zpart (one row per 256 columns of K: 391 rows x 400 outputs at the headline size), 64 outputs x
// 4 row groups for the short, wide xpart; 4 independent partial sums per thread either way.
template <typename T, typename TO>
__global__ void __launch_bounds__(256)
kmat_bwd_finish_kernel(const T* __restrict__ zpart, long long nbx, long long nz,
                       const T* __restrict__ xpart, long long nby, long long nx,
                       const T* __restrict__ spart, long long ns, int same, TO* __restrict__ dZ,
                       TO* __restrict__ dX, TO* __restrict__ dparams, long long sdZ, long long sdX) {
  {  // problem blockIdx.y of a batch (strides of the results in elements; partials are back to back)
    const long long b = blockIdx.y;
    zpart += b * nbx * nz;
    if (xpart != nullptr) xpart += b * nby * nx;
    spart += b * ns * 2;
    dZ += b * sdZ;
    if (dX != nullptr) dX += b * sdX;
    dparams += b * 2;
  }
  __shared__ double red[256];
  const long long zb = (nz + 15) / 16, xb = (nx + 63) / 64;
  const T* part;
  long long rows, stride, n, col;
  int ng, grp, sub;  // row groups, this thread's group, its output within the block
  TO* dst;
  long long blk = blockIdx.x;
  bool fold = false;
  if (blk < zb) {
    ng = 16; grp = threadIdx.x >> 4; sub = threadIdx.x & 15;
    part = zpart; rows = nbx; stride = nz; n = nz; col = blk * 16 + sub; dst = dZ; fold = same != 0;
  } else if (blk < zb + xb) {
    blk -= zb;
    if (xpart == nullptr || dX == nullptr || same) return;  // not requested / folded into dZ (uniform)
    ng = 4; grp = threadIdx.x >> 6; sub = threadIdx.x & 63;
    part = xpart; rows = nby; stride = nx; n = nx; col = blk * 64 + sub; dst = dX;
  } else {
    ng = 16; grp = threadIdx.x >> 4; sub = threadIdx.x & 15;
    part = spart; rows = ns; stride = 2; n = 2; col = sub; dst = dparams;
  }
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (col < n) {
    long long r = grp;
    for (; r + 3 * ng < rows; r += 4 * ng) {
      a0 += (double)part[r * stride + col];
      a1 += (double)part[(r + ng) * stride + col];
      a2 += (double)part[(r + 2 * ng) * stride + col];
      a3 += (double)part[(r + 3 * ng) * stride + col];
    }
    for (; r < rows; r += ng) a0 += (double)part[r * stride + col];
    if (fold)  // K_uu: both arguments are the inducing points (nx == nz)
      for (long long q = grp; q < nby; q += ng) a1 += (double)xpart[q * nx + col];
  }
  const int width = 256 / ng;
  red[grp * width + sub] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && col < n) {
    double t = 0.0;
    for (int q = 0; q < ng; ++q) t += red[q * width + sub];
    dst[col] = (TO)t;
  }
}

template <typename TI, typename TX, typename T>
int kmat_launch(int kind, const TI* Z, int M, const TX* X, long long C, int D, const TI* ls_u,
                const TI* var_u, double jitter, T* K, hipStream_t st, int batch = 1,
                KmatBatch kb = kmat_single()) {
  if (batch < 1 || batch > KM_MAXB) return GPSA_EINVAL;
  dim3 grid((unsigned)cdiv(C, 256), (unsigned)cdiv(M, KM_ROWS), (unsigned)batch);
  switch (kind) {
    case GPSA_K_RBF:
      kmat_fwd_kernel<TI, TX, T, GPSA_K_RBF><<<grid, 256, 0, st>>>(Z, M, X, C, D, ls_u, var_u, (T)jitter, K, kb);
      break;
    case GPSA_K_MATERN12:
      kmat_fwd_kernel<TI, TX, T, GPSA_K_MATERN12><<<grid, 256, 0, st>>>(Z, M, X, C, D, ls_u, var_u, (T)jitter, K, kb);
      break;
    case GPSA_K_MATERN32:
      kmat_fwd_kernel<TI, TX, T, GPSA_K_MATERN32><<<grid, 256, 0, st>>>(Z, M, X, C, D, ls_u, var_u, (T)jitter, K, kb);
      break;
    default:
      return GPSA_EINVAL;
  }
  GPSA_LAUNCH_CHECK();
  return 0;
}

// TO: storage type of the gradients (the fp64 backward of the data GP keeps them fp64 from fp32 inputs)
template <typename TI, typename T, typename TO = TI, typename TK = T, typename TX = TI>
int kmat_bwd_launch(int kind, const TI* Z, int M, const TX* X, long long C, int D, const TI* ls_u,
                    const TI* var_u, const TK* Kbar, TO* dZ, TO* dX, TO* dparams, int same, void* ws,
                    long long ws_bytes, hipStream_t st, int batch = 1, KmatBatch kb = kmat_single(),
                    long long sdZ = 0, long long sdX = 0) {
  if (batch < 1 || batch > KM_MAXB) return GPSA_EINVAL;
  const int mch = kb_rows(M, C * batch);
  const long long ncb = cdiv(C, 256), nby = cdiv(M, mch);
  long long nbx = ncb < KB_MAXBX ? ncb : KB_MAXBX;
  const long long nz = (long long)M * D, nx = C * D;
  const long long need = (nbx * nz + nby * nx + nbx * nby * 2) * (long long)sizeof(T) * batch;
  const bool d2 = kmat_bwd_d2(M, C * batch, D);
  if (!d2 && ws_bytes < need) return GPSA_EWORKSPACE;
  T* zpart = reinterpret_cast<T*>(ws);
  T* xpart = zpart + nbx * nz * batch;
  T* spart = xpart + nby * nx * batch;
  T* xp = (dX || same) ? xpart : nullptr;  // K_uu: the X-side partials are folded into dZ
  if (d2) {
    // register-accumulating form (its own, finer row chunks: see kmat_bwd_d2_kernel)
    const int mr = kmat_bwd_d2_rows();
    const long long nby2 = cdiv(M, mr);
    static const int per_env = [] { const char* e = getenv("GPSA_KMAT_BWD_PER"); return e ? atoi(e) : 0; }();
    // column blocks per workgroup: the fewest with which the whole grid is resident at once - two workgroups per CU
    // (launch bounds; ~200 registers).  Round 6, the exact data-GP backward at C = 100k (391 column blocks x 13 row
    // chunks): 8 blocks each = 637 workgroups ran as one full round and a quarter-full second one, 88 - 90 us; 10 (520
    // workgroups, 8 too many) 95 us; 11 (468) 74 us; 12 / 13 / 16 / 20: 76 / 78 / 88 / 98 us (profiles/r06_kmat_bwd_per.txt)
    const long long slots = 2LL * num_cus();
    long long per = 1;
    while (per < ncb && cdiv(ncb, per) * nby2 * batch > slots) ++per;
    if (per_env > 0) per = per_env;
    nbx = cdiv(ncb, per);
    const int rows = (int)cdiv(M, nby2);  // M spread evenly over the row chunks
    if ((nbx * nz + nby2 * nx + nbx * nby2 * 2) * (long long)sizeof(T) * batch > ws_bytes) return GPSA_EWORKSPACE;
    xpart = zpart + nbx * nz * batch;
    spart = xpart + nby2 * nx * batch;
    xp = (dX || same) ? xpart : nullptr;
    dim3 grid((unsigned)nbx, (unsigned)nby2, (unsigned)batch);
#define GPSA_KB_CASE(KIND)                                                                                     \
  kmat_bwd_d2_kernel<TI, T, KIND, KB_D2_ROWS, TK, TX><<<grid, 256, 0, st>>>(Z, M, rows, X, C, ls_u, var_u, Kbar, \
                                                                            zpart, xp, spart, kb);
    switch (kind) {
      case GPSA_K_RBF: GPSA_KB_CASE(GPSA_K_RBF) break;
      case GPSA_K_MATERN12: GPSA_KB_CASE(GPSA_K_MATERN12) break;
      case GPSA_K_MATERN32: GPSA_KB_CASE(GPSA_K_MATERN32) break;
      default:
        return GPSA_EINVAL;
    }
#undef GPSA_KB_CASE
    GPSA_LAUNCH_CHECK();
    const bool fold2 = same != 0;
    if (fold2 && (nx != nz)) return GPSA_EINVAL;
    dim3 fgrid2((unsigned)(cdiv(nz, 16) + cdiv(nx, 64) + 1), (unsigned)batch);
    kmat_bwd_finish_kernel<T, TO><<<fgrid2, 256, 0, st>>>(zpart, nbx, nz, xp, nby2, nx, spart, nbx * nby2,
                                                          fold2 ? 1 : 0, dZ, dX, dparams, sdZ, sdX);
    GPSA_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((unsigned)nbx, (unsigned)nby, (unsigned)batch);
#define GPSA_KB_CASE(KIND)                                                                          \
  if (mch == KB_MCHUNK)                                                                             \
    kmat_bwd_kernel<TI, T, KIND, KB_MCHUNK, TK, TX><<<grid, 256, 0, st>>>(Z, M, X, C, D, ls_u, var_u, \
                                                                      Kbar, zpart, xp, spart, kb);  \
  else                                                                                              \
    kmat_bwd_kernel<TI, T, KIND, KB_MCHUNK_SMALL, TK, TX><<<grid, 256, 0, st>>>(Z, M, X, C, D, ls_u, \
                                                                            var_u, Kbar, zpart, xp, \
                                                                            spart, kb);
  switch (kind) {
    case GPSA_K_RBF: GPSA_KB_CASE(GPSA_K_RBF) break;
    case GPSA_K_MATERN12: GPSA_KB_CASE(GPSA_K_MATERN12) break;
    case GPSA_K_MATERN32: GPSA_KB_CASE(GPSA_K_MATERN32) break;
    default:
      return GPSA_EINVAL;
  }
#undef GPSA_KB_CASE
  GPSA_LAUNCH_CHECK();
  const bool fold = same != 0;
  if (fold && (nx != nz)) return GPSA_EINVAL;
  dim3 fgrid((unsigned)(cdiv(nz, 16) + cdiv(nx, 64) + 1), (unsigned)batch);
  kmat_bwd_finish_kernel<T, TO><<<fgrid, 256, 0, st>>>(zpart, nbx, nz, xp, nby, nx, spart, nbx * nby,
                                                       fold ? 1 : 0, dZ, dX, dparams, sdZ, sdX);
  GPSA_LAUNCH_CHECK();
  return 0;
}

}  // namespace gpsa

extern "C" {

int gpsa_kmat(int dtype, int in_dtype, int kind, const void* Z, int M, const void* X, long long C,
              int D, const void* ls_u, const void* var_u, double jitter, void* K, void* stream) {
  if (D < 1 || D > gpsa::MAXD || M < 1 || C < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32 && in_dtype == GPSA_F32)
    return gpsa::kmat_launch<float, float, float>(kind, (const float*)Z, M, (const float*)X, C, D,
                                           (const float*)ls_u, (const float*)var_u, jitter, (float*)K, st);
  if (dtype == GPSA_F64 && in_dtype == GPSA_F64)
    return gpsa::kmat_launch<double, double, double>(kind, (const double*)Z, M, (const double*)X, C, D,
                                             (const double*)ls_u, (const double*)var_u, jitter, (double*)K, st);
  if (dtype == GPSA_F64 && in_dtype == GPSA_F32)
    return gpsa::kmat_launch<float, float, double>(kind, (const float*)Z, M, (const float*)X, C, D,
                                                   (const float*)ls_u, (const float*)var_u, jitter, (double*)K, st);
  if (dtype == GPSA_F64 && in_dtype == GPSA_F32_X64)
    return gpsa::kmat_launch<float, double, double>(kind, (const float*)Z, M, (const double*)X, C, D,
                                                    (const float*)ls_u, (const float*)var_u, jitter, (double*)K, st);
  return GPSA_EINVAL;
}

long long gpsa_kmat_bwd_workspace(int dtype, int M, long long C, int D) {
  const long long nbx = cdiv(C, 256);
  long long nby = cdiv(M, gpsa::kb_rows(M, C));
  if (gpsa::kmat_bwd_d2(M, C, D)) nby = cdiv(M, gpsa::kmat_bwd_d2_rows());  // (finer row chunks: more dX partials)
  return (nbx * M * D + nby * C * D + nbx * nby * 2) * (dtype == GPSA_F64 ? 8 : 4);
}

int gpsa_kmat_bwd(int dtype, int in_dtype, int kind, const void* Z, int M, const void* X, long long C,
                  int D, const void* ls_u, const void* var_u, const void* Kbar, int same, void* dZ,
                  void* dX, void* dparams, void* workspace, long long workspace_bytes, void* stream) {
  if (D < 1 || D > gpsa::MAXD || M < 1 || C < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32 && in_dtype == GPSA_F32)
    return gpsa::kmat_bwd_launch<float, float>(kind, (const float*)Z, M, (const float*)X, C, D,
                                               (const float*)ls_u, (const float*)var_u, (const float*)Kbar,
                                               (float*)dZ, (float*)dX, (float*)dparams, same, workspace,
                                               workspace_bytes, st);
  if (dtype == GPSA_F64 && in_dtype == GPSA_F64)
    return gpsa::kmat_bwd_launch<double, double>(kind, (const double*)Z, M, (const double*)X, C, D,
                                                 (const double*)ls_u, (const double*)var_u,
                                                 (const double*)Kbar, (double*)dZ, (double*)dX,
                                                 (double*)dparams, same, workspace, workspace_bytes, st);
  if (dtype == GPSA_F64 && in_dtype == GPSA_F32)
    return gpsa::kmat_bwd_launch<float, double>(kind, (const float*)Z, M, (const float*)X, C, D,
                                                (const float*)ls_u, (const float*)var_u,
                                                (const double*)Kbar, (float*)dZ, (float*)dX,
                                                (float*)dparams, same, workspace, workspace_bytes, st);
  if (dtype == GPSA_F64 && in_dtype == GPSA_F32_OUT64)
    return gpsa::kmat_bwd_launch<float, double, double>(kind, (const float*)Z, M, (const float*)X, C, D,
                                                        (const float*)ls_u, (const float*)var_u,
                                                        (const double*)Kbar, (double*)dZ, (double*)dX,
                                                        (double*)dparams, same, workspace, workspace_bytes, st);
  if (dtype == GPSA_F64 && in_dtype == GPSA_F32_ACC64)
    return gpsa::kmat_bwd_launch<float, double, double, float>(
        kind, (const float*)Z, M, (const float*)X, C, D, (const float*)ls_u, (const float*)var_u,
        (const float*)Kbar, (double*)dZ, (double*)dX, (double*)dparams, same, workspace, workspace_bytes, st);
  return GPSA_EINVAL;
}

/* ---- batched forms (the views' warp GPs of a step in ONE launch each way; see KmatBatch) -------------- */
static gpsa::KmatBatch make_batch(long long sZ, long long sX, long long sK, int sP, const long long* n_live,
                                  int batch) {
  gpsa::KmatBatch kb = gpsa::kmat_single();
  kb.sZ = sZ;
  kb.sX = sX;
  kb.sK = sK;
  kb.sP = sP;
  kb.ragged = n_live != nullptr;
  for (int i = 0; i < batch && i < gpsa::KM_MAXB; ++i) kb.n[i] = n_live ? n_live[i] : 0;
  return kb;
}

int gpsa_kmat_batched(int kind, const float* Z, long long strideZ, int M, const float* X, long long strideX,
                      long long C, int D, const float* ls_u, const float* var_u, int stride_par,
                      const long long* n_live, int batch, double jitter, double* K, long long strideK,
                      void* stream) {
  if (D < 1 || D > gpsa::MAXD || M < 1 || C < 1 || batch < 1 || batch > gpsa::KM_MAXB) return GPSA_EINVAL;
  return gpsa::kmat_launch<float, float, double>(kind, Z, M, X, C, D, ls_u, var_u, jitter, K, as_stream(stream),
                                                 batch, make_batch(strideZ, strideX, strideK, stride_par, n_live, batch));
}

long long gpsa_kmat_bwd_batched_workspace(int M, long long C, int D, int batch) {
  const int mch = gpsa::kb_rows(M, C * batch);
  const long long ncb = cdiv(C, 256), nbx = ncb < gpsa::KB_MAXBX ? ncb : gpsa::KB_MAXBX;
  long long nby = cdiv(M, mch);
  if (gpsa::kmat_bwd_d2(M, C * batch, D)) nby = cdiv(M, gpsa::kmat_bwd_d2_rows());
  return (nbx * M * D + nby * C * D + nbx * nby * 2) * 8LL * batch;
}

int gpsa_kmat_bwd_batched(int kind, const float* Z, long long strideZ, int M, const float* X, long long strideX,
                          long long C, int D, const float* ls_u, const float* var_u, int stride_par,
                          const long long* n_live, int batch, const double* Kbar, long long strideK, int same,
                          double* dZ, long long stride_dZ, double* dparams, void* workspace,
                          long long workspace_bytes, void* stream) {
  if (D < 1 || D > gpsa::MAXD || M < 1 || C < 1 || batch < 1 || batch > gpsa::KM_MAXB) return GPSA_EINVAL;
  return gpsa::kmat_bwd_launch<float, double, double, double, float>(
      kind, Z, M, X, C, D, ls_u, var_u, Kbar, dZ, (double*)nullptr, dparams, same, workspace, workspace_bytes,
      as_stream(stream), batch, make_batch(strideZ, strideX, strideK, stride_par, n_live, batch), stride_dZ, 0);
}

/* The data GP's covariance backward: fp32 Z / hyper-parameters, X = the warp GP's unrounded fp64 draws, an
 * fp32 gradient panel Kbar; fp64 arithmetic, partial sums and results (dZ [M,D], dX [C,D] or NULL,
 * dparams[2]).  workspace >= gpsa_kmat_bwd_workspace(GPSA_F64, M, C, D). */
int gpsa_kmat_bwd_x64(int kind, const float* Z, int M, const double* X, long long C, int D, const float* ls_u,
                      const float* var_u, const float* Kbar, double* dZ, double* dX, double* dparams,
                      void* workspace, long long workspace_bytes, void* stream) {
  return gpsa_kmat_bwd_x64_axpy(kind, Z, M, X, C, D, ls_u, var_u, Kbar, nullptr, nullptr, 0.0, dZ, dX, dparams,
                                workspace, workspace_bytes, stream);
}

int gpsa_kmat_bwd_x64_f64(int kind, const float* Z, int M, const double* X, long long C, int D, const float* ls_u,
                          const float* var_u, const double* Kbar, double* dZ, double* dX, double* dparams,
                          void* workspace, long long workspace_bytes, void* stream) {
  return gpsa_kmat_bwd_x64_f64_axpy(kind, Z, M, X, C, D, ls_u, var_u, Kbar, nullptr, nullptr, 0.0, dZ, dX, dparams,
                                    workspace, workspace_bytes, stream);
}

/* ... with the fp64 panel given in two pieces, Kbar[m,c] + s * d[c] * X2[m,c] (X2 [M,C] fp64, d [C] fp32): the exact
 * inducing-point gradient's dK_uf = K^-1 abar + 2 qbar o alpha on the UNROUNDED projection, without the pass that
 * wrote it out (round 5) */
int gpsa_kmat_bwd_x64_f64_axpy(int kind, const float* Z, int M, const double* X, long long C, int D, const float* ls_u,
                               const float* var_u, const double* Kbar, const double* X2, const float* d, double s,
                               double* dZ, double* dX, double* dparams, void* workspace, long long workspace_bytes,
                               void* stream) {
  if (D < 1 || D > gpsa::MAXD || M < 1 || C < 1) return GPSA_EINVAL;
  if ((X2 == nullptr) != (d == nullptr)) return GPSA_EINVAL;
  gpsa::KmatBatch kb = gpsa::kmat_single();
  kb.axX = X2;
  kb.axd = d;
  kb.axs = (float)s;
  return gpsa::kmat_bwd_launch<float, double, double, double, double>(
      kind, Z, M, X, C, D, ls_u, var_u, Kbar, dZ, dX, dparams, 0, workspace, workspace_bytes, as_stream(stream), 1, kb);
}

/* the same with the gradient panel given in two pieces: Kbar[m,c] + s * d[c] * X2[m,c]  (X2 [M,C], d [C] fp32; both
 * NULL: Kbar alone) - the data GP's dK_uf = K^-1 abar + 2 qbar o alpha without a pass that writes it out */
int gpsa_kmat_bwd_x64_axpy(int kind, const float* Z, int M, const double* X, long long C, int D, const float* ls_u,
                           const float* var_u, const float* Kbar, const float* X2, const float* d, double s,
                           double* dZ, double* dX, double* dparams, void* workspace, long long workspace_bytes,
                           void* stream) {
  if (D < 1 || D > gpsa::MAXD || M < 1 || C < 1) return GPSA_EINVAL;
  if ((X2 == nullptr) != (d == nullptr)) return GPSA_EINVAL;
  gpsa::KmatBatch kb = gpsa::kmat_single();
  kb.axX = X2;
  kb.axd = d;
  kb.axs = (float)s;
  return gpsa::kmat_bwd_launch<float, double, double, float, double>(
      kind, Z, M, X, C, D, ls_u, var_u, Kbar, dZ, dX, dparams, 0, workspace, workspace_bytes, as_stream(stream), 1, kb);
}

}  // extern "C"
