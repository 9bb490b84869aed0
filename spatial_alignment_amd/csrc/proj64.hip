// The fp64 projection alpha = K_uu^-1 K_uf (whiten.hip) as a PERSISTENT, output-stationary kernel (round 5).
//
// whiten_mfma_kernel walks K: a workgroup owns 64 columns, keeps all MB row tiles of the result in registers and
// leaves with 52-104 stores per lane at the very end; its grid comes in rounds (1563 workgroups on 512 slots at the
// headline size: the fourth round is 27 workgroups) and nothing covers a workgroup's own prologue (the column loads)
// and epilogue but the one other workgroup of its CU.  Here the ROW TILE is the outer loop instead:
//   * a wave keeps its 16 columns of the right-hand side in registers for a whole column tile (52 doubles per lane at
//     MB = 13 - the B operand of every MFMA of the tile, and the factor that closes q = k^T alpha);
//   * a stage is one row tile rt: the 4 MB fragments Kinv[16 rt .. +15][all k] (26 KiB at MB = 13) arrive by LDS-DMA
//     from the SAME packed inverse whiten.hip uses (fragment (kc, ks, rt) is 512 contiguous bytes: a 1-KiB piece takes
//     two of them, half a wave each), 4 MB MFMAs per wave run into two alternating accumulators, and the finished 16 x 16
//     block leaves at once: 4 (8 with the fp32 copy) stores per lane and stage, spread over the whole kernel and
//     draining under the next stage's MFMAs;
//   * the unit of work is (column tile, row tile): the grid is 3 workgroups per CU (<= 168 registers, 52 KiB of LDS
//     each), and workgroup i takes units [i U / G, (i + 1) U / G) - within one unit of each other whatever C is.  A
//     column tile cut between two workgroups costs a second read of its columns; its two partial q meet by
//     atomicAdd (two addends onto a zeroed word: order-independent).
// One barrier per stage; the only vmcnt wait is a vmcnt(0) between a stage's last MFMA and its stores, when everything
// older (the next stage's pieces, the previous stage's stores) has been in flight for a whole stage.
#include "internal.hpp"
#include "kmat_cov.hpp"

#include <stdlib.h>

namespace gpsa {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* proj_lds_ptr_t;

struct ProjArgs {
  const double* Apk;  // [batch][kc][ks][rt][64]: pack_whiten_kernel's layout
  const void* X;      // [batch] panels [M][C], stride sX elements
  double* alpha;      // [batch] results [M][C]
  float* out32;       // optional fp32-rounded copy of the result (same strides)
  double* q;          // optional [batch][C], zeroed by the launcher
  long long C, sX;
  long long U;        // units = batch * T * MB
  int M, T;           // T = column tiles of 64 per problem
  int dbg;            // timing experiments (GPSA_PROJ64_SKIP; results are then wrong): 1 no stores, 2 one slab load per
                      // workgroup, 4 no LDS-DMA after the first stage, 8 no barrier / vmcnt wait
  // GEN (round 6): the right-hand side is not read but FORMED - K_uf[m, c] = k(Z_m, x_c) as the wave loads its 16
  // columns: the covariance launch in front (kmat_fwd: C M doubles written, then read back here) disappears
  const float* gZ;    // [M][gD] inducing points (fp32 parameters)
  const double* gX;   // [C][gD] points (the warp GPs' unrounded draws)
  const float *g_ls, *g_var;  // log lengthscale, log variance
  int gD;
};

template <int IMM>
__device__ __forceinline__ void proj_glds16(const char* gsrc_minus_imm) {
  static_assert(IMM >= 0 && IMM < 4096, "immediate offset of a global instruction");
  asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(gsrc_minus_imm), "n"(IMM) : "memory");
}
__device__ __forceinline__ void proj_set_m0(unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_base) : "memory");
}

// LDS position (KiB) of piece p of a stage: wave-major (wave p & 3 issues it as its (p >> 2)-th piece; a wave's pieces
// are consecutive KiB so that one m0 value reaches four of them through the immediate offset)
template <int NPIECE>
__host__ __device__ constexpr int proj_wave_start(int w) {
  int s = 0;
  for (int v = 0; v < w; ++v) s += (NPIECE - v + 3) / 4;
  return s;
}
template <int NPIECE>
__host__ __device__ constexpr int proj_pos(int p) {
  return proj_wave_start<NPIECE>(p & 3) + (p >> 2);
}

template <int MB, typename TI, int OCC, int GEN = 0>  // GEN: 0 = read X; 1 + GPSA_K_* = form K_uf from (gZ, gX)
__global__ void __launch_bounds__(256, OCC) proj64_kernel(ProjArgs a) {
  constexpr int NF = 4 * MB;             // 512-byte fragments of a stage: f = 4 kc + ks
  constexpr int NPIECE = NF / 2;         // 1-KiB pieces
  constexpr int NPW = (NPIECE + 3) / 4;  // most pieces a wave issues
  constexpr int SLOT = NF * 64;          // doubles per ring slot
  constexpr long long PB = (long long)MB * 4 * MB * 64 * 8;  // bytes of one packed inverse
  __shared__ __attribute__((aligned(16))) double lds[2][SLOT];
  __shared__ double Zs[GEN ? MB * 16 : 1][MAXD];  // GEN: the inducing points (rows >= M repeat row M - 1: any finite value)

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const long long u0 = (long long)blockIdx.x * a.U / gridDim.x, u1 = (long long)(blockIdx.x + 1) * a.U / gridDim.x;
  if (u0 >= u1) return;
  const long long C = a.C;
  const int M = a.M;
  // M <= 16 (MB - 1) + 8: the last two K steps of a row tile meet only the packed inverse's zero padding (M = 200: k =
  // 200 .. 207) - their MFMAs and the piece that carries their fragments are skipped (2 of 52: 4 % of the kernel)
  const bool short_k = M <= 16 * (MB - 1) + 8;
  const int npiece_live = NPIECE - (short_k ? 1 : 0);

  // where this workgroup starts: (problem b, column tile t, row tile rt)
  int rt = (int)(u0 % MB);
  int b, t;
  {
    const long long tile0 = u0 / MB;
    b = (int)(tile0 / a.T);
    t = (int)(tile0 % a.T);
  }
  const unsigned lds0 = (unsigned)(unsigned long long)(proj_lds_ptr_t)(&lds[0][0]);
  int wstart = 0;  // first KiB of this wave's pieces inside a slot
  for (int v = 0; v < w; ++v) wstart += (NPIECE - v + 3) / 4;
  // lane part of a piece's source: lanes 0..31 copy fragment 2 p, lanes 32..63 fragment 2 p + 1 (16 bytes per lane);
  // fragment f of row tile r sits at ((f * MB) + r) * 512 bytes of the packed inverse
  const char* const abase =
      reinterpret_cast<const char*>(a.Apk) + ((long long)(2 * w + (lane >> 5)) * MB) * 512 + (lane & 31) * 16;
#define GPSA_PJ_STAGE(RT_, B_, SLOT_)                                                                        \
  {                                                                                                          \
    const char* sp__ = abase + (long long)(B_) * PB + (RT_) * 512;                                           \
    const unsigned l__ = lds0 + (unsigned)(SLOT_) * (SLOT * 8) + (unsigned)wstart * 1024;                     \
    _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                        \
      if (w + 4 * i < npiece_live) { /* wave-uniform */                                                      \
        if ((i & 3) == 0) proj_set_m0(l__ + i * 1024);                                                       \
        const char* g__ = sp__ + (long long)i * (8LL * MB * 512);                                            \
        if ((i & 3) == 0) proj_glds16<0>(g__);                                                               \
        else if ((i & 3) == 1) proj_glds16<1024>(g__ - 1024);                                                \
        else if ((i & 3) == 2) proj_glds16<2048>(g__ - 2048);                                                \
        else proj_glds16<3072>(g__ - 3072);                                                                  \
      }                                                                                                      \
    }                                                                                                        \
  }
  GPSA_PJ_STAGE(rt, b, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  double g_ell = 1.0, g_inv_ell = 1.0, g_var = 1.0;
  if (GEN) {
    for (int i = tid; i < MB * 16 * MAXD; i += 256) {
      const int r = i / MAXD, d = i % MAXD;
      Zs[r][d] = d < a.gD ? (double)a.gZ[(long long)(r < M ? r : M - 1) * a.gD + d] : 0.0;
    }
    g_ell = exp((double)a.g_ls[0]);
    g_inv_ell = 1.0 / g_ell;
    g_var = exp((double)a.g_var[0]);
    // (the loop's first barrier publishes Zs before any wave forms a slab)
  }

  double slab[MB][4];  // the wave's 16 columns of the right-hand side: slab[kc][ks] = X[16 kc + 4 ks + kq][c]
  double s = 0.0;      // this segment's share of q[c]
  long long c = 0;
  bool okc = false;
  int seg_rt0 = 0;
  const TI* Xb = nullptr;
  double* alb = nullptr;
  float* o32b = nullptr;
  int slot = 0;
  for (long long u = u0; u < u1; ++u) {
    if (!(a.dbg & 8)) __syncthreads();  // stage u has landed (everyone waited for its own pieces); stage u - 1 has been read by all
    {
      int nrt = rt + 1, nb = b;
      if (nrt == MB) {
        nrt = 0;
        if (t + 1 == a.T) nb = b + 1;
      }
      if (u + 1 < u1 && !(a.dbg & 4)) GPSA_PJ_STAGE(nrt, nb, slot ^ 1)
    }
    if (u == u0 || (rt == 0 && !(a.dbg & 2))) {  // a new column tile: its columns of the right-hand side
      c = (long long)t * 64 + w * 16 + j;
      okc = c < C;
      Xb = reinterpret_cast<const TI*>(a.X) + (long long)b * a.sX;
      alb = a.alpha + (long long)b * a.sX;
      o32b = a.out32 != nullptr ? a.out32 + (long long)b * a.sX : nullptr;
      // Unconditional loads from a row clamped into the panel, no select: rows >= M meet the packed inverse's zero
      // columns (any finite value does), a column >= C computes on column 0 and is never stored.  (A select on the
      // loaded value lets the compiler sink every load into its own branch with a vmcnt(0) behind it; offsets that are
      // loop-invariant get hoisted into 104 registers and spilled - hence the opaque copies of M and C.)
      if (GEN) {
        double x[MAXD];
#pragma unroll
        for (int d = 0; d < MAXD; ++d) x[d] = d < a.gD ? a.gX[(okc ? c : 0) * a.gD + d] : 0.0;
#pragma unroll
        for (int kc = 0; kc < MB; ++kc)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            double k, cd, pl;
            cov_eval<double, (GEN ? GEN - 1 : 0)>(Zs[16 * kc + 4 * ks + kq], x, a.gD, g_ell, g_inv_ell, g_var, k, cd, pl);
            slab[kc][ks] = k;
          }
      } else {
      const TI* xp = Xb + (okc ? c : 0);
      int Mv = M;
      long long Cv = C;
      asm volatile("" : "+s"(Mv), "+s"(Cv));
#pragma unroll
      for (int kc = 0; kc < MB; ++kc)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int row = 16 * kc + 4 * ks + kq;
          slab[kc][ks] = (double)xp[(long long)(row < Mv ? row : Mv - 1) * Cv];
        }
      }
      s = 0.0;
      seg_rt0 = rt;
    }
    f64x4 acc0 = (f64x4){0.0, 0.0, 0.0, 0.0}, acc1 = (f64x4){0.0, 0.0, 0.0, 0.0};
    const double* base = &lds[0][0] + slot * SLOT + lane;
    // fragments in groups of four (two ds_read2st64_b64), the NEXT group requested before the current group's MFMAs:
    // left to itself the compiler reads a pair, waits for it, issues its two MFMAs and reads the next pair into the
    // same registers - one LDS round trip per 128 cycles of matrix pipe
    constexpr int NG = NF / 4;
    static_assert(NF % 4 == 0, "fragments per stage");
    double an[4], ac[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ac[e] = base[proj_pos<NPIECE>(e >> 1) * 128 + (e & 1) * 64];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = 4 * (g + 1) + e;
          an[e] = base[proj_pos<NPIECE>(f >> 1) * 128 + (f & 1) * 64];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = 4 * g + e;
        if (g == NG - 1 && e >= 2 && short_k) continue;  // (uniform)
        if (e & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[e], slab[f >> 2][f & 3], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[e], slab[f >> 2][f & 3], acc0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) ac[e] = an[e];
    }
    double y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = acc0[r] + acc1[r];
    if (a.q != nullptr) {  // rows 16 rt + 4 r + kq of the result meet the same rows of the right-hand side
      switch (rt) {
#define GPSA_PJ_Q(K_)                                                                              \
  case K_:                                                                                         \
    if (K_ < MB)                                                                                   \
      s += (y[0] * slab[K_ < MB ? K_ : 0][0] + y[1] * slab[K_ < MB ? K_ : 0][1]) +                 \
           (y[2] * slab[K_ < MB ? K_ : 0][2] + y[3] * slab[K_ < MB ? K_ : 0][3]);                  \
    break;
        GPSA_PJ_Q(0) GPSA_PJ_Q(1) GPSA_PJ_Q(2) GPSA_PJ_Q(3) GPSA_PJ_Q(4) GPSA_PJ_Q(5) GPSA_PJ_Q(6) GPSA_PJ_Q(7)
        GPSA_PJ_Q(8) GPSA_PJ_Q(9) GPSA_PJ_Q(10) GPSA_PJ_Q(11) GPSA_PJ_Q(12) GPSA_PJ_Q(13) GPSA_PJ_Q(14) GPSA_PJ_Q(15)
#undef GPSA_PJ_Q
        default: break;
      }
    }
    // everything older than this point has been in flight for a whole stage: the next stage's pieces have landed
    // (the barrier at the top publishes them), the previous stage's stores are out
    __builtin_amdgcn_sched_barrier(0);
    if (!(a.dbg & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(a.dbg & 1)) {
      const int row0 = 16 * rt + kq;
      const long long o = (long long)row0 * C + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (okc && row0 + 4 * r < M) {
          alb[o + (long long)(4 * r) * C] = y[r];
          if (o32b != nullptr) o32b[o + (long long)(4 * r) * C] = (float)y[r];  // uniform branch
        }
      }
    }
    const bool last_rt = rt == MB - 1;
    if (a.q != nullptr && (last_rt || u + 1 == u1)) {  // the segment ends: close its share of q
      double z = s;
      z += __shfl_xor(z, 16, 64);
      z += __shfl_xor(z, 32, 64);
      if (kq == 0 && okc) {
        double* qp = a.q + (long long)b * C + c;
        if (seg_rt0 == 0 && last_rt) *qp = z;  // the whole column tile was this workgroup's
        else (void)__hip_atomic_fetch_add(qp, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (last_rt) {
      rt = 0;
      if (++t == a.T) {
        t = 0;
        ++b;
      }
    } else {
      ++rt;
    }
    slot ^= 1;
  }
#undef GPSA_PJ_STAGE
}

static inline long long proj64_min_units() {
  static const long long v = [] {
    const char* e = getenv("GPSA_PROJ64_MIN_TILES");
    return e ? atoll(e) : -1LL;
  }();
  return v;
}

// shapes the persistent kernel takes (everything else stays on whiten_mfma_kernel)
bool proj64_ok(int MB, long long C, int batch) {
  static const bool off = [] {
    const char* e = getenv("GPSA_PROJ64");
    return e && e[0] == '0';
  }();
  if (off || (MB != 13 && MB != 7)) return false;
  // long panels only: at least two column tiles per workgroup of the full grid.  Below that a workgroup is one tile's
  // 13 dependent stages behind one memory latency and whiten_mfma_kernel's wider spread wins (C = 20 000: 64 vs 72 us,
  // C = 12 500: 41 vs 50; at C = 100 000 / 200 000: 231 -> 218 / 423 -> 410 isolated, 202 -> 192 and 179 -> 164 in the step)
  const long long tiles = (long long)batch * cdiv(C, 64);
  const long long floor_t = proj64_min_units() >= 0 ? proj64_min_units() : 2LL * 3 * num_cus();
  return tiles >= floor_t && C * 8 < (1LL << 40);
}

template <typename TI>
int proj64_launch(int MB, const double* Apk, const TI* X, int M, long long C, double* alpha, float* out32, double* q,
                  int batch, long long sX, hipStream_t st, bool q_zeroed) {
  ProjArgs a;
  a.Apk = Apk;
  a.X = X;
  a.alpha = alpha;
  a.out32 = out32;
  a.q = q;
  a.C = C;
  a.sX = sX;
  a.M = M;
  a.T = (int)cdiv(C, 64);
  a.U = (long long)batch * a.T * MB;
  static const int dbg = [] { const char* e = getenv("GPSA_PROJ64_SKIP"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  a.gZ = nullptr; a.gX = nullptr; a.g_ls = a.g_var = nullptr; a.gD = 0;
  // at least MB units per workgroup: a column tile (MB consecutive units) then meets at most two workgroups
  // workgroups per CU: three for the fp32 right-hand side (168 registers), two for the fp64 one (its three-per-CU
  // instantiation spills five registers; measured equal: LAB_NOTES); GPSA_PROJ64_OCC = 1 / 2 / 3 forces one
  static const int occ_env = [] {
    const char* e = getenv("GPSA_PROJ64_OCC");
    return e && (e[0] == '1' || e[0] == '2' || e[0] == '3') ? e[0] - '0' : 0;
  }();
  const int occ = occ_env ? occ_env : (sizeof(TI) == 8 ? 2 : 3);
  long long grid = (long long)occ * num_cus();
  if (grid > (long long)batch * a.T) grid = (long long)batch * a.T;
  if (q != nullptr && !q_zeroed) {
    const int e = zero_fill_async(q, (size_t)((long long)batch * C * 8), st);
    if (e != 0) return e;
  }
  switch (MB) {
    case 13:
      if (occ == 1) proj64_kernel<13, TI, 1><<<(unsigned)grid, 256, 0, st>>>(a);
      else if (occ == 2) proj64_kernel<13, TI, 2><<<(unsigned)grid, 256, 0, st>>>(a);
      else proj64_kernel<13, TI, 3><<<(unsigned)grid, 256, 0, st>>>(a);
      break;
    case 7:
      if (occ == 2) proj64_kernel<7, TI, 2><<<(unsigned)grid, 256, 0, st>>>(a);
      else proj64_kernel<7, TI, 3><<<(unsigned)grid, 256, 0, st>>>(a);
      break;
    default: return GPSA_EUNSUPPORTED;
  }
  GPSA_LAUNCH_CHECK();
  return 0;
}

// alpha = Kinv K_uf with K_uf[m, c] = k(Z_m, x_c) formed inside the kernel (one problem; the data GP's forward)
int proj64_gen_launch(int MB, const double* Apk, int kind, const float* Z, const double* X64, int D, const float* ls_u,
                      const float* var_u, int M, long long C, double* alpha, float* out32, double* q, hipStream_t st,
                      bool q_zeroed) {
  if (D < 1 || D > MAXD) return GPSA_EUNSUPPORTED;
  ProjArgs a;
  a.Apk = Apk;
  a.X = nullptr;
  a.alpha = alpha;
  a.out32 = out32;
  a.q = q;
  a.C = C;
  a.sX = 0;
  a.M = M;
  a.T = (int)cdiv(C, 64);
  a.U = (long long)a.T * MB;
  a.dbg = 0;
  a.gZ = Z;
  a.gX = X64;
  a.g_ls = ls_u;
  a.g_var = var_u;
  a.gD = D;
  long long grid = 2LL * num_cus();  // (three workgroups per CU do not fit: 60 KB of LDS each with the inducing points)
  if (grid > a.T) grid = a.T;
  if (q != nullptr && !q_zeroed) {
    const int e = zero_fill_async(q, (size_t)(C * 8), st);
    if (e != 0) return e;
  }
#define GPSA_PJG(MB_, K_)                                                             \
  case K_:                                                                            \
    proj64_kernel<MB_, double, 2, 1 + K_><<<(unsigned)grid, 256, 0, st>>>(a);         \
    break;
  if (MB == 13) {
    switch (kind) {
      GPSA_PJG(13, GPSA_K_RBF) GPSA_PJG(13, GPSA_K_MATERN12) GPSA_PJG(13, GPSA_K_MATERN32)
      default: return GPSA_EUNSUPPORTED;
    }
  } else if (MB == 7) {
    switch (kind) {
      GPSA_PJG(7, GPSA_K_RBF) GPSA_PJG(7, GPSA_K_MATERN12) GPSA_PJG(7, GPSA_K_MATERN32)
      default: return GPSA_EUNSUPPORTED;
    }
  } else {
    return GPSA_EUNSUPPORTED;
  }
#undef GPSA_PJG
  GPSA_LAUNCH_CHECK();
  return 0;
}

template int proj64_launch<double>(int, const double*, const double*, int, long long, double*, float*, double*, int,
                                   long long, hipStream_t, bool);
template int proj64_launch<float>(int, const double*, const float*, int, long long, double*, float*, double*, int,
                                  long long, hipStream_t, bool);

}  // namespace gpsa
