// Shared pieces of the quadratic-form translation units (quadform.hip: generic kernels, launchers and the C ABI;
// qf_panel_*.hip, qf_elbo.hip, qf_sym.hip, qf_gram.hip, qf_big.hip: one kernel family each, explicitly instantiated
// there and declared ``extern template`` here - a launch from another unit calls the host stub its own unit defines).
// The declarations CARRY the launch bounds: an instantiation takes its attributes from the first declaration, and
// without them here every kernel was built for 1024 threads - 128 registers, the accumulators in scratch, 4.7x slower
// (tests/test_cabi.py::test_kernel_resources now reads the code objects' metadata).
#pragma once
#include <stdlib.h>

#include "common.hpp"

namespace gpsa {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
// The same with the LDS target given as  m0 + IMM  (IMM: the instruction's immediate offset, which the hardware adds
// to the LDS address AND to the global address - the caller passes  source - IMM bytes).  Writing m0 is what makes an
// LDS-DMA issue expensive next to a busy matrix pipe: measured (tools/microbench/panel_shape2.hip, the fused kernel's
// loop, fraction of the fp32-MFMA peak) 0.926 with m0 set once, 0.912 with ONE write per ring stage and the stage's
// pieces reached through IMM, 0.878 with a write per piece (round 3's glds16 above, which also saved and restored
// m0: two writes per piece); the m0 set-up alone, without any load, 0.907.  So a wave's pieces of a stage sit within
// 4 KiB of each other (immediate offsets reach +-4 KiB) and m0 is written once per stage.  Nothing the compiler emits
// for these kernels uses m0 (no dynamic register indexing: -Werror=pass-failed; no GWS / sendmsg), and the clobber
// tells it where it changes.
__device__ __forceinline__ void dma_set_m0(unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_base) : "memory");
}
template <int IMM>
__device__ __forceinline__ void glds16_m0(const float* gsrc_minus_imm) {
  static_assert(IMM >= 0 && IMM < 4096, "immediate offset of a global instruction");
  asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(gsrc_minus_imm), "n"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ void glds4_m0(const float* gsrc_minus_imm) {
  static_assert(IMM >= 0 && IMM < 4096, "immediate offset of a global instruction");
  asm volatile("global_load_lds_dword %0, off offset:%1" ::"v"(gsrc_minus_imm), "n"(IMM) : "memory");
}
#define GPSA_DMA_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// wait until at most N of this wave's vector-memory operations are outstanding (N = the LDS-DMA
// operations of the newest stage: everything older, i.e. the stage about to be read, has landed)
#define GPSA_DMA_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(lds_ptr_t)(p);
}

// The alpha slab of a column tile for the register-resident panel kernels: xb[ct][t][r] = X[row][c], row = 16 t +
// (QO ? 4 kq + r : 4 r + kq), c = cw + 16 ct + j, zero beyond M rows / C columns; 4 MB values per lane and column
// group, every one its own 4-byte load.  All loads are issued back to back - one basic block, one wait - and the
// padding is zeroed by selects at the end (round 5): with a bounds test around each load the compiler built a basic
// block per load and - the 64-bit row offsets living in scratch by then - put an s_waitcnt vmcnt(0) in front of a
// third of them: ~34 memory round trips in a row per tile change.  Every address is  (wave-uniform row base, clamped
// to M - 1) + (this lane's 32-bit offset, its row part clamped to what is left of the matrix below the base): no
// per-load 64-bit lane arithmetic (the allocator spilled it), no branch, every address inside the panel whatever MB
// (the next SUPPORTED tile count: whole row tiles can lie beyond M).  The launchers refuse C > GPSA_PANEL_MAX_C
// (the lane offset (12 C + c) * 4 bytes must fit 32 bits).
#define GPSA_PANEL_MAX_C (1LL << 26)
// FULLT: the caller guarantees M > 16 (MB - 1) - only the last row tile can be partial, the others need no clamp (the
// headline shape's instantiation: the clamps of 4 MB row bases are ~700 scalar instructions whose results the
// allocator parks in VGPR lanes and scratch).
template <int MB, int NCT, bool QO, bool FULLT = false>
__device__ __forceinline__ void load_alpha_slab(const float* __restrict__ X, int M, long long C, long long cw, int j,
                                                int kq, float (&xb)[NCT][MB][4], bool (&okc)[NCT]) {
  const int Mm1 = M - 1;
  const unsigned Cu = (unsigned)C;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const long long c = cw + ct * 16 + j;
    okc[ct] = c < C;
    const unsigned cl = (unsigned)(okc[ct] ? c : C - 1);
#pragma unroll
    for (int t = 0; t < MB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int want = t * 16 + (QO ? r : r * 4);      // uniform part of the row
        const int lane_row = QO ? kq * 4 : kq;
        if (FULLT && t < MB - 1) {
          xb[ct][t][r] = (X + (long long)want * C)[(unsigned)lane_row * Cu + cl];
        } else {
          const int bu = want < Mm1 ? want : Mm1;         // (scalar)
          const int lp = lane_row < Mm1 - bu ? lane_row : Mm1 - bu;
          xb[ct][t][r] = (X + (long long)bu * C)[(unsigned)lp * Cu + cl];
        }
      }
  }
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int t = 0; t < MB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        xb[ct][t][r] = (okc[ct] && t * 16 + (QO ? kq * 4 + r : r * 4 + kq) < M) ? xb[ct][t][r] : 0.f;
}
// the last row tile's rows in K-step order (4 r + kq), rows r < RL: xl[ct][r]; clamped lane addresses, zero padding
template <int MB, int NCT, int NR>
__device__ __forceinline__ void load_alpha_last(const float* __restrict__ X, int M, long long C, long long cw, int j,
                                                int kq, int nlive, float (&xl)[NCT][NR]) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const long long c = cw + ct * 16 + j;
    const bool ok = c < C;
    const float* __restrict__ Xc = X + (ok ? c : C - 1);
    float v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int row = (MB - 1) * 16 + r * 4 + kq;
      v[r] = Xc[(long long)(row < M ? row : M - 1) * C];
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) xl[ct][r] = (r < nlive && ok && (MB - 1) * 16 + r * 4 + kq < M) ? v[r] : 0.f;
  }
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
  int phase = 0;  // experiment knob (big_phase()): see big_phase_prologue
};
__global__ void pad_rows_kernel(const float* __restrict__ g, int L, long long C, long long Cpad,
                                float* __restrict__ gpad);
void gram_big_launch(dim3 grid, hipStream_t st, const GramBigArgs& a);  // qf_big.hip (GPSA_BIG_PAIR: the A/B)
// W[l] = P[l] X for large M (see prod_big_kernel)
struct ProdBigArgs {
  const float* P;  // [L][M][Mp], zero for k >= M (Mp = a multiple of 16)
  const float* X;  // [M][C]
  float* W;        // [L][M][C]
  int M, Mp, L;
  long long C;
};
__global__ void __launch_bounds__(256, 2) prod_big_kernel(ProdBigArgs a);
template <typename TO>
__global__ void __launch_bounds__(256) gram_big_reduce_kernel(const float* __restrict__ part, int M, int nsplit, TO* __restrict__ out);
// outputs per run of same-XCD workgroups in the large-M kernels (GPSA_BIG_LB; 0 = the plain 3-D / 2-D grids)
// The in-phase experiment on the two-workgroups-per-CU kernels (round-4 verdict item 7 (i)): the wave that sits in
// an ODD wave slot of its SIMD (HW_ID.wave_id: the second of the two co-resident workgroups) can be started late
// (GPSA_BIG_PHASE = n: n x 256 cycles of s_sleep, half a 64-MFMA chunk = 4) and / or raised to s_setprio 1 for its
// whole life (GPSA_BIG_PRIO = 1; MI355X_MICROARCH.md "Two waves per SIMD" item 4).  Default 0 / 0: measured, no gain
// (docs/LAB_NOTES.md, round 5).
static inline int big_phase() {
  static const int v = [] {
    const char* e = getenv("GPSA_BIG_PHASE");
    const char* q = getenv("GPSA_BIG_PRIO");
    int r = e ? atoi(e) & 0xff : 0;
    if (q && q[0] == '1') r |= 0x100;
    return r;
  }();
  return v;
}
#if defined(__HIPCC__)
__device__ __forceinline__ void big_phase_prologue(int phase) {
  if (phase == 0) return;
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);  // HW_REG_HW_ID bits [3:0]: the SIMD's wave slot
  if (slot & 1) {
    if (phase & 0x100) __builtin_amdgcn_s_setprio(1);
    for (int i = 0; i < (phase & 0xff); ++i) __builtin_amdgcn_s_sleep(4);
  }
}
#endif
static inline int big_remap_lb() {
  static const int v = [] { const char* e = getenv("GPSA_BIG_LB"); return e ? atoi(e) : 16; }();
  return v;
}
static inline bool gram_big_off() {
  static const bool v = [] { const char* e = getenv("GPSA_GRAM_BIG"); return e && e[0] == '0'; }();
  return v;
}

struct BigQuadArgs {
  const float* P;  // [L][M][Mp]  (TRI: U_l, else Omega_l), zero for k >= M
  const float* X;  // alpha [M][C]
  float* v;        // [L][C]
  float* W;        // STORE: [L][M][C]
  int M, Mp, L;
  long long C;
  int lb;  // > 0: 1-D grid; same-XCD workgroups come in runs of ``lb`` outputs of ONE column tile (they share its
           // alpha tile in that XCD's L2; each U_l / Omega_l is then shared by the few column tiles the XCD works on)
  int phase = 0;
};

struct BigAccumArgs {
  const float* P;  // [L][M][Mp] Omega_l, zero for k >= M
  const float* X;  // alpha [M][C]
  const float* g;  // [L][C]
  float* out;      // [nsplit][M][C]
  int M, Mp, L, nrb, nsplit;
  long long C, ctiles;
  float scale;
  int phase = 0;
};

__global__ void __launch_bounds__(256) big_accum_reduce_kernel(const float* __restrict__ part, int nsplit, long long n4, float* __restrict__ out);
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

template <typename TS>
__global__ void pad_k_kernel(const TS* __restrict__ src, int M, int Mp, long long n, float* __restrict__ dst);
template <typename TS>
__global__ void pad_k_tri_kernel(const TS* __restrict__ src, int M, int Mp, long long n, float* __restrict__ dst);
template <bool TRI, bool STORE, int NS>
__global__ void __launch_bounds__(256, 2) big_quad_kernel(BigQuadArgs a);
template <int NS>
__global__ void __launch_bounds__(256, 2) big_accum_kernel(BigAccumArgs a);
extern template __global__ void pad_k_kernel<float>(const float* __restrict__, int, int, long long, float* __restrict__);
extern template __global__ void pad_k_kernel<double>(const double* __restrict__, int, int, long long, float* __restrict__);
extern template __global__ void pad_k_tri_kernel<float>(const float* __restrict__, int, int, long long, float* __restrict__);
extern template __global__ void pad_k_tri_kernel<double>(const double* __restrict__, int, int, long long, float* __restrict__);
extern template __global__ void big_quad_kernel<true, false, 3>(BigQuadArgs);
extern template __global__ void big_quad_kernel<false, true, 3>(BigQuadArgs);
extern template __global__ void big_accum_kernel<3>(BigAccumArgs);
extern template __global__ void gram_big_reduce_kernel<float>(const float* __restrict__, int, int, float* __restrict__);

// ---- register-resident panel kernels (M <= 512): qf_panel_quad.hip / qf_panel_accum.hip / qf_panel_store.hip
template <int MB, int NCT, int MODE, int RL>
__global__ void __launch_bounds__(256, (MB * NCT >= 24) ? 1 : 2) panel_mfma_kernel(const float* __restrict__ Ppk, const float* __restrict__ X, const float* __restrict__ g, int M, long long C, int L, float* __restrict__ out, float* __restrict__ colsq, float out_scale, float* __restrict__ slab, float* __restrict__ keep);
#define GPSA_PANEL_SIG (const float* __restrict__ Ppk, const float* __restrict__ X, const float* __restrict__ g, int M, long long C, int L, float* __restrict__ out, float* __restrict__ colsq, float out_scale, float* __restrict__ slab, float* __restrict__ keep)
#define GPSA_PANEL_SHAPES(X, MODE) X(2, 4, MODE) X(4, 4, MODE) X(7, 4, MODE) X(13, 3, MODE) X(16, 2, MODE)
#define GPSA_PANEL_SHAPES_BIG(X, MODE) X(24, 1, MODE) X(32, 1, MODE)  // accumulate only; one unit per instantiation:
                                                                      // 24 / 32 row tiles unroll into 2304 / 4096 MFMAs
#define GPSA_PANEL_EXTERN(MB, NCT, MODE)                                              \
  extern template __global__ void panel_mfma_kernel<MB, NCT, MODE, 2> GPSA_PANEL_SIG; \
  extern template __global__ void panel_mfma_kernel<MB, NCT, MODE, 4> GPSA_PANEL_SIG;
#define GPSA_PANEL_DEFINE(MB, NCT, MODE)                                       \
  template __global__ void panel_mfma_kernel<MB, NCT, MODE, 2> GPSA_PANEL_SIG; \
  template __global__ void panel_mfma_kernel<MB, NCT, MODE, 4> GPSA_PANEL_SIG;
GPSA_PANEL_SHAPES(GPSA_PANEL_EXTERN, MODE_QUAD)
GPSA_PANEL_SHAPES(GPSA_PANEL_EXTERN, MODE_STORE)
GPSA_PANEL_SHAPES(GPSA_PANEL_EXTERN, MODE_ACCUM)
GPSA_PANEL_SHAPES_BIG(GPSA_PANEL_EXTERN, MODE_ACCUM)

// ---- forward + likelihood + abar in one pass: qf_elbo.hip
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
  float* FT;             // [L][C]  the draws themselves, F[s][n][l] at [l][s N + n] (optional: nullptr = not wanted)
  float* abar;           // [M][C]
  float* slab;           // 2 partial tiles per workgroup (accum_slab layout with this kernel's NCT)
  double* part;          // [nparts] sum of z^2 = ((Y - F) / s)^2 over the workgroup's items; entries >= gridDim.x: 0
  int nparts;
};

template <int MB, int NCT, int RL, bool FULLT = false, bool PAIRB = false>
__global__ void __launch_bounds__(256, (MB * NCT >= 14) ? 1 : 2) panel_elbo_kernel(ElboArgs a);
#define GPSA_ELBO_SHAPES(X) X(2, 4) X(4, 4) X(7, 4) X(13, 2) X(13, 1) X(16, 2)
#define GPSA_ELBO_EXTERN(MB, NCT)                                          \
  extern template __global__ void panel_elbo_kernel<MB, NCT, 2>(ElboArgs); \
  extern template __global__ void panel_elbo_kernel<MB, NCT, 4>(ElboArgs);
#define GPSA_ELBO_DEFINE(MB, NCT)                                   \
  template __global__ void panel_elbo_kernel<MB, NCT, 2>(ElboArgs); \
  template __global__ void panel_elbo_kernel<MB, NCT, 4>(ElboArgs);
GPSA_ELBO_SHAPES(GPSA_ELBO_EXTERN)
// M > 16 (MB - 1) (every row tile but the last inside the matrix), the 13-tile shape: the headline configuration's
extern template __global__ void panel_elbo_kernel<13, 2, 2, true>(ElboArgs);
extern template __global__ void panel_elbo_kernel<13, 2, 4, true>(ElboArgs);
// ... with one barrier per two K chunks (qf_elbo.hip: PAIRB)
extern template __global__ void panel_elbo_kernel<13, 2, 2, true, true>(ElboArgs);
extern template __global__ void panel_elbo_kernel<13, 2, 4, true, true>(ElboArgs);

// ---- symmetric quadratic form: qf_sym.hip
template <int MB, int NCT, int RL>
__global__ void __launch_bounds__(256, (MB * NCT >= 24) ? 1 : 2) quad_sym_mfma_kernel(const float* __restrict__ Ppk, const float* __restrict__ X, int M, long long C, int L, float* __restrict__ out);
#define GPSA_SYM_SIG (const float* __restrict__ Ppk, const float* __restrict__ X, int M, long long C, int L, float* __restrict__ out)
#define GPSA_SYM_SHAPES(X) X(2, 4) X(4, 4) X(7, 4) X(13, 3) X(16, 2) X(24, 1)
#define GPSA_SYM_EXTERN(MB, NCT)                                                 \
  extern template __global__ void quad_sym_mfma_kernel<MB, NCT, 2> GPSA_SYM_SIG; \
  extern template __global__ void quad_sym_mfma_kernel<MB, NCT, 4> GPSA_SYM_SIG;
#define GPSA_SYM_DEFINE(MB, NCT)                                          \
  template __global__ void quad_sym_mfma_kernel<MB, NCT, 2> GPSA_SYM_SIG; \
  template __global__ void quad_sym_mfma_kernel<MB, NCT, 4> GPSA_SYM_SIG;
GPSA_SYM_SHAPES(GPSA_SYM_EXTERN)

// ---- Gram sums (M <= 256): qf_gram.hip
constexpr int GR_KC = 64;  // columns per staged chunk (four 16-deep MFMA K blocks)
template <int MB, bool ALIGNED, int NL>
__global__ void __launch_bounds__(256, (MB >= 13 || NL > 1) ? 1 : 2) gram_mfma_kernel(const float* __restrict__ alpha, const float* __restrict__ g, long long gstride, int M, long long C, int L, int nsplit, float* __restrict__ part, const float* __restrict__ dmean);
#define GPSA_GRAM_SIG (const float* __restrict__ alpha, const float* __restrict__ g, long long gstride, int M, long long C, int L, int nsplit, float* __restrict__ part, const float* __restrict__ dmean)
#define GPSA_GRAM_SHAPES(X) X(2, 2) X(4, 2) X(7, 2) X(13, 2) X(2, 1) X(4, 1) X(7, 1) X(13, 1) X(16, 1)
#define GPSA_GRAM_EXTERN(MB, NL)                                                \
  extern template __global__ void gram_mfma_kernel<MB, true, NL> GPSA_GRAM_SIG; \
  extern template __global__ void gram_mfma_kernel<MB, false, NL> GPSA_GRAM_SIG;
#define GPSA_GRAM_DEFINE(MB, NL)                                         \
  template __global__ void gram_mfma_kernel<MB, true, NL> GPSA_GRAM_SIG; \
  template __global__ void gram_mfma_kernel<MB, false, NL> GPSA_GRAM_SIG;
GPSA_GRAM_SHAPES(GPSA_GRAM_EXTERN)
template <typename TO>
__global__ void gram_reduce_kernel(const float* __restrict__ part, int M, int MP, int L, int nsplit, TO* __restrict__ out,
                                   float* __restrict__ ddelta, float dbeta);
extern template __global__ void gram_reduce_kernel<float>(const float* __restrict__, int, int, int, int, float* __restrict__, float* __restrict__, float);
extern template __global__ void gram_reduce_kernel<double>(const float* __restrict__, int, int, int, int, double* __restrict__, float* __restrict__, float);

}  // namespace gpsa
