// The 128 x 128 LDS-DMA kernels of the large-M data GP (M > 256: BASELINE configs 4 / 5) - Gram sums, full product,
// block-triangular form, accumulate - and the Omega = A A^T pair of the M x M stage (split out of quadform.hip in
// round 4: one translation unit per kernel family).
#include "qf_common.hpp"

namespace gpsa {

// KiB of piece P_ (0 .. 15) inside a ring slot of the 128 x 128 kernels: wave-major, see gram_big_kernel's stage
#define GPSA_BIG_POS(P_) ((((P_) & 3) << 2) + ((P_) >> 2))

// PAIR (round 6, the lever the round-5 counters pointed at): ONE wait + barrier per TWO 16-column chunks.  Four
// one-chunk slots (64 KB a workgroup: two workgroups still share a CU); iteration k multiplies the chunks 2k, 2k + 1 out
// of the slots (2k, 2k + 1) mod 4 while the stages of the chunks 2k + 2, 2k + 3 - requested at the top of the iteration
// into the slots the barrier that ended iteration k - 1 released - are in flight; the iteration ends on vmcnt(0) + barrier.
// GPSA_BIG_PAIR=1 selects it (A/B: profiles/r06_big_pair_ab.txt).
template <bool PAIR>
__global__ void __launch_bounds__(256, 2) gram_big_kernel_t(GramBigArgs a) {
  constexpr int NSLOT = PAIR ? 4 : 3;
  __shared__ __attribute__((aligned(16))) float lds[NSLOT][16 * 256];
  __shared__ __attribute__((aligned(16))) float sg[NSLOT][16];
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
  big_phase_prologue(a.phase);

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
    /* wave-major ring slot (round 4): the wave's four pieces w, 4+w, 8+w, 12+w are the four consecutive KiB at   \
       4 w - ONE m0 write per stage, the pieces through the immediate offset (qf_common.hpp: glds16_m0; an m0 write \
       next to a busy matrix pipe costs ~40 cycles, and glds16 did two per piece) */                              \
    dma_set_m0(glds0 + (unsigned)(BUF) * (16 * 256 * 4) + (unsigned)w * 4096);                \
    glds16_m0<0>(rowp[0] + col__);                                                            \
    glds16_m0<1024>(rowp[1] + col__ - 256);                                                   \
    glds16_m0<2048>(rowp[2] + col__ - 512);                                                   \
    glds16_m0<3072>(rowp[3] + col__ - 768);                                                   \
    if (lane < 4) {                                                                           \
      dma_set_m0(gsg0 + (unsigned)(BUF) * 64);                                                \
      glds16_m0<0>(gl + (long long)(CH) * 16 + lane * 4);                                     \
    }                                                                                         \
  }
#define GPSA_GB_MMA(F)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                             \
      acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].F, bv[k].F, acc[i][k], 0, 0, 0);
#define GPSA_GB_CHUNK(BUF)                                                                    \
  {                                                                                           \
    const float* base = &lds[BUF][lane * 4];                                                  \
    const float4 gk = *reinterpret_cast<const float4*>(&sg[BUF][kq * 4]);                     \
    float4 av[4], bv[4];                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
      const float4 x = *reinterpret_cast<const float4*>(base + GPSA_BIG_POS(wr * 4 + i) * 256); \
      av[i] = make_float4(x.x * gk.x, x.y * gk.y, x.z * gk.z, x.w * gk.w);                    \
      bv[i] = *reinterpret_cast<const float4*>(base + GPSA_BIG_POS(8 + wc * 4 + i) * 256);    \
    }                                                                                         \
    GPSA_GB_MMA(x)                                                                            \
    GPSA_GB_MMA(y)                                                                            \
    GPSA_GB_MMA(z)                                                                            \
    GPSA_GB_MMA(w)                                                                            \
  }
  if (PAIR) {
    if (ch0 < ch1) {
      GPSA_GB_STAGE(ch0, 0)
      GPSA_GB_STAGE(ch0 + 1 < ch1 ? ch0 + 1 : ch0, 1)
    }
    GPSA_DMA_WAIT(0);
    __syncthreads();
    int buf = 0;  // 0 or 2: the pair's first slot
    for (long long ch = ch0; ch < ch1; ch += 2) {
      const int nb = buf ^ 2;
      GPSA_GB_STAGE(ch + 2 < ch1 ? ch + 2 : ch1 - 1, nb)
      GPSA_GB_STAGE(ch + 3 < ch1 ? ch + 3 : ch1 - 1, nb + 1)
      GPSA_GB_CHUNK(buf)
      if (ch + 1 < ch1) GPSA_GB_CHUNK(buf + 1)
      GPSA_DMA_WAIT(0);
      __syncthreads();
      buf = nb;
    }
  } else {
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
      GPSA_GB_CHUNK(buf)
      GPSA_DMA_WAIT(5);
      __syncthreads();
      buf = (buf == 2) ? 0 : buf + 1;
    }
  }
#undef GPSA_GB_CHUNK
#undef GPSA_GB_MMA
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
template __global__ void gram_big_kernel_t<false>(GramBigArgs);
template __global__ void gram_big_kernel_t<true>(GramBigArgs);
void gram_big_launch(dim3 grid, hipStream_t st, const GramBigArgs& a) {
  static const bool pair = [] { const char* e = getenv("GPSA_BIG_PAIR"); return e && e[0] == '1'; }();
  if (pair) gram_big_kernel_t<true><<<grid, 256, 0, st>>>(a);
  else gram_big_kernel_t<false><<<grid, 256, 0, st>>>(a);
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
      av[i] = *reinterpret_cast<const float4*>(base + (wr * 4 + i) * 256);  // (piece-major slot: this kernel's own stage)
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
  big_phase_prologue(a.phase);
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
    const bool oob__ = last_oob && s_ch == nch - 1;                                           \
    dma_set_m0(lds0 + (unsigned)(BUF) * (16 * 256 * 4) + (unsigned)w * 4096); /* wave-major slot: gram_big_kernel */ \
    glds16_m0<0>(sp0);                                                                        \
    glds16_m0<1024>(sp1 - 256);                                                               \
    glds16_m0<2048>((oob__ ? xclamp0 : sx0) - 512);                                           \
    glds16_m0<3072>((oob__ ? xclamp1 : sx1) - 768);                                           \
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
        av[i] = *reinterpret_cast<const float4*>(base + GPSA_BIG_POS(wr * 4 + i) * 256);
        bv[i] = *reinterpret_cast<const float4*>(base + GPSA_BIG_POS(8 + wc * 4 + i) * 256);
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
  big_phase_prologue(a.phase);
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
    const bool oob__ = last_oob && s_ch == nch - 1;                                           \
    dma_set_m0(lds0 + (unsigned)(BUF) * (16 * 256 * 4) + (unsigned)w * 4096); /* wave-major slot: gram_big_kernel */ \
    glds16_m0<0>(sp0);                                                                        \
    glds16_m0<1024>(sp1 - 256);                                                               \
    glds16_m0<2048>((oob__ ? xclamp0 : sx0) - 512);                                           \
    glds16_m0<3072>((oob__ ? xclamp1 : sx1) - 768);                                           \
    if (lane < 32) {                                                                          \
      dma_set_m0(sg0 + (unsigned)(BUF) * 512);                                                \
      glds16_m0<0>(sgp);                                                                      \
    }                                                                                         \
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
        av[i] = *reinterpret_cast<const float4*>(base + GPSA_BIG_POS(wr * 4 + i) * 256);
        const float4 x = *reinterpret_cast<const float4*>(base + GPSA_BIG_POS(8 + wc * 4 + i) * 256);
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


template __global__ void pad_k_kernel<float>(const float* __restrict__, int, int, long long, float* __restrict__);
template __global__ void pad_k_kernel<double>(const double* __restrict__, int, int, long long, float* __restrict__);
template __global__ void pad_k_tri_kernel<float>(const float* __restrict__, int, int, long long, float* __restrict__);
template __global__ void pad_k_tri_kernel<double>(const double* __restrict__, int, int, long long, float* __restrict__);
template __global__ void big_quad_kernel<true, false, 3>(BigQuadArgs);
template __global__ void big_quad_kernel<false, true, 3>(BigQuadArgs);
template __global__ void big_accum_kernel<3>(BigAccumArgs);
template __global__ void gram_big_reduce_kernel<float>(const float* __restrict__, int, int, float* __restrict__);

}  // namespace gpsa
