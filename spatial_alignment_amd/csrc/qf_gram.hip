// gram_mfma_kernel: dOmega_l = sum_c g[l,c] alpha_c alpha_c^T on the matrix cores (M <= 256).
#include "qf_common.hpp"

namespace gpsa {

// 16 bytes of zeros: where the LDS-DMA of g is sent for the column groups past C (see gram_wave_run)
__device__ __attribute__((aligned(16))) const float gram_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// ------------------------------------------------------------------------------------------------
// MFMA Gram kernel:  dOmega_l = sum_c g[l,c] alpha_c alpha_c^T   (lower-triangle 16x16 tiles)
// grid (L, nsplit): workgroup (l, s) sweeps its share of the columns in 32-column chunks staged in
// LDS; the NT = MB(MB+1)/2 lower tiles are dealt to the 4 waves by whole tile rows (GramPlan).
// Both MFMA operands of a tile are rows of the same LDS image (A: rows of tile-row, scaled by g in
// registers once per row and K block; B: rows of tile-column); the K index (columns c) is permuted as
// in the panel kernel so that one ds_read_b128 feeds four MFMAs.  Partials [L][nsplit][MP][MP] are summed + mirrored by a second
// kernel (deterministic).
// ------------------------------------------------------------------------------------------------

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
// ``stage(kb, gp)``: the caller's LDS-DMA issues for the NEXT chunk, one per group of the first groups of every K
// block, pinned between the groups' MFMAs (in one clump in front of the chunk they cost 8 % of the kernel: 14 issues
// with the matrix pipe idle, 156 times per workgroup; without any staging the kernel ran 1750 instead of 1900 us)
// ``drow`` (>= 0: the d-delta option): row ``drow`` of the LAST tile row - the first padding row, M - 16 (MB - 1) - takes
// dmean[l, c] as its (already "scaled") row fragment instead of g alpha: the tiles of that tile row then carry
// sum_c dmean[l,c] alpha[col,c] = d delta_F[col, l] in their row M (MFMAs the padding runs anyway).  gvec holds the
// NL rows of dmean behind the NL rows of g.
template <int MB, int NKB, int W, int NS, int NL, typename STG>
__device__ __forceinline__ void gram_wave_chunk(const float* __restrict__ img,
                                                const float* __restrict__ gvec, int kq,
                                                f32x4 (&acc)[NL][NS + GR_G], STG&& stage, int drow, int lane15) {
  constexpr GramPlan<MB> P{};
  constexpr int N = P.cnt[W];
  constexpr int NGRP = (((NS + GR_G - 1) / GR_G) + 1) & ~1;
  // LDS image of a chunk, wave-major (round 4): K block kb of all row tiles is staged by wave kb and sits in NPW
  // consecutive KiB at kb * NPW, so that the staging wave reaches its pieces through immediate offsets from a few m0
  // values (qf_common.hpp: glds16_m0) - piece (tile_row, kb) is KiB kb * NPW + tile_row
  constexpr int NPW_ = (MB * NKB + 3) / 4;
  static_assert(NKB == 4, "one K block per staging wave");
  auto frag = [&](int kb, int tile_row) {
    return *reinterpret_cast<const float4*>(img + kb * (NPW_ * 256) + tile_row * 256);
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
  // does this wave own the last tile row?  (compile time)
  constexpr bool OWNS_LAST = [] {
    constexpr GramPlan<MB> Q{};
    for (int s = 0; s < Q.cnt[W]; ++s)
      if (Q.rr[W][s] == MB - 1) return true;
    return false;
  }();
  const bool dsel = OWNS_LAST && drow >= 0 && lane15 == drow;
  // dmean of the K block being multiplied / of the next one (fetched with gn: no LDS round trip at the point of use)
  float4 dk[OWNS_LAST ? NL : 1], dn[OWNS_LAST ? NL : 1];
  if (OWNS_LAST) {
#pragma unroll
    for (int q = 0; q < NL; ++q)
      dn[q] = dk[q] = drow >= 0 ? *reinterpret_cast<const float4*>(gvec + (NL + q) * GR_KC + kq * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
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
    if (OWNS_LAST) {
#pragma unroll
      for (int q = 0; q < NL; ++q) dk[q] = dn[q];
    }
#pragma unroll
    for (int gp = 0; gp < NGRP; ++gp) {
      const int cur = gp & 1, nxt = cur ^ 1;
      if (gp + 1 < NGRP) {
        GPSA_GR_FETCH(nxt, kb, gp + 1)
      } else if (kb + 1 < NKB) {  // first group of the next K block (and its g rows)
#pragma unroll
        for (int q = 0; q < NL; ++q)
          gn[q] = *reinterpret_cast<const float4*>(gvec + q * GR_KC + (kb + 1) * 16 + kq * 4);
        if (OWNS_LAST && drow >= 0) {
#pragma unroll
          for (int q = 0; q < NL; ++q)
            dn[q] = *reinterpret_cast<const float4*>(gvec + (NL + q) * GR_KC + (kb + 1) * 16 + kq * 4);
        }
        GPSA_GR_FETCH(nxt, kb + 1, 0)
      }
      __builtin_amdgcn_sched_barrier(0);
      if (gp < 5) stage(kb, gp);
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
            if (OWNS_LAST && P.rr[W][s_ < N ? s_ : N - 1] == MB - 1 && drow >= 0) {  // (compile time && uniform)
              const float4 d_ = dk[q];
              // (component by component: a select between two float4 objects goes through scratch)
              arow[q] = make_float4(dsel ? d_.x : arow[q].x, dsel ? d_.y : arow[q].y, dsel ? d_.z : arow[q].z,
                                    dsel ? d_.w : arow[q].w);
            }
          }
          a[q][u] = arow[q];
        }
      }
      // (surplus slots of the wave's last group issue nothing: 5 of the 96 slots of a K block at MB = 13 - they used to
      //  repeat the last tile into scratch accumulators, 5 % of the kernel's MFMAs)
#define GPSA_GR_MMA(F)                                                                          \
  _Pragma("unroll") for (int q = 0; q < NL; ++q)                                                \
    _Pragma("unroll") for (int u = 0; u < GR_G; ++u)                                            \
      if (GR_G * gp + u < N)                                                                    \
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
__device__ __forceinline__ void gram_wave_run(const float* __restrict__ alpha, const float* __restrict__ g,
                                              long long gstride, int M, long long C, int L, int nsplit, float* __restrict__ part,
                                              float* __restrict__ sA_, float* __restrict__ sG_,
                                              const float* __restrict__ dmean) {
  constexpr int MP = MB * 16;
  constexpr GramPlan<MB> PLAN{};
  constexpr int NS = PLAN.max_cnt();
  constexpr int NKB = GR_KC / 16, NPIECE = MB * NKB;
  constexpr int NPW = (NPIECE + 3) / 4;
  constexpr int SA_STRIDE = NPW * 4 * 256, SG_STRIDE = 2 * NL * GR_KC;  // g rows, then dmean rows (d-delta option)
  const int nstage = (dmean != nullptr ? 2 : 1) * NL * (GR_KC / 4);  // lanes of the g / dmean piece
  const int drow = dmean != nullptr ? M - 16 * (MB - 1) : -1;
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
  // meet g == 0 (gstride == C: g is the caller's [L, C] panel and the 16-byte groups past column C are fetched
  // from a block of zeros instead - C is a multiple of 4 here, so no group straddles it; otherwise the launcher
  // made a zero-padded copy with gstride = C rounded up to GR_KC columns).  Every wave issues exactly NPW + 1 operations per stage (surplus pieces re-load piece 0 into
  // an unused slot; all four waves DMA the same 128 bytes of g) so that a counted vmcnt(NPW+1) means
  // "everything but the newest stage has landed".
  auto gsrc = [&](int ln, long long col) -> const float* {  // lane ln of the piece: output ln / 16 of g, then of dmean
    const int which = ln / (NL * (GR_KC / 4)), lq = min(l0 + (ln / (GR_KC / 4)) % NL, L - 1);
    if (which == 1) return col >= C ? gram_zero16 : dmean + (long long)lq * C + col;
    return (gstride == C && col >= C) ? gram_zero16 : g + (long long)lq * gstride + col;
  };
#define GPSA_GR_STAGE(CH, BUF)                                                               \
  {                                                                                          \
    const long long cb__ = (long long)(CH) * GR_KC;                                          \
    if (ALIGNED) {                                                                           \
      /* wave w stages K block w of every row tile: piece pc = row tile pc -> KiB w * NPW + pc; m0 once per 4 pieces */ \
      long long col = cb__ + w * 16 + kq * 4;                                                \
      col = col < C - 4 ? col : C - 4;                                                       \
      _Pragma("unroll") for (int pc = 0; pc < NPW; ++pc) {                                   \
        if ((pc & 3) == 0)                                                                   \
          dma_set_m0(__builtin_amdgcn_readfirstlane(lds_addr(sA_ + (BUF) * SA_STRIDE + (w * NPW + pc) * 256))); \
        int row = (pc < MB ? pc : 0) * 16 + j;                                               \
        row = row < M ? row : M - 1;                                                         \
        const float* gp__ = alpha + (long long)row * C + col;                                \
        if ((pc & 3) == 0) glds16_m0<0>(gp__);                                               \
        if ((pc & 3) == 1) glds16_m0<1024>(gp__ - 256);                                      \
        if ((pc & 3) == 2) glds16_m0<2048>(gp__ - 512);                                      \
        if ((pc & 3) == 3) glds16_m0<3072>(gp__ - 768);                                      \
      }                                                                                      \
      if (lane < nstage) {                                                                   \
        dma_set_m0(__builtin_amdgcn_readfirstlane(lds_addr(sG_ + (BUF) * SG_STRIDE)));       \
        glds16_m0<0>(gsrc(lane, cb__ + (lane % (GR_KC / 4)) * 4));                           \
      }                                                                                      \
    } else {                                                                                 \
      for (int e = tid; e < NPIECE * 256; e += 256) {                                        \
        const int piece = e >> 8, ln = (e >> 2) & 63, r = e & 3;                             \
        const int rb = piece / NKB, kb = piece % NKB;                                        \
        const int row = rb * 16 + (ln & 15);                                                 \
        const long long col = cb__ + kb * 16 + (ln >> 4) * 4 + r;                            \
        sA_[(BUF) * SA_STRIDE + (kb * NPW + rb) * 256 + (e & 255)] =                         \
            (row < M && col < C) ? alpha[(long long)row * C + col] : 0.f;                    \
      }                                                                                      \
      if (tid < NL * GR_KC) /* (the d-delta option is for the aligned path only) */          \
        sG_[(BUF) * SG_STRIDE + tid] = g[(long long)min(l0 + tid / GR_KC, L - 1) * gstride + cb__ + tid % GR_KC]; \
    }                                                                                        \
  }

  if (ch1 > ch0) GPSA_GR_STAGE(ch0, 0)
  GPSA_DMA_DRAIN();
  __syncthreads();
  int buf = 0;
  for (long long ch = ch0; ch < ch1; ++ch) {
    // the other slot held chunk ch-1: everyone left it before the barrier that ended that iteration
    const bool more = ch + 1 < ch1;
    constexpr int NGRP_ = (((NS + GR_G - 1) / GR_G) + 1) & ~1;
    constexpr bool SPREAD = ALIGNED && NGRP_ >= 5;  // (the g piece rides in group 4 of K block 0)
    if (more && !SPREAD) GPSA_GR_STAGE(ch + 1, buf ^ 1)
    const float* img = sA_ + buf * SA_STRIDE + lane * 4;
    // next chunk's pieces of this wave (K block w of row tiles 0 .. MB-1 -> KiB w * NPW + row tile), four per K block
    // of the current chunk: m0 with the first of the four, the others through the immediate offset
    long long ncol = (ch + 1) * GR_KC + w * 16 + kq * 4;
    ncol = ncol < C - 4 ? ncol : C - 4;
    const float* nsrc = alpha + ncol;
    float* nimg = sA_ + (buf ^ 1) * SA_STRIDE + w * NPW * 256;
    auto stage = [&](int kb, int gp) {
      if (!SPREAD || !more) return;
      if (gp < 4) {
        const int pc = kb * 4 + gp;
        if (pc < NPW) {  // (uniform)
          if (gp == 0) dma_set_m0(__builtin_amdgcn_readfirstlane(lds_addr(nimg + kb * 4 * 256)));
          int row = pc * 16 + j;
          row = row < M ? row : M - 1;
          const float* gp__ = nsrc + (long long)row * C;
          if (gp == 0) glds16_m0<0>(gp__);
          if (gp == 1) glds16_m0<1024>(gp__ - 256);
          if (gp == 2) glds16_m0<2048>(gp__ - 512);
          if (gp == 3) glds16_m0<3072>(gp__ - 768);
        }
      } else if (kb == 0 && lane < nstage) {
        dma_set_m0(__builtin_amdgcn_readfirstlane(lds_addr(sG_ + (buf ^ 1) * SG_STRIDE)));
        glds16_m0<0>(gsrc(lane, (ch + 1) * GR_KC + (lane % (GR_KC / 4)) * 4));
      }
    };
    static_assert(NPW <= 16, "four pieces per K block of the chunk being multiplied");
    gram_wave_chunk<MB, NKB, W, NS, NL>(img, sG_ + buf * SG_STRIDE, kq, acc, stage, drow, lane & 15);
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
gram_mfma_kernel(const float* __restrict__ alpha, const float* __restrict__ g, long long gstride, int M,
                 long long C, int L, int nsplit, float* __restrict__ part, const float* __restrict__ dmean) {
  constexpr int NKB = GR_KC / 16, NPIECE = MB * NKB;  // 1-KiB pieces (16 rows x 16 cols) per chunk
  // LDS image of a chunk: piece (rb, kb) at float offset (rb*NKB + kb)*256, stored in MFMA-fragment
  // order: lane j + 16 kq holds alpha[16 rb + j][cb + 16 kb + 4 kq .. +3]  => a fragment read is one
  // conflict-free ds_read_b128 at lane*16 bytes.
  constexpr int NPW = (NPIECE + 3) / 4;  // LDS-DMA pieces per wave per stage (uniform; + 1 for g)
  // two slots: the chunk being multiplied and the next one in flight (a chunk is ~12k MFMA cycles per
  // wave, far longer than the DMA latency, so one stage ahead is enough and the chunks can be big)
  __shared__ __attribute__((aligned(16))) float sA[2][NPW * 4 * 256];
  __shared__ __attribute__((aligned(16))) float sG[2][2 * NL * GR_KC];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  switch (w) {
    case 0: gram_wave_run<MB, ALIGNED, NL, 0>(alpha, g, gstride, M, C, L, nsplit, part, &sA[0][0], &sG[0][0], dmean); break;
    case 1: gram_wave_run<MB, ALIGNED, NL, 1>(alpha, g, gstride, M, C, L, nsplit, part, &sA[0][0], &sG[0][0], dmean); break;
    case 2: gram_wave_run<MB, ALIGNED, NL, 2>(alpha, g, gstride, M, C, L, nsplit, part, &sA[0][0], &sG[0][0], dmean); break;
    default: gram_wave_run<MB, ALIGNED, NL, 3>(alpha, g, gstride, M, C, L, nsplit, part, &sA[0][0], &sG[0][0], dmean); break;
  }
}

template <typename TO>
__global__ void gram_reduce_kernel(const float* __restrict__ part, int M, int MP, int L, int nsplit,
                                   TO* __restrict__ out, float* __restrict__ ddelta, float dbeta) {
  // grid: (ceil(M / 32) column blocks, M rows, L); threads 32 x 8 (8 rows per block in y)
  const int jj = blockIdx.x * 32 + (threadIdx.x & 31);
  const int i = blockIdx.y * 8 + (threadIdx.x >> 5);
  const int l = blockIdx.z;
  if (ddelta != nullptr && i == M && jj < M) {  // the d-delta option: row M of the partials is d delta_F[:, l]
    const float* p = part + (long long)l * nsplit * MP * MP + (long long)M * MP + jj;
    const long long mm = (long long)MP * MP;
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += (double)p[sp * mm];
    float* d = ddelta + (long long)jj * L + l;
    *d = dbeta != 0.f ? dbeta * *d + (float)s : (float)s;
    return;
  }
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


GPSA_GRAM_SHAPES(GPSA_GRAM_DEFINE)
template __global__ void gram_reduce_kernel<float>(const float* __restrict__, int, int, int, int, float* __restrict__, float* __restrict__, float);
template __global__ void gram_reduce_kernel<double>(const float* __restrict__, int, int, int, int, double* __restrict__, float* __restrict__, float);

}  // namespace gpsa
