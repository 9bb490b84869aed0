// quad_sym_mfma_kernel: the symmetric quadratic form (forward without a backward).
#include "qf_common.hpp"

namespace gpsa {

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
    /* wave-major slots (round 4): slot sl = pc * 4 + w sits at KiB w * NPW + pc; m0 once per four pieces */ \
    _Pragma("unroll") for (int pc = 0; pc < NPW; ++pc) {                                       \
      const int sl = pc * 4 + w;                                                               \
      const int se = sl < n1__ + n2__ ? sl : 0;                                                \
      const int kc__ = se < n1__ ? q__ : p__;                                                  \
      const int rt__ = se < n1__ ? se : se - n1__;                                             \
      if ((pc & 3) == 0)                                                                       \
        dma_set_m0(__builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][(w * NPW + pc) * 256]))); \
      const float* g__ = m__ + (kc__ * MB + rt__) * 256;                                       \
      if ((pc & 3) == 0) glds16_m0<0>(g__);                                                    \
      if ((pc & 3) == 1) glds16_m0<1024>(g__ - 256);                                           \
      if ((pc & 3) == 2) glds16_m0<2048>(g__ - 512);                                           \
      if ((pc & 3) == 3) glds16_m0<3072>(g__ - 768);                                           \
    }                                                                                          \
  }
#define GPSA_QS_POS(SL) (((SL) & 3) * NPW + ((SL) >> 2))
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
        float4 a_nxt = *reinterpret_cast<const float4*>(base + GPSA_QS_POS(0) * 256);
#pragma unroll
        for (int sl = 0; sl < MB + 1; ++sl) {
          if (sl < n1 + n2) {
            const int kc = sl < n1 ? q : p;
            const int rt = sl < n1 ? sl : sl - n1;
            const float4 a4 = a_nxt;
            if (sl + 1 < n1 + n2) a_nxt = *reinterpret_cast<const float4*>(base + GPSA_QS_POS(sl + 1) * 256);
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
#undef GPSA_QS_POS
#undef GPSA_QS_STAGE_NEXT
}


GPSA_SYM_SHAPES(GPSA_SYM_DEFINE)

}  // namespace gpsa
