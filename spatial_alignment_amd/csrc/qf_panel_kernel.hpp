// panel_mfma_kernel: the register-resident full-product kernel (template body; instantiated per mode by
// qf_panel_quad.hip, qf_panel_accum.hip and qf_panel_store.hip).
#pragma once
#include "qf_common.hpp"

namespace gpsa {

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
                  float* __restrict__ keep) {  // QUAD: optional, the products P_l X in fragment order
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
  // Ring slots are wave-major (round 4, as panel_elbo_kernel): wave w's pieces w, w + 4, ... are the NPW consecutive
  // KiB at w * NPW, so that one m0 write reaches four of them through the instruction's immediate offset
  // (qf_common.hpp: glds16_m0); row tile rt's fragment (piece rt) sits at KiB (rt & 3) * NPW + (rt >> 2).  The stage
  // cursor is a pointer that advances by one chunk and a count of the chunks left in the step (the chunks of a step's
  // outputs are consecutive in the packed operand); the next step is looked up when the count runs out.
#define GPSA_POS(P_) (((P_) & 3) * NPW + ((P_) >> 2))
  const TileOrder ord(it0, it1, L);
  long long sstep = 0, stile_;
  int sa_, sb_;
  ord.get(0, stile_, sa_, sb_);
  const float* sp = Ppk + (long long)sa_ * MB * CHUNK + lane * 4;
  int srem = (sb_ - sa_ + 1) * MB;
#define GPSA_STAGE_PIECE(BUF, PC)                                                              \
  {                                                                                            \
    constexpr int pc__ = (PC);                                                                 \
    if ((pc__ & 3) == 0)                                                                       \
      dma_set_m0(__builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][(w * NPW + pc__) * 256]))); \
    const int piece = pc__ * 4 + w;                                                            \
    glds16_m0<(pc__ & 3) * 1024>(sp + ((piece < MB ? piece : MB - 1) - (pc__ & 3)) * 256);     \
  }
#define GPSA_STAGE_PIECE_RT(BUF, RT)                                                           \
  switch (RT) { /* (a constant after unrolling: the immediate offset must be one) */           \
    case 0: GPSA_STAGE_PIECE(BUF, 0) break;                                                    \
    case 1: GPSA_STAGE_PIECE(BUF, (NPW > 1 ? 1 : 0)) break;                                    \
    case 2: GPSA_STAGE_PIECE(BUF, (NPW > 2 ? 2 : 0)) break;                                    \
    case 3: GPSA_STAGE_PIECE(BUF, (NPW > 3 ? 3 : 0)) break;                                    \
    case 4: GPSA_STAGE_PIECE(BUF, (NPW > 4 ? 4 : 0)) break;                                    \
    case 5: GPSA_STAGE_PIECE(BUF, (NPW > 5 ? 5 : 0)) break;                                    \
    case 6: GPSA_STAGE_PIECE(BUF, (NPW > 6 ? 6 : 0)) break;                                    \
    default: GPSA_STAGE_PIECE(BUF, (NPW > 7 ? 7 : 0)) break;                                   \
  }
  static_assert(NPW <= 8, "pieces per wave and stage");
#define GPSA_STAGE_ADVANCE()                                                                   \
  {                                                                                            \
    if (--srem > 0) {                                                                          \
      sp += CHUNK;                                                                             \
    } else if (sstep + 1 < ord.n) {                                                            \
      ++sstep;                                                                                 \
      ord.get(sstep, stile_, sa_, sb_);                                                        \
      sp = Ppk + (long long)sa_ * MB * CHUNK + lane * 4;                                       \
      srem = (sb_ - sa_ + 1) * MB;                                                             \
    } else {                                                                                   \
      srem = 0x7fffffff; /* nothing left: the surplus stages walk on BEHIND the last chunk (up to AHEAD chunks, never multiplied: every workspace holds slabs or slack there) */         \
    }                                                                                          \
  }
#define GPSA_STAGE_NEXT(BUF)                                                                   \
  {                                                                                            \
    _Pragma("unroll") for (int pc = 0; pc < NPW; ++pc) GPSA_STAGE_PIECE_RT(BUF, pc)            \
    GPSA_STAGE_ADVANCE()                                                                       \
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
        float4 a_nxt = *reinterpret_cast<const float4*>(base + GPSA_POS(0) * 256);
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
              if (rt + 1 < MB) a_nxt = *reinterpret_cast<const float4*>(base + GPSA_POS(rt + 1) * 256);
            } else if (r == 1) {
              if (MB >= NPW + 3) {
                if (rt < NPW) GPSA_STAGE_PIECE_RT(buf == 0 ? 2 : buf - 1, rt)
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
#undef GPSA_POS
#undef GPSA_STAGE_PIECE_RT
#undef GPSA_STAGE_PIECE
#undef GPSA_STAGE_ADVANCE
#undef GPSA_STAGE_NEXT
#undef GPSA_FLUSH
}

}  // namespace gpsa
