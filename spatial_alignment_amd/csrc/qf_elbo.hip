// panel_elbo_kernel: the data GP's forward, its Gaussian likelihood and the backward's abar in one pass over the
// products Omega_l alpha (the dominant kernel of the training step).
#include "qf_common.hpp"

namespace gpsa {

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

// PAIRB (round 5): ONE barrier per TWO K chunks.  The ring then has six one-chunk slots and four stages in flight; the
// wait + barrier in front of a chunk's last row tile is taken only by the odd chunks (and by an output's last chunk:
// MB may be odd) and covers the next two chunks: at most the two newest stages (chunks c + 3, c + 4) stay outstanding.
// A chunk's pieces go to the slot of chunk c - 2, which every wave has finished before the barrier it has last
// passed (at the end of chunk c - 1 or c - 2).
template <int MB, int NCT, int RL, bool FULLT, bool PAIRB>
__global__ void __launch_bounds__(256, (MB * NCT >= 14) ? 1 : 2) panel_elbo_kernel(ElboArgs a) {
  constexpr int MP = MB * 16;
  constexpr int WGCOLS = 64 * NCT;
  constexpr int CHUNK = MP * 16;
  constexpr int NPW = (MB + 3) / 4;
  constexpr int BUFF = NPW * 4 * 256;
  constexpr int NGATHER = (3 * NCT * 16 + 63) / 64;  // 4-byte LDS-DMA operations per wave and output
  constexpr int NSLOT = PAIRB ? 6 : 3, AHEAD = PAIRB ? 4 : 2;  // ring slots, stages in flight
  __shared__ __attribute__((aligned(16))) float lds[NSLOT][BUFF];
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
  // LDS layout of a ring slot: wave-major - wave w's pieces (w, w + 4, ...) are the NPW consecutive KiB at w * NPW, so
  // that one m0 write per stage covers them through the instruction's immediate offset (qf_common.hpp: glds16_m0);
  // row tile rt's fragment (piece rt) sits at KiB (rt & 3) * NPW + (rt >> 2)
#define GPSA_POS(P_) (((P_) & 3) * NPW + ((P_) >> 2))
  const TileOrder ord(it0, it1, L);
  // The stage cursor: the chunks of a step (a column tile's outputs a .. b) are consecutive in the packed operand, so
  // the cursor is a pointer that advances by one chunk and a count of the chunks left in the step; the next step's
  // start is looked up (TileOrder::get: branches, 64-bit compares) only when the count runs out - a few times per
  // workgroup.  (Round 3 re-derived (step, l, kc) -> address with that branchy code at every chunk: ~75 scalar
  // instructions and three taken branches between two MFMAs, 13 times per output.)
  long long sstep = 0, stile_;
  int sa_, sb_;
  ord.get(0, stile_, sa_, sb_);
  const float* sp = Ppk + (long long)sa_ * MB * CHUNK + lane * 4;  // this lane's 16 bytes of piece 0 of the chunk
  int srem = (sb_ - sa_ + 1) * MB;                                  // chunks of the step not staged yet
  // piece PC (compile time) of the stage cursor's chunk -> slot BUF; m0 is written with the first piece of a stage
#define GPSA_STAGE_PIECE(BUF, PC)                                                              \
  {                                                                                            \
    constexpr int pc__ = (PC);                                                                 \
    if (pc__ == 0) dma_set_m0(__builtin_amdgcn_readfirstlane(lds_addr(&lds[BUF][w * NPW * 256]))); \
    const int piece = pc__ * 4 + w;                                                            \
    glds16_m0<pc__ * 1024>(sp + ((piece < MB ? piece : MB - 1) - pc__) * 256);                 \
  }
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
    GPSA_STAGE_PIECE(BUF, 0)                                                                   \
    if (NPW > 1) GPSA_STAGE_PIECE(BUF, (NPW > 1 ? 1 : 0))                                      \
    if (NPW > 2) GPSA_STAGE_PIECE(BUF, (NPW > 2 ? 2 : 0))                                      \
    if (NPW > 3) GPSA_STAGE_PIECE(BUF, (NPW > 3 ? 3 : 0))                                      \
    GPSA_STAGE_ADVANCE()                                                                       \
  }
  static_assert(NPW <= 4, "a wave's pieces of a stage must lie within the 4 KiB an immediate offset reaches");

  // likelihood constants (elementwise.hip: loglik_*_kernel)
  const double sN = exp((double)a.noise_u[0]) + 1e-5;
  const float inv = (float)(1.0 / sN);
  const float coef = (float)(-1.0 / (sN * sN * (double)a.S));  // dLoss/dF = coef (Y - F) at upstream gradient 1
  const double var0 = exp((double)a.var_u[0]);
  double z2 = 0.0;

  int buf = 0;
  GPSA_STAGE_NEXT(0)
  GPSA_STAGE_NEXT(1)
  if (PAIRB) {
    GPSA_STAGE_NEXT(2)
    GPSA_STAGE_NEXT(3)
    GPSA_DMA_WAIT(2 * NPW);
  } else {
    GPSA_DMA_WAIT(NPW);
  }
  __syncthreads();
  // The fragment of row tile 0 of the NEXT chunk is read during the last row tile of the current one: the chunk's
  // wait + barrier sit in FRONT of that last row tile (every wave has then issued - and, by its lgkmcnt wait,
  // received - all its reads of the current slot, and the next slot's pieces, issued two chunks ago, have landed),
  // so no chunk starts with an LDS round trip in the open (13 of them per output before)
  float4 a_nxt = *reinterpret_cast<const float4*>(&lds[0][lane * 4 + GPSA_POS(0) * 256]);

  for (long long step = 0; step < ord.n; ++step) {
    long long tile;
    int l_lo, l_hi;
    ord.get(step, tile, l_lo, l_hi);
    const long long cw = tile * WGCOLS + (long long)w * (16 * NCT);
    float resid[NCT];
    bool okc[NCT];
    load_alpha_slab<MB, NCT, true, FULLT>(X, M, C, cw, j, kq, xb, okc);
    if (RL < 4) load_alpha_last<MB, NCT, 4>(X, M, C, cw, j, kq, RL, xl);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const long long c = cw + ct * 16 + j;
      // sigma^2 - q formed in fp64 before rounding (data_sample_fwd_kernel)
      const double qc = a.q[okc[ct] ? c : C - 1];
      resid[ct] = okc[ct] ? (float)(var0 - qc) : 1.f;
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
      // (meanT == nullptr: the mean is row M of this kernel's own product - the caller packed delta_l^T there - and
      //  the gather's kind-0 slots fetch a second copy of eps that nobody reads)
      gp[o] = kind == 0 ? (a.meanT != nullptr ? a.meanT + (long long)l_lo * C + c : a.eps + c * L + l_lo)
                        : (kind == 1 ? a.eps + c * L + l_lo : a.Y + (c % a.N) * L + l_lo);
      gstep[o] = (kind == 0 && a.meanT != nullptr) ? C : 1;
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
        const float* nbase = &lds[buf == NSLOT - 1 ? 0 : buf + 1][lane * 4];
        const int sbuf = buf + AHEAD >= NSLOT ? buf + AHEAD - NSLOT : buf + AHEAD;  // the slot this chunk's stage fills
        if (kc == 0) {
          // this output's mean / eps / Y: BEFORE the chunk's ring stage is issued, so that the counted wait at the
          // end of the chunk (all but the newest NPW operations) covers them
          dma_set_m0(__builtin_amdgcn_readfirstlane(lds_addr(&sgat[w][0])));
          glds4_m0<0>(gp[0]);
          gp[0] += gstep[0];
          if (NGATHER > 1) {
            glds4_m0<256>(gp[NGATHER > 1 ? 1 : 0] - 64);
            gp[NGATHER > 1 ? 1 : 0] += gstep[NGATHER > 1 ? 1 : 0];
          }
          if (NGATHER > 2) {
            glds4_m0<512>(gp[NGATHER > 2 ? 2 : 0] - 128);
            gp[NGATHER > 2 ? 2 : 0] += gstep[NGATHER > 2 ? 2 : 0];
          }
          static_assert(NGATHER <= 3, "gather operations per wave and output");
        }
#pragma unroll
        for (int rt = 0; rt < MB; ++rt) {
          const float4 a4 = a_nxt;
          const float av[4] = {a4.x, a4.y, a4.z, a4.w};
          if (rt == MB - 1 && (!PAIRB || (kc & 1) || kc == MB - 1)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's last read of the current slot is in
            if (PAIRB) GPSA_DMA_WAIT(2 * NPW);
            else GPSA_DMA_WAIT(NPW);
            __syncthreads();
          }
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
              a_nxt = *reinterpret_cast<const float4*>((rt + 1 < MB ? base + GPSA_POS(rt + 1 < MB ? rt + 1 : 0) * 256
                                                                    : nbase + GPSA_POS(0) * 256));
            } else if (r == 1) {
              if (MB >= NPW + 3) {
                if (rt == 0) GPSA_STAGE_PIECE(sbuf, 0)
                if (rt == 1 && NPW > 1) GPSA_STAGE_PIECE(sbuf, (NPW > 1 ? 1 : 0))
                if (rt == 2 && NPW > 2) GPSA_STAGE_PIECE(sbuf, (NPW > 2 ? 2 : 0))
                if (rt == 3 && NPW > 3) GPSA_STAGE_PIECE(sbuf, (NPW > 3 ? 3 : 0))
                if (rt == NPW) GPSA_STAGE_ADVANCE()
              } else if (rt == 0) {
                GPSA_STAGE_NEXT(sbuf)
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        buf = (buf == NSLOT - 1) ? 0 : buf + 1;
      }
      // closing of output l: v, the draw, its likelihood term and gradient, and g_l W_l into the second set
      float z2l = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        f32x2 s2a = (f32x2){0.f, 0.f}, s2b = (f32x2){0.f, 0.f};  // packed FMAs (VGPR x VGPR), two independent chains
#pragma unroll
        for (int rt = 0; rt < MB; ++rt) {
          const f32x2 x01 = (f32x2){xb[ct][rt][0], xb[ct][rt][1]}, x23 = (f32x2){xb[ct][rt][2], xb[ct][rt][3]};
          s2a = __builtin_elementwise_fma(acc[rt][ct].xy, x01, s2a);
          s2b = __builtin_elementwise_fma(acc[rt][ct].zw, x23, s2b);
        }
        float s = (s2a.x + s2a.y) + (s2b.x + s2b.y);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        float mean;
        if (a.meanT != nullptr) {  // (uniform)
          mean = sgat[w][(ct * 3 + 0) * 16 + j];
        } else {
          // row M = 16 (MB - 1) + lr of the product: C layout row 4 kq + r -> lane j + 16 (lr >> 2), component lr & 3
          const int lr = M - 16 * (MB - 1);
          const f32x4 t4 = acc[MB - 1][ct];
          const int r0 = lr & 3;
          const float pick = r0 == 0 ? t4.x : (r0 == 1 ? t4.y : (r0 == 2 ? t4.z : t4.w));
          mean = __shfl(pick, j + 16 * (lr >> 2), 64);
        }
        const float e = sgat[w][(ct * 3 + 1) * 16 + j];
        const float y = sgat[w][(ct * 3 + 2) * 16 + j];
        const float var = resid[ct] + s + 2e-5f;  // TWO_JITTER (elementwise.hip)
        const float sd = sqrtf(var);
        const float Fd = mean + sd * e;  // the draw (data_sample_fwd_kernel's expression)
        const float rres = y - Fd;
        const float dF = coef * rres;
        const float gv = okc[ct] ? dF * e * 0.5f / sd : 0.f;
        if (okc[ct] && kq == 0) {
          const long long o = (long long)l * C + cw + ct * 16 + j;
          a.g[o] = gv;
          a.dmeanT[o] = dF;
          if (a.FT != nullptr) a.FT[o] = Fd;  // (uniform)
          const float z = rres * inv;
          z2l += z * z;
        }
        // Register files (round 4): this unit is built with -amdgpu-mfma-vgpr-form (__graft_entry__.build), so the FIRST
        // accumulator set (the product being formed) and the alpha slab live in arch VGPRs - the dot product above
        // needs no register-file crossing - and the SECOND set lives in the AGPR file, every access through an
        // "a"-constrained operand (left to itself the allocator would home it in VGPRs and evict the alpha slab).
        // Measured, kernel + pack + slab reduce at the headline size: both sets in AGPRs (round 3) 3468 us, this 3381;
        // tried and not kept: alpha slab in AGPRs as the MFMAs' B operand (full rate: tools/microbench/mfma_operand.hip)
        // with both sets in VGPRs and a packed-FMA update - 60 % fewer closing instructions, 3423 us (3343 against
        // 3330 with the stage cursor below: v_pk_fma_f32 buys nothing here and the allocator spills 900 bytes).
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
#undef GPSA_POS
#undef GPSA_STAGE_PIECE
#undef GPSA_STAGE_ADVANCE
#undef GPSA_STAGE_NEXT
}

GPSA_ELBO_SHAPES(GPSA_ELBO_DEFINE)
template __global__ void panel_elbo_kernel<13, 2, 2, true>(ElboArgs);
template __global__ void panel_elbo_kernel<13, 2, 4, true>(ElboArgs);
template __global__ void panel_elbo_kernel<13, 2, 2, true, true>(ElboArgs);
template __global__ void panel_elbo_kernel<13, 2, 4, true, true>(ElboArgs);

}  // namespace gpsa
