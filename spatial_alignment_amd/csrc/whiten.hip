// fp64 projection of the cross-covariance onto the inducing points on the fp64 matrix cores:
//     alpha = K_uu^-1 K_uf ,   q[c] = k_c^T K_uu^-1 k_c = sum_m K_uf[m,c] alpha[m,c]
// (gpsa/models/vgpsa.py:179-183, 194-196: the K_fu K_uu^-1 factors of the conditional mean and
// covariance of both GP layers).  The M x M inverse comes from the fp64 Cholesky factor
// (linalg.hip); this kernel is the only N-scaled fp64 product of a step, so it is written against
// v_mfma_f64_16x16x4_f64 (78.6 TFLOP/s dense on MI355X - the same peak as the fp64 vector pipe, but
// reachable, and it leaves the vector pipe to the address arithmetic).
//
// One workgroup = 4 waves = 64 columns.  Each wave keeps its 16 columns of K_uf (the B operand, one
// fp64 per lane and K step) and all MB row tiles of the result in registers; the packed inverse
// streams through a 2-deep LDS ring in chunks of 16 K values by LDS-DMA, shared by the 4 waves.  The
// C/D layout of the fp64 MFMA (row = (lane>>4) + 4*reg) coincides with the B operand's K layout
// (k = 4*step + (lane>>4)), so q closes in registers: q = sum acc[rt][reg] * xb[4*rt + reg], then two
// cross-lane adds.
#include <type_traits>

#include "internal.hpp"

namespace gpsa {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr64_t;

// LDS-DMA of 16 bytes per lane to  m0 + IMM + 16 lane  (IMM also added to the global address: the caller passes
// source - IMM bytes).  m0 is written once per group of up to four 1-KiB pieces: an s_mov to m0 next to a busy matrix
// pipe costs the wave ~40 cycles (tools/microbench/panel_shape2.hip), and round 3's helper saved, set and restored it
// around every piece - 14 writes per wave and K chunk here, against 52 fp64 MFMAs.
__device__ __forceinline__ void dma64_set_m0(unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_base) : "memory");
}
template <int IMM>
__device__ __forceinline__ void glds16_f64_m0(const double* gsrc_minus_imm) {
  static_assert(IMM >= 0 && IMM < 4096, "immediate offset of a global instruction");
  asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(gsrc_minus_imm), "n"(IMM) : "memory");
}

// Kinv [M][M] row-major -> Apk[kc][ks][rt][lane] = Kinv[16 rt + (lane&15)][16 kc + 4 ks + (lane>>4)]
// (zero padded to MP = 16 MB): the A-operand fragment of every MFMA is one lane-linear 512-byte row.
// qz (optional): nq doubles to zero on the way - the persistent projection kernel (proj64.hip) closes q by atomic adds
__global__ void pack_whiten_kernel(const double* __restrict__ Kinv, int M, int MB, double* __restrict__ Apk,
                                   long long sKinv, double* __restrict__ qz, long long nq) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long tot = (long long)MB * 4 * MB * 64;
  if (qz != nullptr) {
    const long long nth = (long long)gridDim.x * gridDim.y * blockDim.x;
    for (long long i = (blockIdx.y * (long long)gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < nq; i += nth)
      qz[i] = 0.0;
  }
  if (idx >= tot) return;
  Kinv += blockIdx.y * sKinv;  // problem blockIdx.y of a batch (the views' warp GPs)
  Apk += blockIdx.y * tot;
  const int lane = (int)(idx & 63);
  long long t = idx >> 6;
  const int rt = (int)(t % MB);
  t /= MB;
  const int ks = (int)(t & 3);
  const int kc = (int)(t >> 2);
  const int row = rt * 16 + (lane & 15), k = kc * 16 + ks * 4 + (lane >> 4);
  Apk[idx] = (row < M && k < M) ? Kinv[(long long)row * M + k] : 0.0;
}

// TI: storage type of the right-hand side (fp64 covariance in the forward; the fp32 gradient panel in the
// data layer's backward, widened on the fly: K^-1 in ONE fp64 pass there instead of two fp32
// triangular passes), TO: storage type of the result.
//
// STREAM: the right-hand side streams through registers one K chunk (16 rows) ahead of the MFMAs that
// consume it, so the accumulators are the only resident state and two workgroups share a CU (q then
// closes against a second, cache-warm read of the column); otherwise the whole 16-column slab stays in
// registers, one workgroup per CU -- the shorter dependency chain when the grid does not fill the chip.
// optional fused epilogue: alpha[m,c] = (K^-1 X)[m,c] + s * d[c] * X2[m,c]  (the data GP's backward:
// dK_uf = K^-1 abar + 2 qbar o alpha in the solve's own pass instead of a second sweep over two panels)
struct WhitenAxpy {
  const float* X2;
  const float* d;
  float s;
  float* out32 = nullptr;  // a second, fp32-rounded copy of the result (the exact inducing-point gradient keeps both)
};

// RS (row split, resident right-hand side only): a workgroup covers 32 columns instead of 64 and each pair of
// waves shares 16 of them, one wave taking the upper half of the row tiles, the other the lower half (an odd MB
// makes the halves overlap by one tile: both write it, one counts it in q).  Half the dependent MFMAs per wave:
// when even the doubled grid leaves every workgroup a CU of its own (the warp GPs of a 1/8 shard: a wave's 676
// dependent MFMAs are the whole kernel time) the latency drops by a third; the packed inverse is streamed by twice
// as many workgroups, which is why it is not the rule.
template <int MB, typename TI, typename TO, bool STREAM, bool RS = false>
__global__ void __launch_bounds__(256, (MB >= (STREAM ? 14 : 13) && !RS) ? 1 : 2)
whiten_mfma_kernel(const double* __restrict__ Apk0, const TI* __restrict__ X0, int M, long long C,
                   TO* __restrict__ alpha0, double* __restrict__ q0, long long sX, WhitenAxpy ax) {
  static_assert(!(RS && STREAM), "row split: resident right-hand side only");
  constexpr int NRT = RS ? (MB + 1) / 2 : MB;   // row tiles of this wave
  constexpr int OVER = RS ? 2 * NRT - MB : 0;   // tiles both halves compute
  constexpr int CHUNK = 4 * MB * 64;          // doubles per K chunk (2*MB pieces of 1 KiB)
  // problem blockIdx.y of a batch: its own packed inverse, panels at stride sX, q at stride C
  const long long pb = blockIdx.y;
  const double* __restrict__ Apk = Apk0 + pb * (long long)MB * CHUNK;
  const TI* __restrict__ X = X0 + pb * sX;
  TO* __restrict__ alpha = alpha0 + pb * sX;
  double* __restrict__ q = q0 + pb * C;  // only dereferenced when q0 != nullptr
  constexpr int NPIECE = 2 * MB;
  constexpr int NPW = (NPIECE + 3) / 4;       // LDS-DMA operations per wave per stage (uniform)
  constexpr int BUFD = NPW * 4 * 128;         // doubles per ring slot (incl. dummy pieces)
  __shared__ __attribute__((aligned(16))) double lds[2][BUFD];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const long long c = RS ? blockIdx.x * 32LL + (w & 1) * 16 + j : blockIdx.x * 64LL + w * 16 + j;
  const int rt0 = (RS && (w >> 1)) ? MB - NRT : 0;  // first row tile of this wave (wave-uniform)
  const bool okc = c < C;
  const TI* xcol = X + (okc ? c : 0);
  // LDS byte address of the ring, taken ONCE from the array's base (one foldable address-space cast)
  const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr64_t)(&lds[0][0]);

  // ring slot, wave-major: wave w's pieces w, w + 4, ... are the NPW consecutive KiB at w * NPW (one m0 value reaches
  // four of them through the immediate offset); element e = (ks * MB + rt) * 64 + lane of a chunk sits at
  //   GPSA_WPOS(e >> 7) * 128 + (e & 127)
#define GPSA_WPOS(P_) (((P_) & 3) * NPW + ((P_) >> 2))
#define GPSA_WELEM(KS, RT) (GPSA_WPOS(((KS) * MB + (RT)) >> 1) * 128 + ((((KS) * MB + (RT)) & 1) << 6))
  // piece PC (compile time) of chunk Q -> slot BUF
#define GPSA_WSTAGE_PIECE(Q, BUF, PC)                                                          \
  {                                                                                            \
    constexpr int pc__ = (PC);                                                                 \
    if ((pc__ & 3) == 0)                                                                       \
      dma64_set_m0(__builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((BUF) * BUFD + (w * NPW + pc__) * 128) * 8))); \
    const int piece = pc__ * 4 + w;                                                            \
    glds16_f64_m0<(pc__ & 3) * 1024>(Apk + (long long)(Q) * CHUNK + lane * 2 +                 \
                                     ((piece < NPIECE ? piece : NPIECE - 1) - (pc__ & 3)) * 128); \
  }
#define GPSA_WSTAGE(Q, BUF)                                                                    \
  {                                                                                            \
    GPSA_WSTAGE_PIECE(Q, BUF, 0)                                                               \
    if (NPW > 1) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 1 ? 1 : 0))                                  \
    if (NPW > 2) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 2 ? 2 : 0))                                  \
    if (NPW > 3) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 3 ? 3 : 0))                                  \
    if (NPW > 4) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 4 ? 4 : 0))                                  \
    if (NPW > 5) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 5 ? 5 : 0))                                  \
    if (NPW > 6) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 6 ? 6 : 0))                                  \
    if (NPW > 7) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 7 ? 7 : 0))                                  \
    if (NPW > 8) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 8 ? 8 : 0))                                  \
    if (NPW > 9) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 9 ? 9 : 0))                                  \
    if (NPW > 10) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 10 ? 10 : 0))                               \
    if (NPW > 11) GPSA_WSTAGE_PIECE(Q, BUF, (NPW > 11 ? 11 : 0))                               \
  }
  static_assert(NPW <= 12, "pieces per wave and stage");
#define GPSA_WLOADB(DST, KC, T)                                                                \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                           \
    const int row = 16 * (KC) + 4 * ks + kq;                                                   \
    DST[ks] = (okc && row < M) ? (T)xcol[(long long)row * C] : (T)0;                           \
  }
  GPSA_WSTAGE(0, 0)

  double xb[STREAM ? 1 : MB][4];  // resident slab (!STREAM)
  TI xr[2][4];                    // STREAM: raw values of this chunk and the next
  if constexpr (STREAM) {
    GPSA_WLOADB(xr[0], 0, TI)
  } else {
#pragma unroll
    for (int kc = 0; kc < MB; ++kc) GPSA_WLOADB(xb[kc], kc, double)
  }
  f64x4 acc[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) acc[rt] = (f64x4){0.0, 0.0, 0.0, 0.0};

#pragma unroll
  for (int kc = 0; kc < MB; ++kc) {
    // chunk kc has landed (this wave's share; the barrier publishes everyone's) and every wave is
    // done with chunk kc-1, whose slot the next stage overwrites
    if constexpr (STREAM) {
      // the wait also hands this chunk's right-hand-side registers to the compiler as ready, so it
      // neither waits again behind the loads issued below nor widens fp32 values before they land
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(xr[kc & 1][0]), "+v"(xr[kc & 1][1]), "+v"(xr[kc & 1][2]), "+v"(xr[kc & 1][3])
                   :
                   : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (kc + 1 < MB) {
      if constexpr (STREAM) GPSA_WLOADB(xr[(kc + 1) & 1], kc + 1, TI)
    }
    const double* base = &lds[kc & 1][lane];
    constexpr int PPK = (NPW + 3) / 4;  // the next stage's pieces are issued PPK at a time between the four K steps
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (kc + 1 < MB) {
#pragma unroll
        for (int u = 0; u < PPK; ++u) {
          if (ks * PPK + u < NPW) {
            switch (ks * PPK + u) {  // (compile time after unrolling: the immediate offset must be a constant)
              case 0: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 0) break;
              case 1: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 1) break;
              case 2: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 2) break;
              case 3: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 3) break;
              case 4: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 4) break;
              case 5: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 5) break;
              case 6: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 6) break;
              case 7: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 7) break;
              case 8: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 8) break;
              case 9: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 9) break;
              case 10: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 10) break;
              default: GPSA_WSTAGE_PIECE(kc + 1, (kc + 1) & 1, 11) break;
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const double b = STREAM ? (double)xr[kc & 1][ks] : xb[STREAM ? 0 : kc][ks];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
        acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(base[GPSA_WELEM(ks, rt0 + rt)], b, acc[rt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef GPSA_WSTAGE
#undef GPSA_WSTAGE_PIECE

  // the accumulator row of register r is the row the lane held as B operand (4 r + kq within the
  // tile), so q = k^T alpha closes lane-locally
  double s = 0.0;
  const double axd = (ax.X2 != nullptr && okc) ? (double)ax.s * (double)ax.d[c] : 0.0;
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    double kb[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (STREAM) {
      if (q0 != nullptr) GPSA_WLOADB(kb, rt, double)
    } else if constexpr (RS) {
      // the resident slab is indexed by the GLOBAL tile: rt0 is wave-uniform, the select below is a uniform branch
#pragma unroll
      for (int r = 0; r < 4; ++r) kb[r] = (rt0 == 0) ? xb[rt][r] : xb[MB - NRT + rt][r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) kb[r] = xb[STREAM ? 0 : rt][r];
    }
    const bool counted = !(RS && rt0 != 0 && rt < OVER);  // the overlapping tile counts once in q
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (rt0 + rt) * 16 + 4 * r + kq;
      double y = acc[rt][r];
      if (counted) s += y * kb[r];
      if (okc && row < M) {
        if (ax.X2 != nullptr) y += axd * (double)ax.X2[(long long)row * C + c];  // uniform branch
        alpha[(long long)row * C + c] = (TO)y;
        if (ax.out32 != nullptr) ax.out32[(long long)row * C + c] = (float)y;  // uniform branch
      }
    }
  }
#undef GPSA_WLOADB
  if (q0 != nullptr) {
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if constexpr (RS) {  // the two row halves of a column meet through LDS (the ring is idle by now)
      __syncthreads();
      double* qx = &lds[0][0];
      if ((w >> 1) == 1 && kq == 0) qx[(w & 1) * 16 + j] = s;
      __syncthreads();
      if ((w >> 1) == 0 && kq == 0 && okc) q[c] = s + qx[(w & 1) * 16 + j];
    } else {
      if (kq == 0 && okc) q[c] = s;
    }
  }
}

static inline int whiten_mb_for(int M) {
  const int mb = (M + 15) / 16;
  if (mb <= 2) return 2;
  if (mb <= 4) return 4;
  if (mb <= 7) return 7;
  if (mb <= 13) return 13;
  if (mb <= 16) return 16;
  if (mb <= 24) return 24;  // M <= 384: streamed right-hand side only (the accumulators alone are 8 MB registers;
  return 0;                 // 32 row tiles do not fit the register file next to the operands)
}

template <typename TI, typename TO>
static int whiten_launch(int MB, const double* Apk, const TI* X, int M, long long C, TO* alpha,
                         double* q, hipStream_t st, int batch = 1, long long sX = 0,
                         WhitenAxpy ax = WhitenAxpy{nullptr, nullptr, 0.f}, bool q_zeroed = false) {
  if constexpr (std::is_same<TO, double>::value) {  // long panels: the persistent output-stationary kernel (proj64.hip)
    if (ax.X2 == nullptr && proj64_ok(MB, C, batch))
      return proj64_launch<TI>(MB, Apk, X, M, C, alpha, ax.out32, q, batch, sX, st, q_zeroed);
  }
  const dim3 grid((unsigned)cdiv(C, 64), (unsigned)batch);
  bool stream = q == nullptr || (long long)grid.x * batch > num_cus();
  // at most one workgroup per CU and no second pass needed for q: split the rows over wave pairs (RS)
  static const bool rs_off = [] { const char* e = getenv("GPSA_WHITEN_RS"); return e && e[0] == '0'; }();
  // (only while the doubled grid still leaves every workgroup a CU of its own: measured at a 1/8 shard, 80
  //  workgroups 28 -> 19 us, but 391 instead of 196 workgroups 34 -> 44 us)
  const bool rs = !rs_off && cdiv(C, 32) * batch <= num_cus() && (MB == 13 || MB == 7 || MB == 16 || MB == 4) &&
                  ax.X2 == nullptr;
  if (rs) {
    const dim3 grid2((unsigned)cdiv(C, 32), (unsigned)batch);
#define GPSA_WRS(V)                                                                                    \
  case V:                                                                                              \
    whiten_mfma_kernel<V, TI, TO, false, true><<<grid2, 256, 0, st>>>(Apk, X, M, C, alpha, q, sX, ax); \
    GPSA_LAUNCH_CHECK();                                                                               \
    return 0;
    switch (MB) {
      GPSA_WRS(4) GPSA_WRS(7) GPSA_WRS(13) GPSA_WRS(16)
      default: break;
    }
#undef GPSA_WRS
  }
  if (const char* e = getenv("GPSA_WHITEN_STREAM")) stream = atoi(e) != 0;
  if (MB > 16) stream = true;
#define GPSA_WCASE(V)                                                                     \
  case V:                                                                                 \
    if (stream)                                                                           \
      whiten_mfma_kernel<V, TI, TO, true><<<grid, 256, 0, st>>>(Apk, X, M, C, alpha, q, sX, ax);  \
    else                                                                                      \
      whiten_mfma_kernel<V, TI, TO, false><<<grid, 256, 0, st>>>(Apk, X, M, C, alpha, q, sX, ax); \
    break;
#define GPSA_WCASE_STREAM(V)                                                              \
  case V:                                                                                 \
    whiten_mfma_kernel<V, TI, TO, true><<<grid, 256, 0, st>>>(Apk, X, M, C, alpha, q, sX, ax); \
    break;
  switch (MB) {
    GPSA_WCASE(2)
    GPSA_WCASE(4)
    GPSA_WCASE(7)
    GPSA_WCASE(13)
    GPSA_WCASE(16)
    GPSA_WCASE_STREAM(24)
    default:
      return GPSA_EUNSUPPORTED;
  }
#undef GPSA_WCASE
#undef GPSA_WCASE_STREAM
  GPSA_LAUNCH_CHECK();
  return 0;
}

// the pack launch of a call that the persistent kernel will take zeroes q for it
static inline double* pack_zeroes_q(int MB, long long C, int batch, double* q, bool fp64_out, bool axpy) {
  return (q != nullptr && fp64_out && !axpy && proj64_ok(MB, C, batch)) ? q : nullptr;
}

}  // namespace gpsa

extern "C" {

long long gpsa_whiten_workspace(int M) {
  const int MB = gpsa::whiten_mb_for(M);
  return MB ? (long long)MB * 16 * MB * 16 * 8 : 0;
}

int gpsa_whiten_f64(const double* Kinv, int in_dtype, const void* Kuf, int M, long long C,
                    int alpha_dtype, void* alpha, double* q, void* workspace, long long workspace_bytes,
                    void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || Kuf == nullptr || alpha == nullptr) return GPSA_EINVAL;
  if (alpha_dtype != GPSA_F32 && alpha_dtype != GPSA_F64) return GPSA_EINVAL;
  if (in_dtype != GPSA_F32 && in_dtype != GPSA_F64) return GPSA_EINVAL;
  const int MB = whiten_mb_for(M);
  if (MB == 0) return GPSA_EUNSUPPORTED;
  if (workspace_bytes < gpsa_whiten_workspace(M)) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* Apk = (double*)workspace;
  const long long tot = (long long)MB * 4 * MB * 64;
  double* qz = nullptr;
  if (Kinv != nullptr) {  // NULL: the workspace still holds the packed inverse of an earlier call
    qz = pack_zeroes_q(MB, C, 1, q, alpha_dtype == GPSA_F64, false);
    pack_whiten_kernel<<<(unsigned)cdiv(tot, 256), 256, 0, st>>>(Kinv, M, MB, Apk, 0, qz, C);
    GPSA_LAUNCH_CHECK();
  }
  const WhitenAxpy ax0{nullptr, nullptr, 0.f};
  if (in_dtype == GPSA_F64) {
    if (alpha_dtype == GPSA_F32)
      return whiten_launch<double, float>(MB, Apk, (const double*)Kuf, M, C, (float*)alpha, q, st);
    return whiten_launch<double, double>(MB, Apk, (const double*)Kuf, M, C, (double*)alpha, q, st, 1, 0, ax0, qz != nullptr);
  }
  if (alpha_dtype == GPSA_F32)
    return whiten_launch<float, float>(MB, Apk, (const float*)Kuf, M, C, (float*)alpha, q, st);
  return whiten_launch<float, double>(MB, Apk, (const float*)Kuf, M, C, (double*)alpha, q, st, 1, 0, ax0, qz != nullptr);
}

/* alpha = Kinv Kuf (fp64 panel) stored TWICE from the same accumulators: unrounded (alpha64) and rounded to fp32
 * (alpha32, what the matrix-core contractions read); q as gpsa_whiten_f64.  gpsa_step_desc.exact_inducing_grad. */
int gpsa_whiten_f64_dual(const double* Kinv, const double* Kuf, int M, long long C, double* alpha64, float* alpha32,
                         double* q, void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || Kuf == nullptr || alpha64 == nullptr || alpha32 == nullptr) return GPSA_EINVAL;
  const int MB = whiten_mb_for(M);
  if (MB == 0) return GPSA_EUNSUPPORTED;
  if (workspace_bytes < gpsa_whiten_workspace(M)) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* Apk = (double*)workspace;
  const long long tot = (long long)MB * 4 * MB * 64;
  double* qz = nullptr;
  if (Kinv != nullptr) {
    qz = pack_zeroes_q(MB, C, 1, q, true, false);
    pack_whiten_kernel<<<(unsigned)cdiv(tot, 256), 256, 0, st>>>(Kinv, M, MB, Apk, 0, qz, C);
    GPSA_LAUNCH_CHECK();
  }
  WhitenAxpy ax{nullptr, nullptr, 0.f};
  ax.out32 = alpha32;
  return whiten_launch<double, double>(MB, Apk, Kuf, M, C, alpha64, q, st, 1, 0, ax, qz != nullptr);
}

/* gpsa_whiten_f64_dual with the right-hand side FORMED inside the projection kernel instead of read:
 *   K_uf[m, c] = k(Z_m, x_c)   (kind: GPSA_K_*; Z [M,D] fp32, X64 [C,D] fp64, log-parameters fp32 - exactly what
 *   gpsa_kmat(GPSA_F64, GPSA_F32_X64, ...) evaluates, the same device function in the same precision)
 * so the covariance launch in front and its C*M doubles written and read back disappear (vgpsa.py:171-189: Kuf and
 * the product with Kuu^-1 in one pass).  GPSA_EUNSUPPORTED when the persistent kernel does not take the shape (short
 * panels, M beyond 208, D > 4) or GPSA_PROJ64_GEN=0: the caller then runs gpsa_kmat + gpsa_whiten_f64_dual. */
int gpsa_whiten_gen_f64_dual(const double* Kinv, int kind, const float* Z, const double* X64, int D, const float* ls_u,
                             const float* var_u, int M, long long C, double* alpha64, float* alpha32, double* q,
                             void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || !Z || !X64 || !ls_u || !var_u || alpha64 == nullptr || alpha32 == nullptr) return GPSA_EINVAL;
  static const bool off = [] { const char* e = getenv("GPSA_PROJ64_GEN"); return e && e[0] == '0'; }();
  const int MB = whiten_mb_for(M);
  if (off || MB == 0 || D < 1 || D > MAXD || !proj64_ok(MB, C, 1)) return GPSA_EUNSUPPORTED;
  if (kind != GPSA_K_RBF && kind != GPSA_K_MATERN12 && kind != GPSA_K_MATERN32) return GPSA_EINVAL;
  if (workspace_bytes < gpsa_whiten_workspace(M)) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* Apk = (double*)workspace;
  const long long tot = (long long)MB * 4 * MB * 64;
  double* qz = nullptr;
  if (Kinv != nullptr) {
    qz = pack_zeroes_q(MB, C, 1, q, true, false);
    pack_whiten_kernel<<<(unsigned)cdiv(tot, 256), 256, 0, st>>>(Kinv, M, MB, Apk, 0, qz, C);
    GPSA_LAUNCH_CHECK();
  }
  return proj64_gen_launch(MB, Apk, kind, Z, X64, D, ls_u, var_u, M, C, alpha64, alpha32, q, st, qz != nullptr);
}

/* gamma = Kinv X (fp32 panel, fp64 arithmetic) with the column-scaled update fused into the store:
 *   out[m,c] = (Kinv X)[m,c] + s * d[c] * X2[m,c]      (X, X2, out [M,C] fp32; d [C] fp32)
 * the data GP's dK_uf = K^-1 abar + 2 qbar o alpha (autograd of vgpsa.py:177-196) in one pass.
 * M as gpsa_whiten_f64; workspace >= gpsa_whiten_workspace(M). */
int gpsa_whiten_axpy_f32(const double* Kinv, const float* X, int M, long long C, const float* X2, const float* d,
                         double s, float* out, void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || !X || !X2 || !d || !out) return GPSA_EINVAL;
  const int MB = whiten_mb_for(M);
  if (MB == 0) return GPSA_EUNSUPPORTED;
  if (workspace_bytes < gpsa_whiten_workspace(M)) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* Apk = (double*)workspace;
  const long long tot = (long long)MB * 4 * MB * 64;
  if (Kinv != nullptr) {
    pack_whiten_kernel<<<(unsigned)cdiv(tot, 256), 256, 0, st>>>(Kinv, M, MB, Apk, 0, nullptr, 0);
    GPSA_LAUNCH_CHECK();
  }
  return whiten_launch<float, float>(MB, Apk, X, M, C, out, nullptr, st, 1, 0, WhitenAxpy{X2, d, (float)s});
}

/* batch of fp64 -> fp64 projections with one inverse each (the views' warp GPs): problem b reads
 * Kinv + b strideKinv and Kuf + b strideX ([M,C] blocks, strideX >= M*C), writes alpha + b strideX and
 * q + b C (q may be NULL).  workspace >= batch * gpsa_whiten_workspace(M). */
int gpsa_whiten_batched_f64(const double* Kinv, long long strideKinv, const double* Kuf, int M, long long C,
                            long long strideX, double* alpha, double* q, int batch, void* workspace,
                            long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || batch < 1 || Kuf == nullptr || alpha == nullptr) return GPSA_EINVAL;
  const int MB = whiten_mb_for(M);
  if (MB == 0) return GPSA_EUNSUPPORTED;
  if (workspace_bytes < gpsa_whiten_workspace(M) * batch) return GPSA_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* Apk = (double*)workspace;
  const long long tot = (long long)MB * 4 * MB * 64;
  dim3 pgrid((unsigned)cdiv(tot, 256), (unsigned)batch);
  double* qz = nullptr;
  if (Kinv != nullptr) {
    qz = pack_zeroes_q(MB, C, batch, q, true, false);
    pack_whiten_kernel<<<pgrid, 256, 0, st>>>(Kinv, M, MB, Apk, strideKinv, qz, (long long)batch * C);
    GPSA_LAUNCH_CHECK();
  }
  return whiten_launch<double, double>(MB, Apk, Kuf, M, C, alpha, q, st, batch, strideX, WhitenAxpy{nullptr, nullptr, 0.f},
                                       qz != nullptr);
}

}  // extern "C"
