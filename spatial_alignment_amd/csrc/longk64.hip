// Long-K fp64 products with a small square result, on the fp64 matrix cores with LDS-DMA staging (round 5):
//     Out_p [M, M] (+)= alpha * sum_k A_p[:, k] B_p[:, k]^T ,      A_p = G_p + d_p o B_p   (either part may be absent)
// for p = 0 .. nprob-1, every operand an [M, K] row-major panel (K contiguous; K = the spots of a view / the S N
// columns of the data GP: 10^4 .. 10^5) and d_p a length-K vector.  The products of the step that have this shape -
//   * the warp GPs' backward:  dOmega_j = sum_c g_j[c] alpha_c alpha_c^T   (G absent, d = g_j, symmetric),
//                              dK_uu    = -(gamma + qbar o alpha) alpha^T  (vgpsa.py:177-191 through autograd),
//   * the exact inducing-point gradient of the data GP:  dK_uu = -(gamma + qbar o alpha64) alpha64^T  (fp64, C long)
// - ran through the generic 64 x 64 x 16 register-staged product at 0.29 - 0.36 of the fp64-MFMA peak (one barrier
// per 16 MFMAs per wave, the operand tiles staged through VGPRs and ds_write), after a pass of their own that wrote the
// scaled / summed left operand out.  Here:
//   * one workgroup (8 waves, two per SIMD) owns the WHOLE M x M result of one product for a slice of K: a K group
//     of 8 columns of the <= 16 MB rows of B (and of G) is read once from HBM and shared by all 8 waves through LDS
//     (arithmetic intensity M / 8 flop per byte instead of 4);
//   * the panels move global -> LDS by LDS-DMA, 16 bytes per lane: lane (i, kq) of a piece fetches X[16 t + i][k0 +
//     2 kq .. + 1] and reads the same 16 bytes back as its A / B fragment of TWO MFMAs (the MFMA's K slot kq stands
//     for column k0 + 2 kq + t in step t: any assignment works as long as both operands use it) - a lane-linear,
//     conflict-free ds_read_b128, no packing pass, no ds_write, no transposition;
//   * the left operand is formed in registers as the fragments are read: a = G + d o b (two fp64 FMAs per row tile and
//     K group), d riding through LDS with its stage (a compiler-visible load in the loop would drain the ring);
//   * three-slot ring, two stages in flight, one barrier per K group = per 2 T / 8 MFMAs of 64 cycles per wave (T = the
//     product's 16 x 16 tiles: MB^2, or MB (MB + 1) / 2 for a symmetric result - the lower triangle, mirrored by the
//     reduction);
//   * the T tiles are dealt to the 8 waves as contiguous row-major runs (22 / 21 tiles at MB = 13): 96 % balanced,
//     a wave's run shares its A fragment along a row;
//   * partial results [nprob][nsplit][MP, MP] leave once per workgroup; a second kernel adds them in a fixed order
//     (bitwise repeatable), applies alpha / beta and mirrors symmetric results.
#include <stdlib.h>
#include <string.h>

#include "internal.hpp"

namespace gpsa {

typedef double lk_f64x4 __attribute__((ext_vector_type(4)));
typedef double lk_f64x2 __attribute__((ext_vector_type(2)));
typedef float lk_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lk_lds_ptr;

constexpr int LK_WAVES = 8;
constexpr int LK_MAXP = 48;  // products per launch (16 views x (D + 1))

struct LongKArgs {
  const double* G[LK_MAXP];  // [M][ld] or unused (HASG == false)
  const double* B[LK_MAXP];  // [M][ld]
  const void* d[LK_MAXP];    // [K] float (DT == 1) or double (DT == 2); unused with DT == 0
  double* part;              // [nprob][nsplit][MP * MP]
  int M, nprob, nsplit, sym;
  long long K, ld, gper;     // gper: K stages (16 columns) per split
};

__device__ __forceinline__ void lk_set_m0(unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_base) : "memory");
}
template <int IMM>
__device__ __forceinline__ void lk_glds16(const void* gsrc_minus_imm) {
  asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(gsrc_minus_imm), "n"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ void lk_glds4(const void* gsrc_minus_imm) {
  asm volatile("global_load_lds_dword %0, off offset:%1" ::"v"(gsrc_minus_imm), "n"(IMM) : "memory");
}

// MB: row tiles (M <= 16 MB); HASG: the left operand has its own panel G; DT: type of the scale vector d (0: none)
//
// Round-5 notes on the shape of the loop (measured at the exact inducing-point gradient's product, M = 200, K = 10^5:
// the generic product 356 us; the first version of this kernel - K groups of 8 columns, three-slot ring, a test of the
// wave's tile count in front of every tile - 238 us = 0.43 of the fp64-MFMA peak):
//   * a stage is 16 K columns = BOTH 64-byte halves of every row's 128-byte line (a stage of 8 fetched half a line
//     and came back for the other half one stage later, through L2), two slots: one barrier per 4 T / 8 MFMAs a wave;
//   * every wave runs NT_MAX tiles (the last tile of a shorter run is repeated into an accumulator nobody stores: the
//     wave would have waited at the barrier anyway), so the tile loop has no test of the run's length, and the B
//     fragments of tile t + 1 are requested before the MFMAs of tile t.
template <int MB, bool HASG, int DT>
__global__ void __launch_bounds__(512) longk64_kernel(LongKArgs a) {
  static_assert(HASG || DT != 0, "a left operand");
  constexpr int MP = MB * 16;
  constexpr int NOPD = HASG ? 2 : 1;                         // operands staged
  constexpr int NPIECE = MB * NOPD * 2;                       // 1-KiB pieces of a stage: (operand, tile, half)
  constexpr int NPW = (NPIECE + LK_WAVES - 1) / LK_WAVES;     // LDS-DMA operations per wave and stage (uniform)
  static_assert(NPW <= 8, "two groups of four pieces per wave (an immediate offset reaches 4 KiB)");
  constexpr int DAREA = DT == 2 ? 1024 : 256;  // a wave's copy of the stage's 16 scale values (one LDS-DMA operation)
  constexpr int STAGE_B = NPW * LK_WAVES * 1024 + (DT ? LK_WAVES * DAREA : 0);  // bytes per ring slot
  constexpr int NT_MAX = (MB * MB + LK_WAVES - 1) / LK_WAVES;
  constexpr int NT_SYM = (MB * (MB + 1) / 2 + LK_WAVES - 1) / LK_WAVES;  // tiles per wave of a symmetric result
  extern __shared__ __attribute__((aligned(16))) char lk_smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int p = blockIdx.y, sp = blockIdx.x;
  const int M = a.M;
  const long long K = a.K, ld = a.ld;
  const double* __restrict__ Bp = a.B[p];
  const double* __restrict__ Gp = HASG ? a.G[p] : nullptr;
  const long long ng = (K + 15) >> 4;
  const long long g_lo = (long long)sp * a.gper;
  long long g_hi = g_lo + a.gper;
  if (g_hi > ng) g_hi = ng;
  const unsigned lds0 = (unsigned)(unsigned long long)(lk_lds_ptr)(&lk_smem[0]);

  // this wave's run of tiles: [t_lo, t_hi) of the row-major list (sym: the lower triangle, row by row)
  const int T = a.sym ? MB * (MB + 1) / 2 : MB * MB;
  const int t_lo = w * T / LK_WAVES, t_hi = (w + 1) * T / LK_WAVES;
  const int cnt = t_hi - t_lo;
  int i0 = 0, j0 = 0;
  if (a.sym) {
    while ((i0 + 1) * (i0 + 2) / 2 <= t_lo) ++i0;
    j0 = t_lo - i0 * (i0 + 1) / 2;
  } else {
    i0 = t_lo / MB;
    j0 = t_lo - i0 * MB;
  }
  if (cnt == 0) i0 = j0 = 0;  // (more waves than tiles: this one computes a tile into accumulators nobody stores)

  // row pointers of this wave's pieces: piece q = w + 8 u = ((operand * MB + tile) * 2 + half); surplus operations
  // re-fetch the last piece into a slot nobody reads.  Pointer = operand base + clamped row * ld + 8 half + 2 kq
  const double* rp[NPW];
  int koff[NPW];
#pragma unroll
  for (int u = 0; u < NPW; ++u) {
    int q = w + LK_WAVES * u;
    q = q < NPIECE ? q : NPIECE - 1;
    const int h = q & 1, ot = q >> 1;
    const bool isg = HASG && ot >= MB;
    const int t = isg ? ot - MB : ot;
    int row = t * 16 + li;
    row = row < M ? row : M - 1;
    rp[u] = (isg ? Gp : Bp) + (long long)row * ld;
    koff[u] = 8 * h + 2 * kq;
  }

  // stage g -> ring slot s
  auto issue = [&](long long g, int s) {
    const long long kb = g << 4;
#pragma unroll
    for (int u = 0; u < NPW; ++u) {
      if ((u & 3) == 0)
        lk_set_m0(__builtin_amdgcn_readfirstlane(lds0 + (unsigned)(s * STAGE_B + (w * NPW + u) * 1024)));
      long long k = kb + koff[u];
      k = k < K - 2 ? k : K - 2;  // beyond K: an in-bounds (finite) duplicate, masked below
      const char* src = reinterpret_cast<const char*>(rp[u] + k);
      switch (u & 3) {
        case 0: lk_glds16<0>(src); break;
        case 1: lk_glds16<1024>(src - 1024); break;
        case 2: lk_glds16<2048>(src - 2048); break;
        default: lk_glds16<3072>(src - 3072); break;
      }
    }
    if (DT != 0) {
      // the stage's 16 scale values, one copy per wave (no cross-wave dependency): lanes beyond the data re-fetch
      lk_set_m0(__builtin_amdgcn_readfirstlane(lds0 + (unsigned)(s * STAGE_B + NPW * LK_WAVES * 1024 + w * DAREA)));
      if (DT == 1) {
        long long k = kb + (lane & 15);
        k = k < K ? k : K - 1;
        lk_glds4<0>(reinterpret_cast<const float*>(a.d[p]) + k);
      } else {
        long long k = kb + 2 * (lane & 7);
        k = k < K - 1 ? k : K - 2;
        lk_glds16<0>(reinterpret_cast<const double*>(a.d[p]) + k);
      }
    }
  };

  lk_f64x4 acc[NT_MAX];
#pragma unroll
  for (int t = 0; t < NT_MAX; ++t) acc[t] = (lk_f64x4){0.0, 0.0, 0.0, 0.0};

  if (g_lo < g_hi) {
    issue(g_lo, 0);
    int s = 0;
    for (long long g = g_lo; g < g_hi; ++g) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage g has landed
      __syncthreads();  // ... everyone's; and every wave is done with the other slot (stage g - 1)
      if (g + 1 < g_hi) issue(g + 1, s ^ 1);
      const char* slot = lk_smem + s * STAGE_B;
      const char* dslot = slot + NPW * LK_WAVES * 1024 + w * DAREA;
      // this lane's K columns of the stage - half h, MFMA step t: g 16 + 8 h + 2 kq + t - and their scale values
      const long long k_a = (g << 4) + 2 * kq;
      double dv[2][2] = {{1.0, 1.0}, {1.0, 1.0}};
      bool ok[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ok[h][0] = k_a + 8 * h < K;
        ok[h][1] = k_a + 8 * h + 1 < K;
        if (DT == 1) {
          const lk_f32x2 x = *reinterpret_cast<const lk_f32x2*>(dslot + 4 * (8 * h + 2 * kq));
          dv[h][0] = (double)x.x;
          dv[h][1] = (double)x.y;
        } else if (DT == 2) {
          const lk_f64x2 x = *reinterpret_cast<const lk_f64x2*>(dslot + 8 * (8 * h + 2 * kq));
          dv[h][0] = x.x;
          dv[h][1] = x.y;
        }
      }
      // piece (operand O, tile TL, half H) of this slot, this lane's 16 bytes
#define LK_FRAG(O, TL, H)                                                                              \
  (*reinterpret_cast<const lk_f64x2*>(slot + ((((((O) * MB + (TL)) * 2 + (H)) & 7) * NPW +             \
                                                 ((((O) * MB + (TL)) * 2 + (H)) >> 3)) << 10) + lane * 16))
      int i = i0, j = j0, cur = -1;
      double af[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
      lk_f64x2 bn0 = LK_FRAG(0, j, 0), bn1 = LK_FRAG(0, j, 1);
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t) {
        if (t >= NT_SYM && a.sym) break;  // (a test only behind the symmetric run's length: wave-uniform)
        if (i != cur) {  // (wave-uniform: a run of ~T / 8 row-major tiles changes row two or three times)
          cur = i;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const lk_f64x2 bi = LK_FRAG(0, i, h);
            double x0, x1;
            if (HASG) {
              const lk_f64x2 gi = LK_FRAG(1, i, h);
              x0 = DT ? fma(dv[h][0], bi.x, gi.x) : gi.x;
              x1 = DT ? fma(dv[h][1], bi.y, gi.y) : gi.y;
            } else {
              x0 = dv[h][0] * bi.x;
              x1 = dv[h][1] * bi.y;
            }
            af[h][0] = ok[h][0] ? x0 : 0.0;
            af[h][1] = ok[h][1] ? x1 : 0.0;
          }
        }
        const lk_f64x2 b0 = bn0, b1 = bn1;
        // the next tile of the run (a run shorter than NT_MAX repeats its last tile into a spare accumulator)
        if (t + 1 < cnt) {
          ++j;
          if (a.sym ? j > i : j == MB) {
            ++i;
            j = 0;
          }
        }
        if (t + 1 < NT_MAX) {
          bn0 = LK_FRAG(0, j, 0);
          bn1 = LK_FRAG(0, j, 1);
        }
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][0], b0.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][1], b0.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1][0], b1.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1][1], b1.y, acc[t], 0, 0, 0);
      }
#undef LK_FRAG
      s ^= 1;
    }
  }
  // partial result: D layout of the fp64 MFMA - column = lane & 15, rows (lane >> 4) + 4 r
  double* __restrict__ out = a.part + ((long long)p * a.nsplit + sp) * MP * MP;
  int i = i0, j = j0;
#pragma unroll
  for (int t = 0; t < NT_MAX; ++t) {
    if (t < cnt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long long)(i * 16 + kq + 4 * r) * MP + j * 16 + li] = acc[t][r];
      ++j;
      if (a.sym ? j > i : j == MB) {
        ++i;
        j = 0;
      }
    }
  }
}

struct LongKOut {
  double* out[LK_MAXP];
  double alpha[LK_MAXP], beta[LK_MAXP];
};

// out_p[r][c] = beta out_p[r][c] + alpha sum_s part[p][s][r][c]; sym: only the lower triangle was computed - the
// threads of the upper triangle leave, the others write their sum to both halves (reading the mirrored partials
// instead was a stride-MP read per split: 77 us for four 200 x 200 results of 64 splits)
__global__ void __launch_bounds__(256)
longk64_reduce_kernel(const double* __restrict__ part, int M, int MP, int nsplit, int sym, LongKOut o) {
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int p = blockIdx.y;
  const long long e = blockIdx.x * 64LL + lane;
  const int r = (int)(e / M), c = (int)(e % M);
  const bool ok = e < (long long)M * M && !(sym && c > r);
  const long long mm = (long long)MP * MP;
  const double* q = part + (long long)p * nsplit * mm + (long long)r * MP + c;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (ok) {
    int s = grp;
    for (; s + 12 < nsplit; s += 16) {
      a0 += q[s * mm];
      a1 += q[(s + 4) * mm];
      a2 += q[(s + 8) * mm];
      a3 += q[(s + 12) * mm];
    }
    for (; s < nsplit; s += 4) a0 += q[s * mm];
  }
  red[grp][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && ok) {
    const double sum = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const double al = o.alpha[p], be = o.beta[p];
    double* dst = o.out[p] + (long long)r * M + c;
    *dst = (be == 0.0) ? al * sum : al * sum + be * (*dst);
    if (sym && c < r) {
      double* dm = o.out[p] + (long long)c * M + r;
      *dm = (be == 0.0) ? al * sum : al * sum + be * (*dm);
    }
  }
}

static inline int longk_mb_for(int M) {
  const int mb = (M + 15) / 16;
  if (mb <= 2) return 2;
  if (mb <= 4) return 4;
  if (mb <= 7) return 7;
  if (mb <= 13) return 13;
  if (mb <= 16) return 16;
  return 0;
}

// splits of K per product: fill the chip, but leave every workgroup enough K groups to amortise its ring fill and
// its MP x MP partial
static inline int longk_nsplit(long long K, int nprob) {
  const long long ng = (K + 15) / 16;
  long long s = (long long)num_cus() / (nprob > 0 ? nprob : 1);
  if (s > ng / 4) s = ng / 4;
  return (int)(s < 1 ? 1 : s);
}

template <int MB, bool HASG, int DT>
static int longk64_launch_mb(const LongKArgs& a, hipStream_t st) {
  constexpr int NPIECE = MB * (HASG ? 2 : 1) * 2, NPW = (NPIECE + LK_WAVES - 1) / LK_WAVES;
  constexpr int lds = 2 * (NPW * LK_WAVES * 1024 + (DT ? LK_WAVES * (DT == 2 ? 1024 : 256) : 0));
  static per_device_flag attr_flag;
  bool& attr_set = attr_flag.here();
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&longk64_kernel<MB, HASG, DT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return GPSA_EUNSUPPORTED;
    attr_set = true;
  }
  dim3 grid((unsigned)a.nsplit, (unsigned)a.nprob);
  longk64_kernel<MB, HASG, DT><<<grid, 512, lds, st>>>(a);
  GPSA_LAUNCH_CHECK();
  return 0;
}

template <bool HASG, int DT>
static int longk64_launch(int MB, const LongKArgs& a, hipStream_t st) {
  switch (MB) {
    case 2: return longk64_launch_mb<2, HASG, DT>(a, st);
    case 4: return longk64_launch_mb<4, HASG, DT>(a, st);
    case 7: return longk64_launch_mb<7, HASG, DT>(a, st);
    case 13: return longk64_launch_mb<13, HASG, DT>(a, st);
    case 16: return longk64_launch_mb<16, HASG, DT>(a, st);
    default: return GPSA_EUNSUPPORTED;
  }
}

}  // namespace gpsa

extern "C" {

/* bytes of workspace gpsa_longk_f64 needs for nprob products (0: the shape is not covered - the caller keeps its
 * generic path): M <= 256, K even and >= 512, 16-byte aligned panels (checked at the call) */
long long gpsa_longk_f64_workspace(int M, long long K, int nprob) {
  using namespace gpsa;
  const int MB = longk_mb_for(M);
  static const bool off = [] { const char* e = getenv("GPSA_LONGK"); return e && e[0] == '0'; }();
  if (off || MB == 0 || nprob < 1 || nprob > LK_MAXP || K < 512 || (K & 1)) return 0;
  const int ns = longk_nsplit(K, nprob);
  if ((long long)ns * nprob < 48) return 0;  // too few workgroups to be worth a 512-thread launch
  return (long long)nprob * ns * MB * 16 * MB * 16 * 8;
}

int gpsa_longk_f64(int nprob, const double* const* G, const double* const* B, const void* const* d, int d_dtype,
                   int M, long long K, long long ld, int sym, const double* alpha, const double* beta,
                   double* const* out, void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (nprob < 1 || nprob > LK_MAXP || !B || !out || !alpha || !beta || M < 1 || K < 2 || ld < K) return GPSA_EINVAL;
  if (d != nullptr && d_dtype != GPSA_F32 && d_dtype != GPSA_F64) return GPSA_EINVAL;
  if (G == nullptr && d == nullptr) return GPSA_EINVAL;
  const long long need = gpsa_longk_f64_workspace(M, K, nprob);
  if (need == 0 || (ld & 1)) return GPSA_EUNSUPPORTED;
  if (workspace_bytes < need) return GPSA_EWORKSPACE;
  const int MB = longk_mb_for(M);
  LongKArgs a;
  LongKOut o;
  memset(&a, 0, sizeof(a));
  memset(&o, 0, sizeof(o));
  for (int p = 0; p < nprob; ++p) {
    a.B[p] = B[p];
    a.G[p] = G ? G[p] : nullptr;
    a.d[p] = d ? d[p] : nullptr;
    if (!B[p] || !out[p] || (G && !G[p]) || (d && !d[p])) return GPSA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(B[p]) & 15) || (G && (reinterpret_cast<uintptr_t>(G[p]) & 15)) ||
        (d && d_dtype == GPSA_F64 && (reinterpret_cast<uintptr_t>(d[p]) & 15)))
      return GPSA_EUNSUPPORTED;
    o.out[p] = out[p];
    o.alpha[p] = alpha[p];
    o.beta[p] = beta[p];
  }
  a.part = reinterpret_cast<double*>(workspace);
  a.M = M;
  a.nprob = nprob;
  a.nsplit = longk_nsplit(K, nprob);
  a.sym = sym ? 1 : 0;
  a.K = K;
  a.ld = ld;
  a.gper = cdiv(cdiv(K, 16), a.nsplit);
  hipStream_t st = as_stream(stream);
  int rc;
  if (G && d && d_dtype == GPSA_F32) rc = longk64_launch<true, 1>(MB, a, st);
  else if (G && d) rc = longk64_launch<true, 2>(MB, a, st);
  else if (G) rc = longk64_launch<true, 0>(MB, a, st);
  else if (d_dtype == GPSA_F32) rc = longk64_launch<false, 1>(MB, a, st);
  else rc = longk64_launch<false, 2>(MB, a, st);
  if (rc) return rc;
  dim3 rg((unsigned)cdiv((long long)M * M, 64), (unsigned)nprob);
  longk64_reduce_kernel<<<rg, 256, 0, st>>>(a.part, M, MB * 16, a.nsplit, a.sym, o);
  GPSA_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
