// LMC likelihood without F_obs on the matrix cores (round 5; vgpsa.py:428-432 F_obs = F_latent W, :532-538 the Gaussian
// likelihood; the VALU kernel of round 4 stays in elementwise.hip as the fallback).
//
// For an LMC modality the reference forms F_obs[s,n,:] = F_latent[s,n,:] W ([S,N,L] x [L,P]) and the likelihood, its
// gradient and the two LMC gradient products each stream that [S,N,P] tensor again (400 MB each way at BASELINE config
// 3).  One pass over (F_latent, W, Y) does all of it - per tile of 16 spots, sample s and 16 outputs p:
//     fobs = F W                (MFMA: rows c = the tile's spots, columns p, contraction over the L latent outputs)
//     r = Y - fobs;  z^2 += (r / sigma)^2;  dfo = -r / (sigma^2 S)                (on the accumulator registers)
//     dW[l, p] += sum_c F[c, l] dfo[c, p]   (MFMA: the dfo registers ARE the B operand - its K slot kq stands for spot
//                                            4 kq + step, which is how the accumulator holds them; accumulators live
//                                            across all tiles of the workgroup)
//     dF[c, l]  = sum_p dfo[c, p] W[l, p]   (MFMA: contraction over p, which runs across lanes in the accumulator
//                                            layout - the 16 x 16 dfo tile is transposed through a wave-private LDS
//                                            tile: 4 writes + 4 reads per lane, no barrier)
// Round 4's kernel did the three products on the vector pipe: 0.42 of config 3's 5.3 ms.  Here 12 NLT MFMAs per
// (16 spots, 16 outputs); the observations Y of a spot tile are read ONCE and reused for all S samples (the VALU kernel
// walked the S N columns and fetched Y[c mod N] S times: 400 MB instead of 80).
// Everything at upstream gradient 1 (the caller scales; linear).  Deterministic: fixed tile -> workgroup map, dW from
// per-workgroup partials summed in block order, dF written by the tile's owner.
#include "internal.hpp"

namespace gpsa {

typedef float lm_f32x4 __attribute__((ext_vector_type(4)));

// NLT: 16-row tiles of the latent outputs (L <= 16 NLT); NPT: 16-column tiles of the outputs per wave and chunk (a chunk
// = 64 NPT outputs: W's columns are walked in chunks so that its fragments and the dW accumulators stay in registers)
// (two workgroups per CU: 193 - 223 registers; left to itself the compiler took 272 and the kernel - two barriers and an LDS
//  staging per 96 MFMAs of a wave - ran one wave per SIMD at half the MFMA rate)
template <int NLT, int NPT>
__global__ void __launch_bounds__(256, 2)
lmc_mfma_kernel(const float* __restrict__ F, const float* __restrict__ W, const float* __restrict__ Y,
                const float* __restrict__ noise_u, int S, long long N, int L, int P, double* __restrict__ zpart,
                int nparts, float* __restrict__ dF, float* __restrict__ dWpart) {
  constexpr int LT = 16 * NLT, KB = 4 * NLT, PC = 64 * NPT, WS = PC + 4, FS = LT + 1;
  extern __shared__ __attribute__((aligned(16))) float lm_smem[];
  float* sW = lm_smem;                   // [LT][WS]   W[l][p0 + pp], zero padded
  float* sF = sW + LT * WS;              // [16][FS]   the tile's draws F[c][l], zero padded
  float* sT = sF + 16 * FS;              // [4][16][17] a wave's dfo tile, for the transposition
  float* sR = sT + 4 * 16 * 17;          // [4][LT][17] the waves' partial dF^T tiles
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const double sN = exp((double)noise_u[0]) + 1e-5;  // "variance" used as std
  const float inv = (float)(1.0 / sN);
  const float coef = (float)(-1.0 / (sN * sN * (double)S));
  const long long ntiles = (N + 15) / 16;
  double z2 = 0.0;
  float* sTw = sT + w * 16 * 17;

  for (int p0 = 0; p0 < P; p0 += PC) {
    __syncthreads();  // (the previous chunk's readers of sW are done)
    for (int e = tid; e < LT * PC; e += 256) {
      const int l = e / PC, pp = e - l * PC;
      sW[l * WS + pp] = (l < L && p0 + pp < P) ? W[(long long)l * P + p0 + pp] : 0.f;
    }
    __syncthreads();
    const int pw = w * 16 * NPT;  // this wave's first output of the chunk
    float Wf[NPT][KB];            // B fragments of fobs = F W: W[4 ks + kq][p]
#pragma unroll
    for (int i = 0; i < NPT; ++i)
#pragma unroll
      for (int ks = 0; ks < KB; ++ks) Wf[i][ks] = sW[(4 * ks + kq) * WS + pw + 16 * i + li];
    lm_f32x4 dWacc[NPT][NLT];
#pragma unroll
    for (int i = 0; i < NPT; ++i)
#pragma unroll
      for (int lt = 0; lt < NLT; ++lt) dWacc[i][lt] = (lm_f32x4){0.f, 0.f, 0.f, 0.f};
    bool pok[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) pok[i] = p0 + pw + 16 * i + li < P;

    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const long long n0 = t * 16;
      // the tile's observations in the accumulator layout: column p = lane & 15 of output tile i, rows 4 kq + r
      float Yr[NPT][4];
      bool rok[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) rok[r] = n0 + 4 * kq + r < N;
#pragma unroll
      for (int i = 0; i < NPT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long row = rok[r] ? n0 + 4 * kq + r : N - 1;
          const int p = pok[i] ? p0 + pw + 16 * i + li : P - 1;
          Yr[i][r] = Y[row * P + p];
        }
      for (int s = 0; s < S; ++s) {
        const long long c0 = (long long)s * N + n0;
        for (int e = tid; e < 16 * LT; e += 256) {
          const int cc = e / LT, l = e - cc * LT;
          sF[cc * FS + l] = (n0 + cc < N && l < L) ? F[(c0 + cc) * L + l] : 0.f;
        }
        __syncthreads();  // (A) the tile's draws are staged
        float aF[KB], aT[4][NLT];
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) aF[ks] = sF[li * FS + 4 * ks + kq];        // A of fobs: F[c = li][l = 4 ks + kq]
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int lt = 0; lt < NLT; ++lt) aT[r][lt] = sF[(4 * kq + r) * FS + 16 * lt + li];  // A of dW: F[c = 4 kq + r][l]
        lm_f32x4 accF[NLT];
#pragma unroll
        for (int lt = 0; lt < NLT; ++lt) accF[lt] = (lm_f32x4){0.f, 0.f, 0.f, 0.f};
        float z2l = 0.f;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
          lm_f32x4 fo = (lm_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KB; ++ks) fo = __builtin_amdgcn_mfma_f32_16x16x4f32(aF[ks], Wf[i][ks], fo, 0, 0, 0);
          float dfo[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float rr = (rok[r] && pok[i]) ? Yr[i][r] - fo[r] : 0.f;
            const float z = rr * inv;
            z2l = fmaf(z, z, z2l);
            dfo[r] = coef * rr;
          }
          // dW[l, p] += sum_c F[c, l] dfo[c, p]: K step r contracts the spots 4 kq + r
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int lt = 0; lt < NLT; ++lt)
              dWacc[i][lt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[r][lt], dfo[r], dWacc[i][lt], 0, 0, 0);
          // dF^T[l, c] += sum_p W[l, p] dfo[c, p]: the dfo tile transposed through this wave's LDS tile (a wave's LDS
          // operations execute in order: no barrier between its writes and its reads)
#pragma unroll
          for (int r = 0; r < 4; ++r) sTw[(4 * kq + r) * 17 + li] = dfo[r];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const float bT = sTw[li * 17 + 4 * ks + kq];  // B[k = p = 4 ks + kq][n = c = li]
#pragma unroll
            for (int lt = 0; lt < NLT; ++lt) {
              const float aW = sW[(16 * lt + li) * WS + pw + 16 * i + 4 * ks + kq];  // A[m = l][k = p]
              accF[lt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aW, bT, accF[lt], 0, 0, 0);
            }
          }
        }
        z2 += (double)z2l;
        // the four waves' shares of dF^T[l = 16 lt + 4 kq + r][c = li] meet in LDS
        float* sRw = sR + w * LT * 17;
#pragma unroll
        for (int lt = 0; lt < NLT; ++lt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sRw[(16 * lt + 4 * kq + r) * 17 + li] = accF[lt][r];
        __syncthreads();  // (B) every wave is done with sF and has left its share
        for (int e = tid; e < 16 * LT; e += 256) {
          const int l = e >> 4, cc = e & 15;
          if (l < L && n0 + cc < N) {
            const float sum = (sR[l * 17 + cc] + sR[(LT + l) * 17 + cc]) + (sR[(2 * LT + l) * 17 + cc] + sR[(3 * LT + l) * 17 + cc]);
            const long long o = (c0 + cc) * L + l;
            dF[o] = p0 == 0 ? sum : dF[o] + sum;  // (this workgroup owns the tile in every chunk of p)
          }
        }
        // (the next sample's staging of sF may start: nobody reads sF before its barrier (A); sR is rewritten only
        //  after that barrier, when these sums are done)
      }
    }
    // this workgroup's share of dW for the chunk: rows l = 16 lt + 4 kq + r, column p = lane & 15 of tile i
#pragma unroll
    for (int i = 0; i < NPT; ++i)
#pragma unroll
      for (int lt = 0; lt < NLT; ++lt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int l = 16 * lt + 4 * kq + r, p = p0 + pw + 16 * i + li;
          if (l < L && p < P) dWpart[((long long)blockIdx.x * L + l) * P + p] = dWacc[i][lt][r];
        }
  }
  z2 = block_sum(z2, red);
  if (tid == 0) zpart[blockIdx.x] = z2;
  if (blockIdx.x == 0)
    for (int i = (int)gridDim.x + tid; i < nparts; i += 256) zpart[i] = 0.0;
}

static inline long long lmc_mfma_smem(int NLT, int NPT) {
  const int LT = 16 * NLT, PC = 64 * NPT;
  return (long long)(LT * (PC + 4) + 16 * (LT + 1) + 4 * 16 * 17 + 4 * LT * 17) * 4;
}

template <int NLT, int NPT>
static int lmc_mfma_launch_t(const float* F, const float* W, const float* Y, const float* noise_u, int S, long long N,
                             int L, int P, double* zpart, int nparts, float* dF, float* dWpart, int G, hipStream_t st) {
  const int sm = (int)lmc_mfma_smem(NLT, NPT);
  static per_device_flag attr_flag;
  bool& attr_set = attr_flag.here();
  if (!attr_set && sm > 65536) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lmc_mfma_kernel<NLT, NPT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, sm) != hipSuccess)
      return GPSA_EUNSUPPORTED;
    attr_set = true;
  }
  lmc_mfma_kernel<NLT, NPT><<<G, 256, (size_t)sm, st>>>(F, W, Y, noise_u, S, N, L, P, zpart, nparts, dF, dWpart);
  GPSA_LAUNCH_CHECK();
  return 0;
}

// G workgroups (the caller's partial arrays have G rows); GPSA_EUNSUPPORTED: L > 64
int lmc_mfma_launch(const float* F, const float* W, const float* Y, const float* noise_u, int S, long long N, int L,
                    int P, double* zpart, int nparts, float* dF, float* dWpart, int G, hipStream_t st) {
  if (L <= 16) return lmc_mfma_launch_t<1, 8>(F, W, Y, noise_u, S, N, L, P, zpart, nparts, dF, dWpart, G, st);
  if (L <= 32) return lmc_mfma_launch_t<2, 4>(F, W, Y, noise_u, S, N, L, P, zpart, nparts, dF, dWpart, G, st);
  if (L <= 64) return lmc_mfma_launch_t<4, 2>(F, W, Y, noise_u, S, N, L, P, zpart, nparts, dF, dWpart, G, st);
  return GPSA_EUNSUPPORTED;
}

}  // namespace gpsa
