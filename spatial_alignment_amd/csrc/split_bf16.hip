// EXPERIMENT - not on the product path, never timed as the headline (round-5 verdict item 9).
// The fp32 matrix instructions of gfx950 run at 1/16 of the bf16 rate.  Writing every fp32 operand as a sum of three
// bf16 pieces, a = a1 + a2 + a3 (8 + 8 + 8 significant bits, each piece the rounding of what the earlier ones left), the
// product a b is the nine a_i b_j; the six with i + j <= 4 carry everything above 2^-24 |a b|, each a_i b_j is EXACT in
// the fp32 accumulator's product stage, and the accumulation is the same fp32 one the fp32 instruction does.  Six
// v_mfma_f32_16x16x32_bf16 (16 cycles, K = 32) replace eight v_mfma_f32_16x16x4_f32 (32 cycles, K = 4) per 32 k:
// 96 cycles against 256.  Two entry points:
//   gpsa_experiment_split_bf16_product  W_l = Omega_l alpha (vgpsa.py:192-196's contraction) computed that way by a
//       plain kernel (operands from global memory, split in registers, no staging: numerics, not speed) - the
//       kernel-level parity table of tools/split_bf16_parity.py and bench.py's ``split_bf16_experiment``;
//   gpsa_experiment_split_bf16_rate     the MFMA + LDS-fragment-read loop of panel_elbo_kernel's tile (13 x 2
//       accumulators, B operand resident) in fp32 and in split form: the ceiling a split kernel's loop would have.
#include "common.hpp"

namespace gpsa {

typedef float sb_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sb_bf16x8 __attribute__((ext_vector_type(8)));

// x[0..7] -> up to three bf16 pieces (round to nearest even), x ~= p0 + p1 + p2
template <int NP>
__device__ __forceinline__ void sb_split(const float (&x)[8], sb_bf16x8 (&p)[3]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h0 = (__bf16)x[e];
    p[0][e] = h0;
    float r = x[e] - (float)h0;
    const __bf16 h1 = (__bf16)r;
    p[1][e] = NP > 1 ? h1 : (__bf16)0.f;
    r -= (float)h1;
    p[2][e] = NP > 2 ? (__bf16)r : (__bf16)0.f;
  }
}

// one wave = one 16 x 16 tile of W_l; workgroup = 4 column tiles; grid (column tiles / 4, row tiles, outputs)
// NPROD: 6 = three pieces, i + j <= 4; 4 = two pieces, all four products; 3 = two pieces without a2 b2; 1 = plain bf16;
//        0 = the fp32 instruction (v_mfma_f32_16x16x4_f32) on the same tiling, for comparison
template <int NPROD>
__global__ void __launch_bounds__(256) split_product_kernel(const float* __restrict__ Om, const float* __restrict__ X,
                                                            int M, long long C, float* __restrict__ W) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int i0 = blockIdx.y * 16, l = blockIdx.z;
  const long long c0 = ((long long)blockIdx.x * 4 + w) * 16;
  if (c0 >= C) return;  // (wave-uniform)
  const float* A = Om + (long long)l * M * M;
  const int row = i0 + i;
  const long long col = c0 + i;  // (the B / D operands index columns by the same lane & 15)
  sb_f32x4 acc = (sb_f32x4){0.f, 0.f, 0.f, 0.f};
  if (NPROD == 0) {
    for (int k0 = 0; k0 < M; k0 += 4) {
      const int k = k0 + kq;
      const float a = (row < M && k < M) ? A[(long long)row * M + k] : 0.f;
      const float b = (k < M && col < C) ? X[(long long)k * C + col] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
  } else {
    constexpr int NP = NPROD == 6 ? 3 : (NPROD == 1 ? 1 : 2);
    for (int k0 = 0; k0 < M; k0 += 32) {
      float av[8], bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + 8 * kq + e;
        av[e] = (row < M && k < M) ? A[(long long)row * M + k] : 0.f;
        bv[e] = (k < M && col < C) ? X[(long long)k * C + col] : 0.f;
      }
      sb_bf16x8 ap[3], bp[3];
      sb_split<NP>(av, ap);
      sb_split<NP>(bv, bp);
      // smallest products first
      if (NPROD == 6) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[2], bp[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[2], acc, 0, 0, 0);
      }
      if (NPROD == 4) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[1], acc, 0, 0, 0);
      if (NPROD != 1) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[1], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[0], acc, 0, 0, 0);
    }
  }
  // D layout: lane (column lane & 15, row group lane >> 4) holds rows 4 (lane >> 4) .. + 3
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int orow = i0 + 4 * kq + r;
    if (orow < M && col < C) W[((long long)l * M + orow) * C + col] = acc[r];
  }
}

// ---- loop-level rate (see the header comment): LDS holds a fixed image, nothing is refilled
constexpr int SB_MB = 13, SB_NCT = 2;

__global__ void __launch_bounds__(256, 1) split_rate_f32_kernel(float* out, int outputs, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[SB_MB * 256];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < SB_MB * 256; i += 256) lds[i] = seed * (float)(i % 7);
  __syncthreads();
  float xb[SB_NCT][SB_MB][4];
#pragma unroll
  for (int ct = 0; ct < SB_NCT; ++ct)
#pragma unroll
    for (int kc = 0; kc < SB_MB; ++kc)
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[ct][kc][r] = seed + (float)(ct + kc + r + lane);
  sb_f32x4 acc[SB_MB][SB_NCT];
  float keep = 0.f;
  for (int l = 0; l < outputs; ++l) {
#pragma unroll
    for (int kc = 0; kc < SB_MB; ++kc)
#pragma unroll
      for (int rt = 0; rt < SB_MB; ++rt) {
        const float4 a4 = *reinterpret_cast<const float4*>(&lds[rt * 256 + lane * 4]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ct = 0; ct < SB_NCT; ++ct)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                av[r], xb[ct][kc][r], (kc == 0 && r == 0) ? (sb_f32x4){0.f, 0.f, 0.f, 0.f} : acc[rt][ct], 0, 0, 0);
      }
#pragma unroll
    for (int rt = 0; rt < SB_MB; ++rt)
#pragma unroll
      for (int ct = 0; ct < SB_NCT; ++ct) keep += acc[rt][ct][0] + acc[rt][ct][3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = keep;
}

template <int NPROD>
__global__ void __launch_bounds__(256, 1) split_rate_kernel(float* out, int outputs, float seed) {
  constexpr int NP = NPROD == 6 ? 3 : 2, KB = 7;  // K = 224 = 7 blocks of 32 (M = 200 padded)
  __shared__ __attribute__((aligned(16))) unsigned short lds[SB_MB * NP * 512];  // [rt][piece][64 lanes x 8 bf16]
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < SB_MB * NP * 512; i += 256) lds[i] = (unsigned short)(0x3f80 + (i % 5));
  __syncthreads();
  sb_bf16x8 xb[NP][SB_NCT][KB];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int ct = 0; ct < SB_NCT; ++ct)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) xb[p][ct][kb][e] = (__bf16)(seed + (float)(p + ct + kb + e + (lane & 3)));
  sb_f32x4 acc[SB_MB][SB_NCT];
  float keep = 0.f;
  for (int l = 0; l < outputs; ++l) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int rt = 0; rt < SB_MB; ++rt) {
        sb_bf16x8 av[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
          av[p] = *reinterpret_cast<const sb_bf16x8*>(&lds[(rt * NP + p) * 512 + lane * 8]);
#pragma unroll
        for (int ct = 0; ct < SB_NCT; ++ct) {
          sb_f32x4 c = (kb == 0) ? (sb_f32x4){0.f, 0.f, 0.f, 0.f} : acc[rt][ct];
          if (NPROD == 6) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[NP - 1], xb[0][ct][kb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], xb[1][ct][kb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], xb[NP - 1][ct][kb], c, 0, 0, 0);
          }
          if (NPROD == 4) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], xb[1][ct][kb], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], xb[0][ct][kb], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], xb[1][ct][kb], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], xb[0][ct][kb], c, 0, 0, 0);
          acc[rt][ct] = c;
        }
      }
#pragma unroll
    for (int rt = 0; rt < SB_MB; ++rt)
#pragma unroll
      for (int ct = 0; ct < SB_NCT; ++ct) keep += acc[rt][ct][0] + acc[rt][ct][3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = keep;
}

}  // namespace gpsa

extern "C" {

int gpsa_experiment_split_bf16_product(const float* Omega, const float* alpha, int M, long long C, int L, int nprod,
                                       float* W, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)cdiv(cdiv(C, 16), 4), (unsigned)cdiv(M, 16), (unsigned)L);
  switch (nprod) {
    case 0: split_product_kernel<0><<<grid, 256, 0, st>>>(Omega, alpha, M, C, W); break;
    case 1: split_product_kernel<1><<<grid, 256, 0, st>>>(Omega, alpha, M, C, W); break;
    case 3: split_product_kernel<3><<<grid, 256, 0, st>>>(Omega, alpha, M, C, W); break;
    case 4: split_product_kernel<4><<<grid, 256, 0, st>>>(Omega, alpha, M, C, W); break;
    case 6: split_product_kernel<6><<<grid, 256, 0, st>>>(Omega, alpha, M, C, W); break;
    default: return GPSA_EINVAL;
  }
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_experiment_split_bf16_rate(int nprod, int outputs, float* out, void* stream) {
  using namespace gpsa;
  if (outputs < 1 || out == nullptr) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (nprod) {
    case 0: split_rate_f32_kernel<<<256, 256, 0, st>>>(out, outputs, 1.f); break;
    case 3: split_rate_kernel<3><<<256, 256, 0, st>>>(out, outputs, 1.f); break;
    case 4: split_rate_kernel<4><<<256, 256, 0, st>>>(out, outputs, 1.f); break;
    case 6: split_rate_kernel<6><<<256, 256, 0, st>>>(out, outputs, 1.f); break;
    default: return GPSA_EINVAL;
  }
  GPSA_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
