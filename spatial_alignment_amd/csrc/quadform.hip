// The dominant contraction of the GPSA step: the variational variance term
//     v[l,c] = alpha_c^T Omega_l alpha_c            (forward,  gpsa/models/vgpsa.py:192-196)
//     dalpha_c = 2 sum_l g[l,c] Omega_l alpha_c     (backward wrt alpha)
//     dOmega_l = sum_c g[l,c] alpha_c alpha_c^T     (backward wrt Omega)
// plus the whitening products Y = P X (vgpsa.py:177-180).  The reference materialises the
// [S,L,N,M] tensor a_t_Omega_tril; here nothing of that size ever reaches HBM.
//
// fp32, M <= 256:  MFMA path (v_mfma_f32_16x16x4_f32, exact fp32).  One wave owns a slab of 16*NCT
//   columns of alpha, kept in registers for the whole kernel as the MFMA B operand; the M x M left
//   operand streams through LDS in 16-deep K chunks (double buffered, register-staged prefetch).
//   The K index is permuted so that lane quarter q owns k = 16t+4q+r: the accumulator rows a lane
//   holds are then exactly the alpha rows it holds, and the quadratic form closes in registers
//   (2 shuffles per column tile, no LDS round trip).
// otherwise: generic tiled path built from gemm.hip + small fused elementwise kernels.
#include <stdlib.h>

#include "common.hpp"

namespace gpsa {

template <typename T>
int gemm_launch(int transA, int transB, int m, int n, long long k, double alpha, const T* A,
                long long lda, long long sA, const T* B, long long ldb, long long sB, double beta,
                T* C, long long ldc, long long sC, int batch, int splitk, void* ws,
                long long ws_bytes, hipStream_t st);

// ------------------------------------------------------------------------------------------------
// generic helpers
// ------------------------------------------------------------------------------------------------
// out[m,c] = X[m,c] * g[c]
template <typename T>
__global__ void colscale_kernel(const T* __restrict__ X, const T* __restrict__ g, int M, long long C,
                                T* __restrict__ out) {
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  if (c >= C) return;
  const T gv = g[c];
  for (int m = blockIdx.y; m < M; m += gridDim.y) out[(long long)m * C + c] = X[(long long)m * C + c] * gv;
}

// out[m,c] = Y[m,c] + s * d[c] * X[m,c]
template <typename T>
__global__ void col_axpy_kernel(const T* __restrict__ Y, const T* __restrict__ X,
                                const T* __restrict__ d, T s, int M, long long C,
                                T* __restrict__ out) {
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  if (c >= C) return;
  const T dv = s * d[c];
  for (int m = blockIdx.y; m < M; m += gridDim.y) {
    const long long o = (long long)m * C + c;
    out[o] = Y[o] + dv * X[o];
  }
}

// v[b,c] = sum_m X[m,c] * Tm[b,m,c]
template <typename T>
__global__ void coldot_kernel(const T* __restrict__ X, const T* __restrict__ Tm, int M, long long C,
                              T* __restrict__ v, long long vstride) {
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  if (c >= C) return;
  const T* t = Tm + (long long)blockIdx.y * M * C;
  T s = T(0);
  for (int m = 0; m < M; ++m) s += X[(long long)m * C + c] * t[(long long)m * C + c];
  v[(long long)blockIdx.y * vstride + c] = s;
}

template <typename T>
int generic_quadform_fwd(const T* alpha, const T* Omega, int M, long long C, int L, T* v, void* ws,
                         long long ws_bytes, hipStream_t st) {
  const long long per = (long long)M * C * (long long)sizeof(T);
  int lc = (int)(ws_bytes / per);
  if (lc < 1) return GPSA_EWORKSPACE;
  if (lc > L) lc = L;
  T* Tm = reinterpret_cast<T*>(ws);
  for (int l0 = 0; l0 < L; l0 += lc) {
    const int nb = (L - l0 < lc) ? L - l0 : lc;
    int rc = gemm_launch<T>(0, 0, M, (int)C, M, 1.0, Omega + (long long)l0 * M * M, M,
                            (long long)M * M, alpha, C, 0, 0.0, Tm, C, (long long)M * C, nb, 1,
                            nullptr, 0, st);
    if (rc) return rc;
    dim3 grid((unsigned)cdiv(C, 256), (unsigned)nb);
    coldot_kernel<T><<<grid, 256, 0, st>>>(alpha, Tm, M, C, v + (long long)l0 * C, C);
    GPSA_LAUNCH_CHECK();
  }
  return 0;
}

template <typename T>
int generic_quadform_bwd_alpha(const T* alpha, const T* Omega, const T* g, int M, long long C, int L,
                               T* dalpha, void* ws, long long ws_bytes, hipStream_t st) {
  if (ws_bytes < (long long)M * C * (long long)sizeof(T)) return GPSA_EWORKSPACE;
  T* tmp = reinterpret_cast<T*>(ws);
  for (int l = 0; l < L; ++l) {
    dim3 grid((unsigned)cdiv(C, 256), (unsigned)((M < 64) ? M : 64));
    colscale_kernel<T><<<grid, 256, 0, st>>>(alpha, g + (long long)l * C, M, C, tmp);
    GPSA_LAUNCH_CHECK();
    int rc = gemm_launch<T>(0, 0, M, (int)C, M, 2.0, Omega + (long long)l * M * M, M, 0, tmp, C, 0,
                            l == 0 ? 0.0 : 1.0, dalpha, C, 0, 1, 1, nullptr, 0, st);
    if (rc) return rc;
  }
  return 0;
}

static inline int gram_splitk(long long C) {
  long long s = C / 2048;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

template <typename T>
int generic_quadform_bwd_omega(const T* alpha, const T* g, int M, long long C, int L, T* dOmega,
                               void* ws, long long ws_bytes, hipStream_t st) {
  const int sk = gram_splitk(C);
  const long long tmp_b = (long long)M * C * (long long)sizeof(T);
  const long long part_b = (sk > 1) ? (long long)sk * M * M * (long long)sizeof(T) : 0;
  if (ws_bytes < tmp_b + part_b) return GPSA_EWORKSPACE;
  T* tmp = reinterpret_cast<T*>(ws);
  void* part = reinterpret_cast<char*>(ws) + tmp_b;
  for (int l = 0; l < L; ++l) {
    dim3 grid((unsigned)cdiv(C, 256), (unsigned)((M < 64) ? M : 64));
    colscale_kernel<T><<<grid, 256, 0, st>>>(alpha, g + (long long)l * C, M, C, tmp);
    GPSA_LAUNCH_CHECK();
    int rc = gemm_launch<T>(0, 1, M, M, C, 1.0, tmp, C, 0, alpha, C, 0, 0.0,
                            dOmega + (long long)l * M * M, M, 0, 1, sk, part, part_b, st);
    if (rc) return rc;
  }
  return 0;
}

template <typename T>
__global__ void colsq_kernel(const T* __restrict__ Y, int M, long long C, T* __restrict__ q) {
  const long long c = blockIdx.x * 256LL + threadIdx.x;
  if (c >= C) return;
  T s = T(0);
  for (int m = 0; m < M; ++m) {
    T y = Y[(long long)m * C + c];
    s += y * y;
  }
  q[c] = s;
}

// ------------------------------------------------------------------------------------------------
// MFMA panel kernels (fp32)
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MODE_QUAD = 0, MODE_ACCUM = 1, MODE_STORE = 2 };
constexpr int PK_LDS_STRIDE = 20;  // dwords per LDS row of a 16-deep K chunk (16 + 4 pad)

// src [L][M][M] (row-major) -> dst [L][MB][MP][16] fp32, zero padded; chunk kc holds columns
// 16kc..16kc+15 of every row, so that one K chunk is one contiguous block.
template <typename TS>
__global__ void pack_panels_kernel(const TS* __restrict__ src, int M, int MB, int L, int transpose,
                                   float* __restrict__ dst) {
  const int MP = MB * 16;
  const long long per = (long long)MP * MP;
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= per * L) return;
  const int l = (int)(idx / per);
  const long long e = idx % per;
  const int kc = (int)(e / (MP * 16));
  const int rem = (int)(e % (MP * 16));
  const int i = rem / 16, kk = rem % 16, k = kc * 16 + kk;
  float v = 0.f;
  if (i < M && k < M) {
    const TS* s = src + (long long)l * M * M;
    v = (float)(transpose ? s[(long long)k * M + i] : s[(long long)i * M + k]);
  }
  dst[idx] = v;
}

template <int MB, int NCT, int MODE>
__global__ void __launch_bounds__(256, (MB * NCT >= 24) ? 1 : 2)
panel_mfma_kernel(const float* __restrict__ Ppk,  // [L][MB][MP][16]
                  const float* __restrict__ X,    // [M][C]
                  const float* __restrict__ g,    // [L][C]   (ACCUM)
                  int M, long long C, int L,
                  float* __restrict__ out,        // QUAD: v [L][C]; ACCUM/STORE: Y [M][C]
                  float* __restrict__ colsq,      // STORE: optional [C]
                  float out_scale) {
  constexpr int MP = MB * 16;
  constexpr int CHUNK_F4 = MP * 4;                  // float4 per K chunk
  constexpr int NST = (CHUNK_F4 + 255) / 256;       // staging float4 per thread
  __shared__ __attribute__((aligned(16))) float lds[2][MP * PK_LDS_STRIDE];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const long long cw = (long long)blockIdx.x * (64 * NCT) + (long long)w * (16 * NCT);  // wave's first column

  // ---- B operand: this wave's slab of X, resident in registers -------------------------------
  float xb[NCT][MB][4];
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
  }

  f32x4 acc[MB][NCT];
#pragma unroll
  for (int rt = 0; rt < MB; ++rt)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 stage[NST];
#define GPSA_STAGE_LOAD(Q)                                                                  \
  {                                                                                         \
    const float4* src__ = reinterpret_cast<const float4*>(Ppk) + (long long)(Q) * CHUNK_F4; \
    _Pragma("unroll") for (int u = 0; u < NST; ++u) {                                       \
      const int f = tid + u * 256;                                                          \
      stage[u] = src__[f < CHUNK_F4 ? f : CHUNK_F4 - 1];                                    \
    }                                                                                       \
  }
#define GPSA_STAGE_WRITE(BUF)                                                               \
  {                                                                                         \
    _Pragma("unroll") for (int u = 0; u < NST; ++u) {                                       \
      const int f = tid + u * 256;                                                          \
      if (f < CHUNK_F4)                                                                     \
        *reinterpret_cast<float4*>(&lds[BUF][(f >> 2) * PK_LDS_STRIDE + (f & 3) * 4]) =     \
            stage[u];                                                                       \
    }                                                                                       \
  }

  const long long NQ = (long long)L * MB;
  GPSA_STAGE_LOAD(0)
  GPSA_STAGE_WRITE(0)
  __syncthreads();

  for (int l = 0; l < L; ++l) {
    float gv[NCT];
    if (MODE == MODE_ACCUM) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const long long c = cw + ct * 16 + j;
        gv[ct] = (c < C) ? g[(long long)l * C + c] : 0.f;
      }
    }
    if (MODE == MODE_QUAD) {
#pragma unroll
      for (int rt = 0; rt < MB; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kc = 0; kc < MB; ++kc) {
      const long long q = (long long)l * MB + kc;
      const int buf = (int)(q & 1);
      GPSA_STAGE_LOAD((q + 1 < NQ) ? q + 1 : q)  // last prefetch is a harmless re-read
      // B values of this chunk (scaled by g in ACCUM mode)
      float bv[NCT][4];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          bv[ct][r] = (MODE == MODE_ACCUM) ? xb[ct][kc][r] * gv[ct] : xb[ct][kc][r];
      const float* base = &lds[buf][(kq * 4)];
#pragma unroll
      for (int rt = 0; rt < MB; ++rt) {
        const float4 a4 = *reinterpret_cast<const float4*>(base + (rt * 16 + j) * PK_LDS_STRIDE);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[ct][r], acc[rt][ct], 0, 0, 0);
      }
      GPSA_STAGE_WRITE(buf ^ 1)
      __syncthreads();
    }
    if (MODE == MODE_QUAD) {
      // v[l,c] = sum over the rows this lane holds of acc * alpha, then across the 4 lane quarters
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        float s = 0.f;
#pragma unroll
        for (int rt = 0; rt < MB; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s += acc[rt][ct][r] * xb[ct][rt][r];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const long long c = cw + ct * 16 + j;
        if (kq == 0 && c < C) out[(long long)l * C + c] = s;
      }
    }
  }
  if (MODE != MODE_QUAD) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const long long c = cw + ct * 16 + j;
      float s = 0.f;
#pragma unroll
      for (int rt = 0; rt < MB; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + kq * 4 + r;
          const float y = acc[rt][ct][r] * out_scale;
          s += y * y;
          if (c < C && row < M) out[(long long)row * C + c] = y;
        }
      if (MODE == MODE_STORE && colsq != nullptr) {
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (kq == 0 && c < C) colsq[c] = s;
      }
    }
  }
}

static inline int mfma_mb_for(int M) {
  const int mb = (M + 15) / 16;
  if (mb <= 2) return 2;
  if (mb <= 4) return 4;
  if (mb <= 7) return 7;
  if (mb <= 13) return 13;
  if (mb <= 16) return 16;
  return 0;
}

static inline bool force_generic() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GPSA_FORCE_GENERIC");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

template <int MODE>
int panel_mfma_launch(int MBsel, const float* Ppk, const float* X, const float* g, int M,
                      long long C, int L, float* out, float* colsq, float scale, hipStream_t st) {
#define GPSA_PANEL_CASE(MBV, NCTV)                                                              \
  case MBV: {                                                                                   \
    const unsigned grid = (unsigned)cdiv(C, 64 * NCTV);                                         \
    panel_mfma_kernel<MBV, NCTV, MODE><<<grid, 256, 0, st>>>(Ppk, X, g, M, C, L, out, colsq, scale); \
  } break;
  switch (MBsel) {
    GPSA_PANEL_CASE(2, 4)
    GPSA_PANEL_CASE(4, 4)
    GPSA_PANEL_CASE(7, 4)
    GPSA_PANEL_CASE(13, 3)
    GPSA_PANEL_CASE(16, 2)
    default:
      return GPSA_EUNSUPPORTED;
  }
#undef GPSA_PANEL_CASE
  GPSA_LAUNCH_CHECK();
  return 0;
}

static int pack_f32(const float* src, int M, int MB, int L, int transpose, float* dst,
                    hipStream_t st) {
  const long long tot = (long long)L * MB * 16 * MB * 16;
  pack_panels_kernel<float><<<(unsigned)cdiv(tot, 256), 256, 0, st>>>(src, M, MB, L, transpose, dst);
  GPSA_LAUNCH_CHECK();
  return 0;
}

}  // namespace gpsa

extern "C" {

long long gpsa_quadform_workspace(int dtype, int M, long long C, int L) {
  const long long sz = (dtype == GPSA_F64) ? 8 : 4;
  const int MB = gpsa::mfma_mb_for(M);
  long long mfma = 0;
  if (dtype == GPSA_F32 && MB) mfma = (long long)L * MB * 16 * MB * 16 * 4;
  int lc = L < 4 ? L : 4;
  long long generic = (long long)M * C * sz * lc;                         // fwd: lc slabs of T
  long long bo = (long long)M * C * sz + (long long)gpsa::gram_splitk(C) * M * M * sz;  // bwd_omega
  long long r = generic > bo ? generic : bo;
  return (r > mfma ? r : mfma) + 256;
}

int gpsa_quadform_fwd(int dtype, const void* alpha, const void* Omega, int M, long long C, int L,
                      void* v, void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32) {
    const int MB = mfma_mb_for(M);
    if (MB && !force_generic()) {
      if (workspace_bytes < (long long)L * MB * 16 * MB * 16 * 4) return GPSA_EWORKSPACE;
      float* Ppk = (float*)workspace;
      int rc = pack_f32((const float*)Omega, M, MB, L, 0, Ppk, st);
      if (rc) return rc;
      return panel_mfma_launch<MODE_QUAD>(MB, Ppk, (const float*)alpha, nullptr, M, C, L, (float*)v,
                                          nullptr, 1.f, st);
    }
    return generic_quadform_fwd<float>((const float*)alpha, (const float*)Omega, M, C, L, (float*)v,
                                       workspace, workspace_bytes, st);
  }
  if (dtype == GPSA_F64)
    return generic_quadform_fwd<double>((const double*)alpha, (const double*)Omega, M, C, L,
                                        (double*)v, workspace, workspace_bytes, st);
  return GPSA_EINVAL;
}

int gpsa_quadform_bwd_alpha(int dtype, const void* alpha, const void* Omega, const void* g, int M,
                            long long C, int L, void* dalpha, void* workspace,
                            long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32) {
    const int MB = mfma_mb_for(M);
    if (MB && !force_generic()) {
      if (workspace_bytes < (long long)L * MB * 16 * MB * 16 * 4) return GPSA_EWORKSPACE;
      float* Ppk = (float*)workspace;
      int rc = pack_f32((const float*)Omega, M, MB, L, 0, Ppk, st);
      if (rc) return rc;
      return panel_mfma_launch<MODE_ACCUM>(MB, Ppk, (const float*)alpha, (const float*)g, M, C, L,
                                           (float*)dalpha, nullptr, 2.f, st);
    }
    return generic_quadform_bwd_alpha<float>((const float*)alpha, (const float*)Omega,
                                             (const float*)g, M, C, L, (float*)dalpha, workspace,
                                             workspace_bytes, st);
  }
  if (dtype == GPSA_F64)
    return generic_quadform_bwd_alpha<double>((const double*)alpha, (const double*)Omega,
                                              (const double*)g, M, C, L, (double*)dalpha, workspace,
                                              workspace_bytes, st);
  return GPSA_EINVAL;
}

int gpsa_quadform_bwd_omega(int dtype, const void* alpha, const void* g, int M, long long C, int L,
                            void* dOmega, void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1 || L < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32)
    return generic_quadform_bwd_omega<float>((const float*)alpha, (const float*)g, M, C, L,
                                             (float*)dOmega, workspace, workspace_bytes, st);
  if (dtype == GPSA_F64)
    return generic_quadform_bwd_omega<double>((const double*)alpha, (const double*)g, M, C, L,
                                              (double*)dOmega, workspace, workspace_bytes, st);
  return GPSA_EINVAL;
}

int gpsa_col_axpy(int dtype, const void* Y, const void* X, const void* d, double s, int M,
                  long long C, void* out, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)cdiv(C, 256), (unsigned)((M < 64) ? M : 64));
  if (dtype == GPSA_F32)
    col_axpy_kernel<float><<<grid, 256, 0, st>>>((const float*)Y, (const float*)X, (const float*)d,
                                                 (float)s, M, C, (float*)out);
  else if (dtype == GPSA_F64)
    col_axpy_kernel<double><<<grid, 256, 0, st>>>((const double*)Y, (const double*)X,
                                                  (const double*)d, s, M, C, (double*)out);
  else
    return GPSA_EINVAL;
  GPSA_LAUNCH_CHECK();
  return 0;
}

int gpsa_panel_mm(int dtype, const void* P, const void* X, int M, long long C, void* Y, void* colsq,
                  void* workspace, long long workspace_bytes, void* stream) {
  using namespace gpsa;
  if (M < 1 || C < 1) return GPSA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (dtype == GPSA_F32) {
    const int MB = mfma_mb_for(M);
    if (MB && !force_generic()) {
      if (workspace_bytes < (long long)MB * 16 * MB * 16 * 4) return GPSA_EWORKSPACE;
      float* Ppk = (float*)workspace;
      int rc = pack_f32((const float*)P, M, MB, 1, 0, Ppk, st);
      if (rc) return rc;
      return panel_mfma_launch<MODE_STORE>(MB, Ppk, (const float*)X, nullptr, M, C, 1, (float*)Y,
                                           (float*)colsq, 1.f, st);
    }
    int rc = gemm_launch<float>(0, 0, M, (int)C, M, 1.0, (const float*)P, M, 0, (const float*)X, C,
                                0, 0.0, (float*)Y, C, 0, 1, 1, nullptr, 0, st);
    if (rc) return rc;
    if (colsq) {
      colsq_kernel<float><<<(unsigned)cdiv(C, 256), 256, 0, st>>>((const float*)Y, M, C, (float*)colsq);
      GPSA_LAUNCH_CHECK();
    }
    return 0;
  }
  if (dtype == GPSA_F64) {
    int rc = gemm_launch<double>(0, 0, M, (int)C, M, 1.0, (const double*)P, M, 0, (const double*)X,
                                 C, 0, 0.0, (double*)Y, C, 0, 1, 1, nullptr, 0, st);
    if (rc) return rc;
    if (colsq) {
      colsq_kernel<double><<<(unsigned)cdiv(C, 256), 256, 0, st>>>((const double*)Y, M, C, (double*)colsq);
      GPSA_LAUNCH_CHECK();
    }
    return 0;
  }
  return GPSA_EINVAL;
}

}  // extern "C"
