// Shared device helpers for libgpsa_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpsa_hip.h"

#define GPSA_LAUNCH_CHECK()                      \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    if (e__ != hipSuccess) return (int)e__;      \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// "set once" flags (hipFuncSetAttribute: dynamic LDS beyond 64 KiB) are per DEVICE: a process that drives two GPUs must
// set the attribute on each (a per-process flag left the second device launching without it)
struct per_device_flag {
  bool done[64] = {};
  bool& here() {
    int d = 0;
    (void)hipGetDevice(&d);
    return done[d & 63];
  }
};

namespace gpsa {

constexpr int WAVE = 64;
constexpr int MAXD = 4;  // spatial dims supported by the fused covariance kernels

// Cross-lane moves on the DPP path (no LDS crossbar round trip as with ds_bpermute / __shfl_xor).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane_value(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double lane_value(double v, int l) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes of a wave, returned in every lane: quad butterflies (xor 1, xor 2), half-row and
// row mirrors give each row of 16 its total; the four row totals meet through scalar registers.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// Block-wide sum; result valid in thread 0.  `red` holds >= blockDim.x/64 elements of T.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  T r = T(0);
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}

template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }

// out[j] = sum_i part[i*stride + j]   (deterministic second pass of two-pass reductions).
// Launch with 256 threads: 64 columns x 4 row groups; each group keeps 4 independent partial sums so
// that a tall, narrow partial array (hundreds of rows) is not a chain of dependent loads.
template <typename TI, typename TO>
__global__ void reduce_rows_kernel(const TI* __restrict__ part, long long rows, long long stride,
                                   long long n, TO* __restrict__ out, double scale) {
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long long j = blockIdx.x * 64LL + lane;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (j < n) {
    long long i = grp;
    for (; i + 12 < rows; i += 16) {
      a0 += (double)part[i * stride + j];
      a1 += (double)part[(i + 4) * stride + j];
      a2 += (double)part[(i + 8) * stride + j];
      a3 += (double)part[(i + 12) * stride + j];
    }
    for (; i < rows; i += 4) a0 += (double)part[i * stride + j];
  }
  red[grp][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && j < n) out[j] = (TO)(((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) * scale);
}

// compute units of the current device (256 on MI355X); cached after the first call
inline int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

}  // namespace gpsa
