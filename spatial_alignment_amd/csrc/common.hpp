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

namespace gpsa {

constexpr int WAVE = 64;
constexpr int MAXD = 4;  // spatial dims supported by the fused covariance kernels

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
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

// out[j] = sum_i part[i*stride + j]   (deterministic second pass of two-pass reductions)
template <typename TI, typename TO>
__global__ void reduce_rows_kernel(const TI* __restrict__ part, long long rows, long long stride,
                                   long long n, TO* __restrict__ out, double scale) {
  long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (j >= n) return;
  double acc = 0.0;
  for (long long i = 0; i < rows; ++i) acc += (double)part[i * stride + j];
  out[j] = (TO)(acc * scale);
}

}  // namespace gpsa
