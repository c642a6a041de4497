// Shared helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/diffhandles_hip.h"

namespace dh {

void set_error(const std::string& s);

#define DH_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      dh::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
      return DH_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

#define DH_REQUIRE(cond, msg)                                                           \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      dh::set_error(std::string(__func__) + ": " + (msg));                              \
      return DH_ERR_ARG;                                                                \
    }                                                                                   \
  } while (0)

#define DH_LAUNCH_CHECK()                                                               \
  do {                                                                                  \
    hipError_t _e = hipGetLastError();                                                  \
    if (_e != hipSuccess) {                                                             \
      dh::set_error(std::string(__func__) + " launch: " + hipGetErrorString(_e));       \
      return DH_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Bump allocator over a caller-provided workspace.
struct Arena {
  char* base;
  size_t cap, off;
  Arena(void* p, size_t n) : base((char*)p), cap(n), off(0) {}
  template <class T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* r = (T*)(base + off);
    off += n * sizeof(T);
    return r;
  }
  bool ok() const { return off <= cap; }
};

// ---- 16-bit storage types ------------------------------------------------------------
typedef _Float16 f16;
typedef __bf16 bf16;

// m / d for 0 <= m < 2^22 and a quotient below a few hundred, from a float reciprocal of d (1 ulp): the product
// (m + 0.5) * inv is at least 0.5 / d away from an integer, far more than its rounding error, so truncation is exact.
// An integer division is ~30 dependent instructions and sits in front of the first load of every conv tile.
__device__ __forceinline__ int div_small(int m, float inv) { return (int)(((float)m + 0.5f) * inv); }
__device__ __forceinline__ float rcp_fast(int d) { return __builtin_amdgcn_rcpf((float)d); }

template <class T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<f16>(f16 x) { return (float)x; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 x) { return (float)x; }
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <class T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ f16 from_f32<f16>(float x) { return (f16)x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }

// ---- wave / block reductions (wave = 64 lanes) ---------------------------------------
template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <class T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_xor(v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}
// block sum with a fixed tree (deterministic); sm must hold blockDim.x/64 elements.
template <class T>
__device__ __forceinline__ T block_sum(T v, T* sm) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  T r = 0;
  for (int i = 0; i < nw; ++i) r += sm[i];
  return r;
}

}  // namespace dh
