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

// f32 cross-lane steps on the VALU instead of ds_bpermute (an LDS round trip of >100 cycles per step, six dependent
// ones per wave reduction): DPP quad permutes / mirrors inside a row of 16 lanes, v_permlane16_swap / v_permlane32_swap
// (gfx950) across rows and wave halves.  x_swap(v, v) returns {v with its odd rows (upper half) replaced by the even rows
// (lower half), v with the even rows replaced by the odd ones}: their sum is the pairwise total, identical in both lanes.
typedef unsigned lane_u2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141;
struct LanePair { float a, b; };
__device__ __forceinline__ LanePair half_pair(float v) {          // {value of the lower-half lane, of the upper-half lane} of (l, l ^ 32)
  const lane_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return {__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ LanePair row_pair(float v) {           // the same for (l, l ^ 16)
  const lane_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return {__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ float xor32_sum(float v) { const LanePair p = half_pair(v); return p.a + p.b; }
__device__ __forceinline__ float xor32_max(float v) { const LanePair p = half_pair(v); return fmaxf(p.a, p.b); }
// sum over aligned groups of 8 lanes, in every lane of the group
__device__ __forceinline__ float oct_sum(float v) {
  v += dpp_f32<DPP_QUAD_XOR1>(v);
  v += dpp_f32<DPP_QUAD_XOR2>(v);
  v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = oct_sum(v);
  v += dpp_f32<DPP_ROW_MIRROR>(v);
  { const LanePair p = row_pair(v); v = p.a + p.b; }
  return xor32_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<DPP_QUAD_XOR1>(v));
  v = fmaxf(v, dpp_f32<DPP_QUAD_XOR2>(v));
  v = fmaxf(v, dpp_f32<DPP_ROW_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_f32<DPP_ROW_MIRROR>(v));
  { const LanePair p = row_pair(v); v = fmaxf(p.a, p.b); }
  return xor32_max(v);
}
// lanes l and l ^ 32 each hold two 8-byte values (a, b); afterwards the lower lane holds {its a, the upper lane's a} and
// the upper lane {the lower lane's b, its b}: the 16-byte chunks of the wide epilogue stores
__device__ __forceinline__ uint4 half_exchange(uint2 a, uint2 b) {
  const lane_u2 x = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
  const lane_u2 y = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
  return make_uint4(x[0], y[0], x[1], y[1]);
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
