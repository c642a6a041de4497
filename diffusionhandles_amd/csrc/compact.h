// Ordered (stable) stream compaction: indices of non-zero flags, ascending, batched.
#pragma once
#include "common.h"

namespace dh {

// ------------------------------------------------------------------ ordered compaction
constexpr int CP_THREADS = 256;
constexpr int CP_ITEMS = 8;
constexpr int CP_TILE = CP_THREADS * CP_ITEMS;

__device__ __forceinline__ int block_excl_scan_256(int v, int* sm, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < 4; ++i) {
    if (i < w) base += sm[i];
    tot += sm[i];
  }
  *total = tot;
  return base + inc - v;
}

static __global__ void k_cp_count(const uint8_t* flags, int n, int* block_counts, size_t flag_stride, int nb) {
  __shared__ int sm[4];
  const uint8_t* f = flags + (size_t)blockIdx.y * flag_stride;
  int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS, c = 0;
#pragma unroll
  for (int k = 0; k < CP_ITEMS; ++k) c += (base + k < n && f[base + k]) ? 1 : 0;
  int tot;
  block_excl_scan_256(c, sm, &tot);
  if (threadIdx.x == 0) block_counts[blockIdx.y * nb + blockIdx.x] = tot;
}

static __global__ void k_cp_scan(int* block_counts, int nb, int* totals, int total_stride) {
  if (threadIdx.x == 0) {
    int* bc = block_counts + blockIdx.x * nb;
    int run = 0;
    for (int i = 0; i < nb; ++i) {
      int c = bc[i];
      bc[i] = run;
      run += c;
    }
    totals[blockIdx.x * total_stride] = run;
  }
}

static __global__ void k_cp_scatter(const uint8_t* flags, int n, const int* block_offsets, int* out, size_t flag_stride,
                             size_t out_stride, int nb) {
  __shared__ int sm[4];
  const uint8_t* f = flags + (size_t)blockIdx.y * flag_stride;
  int* o = out + (size_t)blockIdx.y * out_stride;
  int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS, c = 0;
  bool fl[CP_ITEMS];
#pragma unroll
  for (int k = 0; k < CP_ITEMS; ++k) {
    fl[k] = (base + k < n) && f[base + k];
    c += fl[k] ? 1 : 0;
  }
  int tot;
  int pos = block_excl_scan_256(c, sm, &tot) + block_offsets[blockIdx.y * nb + blockIdx.x];
#pragma unroll
  for (int k = 0; k < CP_ITEMS; ++k)
    if (fl[k]) o[pos++] = base + k;
}

// out[e][0..count) = indices i with flags[e][i] != 0, ascending.  totals[e*total_stride] = count.
static int compact(const uint8_t* flags, int n, int batch, size_t flag_stride, int* out, size_t out_stride,
                   int* totals, int total_stride, int* block_counts, hipStream_t st) {
  int nb = cdiv(n, CP_TILE);
  if (nb == 0) nb = 1;
  hipLaunchKernelGGL(k_cp_count, dim3(nb, batch), dim3(CP_THREADS), 0, st, flags, n, block_counts, flag_stride, nb);
  hipLaunchKernelGGL(k_cp_scan, dim3(batch), dim3(64), 0, st, block_counts, nb, totals, total_stride);
  hipLaunchKernelGGL(k_cp_scatter, dim3(nb, batch), dim3(CP_THREADS), 0, st, flags, n, block_counts, out,
                     flag_stride, out_stride, nb);
  return nb;
}


}  // namespace dh
