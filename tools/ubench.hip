// Micro-benchmarks of per-iteration costs on one CU-resident workgroup (diagnostics).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k_barrier(int n, long long* cyc) {
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_barrier();
  if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
}
__global__ void __launch_bounds__(256) k_mfma(int n, long long* cyc, float* out) {
  v16f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  v8h x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 0.001f + i); y[i] = (_Float16)(i * 0.5f); }
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a3, 0, 0, 0);
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
template <int MODE>
__global__ void __launch_bounds__(256) k_ldsread(int n, long long* cyc, float* out) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[32768];
  for (int i = threadIdx.x; i < 8192; i += 256) ((float*)sm)[i] = i;
  __syncthreads();
  float acc = 0;
  const int lane = threadIdx.x & 63, ln = lane & 31, hi = lane >> 5;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      int pc, rs = 128;
      if (MODE == 0) pc = ((2 * kk + hi) ^ (ln & 7)) << 4;
      else if (MODE == 1) pc = ((2 * kk + hi) ^ ((ln >> 1) & 7)) << 4;
      else if (MODE == 2) { pc = (2 * kk + hi) << 4; rs = 144; }
      else pc = (2 * kk + hi) << 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 v = *reinterpret_cast<const uint4*>(sm + ((j * 32 + ln) * rs + pc + (i & 1) * 8192));
        acc += __uint_as_float(v.x ^ v.w);
      }
    }
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// stream global -> LDS with the asm DMA, 8 pieces per wave per iteration, 3 iterations in flight
__device__ __forceinline__ void dma16(const void* g, unsigned l) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}
__global__ void __launch_bounds__(256) k_dma(int n, const unsigned char* src, size_t stride, long long* cyc) {
  __shared__ __attribute__((aligned(1024))) unsigned char sm[4 * 32768];
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)sm);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char* base = src + (size_t)blockIdx.x * stride + (size_t)(wave * 8 + (lane >> 3)) * 4096 + (lane & 7) * 16;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    if (i >= 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned sb = __builtin_amdgcn_readfirstlane(lds0 + (i & 3) * 32768 + wave * 1024);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma16(base + (size_t)i * 128 + (size_t)j * 32 * 4096, sb + j * 4096);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
}

int main() {
  long long* cyc; float* out; unsigned char* src;
  hipMalloc(&cyc, 4096 * 8); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&src, (size_t)3 << 30);
  hipMemset(src, 1, (size_t)3 << 30);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  long long h[4096];
  auto report = [&](const char* name, int n, int blocks, float ms) {
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    fflush(stdout); printf("%-10s blocks %4d iters %6d: %8.1f us  %7.1f ns/iter  clock64/iter %.1f\n", name, blocks, n, ms * 1e3, ms * 1e6 / n, (double)h[0] / n);
  };
  for (int blocks : {1, 160, 256, 512}) {
    float ms; int n = 20000;
    k_barrier<<<blocks, 256>>>(n, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); k_barrier<<<blocks, 256>>>(n, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); report("barrier", n, blocks, ms);
    k_mfma<<<blocks, 256>>>(n, cyc, out); hipDeviceSynchronize();
    hipEventRecord(e0); k_mfma<<<blocks, 256>>>(n, cyc, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); report("mfma x4", n, blocks, ms);
    k_ldsread<0><<<blocks, 256>>>(n, cyc, out); hipDeviceSynchronize();
    hipEventRecord(e0); k_ldsread<0><<<blocks, 256>>>(n, cyc, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); report("lds r&7", n, blocks, ms);
    hipEventRecord(e0); k_ldsread<1><<<blocks, 256>>>(n, cyc, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); report("lds r>>1", n, blocks, ms);
    hipEventRecord(e0); k_ldsread<2><<<blocks, 256>>>(n, cyc, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); report("lds pad144", n, blocks, ms);
    hipEventRecord(e0); k_ldsread<3><<<blocks, 256>>>(n, cyc, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); report("lds linear", n, blocks, ms);
    n = 60;
    for (size_t stride : {(size_t)1 << 22, (size_t)1 << 15, (size_t)0}) {
      k_dma<<<blocks, 256>>>(n, src, stride, cyc); hipDeviceSynchronize();
      hipEventRecord(e0); for (int r = 0; r < 20; ++r) k_dma<<<blocks, 256>>>(n, src, stride, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      printf("stride %8zu: ", stride); report("dma 32KB", n * 20, blocks, ms);
    }
  }
  return 0;
}
