// Micro-benchmark (round 5): what a CU can STORE per clock, by the shape of a wave's store instruction.
// Every wave of a 512-thread workgroup issues NI global_store_dwordx4 (64 lanes x 16 B = 1 KiB per instruction); the 1 KiB is laid
// out as rows of SEG contiguous bytes (SEG = 16 ... 1024), consecutive rows LD bytes apart (LD = 2560: a row of a [M][1280] fp16
// tensor) -- SEG = 64 is what a 16 x 16 MFMA block pair gives a GEMM epilogue (16 rows x 64 B), 32 what the 32 x 32 blocks of
// k_gemm_dma give, 256+ what a tile transposed through LDS could store.  Reports cycles per instruction per CU (s_memtime around
// the issue loop of wave 0, i.e. including the back-pressure of the store path) and bytes per clock per CU, for 32 ... 512 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_store tools/ubench_store.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int SEG>
__global__ void __launch_bounds__(512) k_store(unsigned char* out, size_t wg_stride, int ni, int ld, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16;                        // lanes per row
  constexpr int ROWS = 64 / LPR;                       // rows per instruction
  const int row = lane / LPR, col = (lane % LPR) * 16;
  // a workgroup owns a private slab; wave w writes rows [w * ni * ROWS, ...); instruction i rows i * ROWS + row
  unsigned char* base = out + (size_t)blockIdx.x * wg_stride + (size_t)(wave * ni * ROWS + row) * ld + col;
  uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int i = 0; i < ni; ++i) {
    *reinterpret_cast<uint4*>(base + (size_t)i * ROWS * ld) = v;
    v.x += 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t0; }
}

template <int SEG>
int run(unsigned char* out, size_t bytes, unsigned long long* cyc, int nwg, int ni, int ld) {
  constexpr int ROWS = 64 / (SEG / 16);
  const size_t wg_stride = (size_t)8 * ni * ROWS * ld;          // 8 waves
  if ((size_t)nwg * wg_stride > bytes) { printf("SEG %4d: buffer too small\n", SEG); return 0; }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k_store<SEG>), dim3(nwg), dim3(512), 0, 0, out, wg_stride, ni, ld, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_store<SEG>), dim3(nwg), dim3(512), 0, 0, out, wg_stride, ni, ld, cyc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[2];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  const double us = ms * 1e3 / reps, total = (double)nwg * 8 * ni * 1024;
  // s_memtime ticks at 100 MHz on this chip's constant clock: report microseconds of wave 0 instead of core cycles
  printf("SEG %4d B x %2d rows, ld %5d, %3d workgroups, %3d stores per wave: kernel %7.1f us  %6.2f TB/s  %6.1f B/clk/CU(busy, 2.4 GHz) | wave 0: issue %6llu ticks, issue + drain %6llu ticks\n",
         SEG, ROWS, ld, nwg, ni, us, total / us / 1e6, total / (nwg < 256 ? nwg : 256) / (us * 2400.0), h[0], h[1]);
  return 0;
}

int main() {
  const size_t bytes = (size_t)3 << 30;
  unsigned char* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, bytes));
  CK(hipMalloc(&cyc, 64));
  CK(hipMemset(out, 0, bytes));
  for (int nwg : {32, 256, 512}) {
    for (int ni : {12, 48}) {
      run<16>(out, bytes, cyc, nwg, ni, 2560);
      run<32>(out, bytes, cyc, nwg, ni, 2560);
      run<64>(out, bytes, cyc, nwg, ni, 2560);
      run<128>(out, bytes, cyc, nwg, ni, 2560);
      run<256>(out, bytes, cyc, nwg, ni, 2560);
      run<1024>(out, bytes, cyc, nwg, ni, 2560);
    }
  }
  // row stride = the segment (fully contiguous 1 KiB per instruction whatever SEG): separates "rows" from "bytes per request"
  run<64>(out, bytes, cyc, 256, 48, 64);
  run<1024>(out, bytes, cyc, 256, 48, 1024);
  return 0;
}
