// Micro-benchmark (round 4): is the ~33 B/clk/CU a CU pulls through LDS-DMA (tools/ubench_ingest.hip) a limit of that path or of
// the CU's vector memory pipe?  Workgroups stream a shared L2-resident 4 MiB buffer (the GEMM's activation operand / a weight
// tile shared by the row tiles) in three ways, one workgroup per CU:
//   dma   4 waves: global_load_lds_dwordx4 ring, one barrier per 16 KiB stage (the k_gemm_dma staging path)
//   vgpr  N waves: global_load_dwordx4 into registers, DEPTH loads in flight per lane, nothing else (an MFMA operand fetched
//         straight into its fragment registers)
//   both  8 waves: waves 0-3 run the dma ring, waves 4-7 the register stream, concurrently (would the two paths add up?)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_ingest2 tools/ubench_ingest2.hip && tools/bin/ubench_ingest2
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int ST = 4;
// mode bit 0: waves 0-3 stream `per_wg` bytes through the LDS ring; bit 1: the other waves (or all, if bit 0 is clear) stream
// `per_wg` bytes into registers
template <int WAVES, int DEPTH>
__global__ void __launch_bounds__(64 * WAVES) k_stream(const unsigned char* src, size_t per_wg, size_t wrap, int mode, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[ST * 16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t msk = wrap - 1;
  const bool dma_wave = (mode & 1) && wave < 4;
  const bool reg_wave = (mode & 2) && (!(mode & 1) || wave >= 4);
  unsigned acc = 0;
  if (dma_wave) {
    constexpr int PPW = 4;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)ring);
    const int tiles = (int)(per_wg / 16384);
    const size_t base = ((size_t)blockIdx.x * 65536) & msk;
    auto issue = [&](int t) {
      const unsigned sbase = __builtin_amdgcn_readfirstlane(lds0 + (t % ST) * 16384 + wave * PPW * 1024);
#pragma unroll
      for (int q = 0; q < PPW; ++q) dma16(src + ((base + (size_t)t * 16384 + (size_t)(wave * PPW + q) * 1024 + lane * 16) & msk), sbase + q * 1024);
    };
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
      if (s < tiles) issue(s);
    for (int t = 0; t < tiles; ++t) {
      if (tiles - 1 - t >= ST - 2) wait_vmcnt<PPW * (ST - 2)>(); else wait_vmcnt<0>();
      // (named barrier of the four DMA waves only is not available: the register waves never enter a barrier, so the DMA waves
      //  use a workgroup barrier only when they are alone; with mode 3 they synchronise through their own vmcnt, which is what
      //  bounds the stream anyway)
      if (!(mode & 2)) __builtin_amdgcn_s_barrier();
      acc += ring[(t % ST) * 16384 + (threadIdx.x & 255) * 4];
      if (t + ST - 1 < tiles) issue(t + ST - 1);
    }
  }
  if (reg_wave) {
    const int nw = (mode & 1) ? WAVES - 4 : WAVES, w = (mode & 1) ? wave - 4 : wave;
    const size_t chunk = per_wg / nw;                         // bytes of this wave
    const size_t base = (((size_t)blockIdx.x * 65536) + (size_t)w * chunk) & msk;
    const int iters = (int)(chunk / (1024 * DEPTH));
    uint4 r[DEPTH];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        r[d] = *reinterpret_cast<const uint4*>(src + ((base + ((size_t)it * DEPTH + d) * 1024 + lane * 16) & msk));
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc ^= r[d].x ^ r[d].y ^ r[d].z ^ r[d].w;
    }
  }
  if (acc == 0xdeadbeefu) sink[0] = acc;
}

template <int WAVES, int DEPTH>
static int run(hipStream_t st, const unsigned char* hot, unsigned* sink, int mode, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t per_wg = 2u << 20;
  for (int wgs : {64, 256, 512}) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipLaunchKernelGGL((k_stream<WAVES, DEPTH>), dim3(wgs), dim3(64 * WAVES), 0, st, hot, per_wg, (size_t)(4u << 20), mode, sink);   // warm
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL((k_stream<WAVES, DEPTH>), dim3(wgs), dim3(64 * WAVES), 0, st, hot, per_wg, (size_t)(4u << 20), mode, sink);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    const double streams = (mode == 3) ? 2.0 : 1.0;
    const double total = (double)per_wg * wgs * streams;
    const int cus = wgs < 256 ? wgs : 256;
    printf("%-46s %d waves, %2d loads in flight, %4d workgroups: %8.1f us  %8.1f GB/s total  %6.1f GB/s per busy CU  (%.1f B/clk/CU at 2.4 GHz)\n", what, WAVES, DEPTH,
           wgs, best * 1e3, total / best / 1e6, total / best / 1e6 / cus, total / best / 1e6 / cus / 2.4);
  }
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned char* hot; unsigned* sink;
  CK(hipMalloc((void**)&hot, 4u << 20)); CK(hipMemset(hot, 2, 4u << 20));
  CK(hipMalloc((void**)&sink, 64));
  if (run<4, 8>(st, hot, sink, 1, "LDS-DMA ring (4 waves)")) return 1;
  if (run<4, 4>(st, hot, sink, 2, "registers only")) return 1;
  if (run<4, 8>(st, hot, sink, 2, "registers only")) return 1;
  if (run<4, 16>(st, hot, sink, 2, "registers only")) return 1;
  if (run<8, 8>(st, hot, sink, 2, "registers only")) return 1;
  if (run<8, 16>(st, hot, sink, 2, "registers only")) return 1;
  if (run<16, 8>(st, hot, sink, 2, "registers only")) return 1;
  if (run<8, 8>(st, hot, sink, 3, "LDS-DMA (waves 0-3) + registers (waves 4-7)")) return 1;
  if (run<8, 16>(st, hot, sink, 3, "LDS-DMA (waves 0-3) + registers (waves 4-7)")) return 1;
  return 0;
}
