// Micro-benchmark (round 4, follow-up of ubench_ingest2): the K-loop staging of a 128x64x64 GEMM tile, without the MFMAs.
//   dma24   today's loop: per K tile the 4 waves issue 24 one-KiB LDS-DMA pieces (A 16 KiB + W 8 KiB), ring of ST stages,
//           counted vmcnt, one barrier per tile
//   mixed   per K tile 16 LDS-DMA pieces (A) + per wave 4 register loads of 16 B per lane (its W fragments: waves 2w, 2w+1 read
//           the same 4 KiB, as the two row halves of a 2 x 2 wave layout do), BOTH kept ST-1 tiles ahead (vmcnt completes in
//           order: a register stream only one tile ahead would drain the DMA ring with it), register ring indexed at compile time
// A from one shared 4 MiB buffer (L2), W from a private cold slice per workgroup (weights) or the shared buffer.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_ingest3 tools/ubench_ingest3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ST, bool MIXED, bool FRAGMAJOR = false>
__global__ void __launch_bounds__(256) k_loop(const unsigned char* a_src, const unsigned char* w_src, size_t w_stride, int tiles, unsigned* sink) {
  constexpr int STAGE = MIXED ? 16384 : 24576;
  constexpr int PPW = STAGE / 1024 / 4;                 // DMA pieces per wave per tile
  constexpr int D = ST - 1;
  __shared__ __attribute__((aligned(1024))) unsigned char ring[ST * STAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)ring);
  const size_t amsk = (4u << 20) - 1;
  const size_t abase = ((size_t)blockIdx.x * 65536) & amsk;
  const unsigned char* wbase = w_src + (size_t)blockIdx.x * w_stride;
  auto issue_dma = [&](int t, int slot) {
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds0 + slot * STAGE + wave * PPW * 1024);
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int piece = wave * PPW + q;
      const unsigned char* src = (MIXED || piece < 16) ? a_src + ((abase + (size_t)t * 16384 + (size_t)piece * 1024 + lane * 16) & amsk)
                                                      : wbase + (size_t)t * 8192 + (size_t)(piece - 16) * 1024 + lane * 16;
      dma16(src, sbase + q * 1024);
    }
  };
  // (the register loads are inline asm as well: hipcc counts only the loads it can see, so its own s_waitcnt in front of the first
  //  use of a compiler-issued load would drain the LDS-DMA pieces issued after it -- measured: 1.14 us per tile instead of 0.34)
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  u4v wr[D][4];
  auto issue_reg = [&](int t, u4v (&r)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {     // 32 rows x 32 B per load: row = lane & 31 (128-byte rows), the 32-byte pair of k-step kk
      // FRAGMAJOR: the weights stored fragment-major ([n block][k step][lane][16 B]: a wave's load is one contiguous KiB); otherwise
      // the row-major 64x64 tile of today's layout (32 rows x 32 B per load: 64 separate 16-byte requests)
      const unsigned char* ptr = wbase + (size_t)t * 8192 + (size_t)(wave & 1) * 4096 +
                                 (FRAGMAJOR ? kk * 1024 + lane * 16 : (lane & 31) * 128 + kk * 32 + (lane >> 5) * 16);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[kk]) : "v"(ptr) : "memory");
    }
  };
  unsigned acc = 0;
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < tiles) { if (MIXED) issue_reg(s, wr[s]); issue_dma(s, s); }
  constexpr int OPS = PPW + (MIXED ? 4 : 0);
  for (int t0 = 0; t0 < tiles; t0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int t = t0 + u;
      if (t >= tiles) break;
      if (MIXED) {      // the counted wait carries the fragment registers as operands: their uses cannot be scheduled in front of it
        if (tiles - 1 - t >= D - 1) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wr[u][0]), "+v"(wr[u][1]), "+v"(wr[u][2]), "+v"(wr[u][3]) : "n"(OPS * (D - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(wr[u][0]), "+v"(wr[u][1]), "+v"(wr[u][2]), "+v"(wr[u][3]) :: "memory");
      } else {
        if (tiles - 1 - t >= D - 1) wait_vmcnt<OPS * (D - 1)>(); else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      acc += ring[((t % ST)) * STAGE + threadIdx.x * 4];
      if (MIXED) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc ^= wr[u][kk].x ^ wr[u][kk].w;
      }
      if (t + D < tiles) { if (MIXED) issue_reg(t + D, wr[u]); issue_dma(t + D, (t + D) % ST); }
    }
  }
  if (acc == 0xdeadbeefu) sink[0] = acc;
}
__global__ void k_thrash(const uint4* src, size_t n16, unsigned* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[1] = 1;
}
template <int ST, bool MIXED, bool FRAGMAJOR = false>
static int run(hipStream_t st, const unsigned char* hot, const unsigned char* big, unsigned* sink, const uint4* evict, size_t evict_n16) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tiles = 90;
  for (int cold = 0; cold < 2; ++cold)
    for (int wgs : {80, 160, 240}) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        if (cold) hipLaunchKernelGGL(k_thrash, dim3(2048), dim3(256), 0, st, evict, evict_n16, sink);
        else hipLaunchKernelGGL((k_loop<ST, MIXED, FRAGMAJOR>), dim3(wgs), dim3(256), 0, st, hot, hot, (size_t)0, tiles, sink);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((k_loop<ST, MIXED, FRAGMAJOR>), dim3(wgs), dim3(256), 0, st, hot, cold ? big : hot, cold ? (size_t)tiles * 8192 : (size_t)0, tiles, sink);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("%-6s ring %d  W %-22s %3d workgroups: %7.1f us for %d K tiles = %5.3f us per tile  (%5.1f GB/s per CU of 24 KiB tiles)\n", MIXED ? (FRAGMAJOR ? "mixedF" : "mixed") : "dma24", ST,
             cold ? "private, HBM-cold" : "shared, L2", wgs, best * 1e3, tiles, best * 1e3 / tiles, 24576.0 * tiles / (best * 1e-3) / 1e9);
    }
  return 0;
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned char *hot, *big; unsigned* sink; uint4* evict;
  CK(hipMalloc((void**)&hot, 4u << 20)); CK(hipMemset(hot, 2, 4u << 20));
  const size_t big_bytes = (size_t)256 * 90 * 8192;
  CK(hipMalloc((void**)&big, big_bytes)); CK(hipMemset(big, 1, big_bytes));
  CK(hipMalloc((void**)&sink, 64));
  const size_t evict_bytes = (size_t)1 << 30;
  CK(hipMalloc((void**)&evict, evict_bytes)); CK(hipMemset(evict, 3, evict_bytes));
  if (run<3, false>(st, hot, big, sink, evict, evict_bytes / 16)) return 1;
  if (run<3, true>(st, hot, big, sink, evict, evict_bytes / 16)) return 1;
  if (run<5, false>(st, hot, big, sink, evict, evict_bytes / 16)) return 1;
  if (run<5, true>(st, hot, big, sink, evict, evict_bytes / 16)) return 1;
  if (run<3, true, true>(st, hot, big, sink, evict, evict_bytes / 16)) return 1;
  if (run<5, true, true>(st, hot, big, sink, evict, evict_bytes / 16)) return 1;
  return 0;
}
