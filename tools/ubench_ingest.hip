// Micro-benchmark: how fast can a CU pull GEMM operands into LDS?  (Round 3: why the B = 1 GEMMs sit at ~12 % of the MFMA
// peak.)  Each workgroup streams `per_wg` bytes through an LDS ring with global_load_lds_dwordx4 (the k_gemm_dma staging path:
// 1 KiB per wave-instruction, counted vmcnt, one barrier per 16 KiB stage) and does nothing else.  Swept over
//   * the number of workgroups (32 ... 1024: how many CUs pull at once),
//   * waves per workgroup (4 / 8),
//   * the source: a PRIVATE slice per workgroup of a 2 GiB buffer behind an eviction pass (HBM-cold, the weight stream of a
//     small-M GEMM), or ONE 4 MiB buffer every workgroup reads (L2 / Infinity-Cache warm after the first touch: the activation
//     operand, or a weight tile shared by the row tiles).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_ingest tools/ubench_ingest.hip && tools/bin/ubench_ingest
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// WAVES waves; a stage = 16 KiB = 16 pieces of 1 KiB; ring of ST stages; every wave issues 16 / WAVES pieces per stage
template <int WAVES, int ST>
__global__ void __launch_bounds__(64 * WAVES) k_ingest(const unsigned char* src, size_t per_wg, size_t wg_stride, size_t wrap, unsigned* sink) {      // wrap: a power of two
  __shared__ __attribute__((aligned(1024))) unsigned char ring[ST * 16384];
  constexpr int PPW = 16 / WAVES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)ring);
  const size_t base = ((size_t)blockIdx.x * wg_stride) & (wrap - 1), msk = wrap - 1;
  const int tiles = (int)(per_wg / 16384);
  auto issue = [&](int t) {
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds0 + (t % ST) * 16384 + wave * PPW * 1024);
#pragma unroll
    for (int q = 0; q < PPW; ++q)
      dma16(src + ((base + (size_t)t * 16384 + (size_t)(wave * PPW + q) * 1024 + lane * 16) & msk), sbase + q * 1024);
  };
#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (s < tiles) issue(s);
  unsigned acc = 0;
  for (int t = 0; t < tiles; ++t) {
    if (tiles - 1 - t >= ST - 2) wait_vmcnt<PPW * (ST - 2)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    acc += ring[(t % ST) * 16384 + threadIdx.x * 4];            // one LDS read per thread: keeps the stage "consumed"
    if (t + ST - 1 < tiles) issue(t + ST - 1);
  }
  if (acc == 0xdeadbeefu) sink[0] = acc;
}

// the same stream through REGISTERS: global_load_dwordx4 -> VGPRs -> ds_write_b128 (what a register-staged GEMM operand costs);
// every wave keeps PPW x DEPTH loads in flight (DEPTH stages ahead), one barrier per stage
template <int WAVES, int DEPTH>
__global__ void __launch_bounds__(64 * WAVES) k_ingest_reg(const unsigned char* src, size_t per_wg, size_t wg_stride, size_t wrap, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[2 * 16384];
  constexpr int PPW = 16 / WAVES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = ((size_t)blockIdx.x * wg_stride) & (wrap - 1), msk = wrap - 1;
  const int tiles = (int)(per_wg / 16384);
  uint4 regs[DEPTH][PPW];
  auto fetch = [&](int t, uint4 (&r)[PPW]) {
#pragma unroll
    for (int q = 0; q < PPW; ++q)
      r[q] = *reinterpret_cast<const uint4*>(src + ((base + (size_t)t * 16384 + (size_t)(wave * PPW + q) * 1024 + lane * 16) & msk));
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < tiles) fetch(d, regs[d]);
  unsigned acc = 0;
  for (int t0 = 0; t0 < tiles; t0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int t = t0 + d;
      if (t >= tiles) break;
      unsigned char* st = ring + (t & 1) * 16384 + (wave * PPW) * 1024 + lane * 16;
#pragma unroll
      for (int q = 0; q < PPW; ++q) *reinterpret_cast<uint4*>(st + q * 1024) = regs[d][q];
      if (t + DEPTH < tiles) fetch(t + DEPTH, regs[d]);
      // raw barrier: __syncthreads() would carry a vmcnt(0) and drain the loads that are meant to stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      acc += ring[(t & 1) * 16384 + threadIdx.x * 4];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
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

template <int WAVES, int ST, bool REG = false>
static int sweep(hipStream_t st, const unsigned char* big, size_t big_bytes, const unsigned char* hot, unsigned* sink, const uint4* evict,
                 size_t evict_n16) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t per_wg = 2u << 20;               // 2 MiB per workgroup = 128 stages
  for (int hotsrc = 0; hotsrc < 2; ++hotsrc)
    for (int wgs : {32, 64, 128, 256, 512, 1024}) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        if (!hotsrc) hipLaunchKernelGGL(k_thrash, dim3(2048), dim3(256), 0, st, evict, evict_n16, sink);     // evict L2 / MALL
        else hipLaunchKernelGGL((k_ingest<WAVES, 4>), dim3(wgs), dim3(64 * WAVES), 0, st, hot, per_wg, (size_t)0, (size_t)(4u << 20), sink);   // warm
        CK(hipEventRecord(e0, st));
        if (REG) {
          if (!hotsrc) hipLaunchKernelGGL((k_ingest_reg<WAVES, ST>), dim3(wgs), dim3(64 * WAVES), 0, st, big, per_wg, per_wg, big_bytes, sink);
          else hipLaunchKernelGGL((k_ingest_reg<WAVES, ST>), dim3(wgs), dim3(64 * WAVES), 0, st, hot, per_wg, (size_t)0, (size_t)(4u << 20), sink);
        } else {
          if (!hotsrc) hipLaunchKernelGGL((k_ingest<WAVES, ST>), dim3(wgs), dim3(64 * WAVES), 0, st, big, per_wg, per_wg, big_bytes, sink);
          else hipLaunchKernelGGL((k_ingest<WAVES, ST>), dim3(wgs), dim3(64 * WAVES), 0, st, hot, per_wg, (size_t)0, (size_t)(4u << 20), sink);
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      const double total = (double)per_wg * wgs;
      const int cus = wgs < 256 ? wgs : 256;
      printf("%s %d waves, depth %d x 16 KiB, %-28s %5d workgroups: %8.1f us  %8.1f GB/s total  %6.1f GB/s per busy CU  (%.1f B/clk/CU at 2.4 GHz)\n",
             REG ? "registers + ds_write" : "LDS-DMA             ", WAVES, ST, hotsrc ? "one shared 4 MiB buffer" : "private 2 MiB slices, cold", wgs, best * 1e3, total / best / 1e6,
             total / best / 1e6 / cus, total / best / 1e6 / cus / 2.4);
    }
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t big_bytes = (size_t)2 << 30;
  unsigned char *big, *hot; unsigned* sink; uint4* evict;
  CK(hipMalloc((void**)&big, big_bytes)); CK(hipMemset(big, 1, big_bytes));
  CK(hipMalloc((void**)&hot, 4u << 20)); CK(hipMemset(hot, 2, 4u << 20));
  CK(hipMalloc((void**)&sink, 64));
  const size_t evict_bytes = (size_t)1 << 30;
  CK(hipMalloc((void**)&evict, evict_bytes)); CK(hipMemset(evict, 3, evict_bytes));
  if (sweep<4, 4>(st, big, big_bytes, hot, sink, evict, evict_bytes / 16)) return 1;
  if (sweep<8, 4>(st, big, big_bytes, hot, sink, evict, evict_bytes / 16)) return 1;
  if (sweep<4, 8>(st, big, big_bytes, hot, sink, evict, evict_bytes / 16)) return 1;
  if (sweep<4, 2, true>(st, big, big_bytes, hot, sink, evict, evict_bytes / 16)) return 1;
  if (sweep<4, 4, true>(st, big, big_bytes, hot, sink, evict, evict_bytes / 16)) return 1;
  if (sweep<8, 4, true>(st, big, big_bytes, hot, sink, evict, evict_bytes / 16)) return 1;
  return 0;
}
