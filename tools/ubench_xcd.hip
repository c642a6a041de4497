// Micro-benchmark: does a consumer KERNEL find its producer kernel's output in the XCD's L2?  Kernel W: block b writes chunk
// (b + sw) % 256 of a buffer; kernel R: block b reads chunk (b + sr) % 256.  Blocks land on XCD b % 8 (observed), so sr - sw = 0 or 8
// reads what the same XCD wrote, 1 reads what the neighbouring XCD wrote.  256 blocks x `chunk` bytes; W and R alternate on one
// stream (kernel boundaries in between, as in the engine's graphs); reported: time of R alone (events) per launch.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/bin/ubench_xcd tools/ubench_xcd.hip && tools/bin/ubench_xcd
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_w(uint4* buf, int chunk16, int shift, unsigned v) {
  uint4* p = buf + (size_t)((blockIdx.x + shift) % gridDim.x) * chunk16;
  for (int i = threadIdx.x; i < chunk16; i += 256) p[i] = make_uint4(v + i, v, v, v);
}
__global__ void __launch_bounds__(256) k_r(const uint4* buf, int chunk16, int shift, unsigned* out) {
  const uint4* p = buf + (size_t)((blockIdx.x + shift) % gridDim.x) * chunk16;
  unsigned acc = 0;
  for (int i = threadIdx.x; i < chunk16; i += 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) out[blockIdx.x] = acc;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  uint4* buf; unsigned* out;
  CK(hipMalloc((void**)&buf, (size_t)64 << 20)); CK(hipMalloc((void**)&out, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int kb : {4, 16, 64, 128}) {
    const int chunk16 = kb * 1024 / 16;
    for (int sr : {0, 8, 1, 3, 128}) {
      float tot = 0;
      const int reps = 50;
      for (int r = 0; r < reps + 5; ++r) {
        hipLaunchKernelGGL(k_w, dim3(256), dim3(256), 0, st, buf, chunk16, 0, (unsigned)r);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_r, dim3(256), dim3(256), 0, st, buf, chunk16, sr, out);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 5) tot += ms;
      }
      printf("chunk %4d KB per block (%5.1f MB total), reader shift %3d (%s): read kernel %6.2f us\n", kb, kb * 256 / 1024.0, sr,
             sr % 8 == 0 ? "same XCD" : "other XCD", tot / reps * 1e3);
    }
  }
  return 0;
}
