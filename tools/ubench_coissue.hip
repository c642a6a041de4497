// Micro-benchmark: do one wave's MFMAs overlap another wave's VALU work on the SAME SIMD?  (Round 3: the attention loop runs at
// the SUM of its MFMA and VALU time.)  512-thread blocks, one per CU: waves 0-3 sit on SIMDs 0-3, waves 4-7 are their partners.
//   mode 1: waves 0-3 run `iters` x 16 independent v_mfma_f32_32x32x16_f16, waves 4-7 idle
//   mode 2: waves 0-3 idle, waves 4-7 run `iters` x (VN x v_fma_f32 + EN x v_exp_f32)
//   mode 3: both;  mode 4: every wave runs MFMA block then VALU block per iteration (what one in-order wave does)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_coissue tools/ubench_coissue.hip && tools/bin/ubench_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int VN, int EN>
__device__ __forceinline__ void valu_block(float (&x)[8]) {
#pragma unroll
  for (int i = 0; i < VN; ++i) x[i & 7] = __builtin_fmaf(x[i & 7], 1.0001f, 0.5f);
#pragma unroll
  for (int i = 0; i < EN; ++i) x[i & 7] = __builtin_amdgcn_exp2f(x[i & 7]);
}
__device__ __forceinline__ void mfma_block(v16f (&acc)[4], v8h a, v8h b) {
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
}

template <int MODE, int VN, int EN>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  v16f acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  v8h a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  const bool do_m = MODE == 4 || ((MODE & 1) && wave < 4), do_v = MODE == 4 || ((MODE & 2) && wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (do_m) mfma_block(acc, a, b);
    if (do_v) valu_block<VN, EN>(x);
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE, int VN, int EN>
static float run(float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, VN, EN>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best * 1e3f;
}
template <int VN, int EN>
static void sweep(float* out) {
  const int iters = 2000;
  const float m = run<1, VN, EN>(out, iters), v = run<2, VN, EN>(out, iters), both = run<3, VN, EN>(out, iters), ser = run<4, VN, EN>(out, iters);
  const double cyc = 2.4e3 / iters;      // us -> cycles per iteration at 2.4 GHz
  printf("VALU block = %3d fma + %2d exp: MFMA wave alone %7.1f us (%5.0f cyc/iter = %4.1f per MFMA), VALU wave alone %7.1f us (%5.0f cyc/iter), "
         "both on one SIMD %7.1f us (%5.0f cyc/iter; max %5.0f, sum %5.0f), one wave doing both %7.1f us (%5.0f)\n",
         VN, EN, m, m * cyc, m * cyc / 16, v, v * cyc, both, both * cyc, (m > v ? m : v) * cyc, (m + v) * cyc, ser, ser * cyc);
}
int main() {
  float* out; CK(hipMalloc((void**)&out, 64));
  sweep<128, 0>(out);
  sweep<0, 32>(out);
  sweep<128, 32>(out);
  sweep<64, 16>(out);
  return 0;
}
