// Micro-benchmark: what one dependent kernel costs on this box, by launch path (diagnostics for the launch-bound B=1 step).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_launch tools/ubench_launch.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

struct Big { float v[60]; int n; float* out; };

__global__ void k_triv(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
// no kernel arguments at all: what the argument fetch (s_load of the kernarg segment, cold in every kernel) adds to a launch
__device__ float g_sink[64];
__global__ void k_noarg() { if (threadIdx.x == 0 && blockIdx.x == 0) g_sink[0] += 1.f; }
__global__ void k_empty() {}
template <int I> __global__ void k_var(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[I] += (float)I; }
__global__ void k_big(const Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.out[0] += b.v[b.n]; }
// streaming: y = x + 1 over n floats (dependent chain through memory)
__global__ void k_stream(const float4* x, float4* y, int n4) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) { float4 v = x[i]; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; y[i] = v; }
}

// stream S bytes through the memory hierarchy (evicts L2 / MALL content); NT: non-temporal loads
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void k_thrash(const u4* src, size_t n16, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n16; i += stride) {
    u4 v = NT ? __builtin_nontemporal_load(src + i) : src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[1] = 1.f;
}
// a small kernel with some code and a dependent load chain through its kernarg pointer (what a real small op does)
__global__ void k_small(const float* in, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * 2.f + 1.f;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class F>
static int run(const char* name, hipStream_t st, int n, bool graph, F body) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  if (!graph) {
    for (int i = 0; i < n; ++i) body(i);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) body(i);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s eager  %6d launches: %8.2f us each\n", name, n, ms * 1e3 / n);
  } else {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) body(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    const int reps = 5;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s graph  %6d nodes   : %8.2f us each\n", name, n, ms * 1e3 / n / reps);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  }
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float* p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
  const int n4 = 1 << 18;    // 4 MB buffers
  float4 *a, *b; CK(hipMalloc(&a, (size_t)n4 * 16)); CK(hipMalloc(&b, (size_t)n4 * 16));
  CK(hipMemset(a, 0, (size_t)n4 * 16)); CK(hipMemset(b, 0, (size_t)n4 * 16));
  Big big; big.n = 3; big.out = p; for (int i = 0; i < 60; ++i) big.v[i] = i;
  for (int graph = 0; graph < 2; ++graph) {
    const int N = 2000;
    run("trivial <<<1,64>>>", st, N, graph, [&](int) { hipLaunchKernelGGL(k_triv, dim3(1), dim3(64), 0, st, p); });
    run("trivial <<<256,256>>>", st, N, graph, [&](int) { hipLaunchKernelGGL(k_triv, dim3(256), dim3(256), 0, st, p); });
    run("no-argument kernel, same work <<<256,256>>>", st, N, graph, [&](int) { hipLaunchKernelGGL(k_noarg, dim3(256), dim3(256), 0, st); });
    run("empty kernel (s_endpgm) <<<256,256>>>", st, N, graph, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st); });
    run("trivial <<<2048,256>>>", st, N, graph, [&](int) { hipLaunchKernelGGL(k_triv, dim3(2048), dim3(256), 0, st, p); });
    run("8 different trivial kernels alternating <<<256,256>>>", st, N, graph, [&](int i) {
      switch (i & 7) {
        case 0: hipLaunchKernelGGL(k_var<0>, dim3(256), dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL(k_var<1>, dim3(256), dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(k_var<2>, dim3(256), dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(k_var<3>, dim3(256), dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL(k_var<4>, dim3(256), dim3(256), 0, st, p); break;
        case 5: hipLaunchKernelGGL(k_var<5>, dim3(256), dim3(256), 0, st, p); break;
        case 6: hipLaunchKernelGGL(k_var<6>, dim3(256), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(k_var<7>, dim3(256), dim3(256), 0, st, p); break;
      } });
    run("256-byte kernarg struct <<<256,256>>>", st, N, graph, [&](int) { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, st, big); });
    run("streaming 4 MB -> 4 MB ping-pong <<<1024,256>>>", st, N, graph, [&](int i) {
      if (i & 1) hipLaunchKernelGGL(k_stream, dim3(n4 / 256), dim3(256), 0, st, (const float4*)b, a, n4);
      else hipLaunchKernelGGL(k_stream, dim3(n4 / 256), dim3(256), 0, st, (const float4*)a, b, n4); });
    run("streaming 256 KB -> 256 KB ping-pong <<<64,256>>>", st, N, graph, [&](int i) {
      if (i & 1) hipLaunchKernelGGL(k_stream, dim3(64), dim3(256), 0, st, (const float4*)b, a, 64 * 256);
      else hipLaunchKernelGGL(k_stream, dim3(64), dim3(256), 0, st, (const float4*)a, b, 64 * 256); });
    run("hipMemcpyAsync D2D 64 KB", st, 500, graph, [&](int) { (void)hipMemcpyAsync(b, a, 65536, hipMemcpyDeviceToDevice, st); });
  }
  // cold-start cost: a small kernel right after S bytes of weight-like streaming, vs the streaming alone
  float* src; const size_t SMAX = (size_t)2 << 30; CK(hipMalloc(&src, SMAX)); CK(hipMemset(src, 1, SMAX));
  for (size_t S : {(size_t)16 << 20, (size_t)128 << 20, (size_t)512 << 20, (size_t)2 << 30}) {
    for (int nt = 0; nt < 2; ++nt) {
      const int N = 40;
      auto thrash = [&](int) {
        if (nt) hipLaunchKernelGGL(k_thrash<true>, dim3(2048), dim3(256), 0, st, (const u4*)src, S / 16, p);
        else hipLaunchKernelGGL(k_thrash<false>, dim3(2048), dim3(256), 0, st, (const u4*)src, S / 16, p); };
      char name[128];
      snprintf(name, sizeof name, "thrash %4zu MB %s", S >> 20, nt ? "(nt loads)" : "(plain)   ");
      run(name, st, N, true, thrash);
      snprintf(name, sizeof name, "thrash %4zu MB %s + trivial + small(64 KB) kernels", S >> 20, nt ? "(nt loads)" : "(plain)   ");
      run(name, st, N, true, [&](int i) { thrash(i); hipLaunchKernelGGL(k_triv, dim3(1), dim3(64), 0, st, p);
                                          hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, (const float*)a, (float*)b, 16384); });
    }
  }
  return 0;
}
