// launch-floor micro-benchmark: back-to-back tiny kernels on a stream and inside a graph
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { long a[20]; };
__global__ void k_empty(float* p) { if (threadIdx.x == 9999) p[0] = 1; }
__global__ void __launch_bounds__(256) k_lds(float* p) { __shared__ float s[32768]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (threadIdx.x == 9999) p[0] = s[5]; }
__global__ void k_big(Big b, float* p) { if (threadIdx.x == 9999) p[0] = (float)b.a[3]; }
__global__ void k_copy(const float4* a, float4* b, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
int main() {
  float *p, *a, *b; hipMalloc(&p, 1024); hipMalloc(&a, 1 << 24); hipMalloc(&b, 1 << 24);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 2000; Big big{};
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 10; ++i) launch();
    hipStreamSynchronize(st);
    hipEventRecord(e0, st); for (int i = 0; i < N; ++i) launch(); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-28s stream: %6.2f us/kernel", name, ms * 1e3 / N);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal); for (int i = 0; i < N; ++i) launch(); hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("   graph: %6.2f us/kernel\n", ms * 1e3 / N);
  };
  run("empty 1 block", [&] { k_empty<<<1, 64, 0, st>>>(p); });
  run("empty 256 blocks x256", [&] { k_empty<<<256, 256, 0, st>>>(p); });
  run("empty 2048 blocks x256", [&] { k_empty<<<2048, 256, 0, st>>>(p); });
  run("128KB LDS 160 blocks", [&] { k_lds<<<160, 256, 0, st>>>(p); });
  run("big kernarg 160 blocks", [&] { k_big<<<160, 256, 0, st>>>(big, p); });
  run("copy 2.6MB", [&] { k_copy<<<640, 256, 0, st>>>((const float4*)a, (float4*)b, 163840); });
  return 0;
}
