// Micro-benchmark: GroupNorm over a split-K output, (a) as today: reduce (+ slice statistics) then apply, two launches over many
// workgroups; (b) fused by group: one workgroup per (image, group) sums the slabs into registers, takes the statistics and
// normalises in one launch.  Each variant runs inside a hipGraph behind a producer kernel that rewrites the slabs from all
// CUs (as the split-K GEMM does), so the consumer's reads are as cold as in the step.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_gn tools/ubench_gn.hip
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_produce(float4* slab, size_t n4, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    slab[i] = make_float4(v + (float)(i & 7), v, v - 1.f, v + 2.f);
}

__device__ __forceinline__ float wsum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// (a1) reduce: thread = 4 consecutive columns of one row; sums SPLITS slabs, writes 16-bit y and per-(slice, group) sums
template <int SPLITS>
__global__ void __launch_bounds__(256) k_reduce(const float4* slab, size_t slab4, int M, int C, int cpg, int S, __half* y, float* part) {
  // grid (S slices, G/4 group blocks): a block covers rows of its slice x 4 groups
  const int s = blockIdx.x, gb = blockIdx.y;
  const int rows_per = M / S, c4pg = cpg / 4, cols4 = 4 * c4pg;        // float4 per row of this block
  const int r0 = s * rows_per;
  float sa[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
  for (int e = threadIdx.x; e < rows_per * cols4; e += 256) {
    const int r = r0 + e / cols4, c4 = e % cols4;
    const size_t off = (size_t)r * (C / 4) + gb * cols4 + c4;
    float4 a = slab[off];
#pragma unroll
    for (int k = 1; k < SPLITS; ++k) { const float4 b = slab[off + k * slab4]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    __half2 h0 = __floats2half2_rn(a.x, a.y), h1 = __floats2half2_rn(a.z, a.w);
    *reinterpret_cast<uint2*>(y + off * 4) = make_uint2(*reinterpret_cast<unsigned*>(&h0), *reinterpret_cast<unsigned*>(&h1));
    const int gl = c4 / c4pg;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (gl == q) { sa[q] += a.x + a.y + a.z + a.w; sq[q] += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
  }
  __shared__ float red[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) { sa[q] = wsum(sa[q]); sq[q] = wsum(sq[q]); }
  if ((threadIdx.x & 63) == 0) for (int q = 0; q < 4; ++q) { red[threadIdx.x >> 6][2 * q] = sa[q]; red[threadIdx.x >> 6][2 * q + 1] = sq[q]; }
  __syncthreads();
  if (threadIdx.x < 8) {
    float t = 0; for (int w = 0; w < 4; ++w) t += red[w][threadIdx.x];
    part[((size_t)(gb * 4 + (threadIdx.x >> 1)) * S + s) * 2 + (threadIdx.x & 1)] = t;
  }
}
// (a2) apply: thread = 8 channels of a pixel
__global__ void __launch_bounds__(256) k_apply(const __half* y, const float* part, int M, int C, int cpg, int S, __half* z) {
  __shared__ float2 st[64];
  const int G = C / cpg;
  if (threadIdx.x < G) {
    float a = 0, q = 0;
    for (int s = 0; s < S; ++s) { a += part[((size_t)threadIdx.x * S + s) * 2]; q += part[((size_t)threadIdx.x * S + s) * 2 + 1]; }
    const float n = (float)M * cpg, mean = a / n;
    st[threadIdx.x] = make_float2(mean, rsqrtf(q / n - mean * mean + 1e-5f));
  }
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint4 raw = idx < (size_t)M * C / 8 ? *reinterpret_cast<const uint4*>(y + idx * 8) : make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (idx >= (size_t)M * C / 8) return;
  const int c0 = (int)(idx % (C / 8)) * 8;
  const __half* h = reinterpret_cast<const __half*>(&raw);
  __half o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float2 s2 = st[(c0 + i) / cpg];
    float v = (__half2float(h[i]) - s2.x) * s2.y;
    v = v / (1.f + __expf(-v));
    o[i] = __float2half(v);
  }
  *reinterpret_cast<uint4*>(z + idx * 8) = *reinterpret_cast<uint4*>(o);
}

// (b) fused by group: block = one group; NE float4 elements per thread, every load issued before the first use
template <int SPLITS, int NE, int TH>
__global__ void __launch_bounds__(TH) k_fused(const float4* slab, size_t slab4, int M, int C, int cpg, __half* y, __half* z) {
  const int g = blockIdx.x, c4pg = cpg / 4, total = M * c4pg;
  float4 v[NE][SPLITS];
  size_t off[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int e = threadIdx.x + j * TH;
    const int r = e / c4pg, c4 = e - r * c4pg;
    off[j] = (size_t)r * (C / 4) + g * c4pg + c4;
#pragma unroll
    for (int k = 0; k < SPLITS; ++k) v[j][k] = e < total ? slab[off[j] + k * slab4] : make_float4(0, 0, 0, 0);
  }
  float sa = 0, sq = 0;
#pragma unroll
  for (int j = 0; j < NE; ++j) {
#pragma unroll
    for (int k = 1; k < SPLITS; ++k) { v[j][0].x += v[j][k].x; v[j][0].y += v[j][k].y; v[j][0].z += v[j][k].z; v[j][0].w += v[j][k].w; }
    const float4 a = v[j][0];
    sa += a.x + a.y + a.z + a.w; sq += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  }
  __shared__ float red[TH / 64][2];
  sa = wsum(sa); sq = wsum(sq);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sa; red[threadIdx.x >> 6][1] = sq; }
  __syncthreads();
  float ta = 0, tq = 0;
  for (int w = 0; w < TH / 64; ++w) { ta += red[w][0]; tq += red[w][1]; }
  const float n = (float)M * cpg, mean = ta / n, rstd = rsqrtf(tq / n - mean * mean + 1e-5f);
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int e = threadIdx.x + j * TH;
    if (e >= total) continue;
    const float4 a = v[j][0];
    __half2 h0 = __floats2half2_rn(a.x, a.y), h1 = __floats2half2_rn(a.z, a.w);
    *reinterpret_cast<uint2*>(y + off[j] * 4) = make_uint2(*reinterpret_cast<unsigned*>(&h0), *reinterpret_cast<unsigned*>(&h1));
    float o[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { float t = (o[i] - mean) * rstd; o[i] = t / (1.f + __expf(-t)); }
    h0 = __floats2half2_rn(o[0], o[1]); h1 = __floats2half2_rn(o[2], o[3]);
    *reinterpret_cast<uint2*>(z + off[j] * 4) = make_uint2(*reinterpret_cast<unsigned*>(&h0), *reinterpret_cast<unsigned*>(&h1));
  }
}

template <class F>
static int timed(const char* name, hipStream_t st, int n, F body, float* us_out) {
  hipGraph_t g; hipGraphExec_t ge; hipEvent_t e0, e1; float ms;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) body(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  *us_out = ms * 1e3f / n / 5;
  printf("%-70s %8.2f us per round\n", name, *us_out);
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  return 0;
}

template <int SPLITS, int NE, int TH>
static int shape(hipStream_t st, int M, int C, int cpg, int S) {
  const size_t slab4 = (size_t)M * C / 4;
  const int NB = 8;                      // rotate over buffers
  float4* slab[NB]; __half *y, *z; float* part;
  for (int i = 0; i < NB; ++i) CK(hipMalloc(&slab[i], slab4 * 16 * SPLITS));
  CK(hipMalloc(&y, (size_t)M * C * 2)); CK(hipMalloc(&z, (size_t)M * C * 2)); CK(hipMalloc(&part, 64 * 64 * 2 * 4));
  const int G = C / cpg;
  printf("M=%d C=%d cpg=%d splits=%d (slabs %.1f MB, %d groups)\n", M, C, cpg, SPLITS, slab4 * 16.0 * SPLITS / 1e6, G);
  float t0, t1, t2;
  auto produce = [&](int i) { hipLaunchKernelGGL(k_produce, dim3(256), dim3(256), 0, st, slab[i % NB], slab4 * SPLITS, (float)i); };
  if (timed("  producer only", st, 200, [&](int i) { produce(i); }, &t0)) return 1;
  if (timed("  producer + reduce(+slice sums) + apply", st, 200, [&](int i) {
        produce(i);
        hipLaunchKernelGGL((k_reduce<SPLITS>), dim3(S, G / 4), dim3(256), 0, st, slab[i % NB], slab4, M, C, cpg, S, y, part);
        hipLaunchKernelGGL(k_apply, dim3((unsigned)(((size_t)M * C / 8 + 255) / 256)), dim3(256), 0, st, y, part, M, C, cpg, S, z);
      }, &t1)) return 1;
  if (timed("  producer + fused by-group", st, 200, [&](int i) {
        produce(i);
        hipLaunchKernelGGL((k_fused<SPLITS, NE, TH>), dim3(G), dim3(TH), 0, st, slab[i % NB], slab4, M, C, cpg, y, z);
      }, &t2)) return 1;
  printf("  => reduce + apply %.2f us, fused %.2f us\n", t1 - t0, t2 - t0);
  for (int i = 0; i < NB; ++i) (void)hipFree(slab[i]);
  (void)hipFree(y); (void)hipFree(z); (void)hipFree(part);
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  if (shape<6, 3, 1024>(st, 256, 1280, 40, 16)) return 1;      // 16^2 level
  if (shape<3, 5, 1024>(st, 1024, 640, 20, 32)) return 1;      // 32^2 level
  if (shape<12, 1, 1024>(st, 64, 1280, 40, 16)) return 1;      // 8^2 level (one element per thread, 640 live threads)
  if (shape<3, 3, 1024>(st, 256, 1280, 40, 16)) return 1;      // 16^2 level, fewer splits
  return 0;
}
