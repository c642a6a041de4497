// Non-GEMM U-Net kernels: the two tiny-channel convolutions, GroupNorm(+SiLU) and
// LayerNorm forward/backward, GEGLU forward/backward, concat/split copies, 2x2 sum pooling,
// dtype conversion and the sinusoidal timestep embedding.  Channels-last 16-bit storage,
// f32 math.
#include "unet_kernels.h"

namespace dh {

// Timing-only ablation (tuning builds, -DDH_TUNING; compiled out of the product library): DH_ABLATE_SKIP is a bit mask of
// launcher families that return without launching (results are garbage) -- the step time it removes divided by the
// launches it removes is what a launch of that family costs in situ.  1 = LayerNorm, 2 = GroupNorm, 4 = GEGLU,
// 8 = copies / concat / split / pool
#ifdef DH_TUNING
static int ablate_mask() { static const int m = getenv("DH_ABLATE_SKIP") ? atoi(getenv("DH_ABLATE_SKIP")) : 0; return m; }
// DH_ABLATE_EMPTY=1: the skipped launch is replaced by an EMPTY kernel (one wave, no memory access): what a kernel
// boundary alone costs in situ
__global__ void k_noop() {}
static int ablate_empty() { static const int m = getenv("DH_ABLATE_EMPTY") ? atoi(getenv("DH_ABLATE_EMPTY")) : 0; return m; }
#define DH_ABLATE(bit) do { if (ablate_mask() & (bit)) { if (ablate_empty()) hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, st); return; } } while (0)
#else
#define DH_ABLATE(bit) do { } while (0)
#endif

// ------------------------------------------------------------- tiny-channel convolutions
// few input channels (<= 8), many outputs: y[p][co] = b[co] + sum_{tap,ci} x[p+off][ci] w[co][tap][ci].
// A workgroup owns CF_PIX consecutive pixels so that every weight is read once per CF_PIX pixels (one pixel per
// workgroup re-read the whole matrix per pixel: 236 MB through L2 per 64x64 image).
constexpr int CF_PIX = 4;
template <class T>
__global__ void k_conv_few_in(const float* x, const float* w, const float* bias, T* y, int B, int H, int W, int Ci,
                              int Co) {
  __shared__ float patch[CF_PIX][9 * 8];
  const int K = 9 * Ci, total = B * H * W;
  const int p0 = blockIdx.x * CF_PIX;
  for (int i = threadIdx.x; i < CF_PIX * K; i += blockDim.x) {
    const int lp = i / K, k = i - lp * K, p = p0 + lp;
    float v = 0.f;
    if (p < total) {
      const int b = p / (H * W), r = p - b * H * W, oy = r / W, ox = r - oy * W;
      const int tap = k / Ci, ci = k - tap * Ci;
      const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * H + iy) * W + ix) * Ci + ci];
    }
    patch[lp][k] = v;
  }
  __syncthreads();
  for (int co = threadIdx.x; co < Co; co += blockDim.x) {
    float acc[CF_PIX];
    const float b0 = bias ? bias[co] : 0.f;
#pragma unroll
    for (int lp = 0; lp < CF_PIX; ++lp) acc[lp] = b0;
    const float* wr = w + (size_t)co * K;
    for (int k = 0; k < K; ++k) {
      const float wk = wr[k];
#pragma unroll
      for (int lp = 0; lp < CF_PIX; ++lp) acc[lp] += patch[lp][k] * wk;
    }
#pragma unroll
    for (int lp = 0; lp < CF_PIX; ++lp)
      if (p0 + lp < total) y[(size_t)(p0 + lp) * Co + co] = from_f32<T>(acc[lp]);
  }
}

// many input channels, few outputs (<= 8).  thread = (pixel, 8-channel chunk): per tap one 16-byte activation load and the
// chunk's weights of every output channel, f32 accumulate; the chunks of a pixel are then added in chunk order through LDS
// (deterministic).  A workgroup owns 256 / (Ci / 8) consecutive pixels -- 683 workgroups at 64 x 64 x 320 where the round-2
// kernel (one wave per four pixels, lanes strided over the channels with 2-byte loads) ran 256 and took 47 us.
constexpr int CFO_MAX = 8;
template <class T>
__global__ void __launch_bounds__(256) k_conv_few_out(const T* x, const float* w, const float* bias, float* y, int B, int H, int W,
                                                      int Ci, int Co, int accumulate) {
  __shared__ float sm[256 * CFO_MAX];
  const int nch = Ci >> 3, ppb = 256 / nch;
  const int pl = threadIdx.x / nch, ch = threadIdx.x - pl * nch;
  const int total = B * H * W;
  const int p = blockIdx.x * ppb + pl;
  float acc[CFO_MAX];
#pragma unroll
  for (int co = 0; co < CFO_MAX; ++co) acc[co] = 0.f;
  if (pl < ppb && p < total) {
    const int b = p / (H * W), r = p - b * H * W, py = r / W, px = r - py * W;
    typedef T T8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const uint4 raw = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * Ci + ch * 8);
      const T8 xv = __builtin_bit_cast(T8, raw);
      float xf[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) xf[i] = to_f32<T>(xv[i]);
#pragma unroll
      for (int co = 0; co < CFO_MAX; ++co) {
        if (co >= Co) break;
        const float* wp = w + ((size_t)co * 9 + tap) * Ci + ch * 8;
        const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
        acc[co] += xf[0] * w0.x + xf[1] * w0.y + xf[2] * w0.z + xf[3] * w0.w + xf[4] * w1.x + xf[5] * w1.y + xf[6] * w1.z +
                   xf[7] * w1.w;
      }
    }
  }
#pragma unroll
  for (int co = 0; co < CFO_MAX; ++co)
    if (co < Co) sm[co * 256 + threadIdx.x] = acc[co];
  __syncthreads();
  // thread (pixel pl2, output co2) adds the nch chunk sums of its pixel in chunk order
  const int pl2 = threadIdx.x / Co, co2 = threadIdx.x - pl2 * Co;
  const int p2 = blockIdx.x * ppb + pl2;
  if (pl2 < ppb && p2 < total) {
    float s = 0.f;
    const float* row = sm + co2 * 256 + pl2 * nch;
    for (int c = 0; c < nch; ++c) s += row[c];
    s += bias ? bias[co2] : 0.f;
    float* o = y + (size_t)p2 * Co + co2;
    *o = accumulate ? *o + s : s;
  }
}
static inline unsigned conv_few_out_blocks(int pixels, int Ci) { return (unsigned)cdiv(pixels, 256 / (Ci >> 3)); }
// k_conv_few_out's shape limits (advisor, round 3): 16-byte channel chunks, 256 / (Ci / 8) >= 1 pixels per block and
// pixels-per-block x Co <= 256 reducing threads.  The engines' create calls enforce them (dh_unet_create, dh_vae_*_create);
// a call outside them is refused here instead of dividing by zero or leaving pixels unwritten.
static inline bool conv_few_out_ok(int Ci, int Co, const void* w) {
  const bool ok = Ci % 8 == 0 && Ci >= 64 && Ci <= 2048 && Co >= 1 && Co <= 8 && ((size_t)w & 15) == 0;
  if (!ok) set_error("few-output convolution: needs 64 <= Cin <= 2048, Cin % 8 == 0, Cout <= 8, 16-byte aligned weights");
  return ok;
}

void launch_conv_small_fwd(int dtype, const void* x, int x_is_f32, const float* w, const float* bias, void* y,
                           int y_is_f32, int B, int H, int W, int Cin, int Cout, hipStream_t st) {
  if (x_is_f32) {   // few-in
    if (dtype == DH_DTYPE_F16)
      hipLaunchKernelGGL((k_conv_few_in<f16>), dim3(cdiv(B * H * W, CF_PIX)), dim3(256), 0, st, (const float*)x, w, bias, (f16*)y, B, H, W, Cin, Cout);
    else
      hipLaunchKernelGGL((k_conv_few_in<bf16>), dim3(cdiv(B * H * W, CF_PIX)), dim3(256), 0, st, (const float*)x, w, bias, (bf16*)y, B, H, W, Cin, Cout);
  } else {
    if (!conv_few_out_ok(Cin, Cout, w)) return;
    if (dtype == DH_DTYPE_F16)
      hipLaunchKernelGGL((k_conv_few_out<f16>), dim3(conv_few_out_blocks(B * H * W, Cin)), dim3(256), 0, st, (const f16*)x, w, bias, (float*)y, B, H, W, Cin, Cout, 0);
    else
      hipLaunchKernelGGL((k_conv_few_out<bf16>), dim3(conv_few_out_blocks(B * H * W, Cin)), dim3(256), 0, st, (const bf16*)x, w, bias, (float*)y, B, H, W, Cin, Cout, 0);
  }
}

// input-gradients reuse the two kernels with the flipped/transposed weights prepared at load
void launch_conv_small_bwd(int dtype, const void* dy, int dy_is_f32, const float* w, void* dx, int dx_is_f32,
                           int accumulate, int B, int H, int W, int Cin, int Cout, hipStream_t st) {
  // Cin/Cout here are those of the GRADIENT convolution: dy has Cin channels, dx has Cout
  if (dy_is_f32) {
    if (dtype == DH_DTYPE_F16)
      hipLaunchKernelGGL((k_conv_few_in<f16>), dim3(cdiv(B * H * W, CF_PIX)), dim3(256), 0, st, (const float*)dy, w, (const float*)nullptr, (f16*)dx, B, H, W, Cin, Cout);
    else
      hipLaunchKernelGGL((k_conv_few_in<bf16>), dim3(cdiv(B * H * W, CF_PIX)), dim3(256), 0, st, (const float*)dy, w, (const float*)nullptr, (bf16*)dx, B, H, W, Cin, Cout);
  } else {
    if (!conv_few_out_ok(Cin, Cout, w)) return;
    if (dtype == DH_DTYPE_F16)
      hipLaunchKernelGGL((k_conv_few_out<f16>), dim3(conv_few_out_blocks(B * H * W, Cin)), dim3(256), 0, st, (const f16*)dy, w, (const float*)nullptr, (float*)dx, B, H, W, Cin, Cout, accumulate);
    else
      hipLaunchKernelGGL((k_conv_few_out<bf16>), dim3(conv_few_out_blocks(B * H * W, Cin)), dim3(256), 0, st, (const bf16*)dy, w, (const float*)nullptr, (float*)dx, B, H, W, Cin, Cout, accumulate);
  }
}

// ----------------------------------------------------------------------------- GroupNorm
__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
// silu_grad: unet_kernels.h (shared with the split-K reduce that produces the backward statistics)

// GroupNorm statistics in two deterministic stages:
//   k_gn_partial  grid (S, G/4, B): a workgroup owns one row slice x 4 adjacent groups, reads the 4*cpg
//                 channel window of every row with 16-byte chunks and leaves (n, mean, M2) per group
//                 [f32 sums shifted by a per-group pivot = the group's first element of the slice; wave
//                 butterflies + a fixed-order fold over the 4 waves, no atomics]
//   the apply kernel combines the S <= 16 slices (8 lanes per group, butterfly, Chan's formula) before
//   normalising; block x == 0 also publishes (mean, rstd) for backward.
// partial layout: [b][g][3][S]  (n | mean | M2 planes)
// slices per image: 32 for one or two images (the statistics kernels then launch 256 workgroups, one per CU), 16 for
// bigger batches, where the batch dimension already fills the chip and the apply kernels' combine costs more
static int gn_slice_cap(int B) {
#ifdef DH_TUNING
  static const int v = getenv("DH_GN_SLICES") ? atoi(getenv("DH_GN_SLICES")) : 0;
  if (v > 0) return v > 64 ? 64 : v;                          // <= 64: gn_combine holds 8 slices per lane
#endif
  return B <= 2 ? 32 : 16;
}
int gn_slices(int HW, int B) { const int cap = gn_slice_cap(B); int s = HW / 4; return s < 1 ? 1 : (s > cap ? cap : s); }

template <class T, bool BWD>
__global__ void __launch_bounds__(256) k_gn_partial(const T* x, const T* dy, const float* gamma, const float* beta,
                                                    const float* stats, float* part, int HW, int C, int G, int S, int silu) {
  __shared__ float sm_red[4][2 * GN_GB];
  // index arithmetic by float reciprocals (div_small): the integer divisions of this prologue were ~500 instructions
  // in front of the first load
  const int s = blockIdx.x, g0 = blockIdx.y * GN_GB, b = blockIdx.z, cpg = div_small(C, rcp_fast(G));
  const int W = GN_GB * cpg, nch = W >> 3;
  const float inv_nch = rcp_fast(nch), inv_S = rcp_fast(S);
  const int RP = div_small((int)blockDim.x, inv_nch);
  const int r0 = div_small(HW * s, inv_S), r1 = div_small(HW * (s + 1), inv_S);
  const size_t base = (size_t)b * HW * C + (size_t)g0 * cpg;
  const int rr = div_small((int)threadIdx.x, inv_nch), ch = threadIdx.x - rr * nch;
  float ga[GN_GB], gq[GN_GB], pg[GN_GB];
#pragma unroll
  for (int gl = 0; gl < GN_GB; ++gl) {
    ga[gl] = 0.f; gq[gl] = 0.f;
    pg[gl] = (BWD || g0 + gl >= G) ? 0.f : to_f32<T>(x[base + (size_t)r0 * C + gl * cpg]);
  }
  if (rr < RP) {
    int gi[8];
    const float inv_cpg = rcp_fast(cpg);
    float a[8], q[8], pv[8], gm[8], bt[8], mu[8], rs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {          // a chunk may straddle groups: per-element group index
      gi[i] = div_small(ch * 8 + i, inv_cpg);
      a[i] = 0.f; q[i] = 0.f; pv[i] = 0.f;
#pragma unroll
      for (int gl = 0; gl < GN_GB; ++gl) pv[i] = gi[i] == gl ? pg[gl] : pv[i];
    }
    if (BWD) {
      // vector loads: the affine parameters of the 8 channels and the (mean, rstd) pairs of the 4 groups
      *reinterpret_cast<float4*>(gm) = *reinterpret_cast<const float4*>(gamma + g0 * cpg + ch * 8);
      *reinterpret_cast<float4*>(gm + 4) = *reinterpret_cast<const float4*>(gamma + g0 * cpg + ch * 8 + 4);
      *reinterpret_cast<float4*>(bt) = *reinterpret_cast<const float4*>(beta + g0 * cpg + ch * 8);
      *reinterpret_cast<float4*>(bt + 4) = *reinterpret_cast<const float4*>(beta + g0 * cpg + ch * 8 + 4);
      float2 st4[GN_GB];
#pragma unroll
      for (int gl = 0; gl < GN_GB; ++gl)
        st4[gl] = g0 + gl < G ? *reinterpret_cast<const float2*>(stats + 2 * (b * G + g0 + gl)) : make_float2(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        mu[i] = 0.f; rs[i] = 0.f;
#pragma unroll
        for (int gl = 0; gl < GN_GB; ++gl) { mu[i] = gi[i] == gl ? st4[gl].x : mu[i]; rs[i] = gi[i] == gl ? st4[gl].y : rs[i]; }
      }
    }
    for (int r = r0 + rr; r < r1; r += RP) {
      uint4 raw = *reinterpret_cast<const uint4*>(x + base + (size_t)r * C + ch * 8);
      const T* v = reinterpret_cast<const T*>(&raw);
      if (!BWD) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = to_f32<T>(v[i]) - pv[i]; a[i] += d; q[i] += d * d; }
      } else {
        uint4 rawd = *reinterpret_cast<const uint4*>(dy + base + (size_t)r * C + ch * 8);
        const T* dv = reinterpret_cast<const T*>(&rawd);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (to_f32<T>(v[i]) - mu[i]) * rs[i];
          float d = to_f32<T>(dv[i]);
          if (silu) d *= silu_grad(xh * gm[i] + bt[i]);
          d *= gm[i];
          a[i] += d;
          q[i] += d * xh;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int gl = 0; gl < GN_GB; ++gl) {
        ga[gl] += gi[i] == gl ? a[i] : 0.f;
        gq[gl] += gi[i] == gl ? q[i] : 0.f;
      }
  }
#pragma unroll
  for (int gl = 0; gl < GN_GB; ++gl) { ga[gl] = wave_sum(ga[gl]); gq[gl] = wave_sum(gq[gl]); }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int gl = 0; gl < GN_GB; ++gl) { sm_red[threadIdx.x >> 6][2 * gl] = ga[gl]; sm_red[threadIdx.x >> 6][2 * gl + 1] = gq[gl]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < GN_GB && g0 + (int)threadIdx.x < G) {
    const int gl = threadIdx.x, g = g0 + gl;
    float sa = 0.f, sq = 0.f;
    for (int w = 0; w < 4; ++w) { sa += sm_red[w][2 * gl]; sq += sm_red[w][2 * gl + 1]; }
    if (!BWD) {
      const float n = (float)(r1 - r0) * (float)cpg;
      float piv = 0.f;
#pragma unroll
      for (int k = 0; k < GN_GB; ++k) piv = gl == k ? pg[k] : piv;
      float* o = part + ((size_t)(b * G + g) * 3) * S + s;
      o[0] = n; o[S] = piv + sa / n; o[2 * S] = sq - sa * sa / n;
    } else {
      part[((size_t)(b * G + g) * S + s) * 2] = sa;
      part[((size_t)(b * G + g) * S + s) * 2 + 1] = sq;
    }
  }
}

// combine the S slices of every group of batch item b into sm_stats[g] = (mean, rstd).  Every (n, mean, M2) triple of
// the lane is loaded before the first use: one memory round trip instead of two dependent ones (the partials were
// written by the previous kernel on other XCDs, so each trip goes to memory)
constexpr int GN_COMBINE_K = 8;      // slices per lane: S <= 64
__device__ __forceinline__ void gn_combine(const float* part, int b, int G, int S, float eps, float2* sm_stats,
                                           float* stats_out) {
  const int g = threadIdx.x >> 3, sub = threadIdx.x & 7;
  const bool act = g < G;
  const float* p = part + ((size_t)(b * G + (act ? g : 0)) * 3) * S;
  float cn[GN_COMBINE_K], cm[GN_COMBINE_K], cq[GN_COMBINE_K];
#pragma unroll
  for (int k = 0; k < GN_COMBINE_K; ++k) {
    const int s = sub + 8 * k;
    const bool on = act && s < S;
    cn[k] = on ? p[s] : 0.f;
    cm[k] = on ? p[S + s] : 0.f;
    cq[k] = on ? p[2 * S + s] : 0.f;
  }
  float n = 0.f, sm = 0.f;
#pragma unroll
  for (int k = 0; k < GN_COMBINE_K; ++k) { n += cn[k]; sm += cn[k] * cm[k]; }
  n = oct_sum(n); sm = oct_sum(sm);
  const float mean = sm / n;
  float m2 = 0.f;
#pragma unroll
  for (int k = 0; k < GN_COMBINE_K; ++k) { const float d = cm[k] - mean; m2 += cq[k] + cn[k] * d * d; }
  m2 = oct_sum(m2);
  if (act && sub == 0) {
    const float rstd = rsqrtf(m2 / n + eps);
    sm_stats[g] = make_float2(mean, rstd);
    if (stats_out) { stats_out[2 * (b * G + g)] = mean; stats_out[2 * (b * G + g) + 1] = rstd; }
  }
  __syncthreads();
}

// thread = 8 consecutive channels of one pixel, `iters` such chunks per thread (blockDim apart); grid (blocks, B).
// Every workgroup first merges the slice statistics of its image (a dependent chain of small loads): on big tensors
// `iters` > 1 spreads that prologue over more elements (one chunk per thread left the batched passes at 2.4 TB/s).
template <class T, bool WIDE>      // WIDE: >= 8 channels per group
__global__ void __launch_bounds__(256) k_gn_apply(const T* x, const float* gamma, const float* beta, const float* part,
                                                  float* stats, T* y, int HW, int C, int G, int S, float eps, int silu,
                                                  int iters) {
  __shared__ float2 sm_stats[64];
  const int b = blockIdx.y;
  const int cchunks = C / 8, cpg = C / G;
  const float inv_cpg = 1.f / (float)cpg;
  const unsigned total = (unsigned)HW * cchunks;                      // 32-bit element indices (a 64-bit division per chunk is ~100 instructions)
  unsigned idx = blockIdx.x * iters * blockDim.x + threadIdx.x;
  bool live = idx < total;
  unsigned pix = live ? idx / cchunks : 0;
  size_t row = (size_t)b * HW + pix;
  int c0 = (int)(idx - pix * cchunks) * 8;
  uint4 raw = *reinterpret_cast<const uint4*>(x + row * C + c0);      // in flight under the slice combine
  float gmv[8], btv[8];                                               // so are the affine parameters of the 8 channels
  *reinterpret_cast<float4*>(gmv) = *reinterpret_cast<const float4*>(gamma + c0);
  *reinterpret_cast<float4*>(gmv + 4) = *reinterpret_cast<const float4*>(gamma + c0 + 4);
  *reinterpret_cast<float4*>(btv) = *reinterpret_cast<const float4*>(beta + c0);
  *reinterpret_cast<float4*>(btv + 4) = *reinterpret_cast<const float4*>(beta + c0 + 4);
  gn_combine(part, b, G, S, eps, sm_stats, blockIdx.x == 0 ? stats : nullptr);
  for (int it = 0; it < iters; ++it) {
    if (!live) return;                                                // chunks are ascending: nothing live follows
    uint4 nraw = raw;
    float ngm[8], nbt[8];
    const unsigned nidx = idx + blockDim.x;
    const bool nlive = it + 1 < iters && nidx < total;
    size_t nrow = row;
    int nc0 = c0;
    if (nlive) {                                                      // next chunk's loads fly under this chunk's math
      const unsigned npix = nidx / cchunks;
      nrow = (size_t)b * HW + npix;
      nc0 = (int)(nidx - npix * cchunks) * 8;
      nraw = *reinterpret_cast<const uint4*>(x + nrow * C + nc0);
      *reinterpret_cast<float4*>(ngm) = *reinterpret_cast<const float4*>(gamma + nc0);
      *reinterpret_cast<float4*>(ngm + 4) = *reinterpret_cast<const float4*>(gamma + nc0 + 4);
      *reinterpret_cast<float4*>(nbt) = *reinterpret_cast<const float4*>(beta + nc0);
      *reinterpret_cast<float4*>(nbt + 4) = *reinterpret_cast<const float4*>(beta + nc0 + 4);
    }
    typedef T T8 __attribute__((ext_vector_type(8)));
    const T8 xv = __builtin_bit_cast(T8, raw);
    T8 o;
    // the 8 channels lie in at most two groups (cpg >= 8): one reciprocal multiply instead of eight integer divisions
    // (the divisions made this kernel VALU-bound: 2.4 TB/s on big tensors)
    const int g0 = (int)(((float)c0 + 0.5f) * inv_cpg), bnd = (g0 + 1) * cpg - c0;
    const float2 st0 = sm_stats[g0], st1 = sm_stats[g0 + 1 < G ? g0 + 1 : g0];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float2 st = WIDE ? (i < bnd ? st0 : st1) : sm_stats[(c0 + i) / cpg];
      float z = (to_f32<T>(xv[i]) - st.x) * st.y * gmv[i] + btv[i];
      if (silu) z = silu_f(z);
      o[i] = from_f32<T>(z);
    }
    *reinterpret_cast<uint4*>(y + row * C + c0) = __builtin_bit_cast(uint4, o);
    if (!nlive) return;
    idx = nidx; row = nrow; c0 = nc0; raw = nraw;
#pragma unroll
    for (int i = 0; i < 8; ++i) { gmv[i] = ngm[i]; btv[i] = nbt[i]; }
  }
}

// chunks per thread of the two apply kernels: keep >= ~1024 workgroups in flight
static inline int gn_apply_iters(size_t blocks_total) {
  size_t it = blocks_total / 1024;
  return it < 1 ? 1 : (it > 8 ? 8 : (int)it);
}

void launch_groupnorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats,
                          float* scratch, int B, int HW, int C, int G, float eps, int silu, hipStream_t st,
                          int have_partials) {
  DH_ABLATE(2);
  // have_partials > 1: the producer was a GEMM whose epilogue left that many slices per group (two per row tile, gemm.hip gn_epi)
  const int S = have_partials > 1 ? have_partials : gn_slices(HW, B);
  const size_t blocks0 = ((size_t)HW * (C / 8) + 255) / 256;
  const int iters = gn_apply_iters(blocks0 * B);
  dim3 g1(S, cdiv(G, GN_GB), B), g2((unsigned)((blocks0 + iters - 1) / iters), B);
  if (dtype == DH_DTYPE_F16) {
    if (!have_partials) hipLaunchKernelGGL((k_gn_partial<f16, false>), g1, dim3(256), 0, st, (const f16*)x, (const f16*)nullptr, gamma, beta, (const float*)nullptr, scratch, HW, C, G, S, silu);
    if (C / G >= 8) hipLaunchKernelGGL((k_gn_apply<f16, true>), g2, dim3(256), 0, st, (const f16*)x, gamma, beta, scratch, stats, (f16*)y, HW, C, G, S, eps, silu, iters);
    else hipLaunchKernelGGL((k_gn_apply<f16, false>), g2, dim3(256), 0, st, (const f16*)x, gamma, beta, scratch, stats, (f16*)y, HW, C, G, S, eps, silu, iters);
  } else {
    if (!have_partials) hipLaunchKernelGGL((k_gn_partial<bf16, false>), g1, dim3(256), 0, st, (const bf16*)x, (const bf16*)nullptr, gamma, beta, (const float*)nullptr, scratch, HW, C, G, S, silu);
    if (C / G >= 8) hipLaunchKernelGGL((k_gn_apply<bf16, true>), g2, dim3(256), 0, st, (const bf16*)x, gamma, beta, scratch, stats, (bf16*)y, HW, C, G, S, eps, silu, iters);
    else hipLaunchKernelGGL((k_gn_apply<bf16, false>), g2, dim3(256), 0, st, (const bf16*)x, gamma, beta, scratch, stats, (bf16*)y, HW, C, G, S, eps, silu, iters);
  }
}

template <class T, bool WIDE>
__global__ void __launch_bounds__(256) k_gn_bwd_apply(const T* x, const T* dy, const float* gamma, const float* beta,
                                                      const float* stats, const float* part, T* dx, int HW, int C, int G,
                                                      int S, int silu, int accumulate, int iters, T* split0, T* split1,
                                                      int split_c) {
  __shared__ float4 sm_st[64];     // mean, rstd, mean(dxhat), mean(dxhat*xhat)
  const int b = blockIdx.y, cpg = C / G;
  const float inv_cpg = 1.f / (float)cpg;
  const int cchunks = C / 8;
  const unsigned total = (unsigned)HW * cchunks;                      // 32-bit element indices (a 64-bit division per chunk is ~100 instructions)
  unsigned idx = blockIdx.x * iters * blockDim.x + threadIdx.x;
  bool live = idx < total;
  unsigned pix = live ? idx / cchunks : 0;
  size_t row = (size_t)b * HW + pix;
  int c0 = (int)(idx - pix * cchunks) * 8;
  uint4 rx = *reinterpret_cast<const uint4*>(x + row * C + c0);       // in flight under the slice combine
  uint4 rd = *reinterpret_cast<const uint4*>(dy + row * C + c0);
  uint4 ro = make_uint4(0, 0, 0, 0);
  if (accumulate) ro = *reinterpret_cast<const uint4*>(dx + row * C + c0);
  float gmv[8], btv[8];
  *reinterpret_cast<float4*>(gmv) = *reinterpret_cast<const float4*>(gamma + c0);
  *reinterpret_cast<float4*>(gmv + 4) = *reinterpret_cast<const float4*>(gamma + c0 + 4);
  *reinterpret_cast<float4*>(btv) = *reinterpret_cast<const float4*>(beta + c0);
  *reinterpret_cast<float4*>(btv + 4) = *reinterpret_cast<const float4*>(beta + c0 + 4);
  {   // slice sums of (sum d, sum d * xhat): 8 lanes per group, every pair loaded before the first add (G <= 32, S <= 64)
    const int g = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const bool act = g < G;
    const float2* p = reinterpret_cast<const float2*>(part) + (size_t)(b * G + (act ? g : 0)) * S;
    float2 v[GN_COMBINE_K];
#pragma unroll
    for (int k = 0; k < GN_COMBINE_K; ++k) {
      const int s = sub + 8 * k;
      v[k] = act && s < S ? p[s] : make_float2(0.f, 0.f);
    }
    float2 st = act ? *reinterpret_cast<const float2*>(stats + 2 * (b * G + g)) : make_float2(0.f, 0.f);
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < GN_COMBINE_K; ++k) { a += v[k].x; c += v[k].y; }
    a = oct_sum(a); c = oct_sum(c);
    const float inv = 1.f / ((float)HW * (float)cpg);
    if (act && sub == 0) sm_st[g] = make_float4(st.x, st.y, a * inv, c * inv);
  }
  __syncthreads();
  typedef T T8 __attribute__((ext_vector_type(8)));
  for (int it = 0; it < iters; ++it) {
    if (!live) return;
    const unsigned nidx = idx + blockDim.x;
    const bool nlive = it + 1 < iters && nidx < total;
    size_t nrow = row;
    int nc0 = c0;
    uint4 nrx = rx, nrd = rd, nro = ro;
    float ngm[8], nbt[8];
    if (nlive) {                                                      // next chunk's loads fly under this chunk's math
      const unsigned npix = nidx / cchunks;
      nrow = (size_t)b * HW + npix;
      nc0 = (int)(nidx - npix * cchunks) * 8;
      nrx = *reinterpret_cast<const uint4*>(x + nrow * C + nc0);
      nrd = *reinterpret_cast<const uint4*>(dy + nrow * C + nc0);
      if (accumulate) nro = *reinterpret_cast<const uint4*>(dx + nrow * C + nc0);
      *reinterpret_cast<float4*>(ngm) = *reinterpret_cast<const float4*>(gamma + nc0);
      *reinterpret_cast<float4*>(ngm + 4) = *reinterpret_cast<const float4*>(gamma + nc0 + 4);
      *reinterpret_cast<float4*>(nbt) = *reinterpret_cast<const float4*>(beta + nc0);
      *reinterpret_cast<float4*>(nbt + 4) = *reinterpret_cast<const float4*>(beta + nc0 + 4);
    }
    const T8 xv = __builtin_bit_cast(T8, rx), dv = __builtin_bit_cast(T8, rd), ov = __builtin_bit_cast(T8, ro);
    T8 o;
    const int g0 = (int)(((float)c0 + 0.5f) * inv_cpg), bnd = (g0 + 1) * cpg - c0;     // see k_gn_apply
    const float4 st0 = sm_st[g0], st1 = sm_st[g0 + 1 < G ? g0 + 1 : g0];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 st = WIDE ? (i < bnd ? st0 : st1) : sm_st[(c0 + i) / cpg];
      const float xh = (to_f32<T>(xv[i]) - st.x) * st.y;
      float d = to_f32<T>(dv[i]);
      if (silu) d *= silu_grad(xh * gmv[i] + btv[i]);
      d *= gmv[i];
      float r = st.y * (d - st.z - xh * st.w);
      if (accumulate) r += to_f32<T>(ov[i]);
      o[i] = from_f32<T>(r);
    }
    // (split0 != NULL: x is a concatenation and the gradient goes straight to its two sources; split_c % 8 == 0)
    T* dst = !split0 ? dx + row * C + c0
                     : (c0 < split_c ? split0 + row * split_c + c0 : split1 + row * (C - split_c) + (c0 - split_c));
    *reinterpret_cast<uint4*>(dst) = __builtin_bit_cast(uint4, o);
    if (!nlive) return;
    idx = nidx; row = nrow; c0 = nc0; rx = nrx; rd = nrd; ro = nro;
#pragma unroll
    for (int i = 0; i < 8; ++i) { gmv[i] = ngm[i]; btv[i] = nbt[i]; }
  }
}

void launch_groupnorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* beta,
                          const float* stats, void* dx, float* scratch, int B, int HW, int C, int G, int silu,
                          int accumulate, hipStream_t st, int have_partials, GnBwdSplit split) {
  DH_ABLATE(2);
  // have_partials > 1: the slices were left by a GEMM epilogue, that many per (image, group) (gemm.hip gn_epi == 2)
  const int S = have_partials > 1 ? have_partials : gn_slices(HW, B);
  const size_t blocks0 = ((size_t)HW * (C / 8) + 255) / 256;
  const int iters = gn_apply_iters(blocks0 * B);
  dim3 g1(S, cdiv(G, GN_GB), B), g2((unsigned)((blocks0 + iters - 1) / iters), B);
  if (dtype == DH_DTYPE_F16) {
    if (!have_partials) hipLaunchKernelGGL((k_gn_partial<f16, true>), g1, dim3(256), 0, st, (const f16*)x, (const f16*)dy, gamma, beta, stats, scratch, HW, C, G, S, silu);
    if (C / G >= 8) hipLaunchKernelGGL((k_gn_bwd_apply<f16, true>), g2, dim3(256), 0, st, (const f16*)x, (const f16*)dy, gamma, beta, stats, scratch, (f16*)dx, HW, C, G, S, silu, accumulate, iters, (f16*)split.out0, (f16*)split.out1, split.split_c);
    else hipLaunchKernelGGL((k_gn_bwd_apply<f16, false>), g2, dim3(256), 0, st, (const f16*)x, (const f16*)dy, gamma, beta, stats, scratch, (f16*)dx, HW, C, G, S, silu, accumulate, iters, (f16*)split.out0, (f16*)split.out1, split.split_c);
  } else {
    if (!have_partials) hipLaunchKernelGGL((k_gn_partial<bf16, true>), g1, dim3(256), 0, st, (const bf16*)x, (const bf16*)dy, gamma, beta, stats, scratch, HW, C, G, S, silu);
    if (C / G >= 8) hipLaunchKernelGGL((k_gn_bwd_apply<bf16, true>), g2, dim3(256), 0, st, (const bf16*)x, (const bf16*)dy, gamma, beta, stats, scratch, (bf16*)dx, HW, C, G, S, silu, accumulate, iters, (bf16*)split.out0, (bf16*)split.out1, split.split_c);
    else hipLaunchKernelGGL((k_gn_bwd_apply<bf16, false>), g2, dim3(256), 0, st, (const bf16*)x, (const bf16*)dy, gamma, beta, stats, scratch, (bf16*)dx, HW, C, G, S, silu, accumulate, iters, (bf16*)split.out0, (bf16*)split.out1, split.split_c);
  }
}

// concat(a, b) along channels fused with the GroupNorm slice statistics of the result (the up-path resnets
// normalise the concatenation right away): same workgroup shape and arithmetic as k_gn_partial<fwd>.
template <class T>
__global__ void __launch_bounds__(256) k_concat_gn(const T* a, int Ca, const T* bsrc, int Cb, T* out, float* part, int HW, int G,
                                                   int S) {
  __shared__ float sm_red[4][2 * GN_GB];
  const int C = Ca + Cb;
  const int s = blockIdx.x, g0 = blockIdx.y * GN_GB, b = blockIdx.z, cpg = div_small(C, rcp_fast(G));
  const int W = GN_GB * cpg, nch = W >> 3;
  const float inv_nch = rcp_fast(nch), inv_S = rcp_fast(S);
  const int RP = div_small((int)blockDim.x, inv_nch);
  const int r0 = div_small(HW * s, inv_S), r1 = div_small(HW * (s + 1), inv_S);
  const int rr = div_small((int)threadIdx.x, inv_nch), ch = threadIdx.x - rr * nch;
  auto load8 = [&](size_t m, int n) {          // n: channel of the result, multiple of 8; a chunk never straddles a | b
    return n < Ca ? *reinterpret_cast<const uint4*>(a + m * Ca + n) : *reinterpret_cast<const uint4*>(bsrc + m * Cb + (n - Ca));
  };
  float ga[GN_GB], gq[GN_GB], pg[GN_GB];
#pragma unroll
  for (int gl = 0; gl < GN_GB; ++gl) {
    ga[gl] = 0.f; gq[gl] = 0.f; pg[gl] = 0.f;
    if (g0 + gl < G) {
      const int n = (g0 + gl) * cpg;
      const size_t m = (size_t)b * HW + r0;
      pg[gl] = to_f32<T>(n < Ca ? a[m * Ca + n] : bsrc[m * Cb + (n - Ca)]);
    }
  }
  if (rr < RP && g0 * cpg + ch * 8 < C) {
    const int n = g0 * cpg + ch * 8;
    int gi[8];
    const float inv_cpg = rcp_fast(cpg);
    float av[8], qv[8], pv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      gi[i] = div_small(ch * 8 + i, inv_cpg);
      av[i] = 0.f; qv[i] = 0.f; pv[i] = 0.f;
#pragma unroll
      for (int gl = 0; gl < GN_GB; ++gl) pv[i] = gi[i] == gl ? pg[gl] : pv[i];
    }
    for (int r = r0 + rr; r < r1; r += RP) {
      const size_t m = (size_t)b * HW + r;
      const uint4 raw = load8(m, n);
      *reinterpret_cast<uint4*>(out + m * C + n) = raw;
      const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = to_f32<T>(v[i]) - pv[i]; av[i] += d; qv[i] += d * d; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int gl = 0; gl < GN_GB; ++gl) {
        ga[gl] += gi[i] == gl ? av[i] : 0.f;
        gq[gl] += gi[i] == gl ? qv[i] : 0.f;
      }
  }
#pragma unroll
  for (int gl = 0; gl < GN_GB; ++gl) { ga[gl] = wave_sum(ga[gl]); gq[gl] = wave_sum(gq[gl]); }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int gl = 0; gl < GN_GB; ++gl) { sm_red[threadIdx.x >> 6][2 * gl] = ga[gl]; sm_red[threadIdx.x >> 6][2 * gl + 1] = gq[gl]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < GN_GB && g0 + (int)threadIdx.x < G) {
    const int gl = threadIdx.x, g = g0 + gl;
    float sa = 0.f, sq = 0.f;
    for (int w = 0; w < 4; ++w) { sa += sm_red[w][2 * gl]; sq += sm_red[w][2 * gl + 1]; }
    const float n = (float)(r1 - r0) * (float)cpg;
    float piv = 0.f;
#pragma unroll
    for (int k = 0; k < GN_GB; ++k) piv = gl == k ? pg[k] : piv;
    float* o = part + ((size_t)(b * G + g) * 3) * S + s;
    o[0] = n; o[S] = piv + sa / n; o[2 * S] = sq - sa * sa / n;
  }
}

void launch_concat_gn(int dtype, const void* a, int Ca, const void* b, int Cb, void* out, float* gn_part, int B, int HW,
                      int G, hipStream_t st) {
  DH_ABLATE(8);
  const int S = gn_slices(HW, B);
  dim3 grid(S, cdiv(G, GN_GB), B);
  if (dtype == DH_DTYPE_F16)
    hipLaunchKernelGGL((k_concat_gn<f16>), grid, dim3(256), 0, st, (const f16*)a, Ca, (const f16*)b, Cb, (f16*)out, gn_part, HW, G, S);
  else
    hipLaunchKernelGGL((k_concat_gn<bf16>), grid, dim3(256), 0, st, (const bf16*)a, Ca, (const bf16*)b, Cb, (bf16*)out, gn_part, HW, G, S);
}

// ----------------------------------------------------------------------------- LayerNorm
// one wave per row; C/8 sixteen-byte chunks spread over the lanes (C <= 4096)
constexpr int LN_MAXCH = 8;

template <class T, int NCH>      // NCH: 8-element chunks per lane (C <= 512 * NCH)
__global__ void __launch_bounds__(256, NCH <= 3 ? 4 : 1) k_ln_fwd(const T* x, const float* gamma, const float* beta, T* y, float* stats, int rows, int C,
                         float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = C / 8;
  uint4 raw[NCH];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) {
      raw[k] = *reinterpret_cast<const uint4*>(x + (size_t)row * C + ch * 8);
      const T* v = reinterpret_cast<const T*>(&raw[k]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += to_f32<T>(v[i]);
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) {
      const T* v = reinterpret_cast<const T*>(&raw[k]);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = to_f32<T>(v[i]) - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0 && stats) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) {
      const T* v = reinterpret_cast<const T*>(&raw[k]);
      T o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = ch * 8 + i;
        o[i] = from_f32<T>((to_f32<T>(v[i]) - mean) * rstd * gamma[c] + beta[c]);
      }
      *reinterpret_cast<uint4*>(y + (size_t)row * C + ch * 8) = *reinterpret_cast<uint4*>(o);
    }
  }
}

// One wave per row.  x and dy of the row stay in registers as loaded (16-bit); gamma and the residual gradient are
// fetched where they are used, so the kernel fits 128 VGPRs and four waves share a SIMD (the version that kept every
// operand of the row live needed 272 registers: one wave per SIMD, 0.9 TB/s on the batched passes).
template <class T, int NCH>      // NCH: 8-element chunks per lane (C <= 512 * NCH)
__global__ void __launch_bounds__(256, NCH == 1 ? 8 : (NCH <= 3 ? 4 : 1)) k_ln_bwd(const T* x, const T* dy, const float* gamma, const float* stats,
                                                                  const T* add, T* dx, int rows, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = C / 8;
  const size_t base = (size_t)row * C;
  uint4 rx[NCH], rd[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) {
      rx[k] = *reinterpret_cast<const uint4*>(x + base + ch * 8);
      rd[k] = *reinterpret_cast<const uint4*>(dy + base + ch * 8);
    }
  }
  const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) {
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(gamma + ch * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const T* xv = reinterpret_cast<const T*>(&rx[k]);
      const T* dv = reinterpret_cast<const T*>(&rd[k]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (to_f32<T>(xv[i]) - mean) * rstd;
        const float d = to_f32<T>(dv[i]) * g[i];
        s1 += d;
        s2 += d * xh;
      }
    }
  }
  s1 = wave_sum(s1) / (float)C;
  s2 = wave_sum(s2) / (float)C;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) {
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(gamma + ch * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      uint4 ra = make_uint4(0, 0, 0, 0);
      if (add) ra = *reinterpret_cast<const uint4*>(add + base + ch * 8);
      const T* xv = reinterpret_cast<const T*>(&rx[k]);
      const T* dv = reinterpret_cast<const T*>(&rd[k]);
      const T* av = reinterpret_cast<const T*>(&ra);
      T o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (to_f32<T>(xv[i]) - mean) * rstd;
        const float d = to_f32<T>(dv[i]) * g[i];
        float r = rstd * (d - s1 - xh * s2);
        if (add) r += to_f32<T>(av[i]);
        o[i] = from_f32<T>(r);
      }
      *reinterpret_cast<uint4*>(dx + base + ch * 8) = *reinterpret_cast<uint4*>(o);
    }
  }
}

// The same backward with dy = the sum of `splits` f32 slabs [splits][rows][C] of a split-K input-gradient GEMM (no bias, no
// residual): the reduce launch and the LayerNorm backward launch in one kernel, dy never goes to memory.  The sum is
// rounded to the storage type first, so the result equals the two-kernel path bit for bit.
template <class T, int NCH>
__global__ void __launch_bounds__(256, NCH == 1 ? 4 : (NCH <= 3 ? 2 : 1)) k_splitk_reduce_ln_bwd(const float* part, int splits, const T* x, const float* gamma,
                                                                       const float* stats, const T* add, T* dx, int rows, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = C / 8;
  const size_t base = (size_t)row * C, slab = (size_t)rows * C;
  uint4 rx[NCH];
  float d[NCH][8];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) rx[k] = *reinterpret_cast<const uint4*>(x + base + ch * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[k][i] = 0.f;
  }
  // the slabs in split order (the order of k_splitk_reduce), four splits' loads in flight per chunk
  int z = 0;
  for (; z + 4 <= splits; z += 4) {
    float4 v[NCH][4][2];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int ch = lane + 64 * k;
      if (ch < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* p = part + (size_t)(z + j) * slab + base + ch * 8;
          v[k][j][0] = *reinterpret_cast<const float4*>(p);
          v[k][j][1] = *reinterpret_cast<const float4*>(p + 4);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int ch = lane + 64 * k;
      if (ch < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          d[k][0] += v[k][j][0].x; d[k][1] += v[k][j][0].y; d[k][2] += v[k][j][0].z; d[k][3] += v[k][j][0].w;
          d[k][4] += v[k][j][1].x; d[k][5] += v[k][j][1].y; d[k][6] += v[k][j][1].z; d[k][7] += v[k][j][1].w;
        }
      }
    }
  }
  for (; z < splits; ++z) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int ch = lane + 64 * k;
      if (ch < nch) {
        const float* p = part + (size_t)z * slab + base + ch * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + 4);
        d[k][0] += a0.x; d[k][1] += a0.y; d[k][2] += a0.z; d[k][3] += a0.w;
        d[k][4] += a1.x; d[k][5] += a1.y; d[k][6] += a1.z; d[k][7] += a1.w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) d[k][i] = to_f32<T>(from_f32<T>(d[k][i]));
  const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) {
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(gamma + ch * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const T* xv = reinterpret_cast<const T*>(&rx[k]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (to_f32<T>(xv[i]) - mean) * rstd;
        d[k][i] *= g[i];
        s1 += d[k][i];
        s2 += d[k][i] * xh;
      }
    }
  }
  s1 = wave_sum(s1) / (float)C;
  s2 = wave_sum(s2) / (float)C;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) {
      uint4 ra = make_uint4(0, 0, 0, 0);
      if (add) ra = *reinterpret_cast<const uint4*>(add + base + ch * 8);
      const T* xv = reinterpret_cast<const T*>(&rx[k]);
      const T* av = reinterpret_cast<const T*>(&ra);
      T o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (to_f32<T>(xv[i]) - mean) * rstd;
        float r = rstd * (d[k][i] - s1 - xh * s2);
        if (add) r += to_f32<T>(av[i]);
        o[i] = from_f32<T>(r);
      }
      *reinterpret_cast<uint4*>(dx + base + ch * 8) = *reinterpret_cast<uint4*>(o);
    }
  }
}
void launch_splitk_reduce_ln_bwd(int dtype, const float* partial, int splits, const void* x, const float* gamma,
                                 const float* stats, const void* add, void* dx, int rows, int C, hipStream_t st) {
  // one row per wave; few rows (the 16x16 / 8x8 levels): one-wave blocks so that every row gets its own CU
  const int wpb = rows <= 1024 ? 1 : 4;      // (backward pass -0.4 %)
#define DH_RLN(TT, N) hipLaunchKernelGGL((k_splitk_reduce_ln_bwd<TT, N>), dim3(cdiv(rows, wpb)), dim3(64 * wpb), 0, st, partial, splits, (const TT*)x, gamma, stats, (const TT*)add, (TT*)dx, rows, C)
  const int n = C <= 512 ? 1 : (C <= 1024 ? 2 : (C <= 1536 ? 3 : LN_MAXCH));
  if (dtype == DH_DTYPE_F16) {
    if (n == 1) DH_RLN(f16, 1); else if (n == 2) DH_RLN(f16, 2); else if (n == 3) DH_RLN(f16, 3); else DH_RLN(f16, LN_MAXCH);
  } else {
    if (n == 1) DH_RLN(bf16, 1); else if (n == 2) DH_RLN(bf16, 2); else if (n == 3) DH_RLN(bf16, 3); else DH_RLN(bf16, LN_MAXCH);
  }
#undef DH_RLN
}

void launch_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats,
                          int rows, int C, float eps, hipStream_t st) {
  DH_ABLATE(1);
#define DH_LN_FWD(TT, N) hipLaunchKernelGGL((k_ln_fwd<TT, N>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const TT*)x, gamma, beta, (TT*)y, stats, rows, C, eps)
  const int n = C <= 512 ? 1 : (C <= 1024 ? 2 : (C <= 1536 ? 3 : LN_MAXCH));
  if (dtype == DH_DTYPE_F16) {
    if (n == 1) DH_LN_FWD(f16, 1); else if (n == 2) DH_LN_FWD(f16, 2); else if (n == 3) DH_LN_FWD(f16, 3); else DH_LN_FWD(f16, LN_MAXCH);
  } else {
    if (n == 1) DH_LN_FWD(bf16, 1); else if (n == 2) DH_LN_FWD(bf16, 2); else if (n == 3) DH_LN_FWD(bf16, 3); else DH_LN_FWD(bf16, LN_MAXCH);
  }
#undef DH_LN_FWD
}
void launch_layernorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* stats,
                          const void* add, void* dx, int rows, int C, hipStream_t st) {
  DH_ABLATE(1);
  const int wpb = rows <= 1024 ? 1 : 4;      // (backward pass -0.4 %)
#define DH_LN_BWD(TT, N) hipLaunchKernelGGL((k_ln_bwd<TT, N>), dim3(cdiv(rows, wpb)), dim3(64 * wpb), 0, st, (const TT*)x, (const TT*)dy, gamma, stats, (const TT*)add, (TT*)dx, rows, C)
  const int n = C <= 512 ? 1 : (C <= 1024 ? 2 : (C <= 1536 ? 3 : LN_MAXCH));
  if (dtype == DH_DTYPE_F16) {
    if (n == 1) DH_LN_BWD(f16, 1); else if (n == 2) DH_LN_BWD(f16, 2); else if (n == 3) DH_LN_BWD(f16, 3); else DH_LN_BWD(f16, LN_MAXCH);
  } else {
    if (n == 1) DH_LN_BWD(bf16, 1); else if (n == 2) DH_LN_BWD(bf16, 2); else if (n == 3) DH_LN_BWD(bf16, 3); else DH_LN_BWD(bf16, LN_MAXCH);
  }
#undef DH_LN_BWD
}

// --------------------------------------------------------------------------------- GEGLU

template <class T>
__global__ void k_geglu_fwd(const T* x, T* y, size_t rows, int F) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int fch = F / 8;
  if (idx >= rows * fch) return;
  const unsigned row32 = (unsigned)idx / (unsigned)fch;          // the grid (a 32-bit count of 256-thread blocks) bounds idx below 2^40; tensors here are < 2^32 elements: 32-bit division
  const size_t row = row32;
  const int j0 = (int)((unsigned)idx - row32 * (unsigned)fch) * 8;
  uint4 rh = *reinterpret_cast<const uint4*>(x + row * 2 * F + glu_col(j0, 0));     // (paired layout: 8 outputs lie inside one half block)
  uint4 rg = *reinterpret_cast<const uint4*>(x + row * 2 * F + glu_col(j0, 1));
  const T* h = reinterpret_cast<const T*>(&rh);
  const T* g = reinterpret_cast<const T*>(&rg);
  T o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = from_f32<T>(to_f32<T>(h[i]) * gelu_f(to_f32<T>(g[i])));
  *reinterpret_cast<uint4*>(y + row * F + j0) = *reinterpret_cast<uint4*>(o);
}

template <class T>
__global__ void k_geglu_bwd(const T* x, const T* dy, T* dx, size_t rows, int F) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int fch = F / 8;
  if (idx >= rows * fch) return;
  const unsigned row32 = (unsigned)idx / (unsigned)fch;          // the grid (a 32-bit count of 256-thread blocks) bounds idx below 2^40; tensors here are < 2^32 elements: 32-bit division
  const size_t row = row32;
  const int j0 = (int)((unsigned)idx - row32 * (unsigned)fch) * 8;
  uint4 rh = *reinterpret_cast<const uint4*>(x + row * 2 * F + glu_col(j0, 0));
  uint4 rg = *reinterpret_cast<const uint4*>(x + row * 2 * F + glu_col(j0, 1));
  uint4 rd = *reinterpret_cast<const uint4*>(dy + row * F + j0);
  const T* h = reinterpret_cast<const T*>(&rh);
  const T* g = reinterpret_cast<const T*>(&rg);
  const T* d = reinterpret_cast<const T*>(&rd);
  T oh[8], og[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float gv = to_f32<T>(g[i]), dv = to_f32<T>(d[i]);
    oh[i] = from_f32<T>(dv * gelu_f(gv));
    og[i] = from_f32<T>(dv * to_f32<T>(h[i]) * gelu_grad(gv));
  }
  *reinterpret_cast<uint4*>(dx + row * 2 * F + glu_col(j0, 0)) = *reinterpret_cast<uint4*>(oh);
  *reinterpret_cast<uint4*>(dx + row * 2 * F + glu_col(j0, 1)) = *reinterpret_cast<uint4*>(og);
}

void launch_geglu_fwd(int dtype, const void* x, void* y, int rows, int F, hipStream_t st) {
  DH_ABLATE(4);
  const unsigned nb = (unsigned)(((size_t)rows * (F / 8) + 255) / 256);
  if (dtype == DH_DTYPE_F16) hipLaunchKernelGGL((k_geglu_fwd<f16>), dim3(nb), dim3(256), 0, st, (const f16*)x, (f16*)y, (size_t)rows, F);
  else hipLaunchKernelGGL((k_geglu_fwd<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)x, (bf16*)y, (size_t)rows, F);
}
void launch_geglu_bwd(int dtype, const void* x, const void* dy, void* dx, int rows, int F, hipStream_t st) {
  DH_ABLATE(4);
  const unsigned nb = (unsigned)(((size_t)rows * (F / 8) + 255) / 256);
  if (dtype == DH_DTYPE_F16) hipLaunchKernelGGL((k_geglu_bwd<f16>), dim3(nb), dim3(256), 0, st, (const f16*)x, (const f16*)dy, (f16*)dx, (size_t)rows, F);
  else hipLaunchKernelGGL((k_geglu_bwd<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)x, (const bf16*)dy, (bf16*)dx, (size_t)rows, F);
}

// ---------------------------------------------------------------------------------- misc
template <class T>
__global__ void k_copy_cols(const T* src, long lds_, T* dst, long ldd, size_t rows, int cols, int accumulate) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cch = cols / 8;
  if (idx >= rows * cch) return;
  const unsigned row32 = (unsigned)idx / (unsigned)cch;
  const size_t row = row32;
  const int c0 = (int)((unsigned)idx - row32 * (unsigned)cch) * 8;
  uint4 v = *reinterpret_cast<const uint4*>(src + row * lds_ + c0);
  if (accumulate) {
    uint4 o = *reinterpret_cast<const uint4*>(dst + row * ldd + c0);
    const T* a = reinterpret_cast<const T*>(&v);
    const T* b = reinterpret_cast<const T*>(&o);
    T r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = from_f32<T>(to_f32<T>(a[i]) + to_f32<T>(b[i]));
    v = *reinterpret_cast<uint4*>(r);
  }
  *reinterpret_cast<uint4*>(dst + row * ldd + c0) = v;
}
void launch_copy_cols(int dtype, const void* src, long lds_, void* dst, long ldd, int rows, int cols, int accumulate,
                      hipStream_t st) {
  DH_ABLATE(8);
  const unsigned nb = (unsigned)(((size_t)rows * (cols / 8) + 255) / 256);
  if (dtype == DH_DTYPE_F16) hipLaunchKernelGGL((k_copy_cols<f16>), dim3(nb), dim3(256), 0, st, (const f16*)src, lds_, (f16*)dst, ldd, (size_t)rows, cols, accumulate);
  else hipLaunchKernelGGL((k_copy_cols<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)src, lds_, (bf16*)dst, ldd, (size_t)rows, cols, accumulate);
}

// backward of a channel concat in ONE launch: src [rows][colsA + colsB] -> dstA (=|+=) left part, dstB (=|+=) right part
template <class T>
__global__ void k_split_cols(const T* src, long lds_, T* dstA, long ldA, int colsA, int accA, T* dstB, long ldB, int colsB,
                             int accB, size_t rows) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cch = (colsA + colsB) / 8;
  if (idx >= rows * cch) return;
  const unsigned row32 = (unsigned)idx / (unsigned)cch;
  const size_t row = row32;
  const int c0 = (int)((unsigned)idx - row32 * (unsigned)cch) * 8;
  uint4 v = *reinterpret_cast<const uint4*>(src + row * lds_ + c0);
  const bool left = c0 < colsA;
  T* dst = left ? dstA + row * ldA + c0 : dstB + row * ldB + (c0 - colsA);
  if (left ? accA : accB) {
    uint4 o = *reinterpret_cast<const uint4*>(dst);
    const T* a = reinterpret_cast<const T*>(&v);
    const T* b = reinterpret_cast<const T*>(&o);
    T r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = from_f32<T>(to_f32<T>(a[i]) + to_f32<T>(b[i]));
    v = *reinterpret_cast<uint4*>(r);
  }
  *reinterpret_cast<uint4*>(dst) = v;
}
void launch_split_cols(int dtype, const void* src, long lds_, void* dstA, long ldA, int colsA, int accA, void* dstB, long ldB,
                       int colsB, int accB, int rows, hipStream_t st) {
  DH_ABLATE(8);
  const unsigned nb = (unsigned)(((size_t)rows * ((colsA + colsB) / 8) + 255) / 256);
  if (dtype == DH_DTYPE_F16)
    hipLaunchKernelGGL((k_split_cols<f16>), dim3(nb), dim3(256), 0, st, (const f16*)src, lds_, (f16*)dstA, ldA, colsA, accA, (f16*)dstB, ldB, colsB, accB, (size_t)rows);
  else
    hipLaunchKernelGGL((k_split_cols<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)src, lds_, (bf16*)dstA, ldA, colsA, accA, (bf16*)dstB, ldB, colsB, accB, (size_t)rows);
}

template <class T>
__global__ void k_pool2x2(const T* src, T* dst, int B, int h, int w, int C, int accumulate) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cch = C / 8;
  if (idx >= (size_t)B * h * w * cch) return;
  const unsigned pix = (unsigned)idx / (unsigned)cch;
  const int c0 = (int)((unsigned)idx - pix * (unsigned)cch) * 8;
  const int b = (int)(pix / (unsigned)(h * w));
  const int r = (int)(pix - (unsigned)b * (unsigned)(h * w)), y = r / w, x = r - y * w;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      uint4 v = *reinterpret_cast<const uint4*>(src + (((size_t)b * 2 * h + 2 * y + dy) * 2 * w + 2 * x + dx) * C + c0);
      const T* a = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += to_f32<T>(a[i]);
    }
  if (accumulate) {
    uint4 v = *reinterpret_cast<const uint4*>(dst + pix * C + c0);
    const T* a = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += to_f32<T>(a[i]);
  }
  T o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = from_f32<T>(acc[i]);
  *reinterpret_cast<uint4*>(dst + pix * C + c0) = *reinterpret_cast<uint4*>(o);
}
void launch_pool2x2_sum(int dtype, const void* src, void* dst, int B, int h, int w, int C, int accumulate,
                        hipStream_t st) {
  DH_ABLATE(8);
  const unsigned nb = (unsigned)(((size_t)B * h * w * (C / 8) + 255) / 256);
  if (dtype == DH_DTYPE_F16) hipLaunchKernelGGL((k_pool2x2<f16>), dim3(nb), dim3(256), 0, st, (const f16*)src, (f16*)dst, B, h, w, C, accumulate);
  else hipLaunchKernelGGL((k_pool2x2<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, B, h, w, C, accumulate);
}

template <class T>
__global__ void k_f32_to_t(const float* s, T* d, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = from_f32<T>(s[i]);
}
template <class T>
__global__ void k_t_to_f32(const T* s, float* d, size_t n, int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = accumulate ? d[i] + to_f32<T>(s[i]) : to_f32<T>(s[i]);
}
void launch_f32_to_t(int dtype, const float* src, void* dst, size_t n, hipStream_t st) {
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (dtype == DH_DTYPE_F16) hipLaunchKernelGGL((k_f32_to_t<f16>), dim3(nb), dim3(256), 0, st, src, (f16*)dst, n);
  else hipLaunchKernelGGL((k_f32_to_t<bf16>), dim3(nb), dim3(256), 0, st, src, (bf16*)dst, n);
}
void launch_t_to_f32(int dtype, const void* src, float* dst, size_t n, int accumulate, hipStream_t st) {
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (dtype == DH_DTYPE_F16) hipLaunchKernelGGL((k_t_to_f32<f16>), dim3(nb), dim3(256), 0, st, (const f16*)src, dst, n, accumulate);
  else hipLaunchKernelGGL((k_t_to_f32<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)src, dst, n, accumulate);
}

// Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos(t f_i) | sin(t f_i)], f_i = 10000^(-i/half)
template <class T>
__global__ void k_timestep(const float* tp, int dim, int B, T* out) {
  const float t = *tp;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= B * dim) return;
  const int j = i % dim, k = j < half ? j : j - half;
  const float f = expf(-9.210340371976184f * (float)k / (float)half);
  const float a = t * f;
  out[i] = from_f32<T>(j < half ? cosf(a) : sinf(a));
}
__global__ void k_set_scalar(float* p, float v) { *p = v; }
void launch_set_scalar(float* p, float v, hipStream_t st) { hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, st, p, v); }
void launch_timestep_embedding(int dtype, const float* t, int dim, int B, void* out, hipStream_t st) {
  if (dtype == DH_DTYPE_F16) hipLaunchKernelGGL((k_timestep<f16>), dim3(cdiv(B * dim, 256)), dim3(256), 0, st, t, dim, B, (f16*)out);
  else hipLaunchKernelGGL((k_timestep<bf16>), dim3(cdiv(B * dim, 256)), dim3(256), 0, st, t, dim, B, (bf16*)out);
}

// y = gelu(x), 8 elements per thread (n % 8 == 0)
template <class T>
__global__ void __launch_bounds__(256) k_gelu(const T* x, T* y, size_t n8) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint4 raw = *reinterpret_cast<const uint4*>(x + i * 8);
  const T* v = reinterpret_cast<const T*>(&raw);
  T o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = from_f32<T>(gelu_f(to_f32<T>(v[k])));
  *reinterpret_cast<uint4*>(y + i * 8) = *reinterpret_cast<uint4*>(o);
}
void launch_gelu(int dtype, const void* x, void* y, size_t n, hipStream_t st) {
  const size_t n8 = n / 8;
  const unsigned nb = (unsigned)((n8 + 255) / 256);
  if (dtype == DH_DTYPE_F16) hipLaunchKernelGGL((k_gelu<f16>), dim3(nb), dim3(256), 0, st, (const f16*)x, (f16*)y, n8);
  else hipLaunchKernelGGL((k_gelu<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)x, (bf16*)y, n8);
}

__global__ void k_lane_ops_probe(const float* in, float* out, unsigned* ex) {
  const int lane = threadIdx.x;
  const float v = in[lane];
  out[lane] = wave_sum(v);
  out[64 + lane] = wave_max(v);
  out[128 + lane] = oct_sum(v);
  out[192 + lane] = xor32_sum(v);
  out[256 + lane] = xor32_max(v);
  const uint4 e = half_exchange(make_uint2(4u * lane, 4u * lane + 1), make_uint2(4u * lane + 2, 4u * lane + 3));
  ex[4 * lane] = e.x; ex[4 * lane + 1] = e.y; ex[4 * lane + 2] = e.z; ex[4 * lane + 3] = e.w;
}
void launch_lane_ops_probe(const float* in, float* out, unsigned* ex, hipStream_t st) {
  hipLaunchKernelGGL(k_lane_ops_probe, dim3(1), dim3(64), 0, st, in, out, ex);
}

}  // namespace dh
