// Test hooks: C-ABI wrappers around the individual kernel launchers so that the GPU parity
// tests can check each kernel against a torch fp32 reference in isolation.  Not used by the
// product path.
#include <stdlib.h>

#include "unet_kernels.h"
using namespace dh;

static void* g_dbg_tiled = nullptr;      // the tiled copy of the last weight matrix dh_dbg_gemm was given

// timing runs (tools/bench_gemm_warmth.py): one streaming read over the tiled weights, i.e. what a prefetch would leave in the caches
__global__ void k_dbg_touch(const uint4* p, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x9e3779b9u && sink) sink[0] = acc;
}
extern "C" int dh_dbg_touch_tiled(size_t bytes, void* stream) {
  DH_REQUIRE(g_dbg_tiled != nullptr, "no tiled weights yet (call dh_dbg_gemm first)");
  hipLaunchKernelGGL(k_dbg_touch, dim3(512), dim3(256), 0, (hipStream_t)stream, (const uint4*)g_dbg_tiled, bytes / 16, (unsigned*)nullptr);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_dbg_gemm(int dtype, const void* A, long lda, const void* W, int M, int N, int K, int mode, int Hin,
                           int Win, int Cin, int Hout, int Wout, int stride, int up, const float* bias,
                           const float* rowvec, int rowvec_ld, int rows_per_batch, const void* R, long ldr, void* C,
                           long ldc, int act_silu, float* partial, size_t partial_elems, void* stream) {
  DH_REQUIRE(A && W && C && K % 64 == 0 && N % 64 == 0, "bad arguments (N, K must be multiples of 64)");
  // the hook takes plain [N][K] weights and tiles them into a scratch buffer first
  void*& tiled = g_dbg_tiled;
  static size_t tiled_cap = 0;
  static const void* tiled_from = nullptr;
  const size_t need = (size_t)N * K * 2;
  if (need > tiled_cap) {
    if (tiled) (void)hipFree(tiled);
    DH_CHECK_HIP(hipMalloc(&tiled, need));
    tiled_cap = need;
    tiled_from = nullptr;
  }
  static const bool pretiled = getenv("DH_DBG_PRETILED") != nullptr;   // timing runs: tile a weight matrix once, not per call
  if (!pretiled || tiled_from != W) launch_tile_weights(dtype, W, tiled, N, K, (hipStream_t)stream, mode != 0 ? Cin : 0);
  tiled_from = W;
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = tiled; g.M = M; g.N = N; g.K = K; g.mode = mode; g.Hin = Hin; g.Win = Win; g.Cin = Cin;
  g.Hout = Hout; g.Wout = Wout; g.stride = stride; g.up = up; g.bias = bias; g.rowvec = rowvec; g.rowvec_ld = rowvec_ld;
  g.rows_per_batch = rows_per_batch; g.R = R; g.ldr = ldr; g.C = C; g.ldc = ldc; g.act_silu = act_silu;
  g.partial = partial; g.partial_elems = partial_elems;
  launch_gemm(dtype, g, (hipStream_t)stream);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
// GEMM (dense or 3x3 convolution) followed by the GroupNorm(+SiLU) of its output, the way the engine's forward runs the pair: the
// GroupNorm's slice statistics come from the GEMM launch when it can leave them (split-K reduce, or -- round 6 -- the epilogue of an
// unsplit k_gemm_dma launch: gemm.hip gn_epi) and from the statistics kernel otherwise.  *have_out = what the GEMM reported
// (0 = statistics kernel, 1 = reduce, > 1 = slices left by the epilogue); stats [B * G][2] = (mean, rstd).
extern "C" int dh_dbg_gemm_groupnorm(int dtype, const void* A, long lda, const void* W, int M, int N, int K, int mode, int Hin, int Win,
                                     int Cin, const float* bias, void* C, float* partial, size_t partial_elems, int HW, int G,
                                     const float* gamma, const float* beta, float eps, int silu, void* Y, float* stats, float* scratch,
                                     int* have_out, void* stream) {
  DH_REQUIRE(A && W && C && Y && stats && scratch && gamma && beta && K % 64 == 0 && N % 64 == 0 && HW > 0 && M % HW == 0, "bad arguments");
  static void* tiled = nullptr;        // (this hook's own tiled copy: dh_dbg_gemm keeps the capacity of ITS buffer)
  static size_t cap = 0;
  const size_t need = (size_t)N * K * 2;
  if (need > cap) {
    if (tiled) (void)hipFree(tiled);
    DH_CHECK_HIP(hipMalloc(&tiled, need));
    cap = need;
  }
  launch_tile_weights(dtype, W, tiled, N, K, (hipStream_t)stream, mode != 0 ? Cin : 0);
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = tiled; g.M = M; g.N = N; g.K = K; g.mode = mode; g.Hin = Hin; g.Win = Win; g.Cin = Cin;
  g.Hout = Hin; g.Wout = Win; g.stride = 1; g.up = 0; g.bias = bias; g.C = C; g.ldc = N;
  g.partial = partial; g.partial_elems = partial_elems;
  int have = 0;
  g.gn_part = scratch; g.gn_HW = HW; g.gn_G = G; g.gn_done = &have;
  launch_gemm(dtype, g, (hipStream_t)stream);
  launch_groupnorm_fwd(dtype, C, gamma, beta, Y, stats, scratch, M / HW, HW, N, G, eps, silu, (hipStream_t)stream, have);
  if (have_out) *have_out = have;
  DH_LAUNCH_CHECK();
  return DH_OK;
}
// ... and the backward twin: C = A W^T is dy of GroupNorm(x) (+ SiLU) with saved (mean, rstd) in `stats`; the GEMM (its split-K
// reduce, or its own epilogue: *have_out > 1 = that many slices per group) leaves the backward slice statistics in `scratch` and
// the GroupNorm backward writes dx.  (engine: every input-gradient GEMM in front of a GroupNorm backward)
extern "C" int dh_dbg_gemm_groupnorm_bwd(int dtype, const void* A, long lda, const void* W, int M, int N, int K, int mode, int Hin,
                                         int Win, int Cin, void* C, float* partial, size_t partial_elems, int HW, int G, const void* x,
                                         const float* gamma, const float* beta, const float* stats, int silu, void* dx,
                                         float* scratch, int* have_out, void* stream) {
  DH_REQUIRE(A && W && C && x && dx && stats && scratch && gamma && beta && K % 64 == 0 && N % 64 == 0 && HW > 0 && M % HW == 0, "bad arguments");
  static void* tiled = nullptr;
  static size_t cap = 0;
  const size_t need = (size_t)N * K * 2;
  if (need > cap) {
    if (tiled) (void)hipFree(tiled);
    DH_CHECK_HIP(hipMalloc(&tiled, need));
    cap = need;
  }
  launch_tile_weights(dtype, W, tiled, N, K, (hipStream_t)stream, mode != 0 ? Cin : 0);
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = tiled; g.M = M; g.N = N; g.K = K; g.mode = mode; g.Hin = Hin; g.Win = Win; g.Cin = Cin;
  g.Hout = Hin; g.Wout = Win; g.stride = 1; g.up = 0; g.C = C; g.ldc = N;
  g.partial = partial; g.partial_elems = partial_elems;
  int have = 0;
  g.gnb_x = x; g.gnb_ldx = N; g.gnb_gamma = gamma; g.gnb_beta = beta; g.gnb_stats = stats; g.gnb_silu = silu;
  g.gn_part = scratch; g.gn_HW = HW; g.gn_G = G; g.gn_done = &have;
  launch_gemm(dtype, g, (hipStream_t)stream);
  launch_groupnorm_bwd(dtype, x, C, gamma, beta, stats, dx, scratch, M / HW, HW, N, G, silu, 0, (hipStream_t)stream, have, GnBwdSplit());
  if (have_out) *have_out = have;
  DH_LAUNCH_CHECK();
  return DH_OK;
}
// the LayerNorm-folded form of the dense GEMM (engine: qkv / cross-attention q / GEGLU in-projection at B <= 3): A is the
// LayerNorm INPUT, W already carries gamma, ln_s[n] = sum_k W[n][k], ln_t[n] = sum_k beta[k] W0[n][k] (+ bias);
// out = rstd (A W^T - mean ln_s) + ln_t, and (mean, rstd) per row land in ln_stats
extern "C" int dh_dbg_gemm_lnfold(int dtype, const void* A, long lda, const void* W, int M, int N, int K, const float* ln_s,
                                  const float* ln_t, float* ln_stats, float ln_eps, void* C, long ldc, void* stream) {
  DH_REQUIRE(A && W && C && ln_s && ln_t && ln_stats && K % 64 == 0 && N % 64 == 0, "bad arguments (N, K must be multiples of 64)");
  static void* tiled = nullptr;
  static size_t tiled_cap = 0;
  const size_t need = (size_t)N * K * 2;
  if (need > tiled_cap) {
    if (tiled) (void)hipFree(tiled);
    DH_CHECK_HIP(hipMalloc(&tiled, need));
    tiled_cap = need;
  }
  launch_tile_weights(dtype, W, tiled, N, K, (hipStream_t)stream);
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = tiled; g.M = M; g.N = N; g.K = K; g.mode = A_DENSE; g.C = C; g.ldc = ldc;
  g.ln_s = ln_s; g.ln_t = ln_t; g.ln_stats = ln_stats; g.ln_eps = ln_eps;
  launch_gemm(dtype, g, (hipStream_t)stream);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
// the GEGLU epilogues of the dense GEMM (engine: ff.net.0.proj forward, ff.net.2 input-gradient).  bwd = 0: W [N = 2F][K] and bias
// in the PAIRED row order (unet_kernels.h glu_col), C (may be NULL) receives the pre-activations [M][2F] (paired), y [M][F] =
// value * gelu(gate).  bwd = 1: the GEMM's tile A W^T [M][N = F] is dy of a GEGLU with saved pre-activations x [M][2F] (paired);
// dx [M][2F] (paired) receives d_value | d_gate.
extern "C" int dh_dbg_gemm_glu(int dtype, int bwd, const void* A, long lda, const void* W, int M, int N, int K, const float* bias,
                               void* C, void* y, const void* x, void* dx, void* stream) {
  DH_REQUIRE(A && W && K % 64 == 0 && N % 128 == 0, "bad arguments (K % 64, N % 128)");
  DH_REQUIRE(bwd ? (x && dx) : (y != nullptr), "missing output");
  static void* tiled = nullptr;
  static size_t tiled_cap = 0;
  const size_t need = (size_t)N * K * 2;
  if (need > tiled_cap) {
    if (tiled) (void)hipFree(tiled);
    DH_CHECK_HIP(hipMalloc(&tiled, need));
    tiled_cap = need;
  }
  launch_tile_weights(dtype, W, tiled, N, K, (hipStream_t)stream);
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = tiled; g.M = M; g.N = N; g.K = K; g.mode = A_DENSE; g.bias = bias;
  if (bwd) { g.glub_x = x; g.glub_dx = dx; }
  else { g.C = C; g.ldc = N; g.glu_y = y; g.glu_ldy = N / 2; }
  launch_gemm(dtype, g, (hipStream_t)stream);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
extern "C" int dh_dbg_groupnorm(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                const void* dy, void* dx, float* scratch, int B, int HW, int C, int G, float eps,
                                int silu, int accumulate, void* stream) {
  launch_groupnorm_fwd(dtype, x, gamma, beta, y, stats, scratch, B, HW, C, G, eps, silu, (hipStream_t)stream);
  if (dy && dx)
    launch_groupnorm_bwd(dtype, x, dy, gamma, beta, stats, dx, scratch, B, HW, C, G, silu, accumulate, (hipStream_t)stream);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
extern "C" int dh_dbg_layernorm(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                const void* dy, const void* add, void* dx, int rows, int C, float eps, void* stream) {
  launch_layernorm_fwd(dtype, x, gamma, beta, y, stats, rows, C, eps, (hipStream_t)stream);
  if (dy && dx) launch_layernorm_bwd(dtype, x, dy, gamma, stats, add, dx, rows, C, (hipStream_t)stream);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
extern "C" int dh_dbg_geglu(int dtype, const void* x, void* y, const void* dy, void* dx, int rows, int F, void* stream) {
  launch_geglu_fwd(dtype, x, y, rows, F, (hipStream_t)stream);
  if (dy && dx) launch_geglu_bwd(dtype, x, dy, dx, rows, F, (hipStream_t)stream);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
extern "C" int dh_dbg_attention(int dtype, const void* q, long ldq, const void* k, const void* v, long ldk, void* o,
                                long ldo, float* lse, const void* d_o, float* delta, void* dq, void* dk, void* dv,
                                int B, int H, int Nq, int Nk, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  launch_attention_fwd(dtype, q, ldq, k, v, ldk, o, ldo, lse, B, H, Nq, Nk, st);
  if (d_o) {
    if (dq) launch_attention_bwd_dq(dtype, q, ldq, k, v, ldk, o, ldo, d_o, ldo, lse, delta, dq, ldq, B, H, Nq, Nk, st);
    else launch_attention_delta(dtype, o, ldo, d_o, ldo, delta, B, H, Nq, st);
    if (dk && dv) {
      static float* scratch = nullptr;                 // test hook only: lets the query-chunk path of the few-key case run
      constexpr size_t kScratch = (size_t)16 << 20;
      if (!scratch && hipMalloc((void**)&scratch, kScratch * sizeof(float)) != hipSuccess) scratch = nullptr;
      launch_attention_bwd_dkv(dtype, q, ldq, k, v, ldk, d_o, ldo, lse, delta, dk, dv, ldk, B, H, Nq, Nk, st, scratch,
                               scratch ? kScratch : 0);
    }
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}
// attention backward of one layer, dQ and dK/dV either one after the other on `stream` (stream2 == NULL: dQ writes delta, dK/dV
// reads it) or SIDE BY SIDE: delta by its own kernel, then dQ on `stream` and dK/dV on `stream2` between a fork and a join event
// (measurement hook: what the two kernels gain from each other's idle CUs at B = 1)
extern "C" int dh_dbg_attention_bwd_pair(int dtype, const void* q, long ldq, const void* k, const void* v, long ldk, const void* o,
                                         long ldo, const float* lse, const void* d_o, float* delta, void* dq, void* dk, void* dv,
                                         int B, int H, int Nq, int Nk, void* stream, void* stream2) {
  hipStream_t st = (hipStream_t)stream, s2 = (hipStream_t)stream2;
  if (!s2) {
    launch_attention_bwd_dq(dtype, q, ldq, k, v, ldk, o, ldo, d_o, ldo, lse, delta, dq, ldq, B, H, Nq, Nk, st);
    launch_attention_bwd_dkv(dtype, q, ldq, k, v, ldk, d_o, ldo, lse, delta, dk, dv, ldk, B, H, Nq, Nk, st, nullptr, 0);
  } else {
    static hipEvent_t fork = nullptr, join = nullptr;
    if (!fork) { DH_CHECK_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); DH_CHECK_HIP(hipEventCreateWithFlags(&join, hipEventDisableTiming)); }
    launch_attention_delta(dtype, o, ldo, d_o, ldo, delta, B, H, Nq, st);
    DH_CHECK_HIP(hipEventRecord(fork, st));
    DH_CHECK_HIP(hipStreamWaitEvent(s2, fork, 0));
    launch_attention_bwd_dkv(dtype, q, ldq, k, v, ldk, d_o, ldo, lse, delta, dk, dv, ldk, B, H, Nq, Nk, s2, nullptr, 0);
    launch_attention_bwd_dq(dtype, q, ldq, k, v, ldk, o, ldo, d_o, ldo, lse, nullptr, dq, ldq, B, H, Nq, Nk, st);
    DH_CHECK_HIP(hipEventRecord(join, s2));
    DH_CHECK_HIP(hipStreamWaitEvent(st, join, 0));
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}
extern "C" int dh_dbg_pool2x2(int dtype, const void* src, void* dst, int B, int h, int w, int C, int accumulate, void* stream) {
  launch_pool2x2_sum(dtype, src, dst, B, h, w, C, accumulate, (hipStream_t)stream);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
extern "C" int dh_dbg_lane_ops(const float* in, float* out, unsigned* ex, void* stream) {
  launch_lane_ops_probe(in, out, ex, (hipStream_t)stream);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
