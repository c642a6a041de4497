// Launchers of the U-Net kernels (unet_kernels.hip / attention.hip).  All activations are
// channels-last 16-bit ([rows][C], dtype = DH_DTYPE_F16 or DH_DTYPE_BF16), f32 accumulate.
#pragma once
#include "common.h"

namespace dh {

__device__ inline float silu_grad(float z) {
  const float s = 1.f / (1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}

// GEGLU (reference model/attention.py:345-400 FeedForward with diffusers' GEGLU [ext]: h * gelu(gate), erf form).
// Phi(g) = 0.5 erfc(-g / sqrt 2) from the Abramowitz-Stegun 7.1.26 rational form (|error| <= 1.5e-7, far below a 16-bit
// rounding): one v_exp, one v_rcp and five FMAs, and the SAME exponential exp(-g^2 / 2) is the density the derivative
// needs -- about a third of erff()'s instruction count, which is what lets the activation live in a GEMM epilogue.
// The negative tail uses 0.5 erfc(|x|) directly (no 1 - erf cancellation).
struct GeluParts { float Phi, pdf; };      // Phi(g), exp(-g^2/2) / sqrt(2 pi)
__device__ __forceinline__ GeluParts gelu_parts(float g) {
  const float x = fabsf(g) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.f));
  const float e = __expf(-x * x);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float q = 0.5f * poly * e;                       // 0.5 erfc(|x|)
  return {g < 0.f ? q : 1.f - q, e * 0.3989422804014327f};
}
__device__ __forceinline__ float gelu_f(float g) { return g * gelu_parts(g).Phi; }
__device__ __forceinline__ float gelu_grad(float g) { const GeluParts p = gelu_parts(g); return fmaf(g, p.pdf, p.Phi); }
// Column layout of a GEGLU pre-activation tensor [rows][2F] (and of its gradient): "paired" -- every 32-column block holds
// the 16 value columns of outputs 16b .. 16b+15 followed by their 16 gate columns, so that the lane of the GEMM epilogue that
// owns value columns {8g + 4hi ..+3} (g = 0, 1) also owns their gates (g = 2, 3): the activation is lane-local, and the
// value / gate rows of ff.net.0.proj are permuted accordingly at load time (glu_col).  F % 16 == 0.
__host__ __device__ inline int glu_col(int o, int gate) { return 32 * (o >> 4) + 16 * gate + (o & 15); }
// fused-weight row (= stored column) n' -> row of the torch parameter [2F][K] (value rows first, then gate rows)
__host__ __device__ inline int glu_src_row(int n, int F) { return ((n >> 4) & 1) * F + 16 * (n >> 5) + (n & 15); }

enum { A_DENSE = 0, A_CONV3 = 1, A_CONVT2 = 2 };

struct GemmArgs {
  const void* A = nullptr; long lda = 0;   // dense: row stride; conv: pixel stride (elements)
  const void* W = nullptr;                 // TILED weights: [N/64][K/64] swizzled 64x64 tiles (wt_index in gemm.hip)
  int M = 0, N = 0, K = 0;
  int mode = A_DENSE;
  int Hin = 0, Win = 0, Cin = 0;           // source tensor spatial size, channels per tap
  int Hout = 0, Wout = 0;                  // output spatial size (M = B*Hout*Wout)
  int stride = 1, up = 0;
  int pad = 1;                             // A_CONV3: zero padding on the top / left (1 = the usual 'same' 3x3; 0 with stride 2 = the
                                           // VAE encoder's down-sampler, which pads bottom / right only)
  const float* bias = nullptr;             // [N]
  const float* rowvec = nullptr; int rowvec_ld = 0; int rows_per_batch = 1;
  const void* R = nullptr; long ldr = 0;   // residual added after everything else
  void* C = nullptr; long ldc = 0;
  int act_silu = 0;
  float* partial = nullptr; size_t partial_elems = 0;   // split-K scratch (f32)
  // optional: the output feeds a GroupNorm next.  When the launch goes through the split-K reduce, that kernel
  // also leaves the GroupNorm slice statistics (gn_partial layout) and *gn_done is set to 1.
  float* gn_part = nullptr; int gn_HW = 0, gn_G = 0; int* gn_done = nullptr;
  // gnb_x != NULL: the output is the gradient dy of a GroupNorm(+SiLU) whose INPUT is gnb_x; the reduce then leaves the
  // backward slice statistics (sum d, sum d*xhat with d = dy * gamma * act') in gn_part (k_gn_partial<bwd> layout)
  const void* gnb_x = nullptr; long gnb_ldx = 0; const float *gnb_gamma = nullptr, *gnb_beta = nullptr, *gnb_stats = nullptr;
  int gnb_silu = 0;
  // LayerNorm folded into the GEMM (dense, no split-K): A = the LayerNorm input, W = W * gamma (load-time fold),
  // ln_s[n] = sum_k W'[n][k], ln_t[n] = sum_k beta[k] W[n][k] + bias[n]; the row statistics are written to ln_stats [M][2]
  const float* ln_s = nullptr; const float* ln_t = nullptr; float* ln_stats = nullptr; float ln_eps = 1e-5f;
  // lnb_x != NULL: the output (dense, full rows of width N, nothing accumulated into it) is the gradient dy of a LayerNorm
  // whose INPUT is lnb_x.  When the launch goes through the split-K reduce, that kernel applies the LayerNorm backward to
  // the summed rows and writes lnb_dx (+ lnb_add) instead of C; *lnb_done is set to 1 and the caller skips the LayerNorm op.
  const void* lnb_x = nullptr; const float *lnb_gamma = nullptr, *lnb_stats = nullptr; const void* lnb_add = nullptr;
  void* lnb_dx = nullptr; int* lnb_done = nullptr;
  // GEGLU in the epilogue (dense, no split-K; N = 2F in the paired column layout, see glu_col):
  //   glu_y != NULL (forward, the ff.net.0.proj GEMM): y[m][o] = h * gelu(gate) of the ROUNDED pre-activations goes to glu_y
  //   ([M][F], row stride glu_ldy); C may be NULL then (pre-activations not saved: no backward follows);
  //   glub_x != NULL (the input-gradient GEMM of ff.net.2, N = F): the tile is dy of the GEGLU whose saved pre-activations are
  //   glub_x [M][2F]; d_value = dy gelu(gate), d_gate = dy h gelu'(gate) go to glub_dx [M][2F] (paired layout), C is not written
  void* glu_y = nullptr; long glu_ldy = 0;
  const void* glub_x = nullptr; void* glub_dx = nullptr;
};
// split-K reduce + LayerNorm backward in one pass over the slabs (f32 [splits][rows][C]); dy is rounded to the storage type
// before it is used, exactly as the reduce + k_ln_bwd pair does
void launch_splitk_reduce_ln_bwd(int dtype, const float* partial, int splits, const void* x, const float* gamma,
                                 const float* stats, const void* add, void* dx, int rows, int C, hipStream_t st);
// D[m][n] = sum_k A(m,k) W[n][k] (+bias, +rowvec, silu, +R); returns algorithmic flops
double launch_gemm(int dtype, const GemmArgs& a, hipStream_t st);
bool gemm_profiling_on();   // HIP-event bracket active (bench roofline pass): graphs are bypassed
// plain [N][K] -> tiled weight layout (test hooks); N, K multiples of 64
// (conv_cin > 0: src is in the (tap, channel) order of a 3x3 kernel and lands in the conv_k_index order)
void launch_tile_weights(int dtype, const void* src, void* dst, int N, int K, hipStream_t st, int conv_cin = 0);
// element offset of (n, k) in the tiled weight layout: [N/64][K/64] tiles of 64x64 halves (8 KiB,
// contiguous), each stored as the swizzled LDS image (16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7))
__host__ __device__ inline size_t wt_index(int n, int k, int K) {
  const int r = n & 63, c = (k & 63) >> 3;
  return ((size_t)(n >> 6) * (K >> 6) + (k >> 6)) * 4096 + (size_t)r * 64 + (size_t)((c ^ ((r >> 1) & 7)) << 3) + (k & 7);
}

// K index of (tap, channel) in a 3x3 implicit GEMM.  64-channel chunks outermost, the nine taps of a chunk on consecutive
// K tiles, channel inside the chunk innermost: the shifted rows of a chunk are fetched again by the next eight K tiles while
// they are still in the XCD's L2 (tap-major order re-read them one whole channel sweep later: 20 - 40 tiles of 24 - 56 KB per
// workgroup in between, which does not fit the 4 MB once an XCD holds 20+ workgroups).  C must be a multiple of 64.
__host__ __device__ inline int conv_k_index(int tap, int c) { return (c >> 6) * 576 + tap * 64 + (c & 63); }

// torch-layout f32 parameter [N][C][taps] -> weight storage (unet_engine.cpp): tiled 16-bit (dtype F16 / BF16) or plain f32
void launch_load_weight(int dtype, const float* src, int N, int C, int taps, void* fwd, long fwd_K, long row_off, void* bwd,
                        long bwd_K, long col_off, int Nb, float scale, hipStream_t st);

// direct small convolutions (conv_in: Cin=5 -> C; conv_out: C -> 4) and their input-gradients
void launch_conv_small_fwd(int dtype, const void* x, int x_is_f32, const float* w, const float* bias, void* y,
                           int y_is_f32, int B, int H, int W, int Cin, int Cout, hipStream_t st);
void launch_conv_small_bwd(int dtype, const void* dy, int dy_is_f32, const float* w, void* dx, int dx_is_f32,
                           int accumulate, int B, int H, int W, int Cin, int Cout, hipStream_t st);

// GroupNorm (+SiLU): y = act(gn(x));  stats = [B*G][2] (mean, rstd) saved for backward.
// have_partials != 0: the slice statistics in `scratch` were already left by the producer of x
// (split-K reduce / concat: 1 = gn_slices(HW, B) slices; a GEMM epilogue: the slice count itself, > 1), only the apply kernel runs.
void launch_groupnorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats,
                          float* scratch, int B, int HW, int C, int G, float eps, int silu, hipStream_t st,
                          int have_partials = 0);
int gn_slices(int HW, int B);   // number of row slices of the GroupNorm statistics for HW rows per image, B images
constexpr int GN_GB = 4;        // groups per statistics workgroup
// out[m][0..Ca) = a[m], out[m][Ca..Ca+Cb) = b[m], plus the GroupNorm slice statistics of out (G groups)
void launch_concat_gn(int dtype, const void* a, int Ca, const void* b, int Cb, void* out, float* gn_part, int B, int HW,
                      int G, hipStream_t st);
// dx (=|+=) d gn-act / d x
// (the gradient of a concatenation written in place of the concatenated gradient: columns [0, split_c) of every row go to
// split0 (row stride split_c), the rest to split1 (row stride C - split_c); `accumulate` still reads the addend from dx)
struct GnBwdSplit { void* out0 = nullptr; void* out1 = nullptr; int split_c = 0; };
void launch_groupnorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* beta,
                          const float* stats, void* dx, float* scratch, int B, int HW, int C, int G, int silu,
                          int accumulate, hipStream_t st, int have_partials = 0, GnBwdSplit split = GnBwdSplit());
// y = gelu(x) (erf form), n elements (the text tower's MLP)
void launch_gelu(int dtype, const void* x, void* y, size_t n, hipStream_t st);
// LayerNorm over C per row; stats [rows][2]
void launch_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats,
                          int rows, int C, float eps, hipStream_t st);
// dx = ln_bwd(dy) (+ add)   (add may alias nothing; dx written)
void launch_layernorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* stats,
                          const void* add, void* dx, int rows, int C, hipStream_t st);
// GEGLU: y[m][j] = h * gelu(g), h = x[m][glu_col(j, 0)], g = x[m][glu_col(j, 1)] (paired column layout)
void launch_geglu_fwd(int dtype, const void* x, void* y, int rows, int F, hipStream_t st);
void launch_geglu_bwd(int dtype, const void* x, const void* dy, void* dx, int rows, int F, hipStream_t st);
// misc
void launch_copy_cols(int dtype, const void* src, long lds, void* dst, long ldd, int rows, int cols, int accumulate,
                      hipStream_t st);                       // dst[r][0..cols) (=|+=) src[r][0..cols)
// src [rows][colsA + colsB] split into dstA (=|+=) and dstB (=|+=) in one launch (concat backward)
void launch_split_cols(int dtype, const void* src, long lds, void* dstA, long ldA, int colsA, int accA, void* dstB, long ldB,
                       int colsB, int accB, int rows, hipStream_t st);
// test hook: the cross-lane helpers of common.h on one wave (out[0..63] wave_sum, [64..] wave_max, [128..] oct_sum,
// [192..] xor32_sum, [256..] xor32_max; ex[lane] = half_exchange of (a = {4 lane, 4 lane + 1}, b = {4 lane + 2, 4 lane + 3}))
void launch_lane_ops_probe(const float* in, float* out, unsigned* ex, hipStream_t st);
void launch_pool2x2_sum(int dtype, const void* src, void* dst, int B, int h, int w, int C, int accumulate,
                        hipStream_t st);                     // dst[b][y][x] (=|+=) sum of the 2x2 block of src
void launch_f32_to_t(int dtype, const float* src, void* dst, size_t n, hipStream_t st);
void launch_t_to_f32(int dtype, const void* src, float* dst, size_t n, int accumulate, hipStream_t st);
void launch_timestep_embedding(int dtype, const float* t_dev, int dim, int B, void* out, hipStream_t st);   // [B][dim] = [cos|sin]
void launch_set_scalar(float* p, float v, hipStream_t st);

// flash attention, head dim 64.  q [B*Nq][ldq], k/v [B*Nk][ldk] (head h at column h*64),
// o [B*Nq][ldo]; lse [B][H][Nq] f32 (natural log)
void launch_attention_fwd(int dtype, const void* q, long ldq, const void* k, const void* v, long ldk, void* o,
                          long ldo, float* lse, int B, int H, int Nq, int Nk, hipStream_t st, int causal = 0);
// delta[b][h][q] = sum_d dO*O
void launch_attention_delta(int dtype, const void* o, long ldo, const void* d_o, long lddo, float* delta, int B,
                            int H, int Nq, hipStream_t st);
// dq also WRITES delta[b][h][q] = sum_d dO*O (read by the dkv kernel launched after it) unless delta is NULL
// (then launch_attention_delta has produced it and dq / dkv may run concurrently)
void launch_attention_bwd_dq(int dtype, const void* q, long ldq, const void* k, const void* v, long ldk, const void* o,
                             long ldo, const void* d_o, long lddo, const float* lse, float* delta, void* dq, long lddq,
                             int B, int H, int Nq, int Nk, hipStream_t st);
void launch_attention_bwd_dkv(int dtype, const void* q, long ldq, const void* k, const void* v, long ldk,
                              const void* d_o, long lddo, const float* lse, const float* delta, void* dk, void* dv,
                              long lddk, int B, int H, int Nq, int Nk, hipStream_t st, float* scratch = nullptr,
                              size_t scratch_elems = 0);   // scratch (f32): lets the few-key case split the queries over workgroups

}  // namespace dh
