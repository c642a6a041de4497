// Kernel-side argument block and MFMA wrappers shared by the GEMM translation units (gemm.hip: k_gemm_dma, the tile families of
// rounds 1-4; gemm_pp.hip: k_gemm_pp, the eight-wave ping-pong main loop of round 5 for grids that fill the chip).
#pragma once
#include "unet_kernels.h"

namespace dh {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <class T> struct Mfma;
template <> struct Mfma<f16> {
  static __device__ __forceinline__ v16f run(uint4 a, uint4 b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  }
};
template <> struct Mfma<bf16> {
  static __device__ __forceinline__ v16f run(uint4 a, uint4 b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8b, a), __builtin_bit_cast(v8b, b), c, 0, 0, 0);
  }
};

struct GemmK {   // kernel-side copy of GemmArgs (plain data)
  const void* A; long lda;
  const void* W;
  int M, N, K;
  int mode, Hin, Win, Cin, Hout, Wout, stride, up, pad;
  const float* bias;
  const float* rowvec; int rowvec_ld; int rows_per_batch; float inv_rows_per_batch;
  const void* R; long ldr;
  void* C; long ldc;
  int act_silu;
  int pre_r;        // fetch the residual tile before the K loop
  int wide_store;   // 16-byte epilogue stores (N % 32 == 0, C and ldc 16-byte aligned)
#ifdef DH_TUNING
  int w_nt;         // non-temporal weight DMA (measured: no gain at <= 2 row tiles, a loss beyond; tuning builds only)
  unsigned long long* ts;   // in-kernel timeline of workgroup (0,0,0), lane 0 of wave 0: s_memtime at the phase boundaries
  int lnf_abl;      // timing-only ablation of the folded LayerNorm: 1 = no sums in the K loop, 2 = no exchange, 4 = no epilogue transform
#endif
  // LayerNorm folded into this GEMM (LNF instantiations): A is the LayerNorm INPUT x, W holds W * gamma, and
  // out = rstd * (x W'^T - mean * ln_s) + ln_t with ln_s[n] = sum_k W'[n][k], ln_t[n] = sum_k beta[k] W[n][k] (+ bias);
  // the row statistics come out of the K loop and are saved to ln_stats ([M][2]: mean, rstd) for the LayerNorm backward
  const float* ln_s; const float* ln_t; float* ln_stats; float ln_eps;
  float* partial;
  int splits, k_per_split;
  float* gn_part; int gn_HW, gn_G, gn_S;      // GroupNorm slice statistics of the output (split-K reduce, or the GEMM's own epilogue: gn_epi)
  int gn_epi, gn_cpg;                         // statistics in this launch's epilogue (unsplit k_gemm_dma; 1 = forward, 2 = backward with gnb_*): gn_S = 2 per row tile of an image, gn_cpg = channels per group
  const void* gnb_x; long gnb_ldx; const float *gnb_gamma, *gnb_beta, *gnb_stats; int gnb_silu;   // backward statistics
  const void* lnb_x; const float *lnb_gamma, *lnb_stats; const void* lnb_add; void* lnb_dx;        // LayerNorm backward on the reduce (host side only)
  void* glu_y; long glu_ldy; const void* glub_x; void* glub_dx;                                    // GEGLU epilogues (GLU instantiations)
  // k_gemm_pp only (launch_gemm_pp fills them): row / column tile counts, work-item order (0 = column tile fastest, 1 = row
  // tile fastest), bytes the A / W buffer descriptors cover
  int pp_tm, pp_tn, pp_order, pp_nwork; unsigned pp_a_bytes, pp_w_bytes;
  unsigned long long* pp_ts;        // timeline stamps (measurement variants only)
};


// gemm_pp.hip: the eight-wave ping-pong kernel.  gemm_pp_eligible: whether the shape / epilogue can run on it and the policy
// wants it (`force`: test hook -- 0 policy, 1 never, 2 whenever the kernel can carry the launch); launch_gemm_pp launches it
// (the caller has filled k.splits / k.k_per_split for the tile it reports through bm / bn).
struct PpPlan { int bm = 0, bn = 0, splits = 1; };
bool gemm_pp_plan(const GemmK& k, size_t partial_elems, int force, PpPlan* plan);
void launch_gemm_pp(int dtype, const GemmK& k, const PpPlan& plan, hipStream_t st, hipEvent_t e0, hipEvent_t e1);

}  // namespace dh
