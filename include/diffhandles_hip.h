/*
 * diffhandles_hip.h -- C ABI of the MI355X (gfx950) native library behind the
 * DiffusionHandles guided-denoising edit path.
 *
 * The reference (adobe-research/DiffusionHandles) has no FFI of its own: its boundary is
 * the Python class API (diffhandles/diffusion_handles.py:15-166,
 * guided_stable_diffuser.py:22-610, stable_null_inverter.py:12-181).  These entry points
 * are what a ctypes binding of that path calls; each one cites the reference code whose
 * arithmetic it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream);
 *   - return value 0 = ok, negative = error (dh_last_error() gives the text);
 *   - no hidden device allocation on the hot path: callers pass workspaces whose size the
 *     *_workspace_bytes() queries return.  Engine handles own their weights/workspaces.
 *   - 16-bit activations are passed as raw uint16 storage; `dtype` says how to read them.
 */
#ifndef DIFFHANDLES_HIP_H
#define DIFFHANDLES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DH_OK 0
#define DH_ERR_ARG -1
#define DH_ERR_HIP -2
#define DH_ERR_STATE -3

#define DH_DTYPE_F16 0
#define DH_DTYPE_BF16 1
#define DH_DTYPE_F32 2

const char* dh_last_error(void);
int dh_version(void);
/* number of visible HIP devices (<=0: no GPU) */
int dh_device_count(void);

/* --------------------------------------------------------------------------------------
 * Depth re-projection: unproject -> SE(3) about the masked centroid -> project ->
 * z-buffer -> mask clean-up -> correspondence filter -> harmonic in-fill -> disparity.
 * Replaces transform_depth_pc / depth_to_world_coords / transform_point_cloud /
 * points_to_depth / poisson_solve / normalize_depth
 * (depth_transform.py:198-363, 589-641, 461-533, 643-747, 535-587, 15-28) and the cv2
 * morphology at :308-321, for a batch of n_edits rigid transforms of ONE image.
 *
 * xforms_host: n_edits x 8 doubles {ax, ay, az (unit axis, float32 values widened),
 *              cos(theta), sin(theta), tx, ty, tz} prepared on the host with NumPy so the
 *              transcendental values match the reference's.
 * grid_x/grid_y: the float32 pixel-centre coordinates torch.linspace gives (device).
 * Outputs per edit e (all device):
 *   zmap[e][res*res]       f32  z-buffered depth (inf where empty)
 *   raw_mask[e][res*res]   u8   pixel won by a foreground point
 *   clean_mask[e][res*res] u8   after CLOSE(ellipse res/50) + OPEN(ellipse res/250)
 *   disparity[e][res*res]  f32  255*(1/z - min)/(max - min), harmonically in-filled
 *   vis[e][n_fg]           u8   foreground point j is the (z, index) winner of its pixel
 *   target_xy[e][n_fg][2]  i32  projected pixel (x, y) of every foreground point
 *   corr[e][n_fg][4]       i64  (ox, oy, tx, ty), first counts[e] rows valid, in
 *                               row-major order of the original pixel
 *   counts[e][4]           i32  {n_corr, n_visible, n_inpaint, cg_iterations}
 * bounds: optional [2] f32 {lo, hi} normalisation bounds (use_input_depth_normalization),
 *         NULL = per-edit min/max of the rendered disparity.
 * ------------------------------------------------------------------------------------ */
int dh_reproject_workspace_bytes(int res, int n_fg, int n_edits, size_t* bytes);
int dh_fg_pixel_list(const uint8_t* fg_mask, int res, int32_t* fg_pix, int32_t* n_fg_dev,
                     void* workspace, size_t workspace_bytes, void* stream);
int dh_reproject_edits(const float* depth, const float* bg_depth, const int32_t* fg_pix, int n_fg,
                       int res, const float* grid_x, const float* grid_y, float inv_fx, float inv_fy,
                       double fx, double fy, int n_edits, const double* xforms_host,
                       const float* bounds,
                       float* zmap, uint8_t* raw_mask, uint8_t* clean_mask, float* disparity,
                       uint8_t* vis, int32_t* target_xy, int64_t* corr, int32_t* counts,
                       void* workspace, size_t workspace_bytes, void* stream);
/* unprojected points only (depth_to_world_coords), [res*res][3] f32 */
int dh_unproject(const float* depth, int res, const float* grid_x, const float* grid_y,
                 float inv_fx, float inv_fy, float* points, void* stream);
/* masked centroid with NumPy's sequential float32 accumulation order (f32[3]) */
int dh_masked_centroid(const float* depth, const int32_t* fg_pix, int n_fg, int res,
                       const float* grid_x, const float* grid_y, float inv_fx, float inv_fy,
                       float* centroid, void* stream);

/* Laplacian depth blend of DiffusionHandles.set_foreground (diffusion_handles.py:90-111,
 * utils.solve_laplacian_depth utils.py:49-102): inside binary_dilation(fg_mask, dilate_iters) (cross
 * element) solve 4x - sum(masked nbrs) = sum(known depth nbrs) - laplacian(bg_depth); elsewhere copy depth.
 * counts[4] i32: {_, _, n_unknown, cg_iterations}. */
int dh_laplacian_blend_workspace_bytes(int res, size_t* bytes);
int dh_laplacian_blend(const float* depth, const float* bg_depth, const uint8_t* fg_mask, int res,
                       int dilate_iters, float* out, int32_t* counts, void* workspace,
                       size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------------------
 * Correspondences -> grid x grid cell index lists.  Replaces
 * GuidedStableDiffuser.process_correspondences (guided_stable_diffuser.py:490-584),
 * including scipy.ndimage.binary_erosion (cross element, border 0).
 *   corr [n][4] i64 (ox, oy, tx, ty) in pixels.
 *   pairs [n][2] i32 (orig cell id, target cell id), first counts[0] valid, order kept
 *   bg_lists [3][grid*grid] i32 cell ids (row-major nonzero order) of
 *            {both, orig, trans} background masks; counts[1..3] = their lengths
 *   bg_masks [3][grid*grid] u8
 *   counts [4] i32
 * ------------------------------------------------------------------------------------ */
int dh_cells_workspace_bytes(int n, int grid, size_t* bytes);
int dh_cells_from_correspondences(const int64_t* corr, int n, int img_res, int grid, int bg_erosion,
                                  int32_t* pairs, int32_t* bg_lists, uint8_t* bg_masks, int32_t* counts,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------------------
 * depth_transform_mode = 'mesh' (depth_transform.py:91-195 + depth_to_mesh :30-71 + the pytorch3d
 * renderer outputs 'world_position' / 'flat_vertex_color'): per-pixel-quad triangulation of the
 * background depth and of the rigidly moved foreground depth (transform_points :438-458, float32),
 * nearest-z rasterisation with back-face culling, blur_radius coverage, perspective-correct clipped
 * barycentrics.  pytorch3d is not in the reference tree: parity unpinned (oracle/mesh_ref.py restates
 * the published rule; validated against the point z-buffer on smooth depth).
 *   grid   [res] f32 = linspace(-1,1,res), lin01 [res] f32 = linspace(0,1,res) (host torch values)
 *   xform  [11] f32 host: unit axis x3, cos, sin, translation x3, centroid of the masked points x3
 *   bounds NULL or device {lo, hi} of the input disparity (use_input_depth_normalization)
 * Outputs: zmap [res^2] f32 rendered depth, disparity [res^2] f32 in [0,255], fg_flag [res^2] u8,
 *          corr [<= res^2][4] i64 (src_x, src_y, tgt_x, tgt_y) in row-major target order,
 *          counts[0] = number of correspondences.
 * ------------------------------------------------------------------------------------ */
int dh_mesh_workspace_bytes(int res, size_t* bytes);
int dh_mesh_reproject(const float* depth, const float* bg_depth, const uint8_t* fg_mask, int res,
                      const float* grid, const float* lin01, float inv_f, float f, const float* xform,
                      const float* bounds, float blur_radius, float* zmap, float* disparity,
                      uint8_t* fg_flag, int64_t* corr, int32_t* counts, void* workspace,
                      size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------------------
 * Guidance energy (losses.py:4-84) on channels-last maps [h][w][C].
 * One call evaluates, for ONE activation layer,
 *     loss = fg_w * fg_term + bg_w * bg_term
 * and writes d(loss * grad_scale)/d(cur) for every element (no accumulation, no atomics).
 * cur/orig: [hw_in][C] in `dtype`; when hw_in != grid*grid the maps are bilinearly resized
 * (align_corners=False) to grid x grid first and the gradient is carried back.
 * bg_mode: 0 = global_avg, 1 = local_avg.  patch: fg/bg pooling window (odd, >=1).
 * pairs/bg lists as produced by dh_cells_from_correspondences (device).
 * loss_out: f32[3] = {loss, fg_term, bg_term}.  grad: same shape as cur, in `grad_dtype`.
 * ------------------------------------------------------------------------------------ */
int dh_energy_workspace_bytes(int C, int grid, int n_pairs, size_t* bytes);
int dh_energy_fwd_bwd(const void* cur, const void* orig, int dtype, int C, int h_in, int w_in, int grid,
                      const int32_t* pairs, int n_pairs,
                      const int32_t* bg_both, int n_bg_both,
                      const int32_t* bg_orig, int n_bg_orig,
                      const int32_t* bg_trans, int n_bg_trans,
                      float fg_w, float bg_w, int fg_patch, int bg_patch, int bg_mode,
                      float grad_scale, float* loss_out, void* grad, int grad_dtype,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Planned form of the same evaluation for the default configuration (maps already at the cell grid,
 * fg_patch 1, bg 'global_avg' -- config/default.yaml:5-9), 16-bit activations.  The correspondences of an
 * edit are fixed over its 38 x 3 evaluations, so the target-cell -> (distinct source cell, multiplicity) CSR and the
 * transformed-background flags are built ONCE (dh_energy_plan_build) and every evaluation is three kernels
 * that read the activations once in their own dtype.  Same arithmetic per element as dh_energy_fwd_bwd:
 * the gradient is bit-identical.  loss_out may be NULL (the guided loop only needs the gradient). */
int dh_energy_plan_bytes(int grid, int n_pairs, size_t* bytes);
int dh_energy_plan_build(const int32_t* pairs, int n_pairs, const int32_t* bg_trans, int n_bg_trans, int grid,
                         void* plan, size_t plan_bytes, void* stream);
int dh_energy_planned_workspace_bytes(int C, int grid, size_t* bytes);
int dh_energy_fwd_bwd_planned(const void* cur, const void* orig, int dtype, int C, int grid,
                              const void* plan, size_t plan_bytes, int n_pairs,
                              const int32_t* bg_orig, int n_bg_orig, const int32_t* bg_trans, int n_bg_trans,
                              float fg_w, float bg_w, float grad_scale, float* loss_out, void* grad,
                              int grad_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------------------
 * SD-2-depth U-Net engine (model/unet_2d_condition.py:809-1198 and the block files it
 * calls): forward with the three decoder activation captures, and the backward pass to the
 * input sample / to the text embedding.  Channels-last 16-bit activations, f32 accumulate.
 * ------------------------------------------------------------------------------------ */
typedef struct dh_unet dh_unet;

typedef struct dh_unet_config {
  int in_channels;            /* 5 */
  int out_channels;           /* 4 */
  int n_levels;               /* 4 */
  int block_out_channels[4];  /* 320 640 1280 1280 */
  int layers_per_block;       /* 2 */
  int heads[4];               /* 5 10 20 20 (head dim must be 64) */
  int cross_attention_dim;    /* 1024 */
  int norm_groups;            /* 32 */
  int sample_size;            /* 64 (latent H = W) */
  int text_len;               /* 77 */
  int max_batch;              /* largest batch a forward will see */
  int dtype;                  /* DH_DTYPE_F16 or DH_DTYPE_BF16 */
  int max_diff_batch;         /* largest batch of a forward that is SAVED for a backward pass (0 = max_batch).  A forward nobody
                               * differentiates (the CFG pass at twice the edit batch, initial inference, DDIM inversion) does not keep
                               * a slot per tensor: its tensors share the activation arena by liveness, so only this batch sizes the
                               * activation / gradient arenas (batched edits: max_batch = 2 K, max_diff_batch = K) */
} dh_unet_config;

int dh_unet_create(const dh_unet_config* cfg, dh_unet** out);
/* A second engine on the SAME weights: `parent`'s 16-bit weight arena and f32 parameter arena are used read-only (3.4 GB at
 * fp16, resident once), everything a pass writes -- activations, gradients, split-K slabs, statistics, I/O buffers, graphs --
 * is private (dh_unet_workspace_bytes).  Two such engines driven on two HIP streams run two independent edits concurrently in
 * one process (the reference runs one process per device and one edit at a time, webapp/start_webapps_in_tmux.sh:21-43; a
 * single edit's passes leave most of the chip idle at any instant).  Load every parameter into `parent` BEFORE sharing: the
 * folded LayerNorm weights are finalised here (on `stream`, synchronised), and dh_unet_load_param on a shared engine is an
 * error.  `parent` must outlive the engines that share its weights.  max_batch <= 0: the parent's. */
int dh_unet_create_shared(dh_unet* parent, int max_batch, void* stream, dh_unet** out);
void dh_unet_destroy(dh_unet* u);
/* parameter table: diffusers state-dict names, torch shapes */
int dh_unet_num_params(const dh_unet* u);
int dh_unet_param_info(const dh_unet* u, int i, const char** name, int* ndim, int64_t* shape4);
/* upload parameter i from a DEVICE f32 tensor in torch layout (conv: [Cout,Cin,kh,kw]) */
int dh_unet_load_param(dh_unet* u, int i, const float* src, void* stream);
size_t dh_unet_weight_bytes(const dh_unet* u);
size_t dh_unet_workspace_bytes(const dh_unet* u);

/* forward.  sample [B][H][W][Cin] f32 channels-last; timestep host float; text [B][L][D] f32.
 * eps_out [B][H][W][Cout] f32.  act_out[3] (may be NULL): [B][h][w][C] in engine dtype.
 * save_for_backward != 0 keeps what dh_unet_backward needs (one saved pass at a time). */
int dh_unet_forward(dh_unet* u, const float* sample, float timestep, const float* text, int batch,
                    int save_for_backward, float* eps_out, void* const* act_out, void* stream);
/* backward of the LAST saved forward.  d_act[3]: gradients w.r.t. the three captured
 * activations (engine dtype, may be NULL each); d_eps: gradient w.r.t. eps (f32, may be
 * NULL).  Outputs (either may be NULL): d_sample [B][H][W][Cin] f32, d_text [B][L][D] f32. */
int dh_unet_backward(dh_unet* u, void* const* d_act, const float* d_eps, float* d_sample, float* d_text,
                     void* stream);
/* The engine's own input / output buffers (fixed addresses: its passes replay as hipGraphs).  A caller that writes its
 * inputs there (dh_pack_sample) and hands the SAME pointers to dh_unet_forward / dh_unet_backward, or lets the energy kernels
 * read the captured activations and write their cotangents in place, saves the device-to-device copies either side of a pass
 * (guided_stable_diffuser.py:397-434 builds `torch.cat([latents, depth])` and reads `unet_output[4..6]` every iteration).
 * Every buffer holds max_batch items (DH_IO_ACT_GRAD: max_diff_batch); contents are valid until the next pass that writes them.
 *   which: DH_IO_SAMPLE [B][H][W][Cin] f32 | DH_IO_TEXT [B][L][D] f32 | DH_IO_EPS [B][H][W][Cout] f32 (eps out / d_eps in) |
 *          DH_IO_ACT index 0..2 [B][h][w][C] engine dtype | DH_IO_ACT_GRAD index 0..2 (cotangent of that activation) |
 *          DH_IO_DSAMPLE [B][H][W][Cin] f32 | DH_IO_DTEXT [B][L][D] f32 */
#define DH_IO_SAMPLE 0
#define DH_IO_TEXT 1
#define DH_IO_EPS 2
#define DH_IO_ACT 3
#define DH_IO_ACT_GRAD 4
#define DH_IO_DSAMPLE 5
#define DH_IO_DTEXT 6
int dh_unet_io_ptr(dh_unet* u, int which, int index, void** ptr, size_t* bytes);
/* HIP-event bracket around every launch of the MFMA implicit-GEMM kernel (k_gemm) between begin
 * and end: total milliseconds, number of launches and their algorithmic flops (2*M*N*K).
 * Used by bench.py for the roofline figure; not on the product path. */
int dh_gemm_profile_begin(void);
int dh_gemm_profile_end(double* ms_total, int64_t* launches, double* flops);
/* ALGORITHMIC HBM bytes of the launches bracketed since the last dh_gemm_profile_begin (valid after _end as well): every
 * operand once in its 16-bit storage type -- the A source (dense rows [M][K]; for a convolution the source image
 * [B][Hin][Win][Cin], not its im2col view), the weights [N][K], the output [M][N] and the residual [M][N] when there is one
 * (GEGLU epilogues: the [M][N/2] activation on top, or the [M][2N] pre-activations read and their gradient written in place of
 * the output).  Split-K slabs, im2col tap re-reads and per-XCD re-fetches are NOT in it: `roofline.traffic` (PMC) divided by
 * this figure is the wasted-traffic ratio.  Not on the product path. */
int dh_gemm_profile_bytes(double* bytes);
/* per-kernel-class accumulated launch counts / algorithmic flops of the last forward */
/* Names the text embedding of the following dh_unet_forward calls (0 = unnamed, the default).  The K|V projections of every
 * cross-attention layer depend on the text only; a forward that finds those of the same key, batch and stream already in the
 * engine skips recomputing them.  The caller must change the key whenever the text tensor's content changes
 * (guided_stable_diffuser.py:404-410: the prompt embedding is constant over the optimisation passes of an edit). */
int dh_unet_set_text_key(dh_unet* u, unsigned long long key);
int dh_unet_stats(const dh_unet* u, double* flops_fwd, double* flops_bwd, int64_t* launches);

/* --------------------------------------------------------------------------------------
 * Loop arithmetic (f32 latents, channels-last [B][H][W][4]).
 * ------------------------------------------------------------------------------------ */
/* eps = eps_u + scale*(eps_c - eps_u); x <- sqrt(a_prev)*(x - sqrt(1-a_t) eps)/sqrt(a_t) +
 * sqrt(1-a_prev) eps.   DDIMScheduler.step, eta 0 (guided_stable_diffuser.py:468-474) and
 * prev_step/next_step (stable_null_inverter.py:25-43).  eps_u may be NULL (no CFG). */
int dh_ddim_cfg_step(float* x_out, const float* x, const float* eps_u, const float* eps_c, float scale,
                     float alpha_t, float alpha_prev, int n, void* stream);
/* x_out = x - lr * g / grad_scale                         guided_stable_diffuser.py:434 */
int dh_latent_update(float* x_out, const float* x, const float* g, float lr, float grad_scale, int n,
                     void* stream);
/* the same with g read out of a wider gradient: g [pixels][g_channels], the first `channels` of every pixel (the latent
 * channels of d(sample) = d(cat[latents, depth])) */
int dh_latent_update_strided(float* x_out, const float* x, const float* g, int g_channels, int channels, float lr,
                             float grad_scale, int pixels, void* stream);
/* dst[b][p][:] = concat(latent[b or 0][p][0:latent_channels], depth[b or 0][p][0:depth_channels]) for b < batch: the U-Net
 * input `torch.cat([latents (x batch), depth (x batch)], dim=1)` (guided_stable_diffuser.py:400-401, 451-455) in one launch.
 * latent_batch / depth_batch divide batch: item b reads latent[b mod latent_batch] (1 = broadcast, batch = one each, K of 2K =
 * the K edits repeated for the two halves of the CFG pass); depth may be NULL (use_depth false). */
int dh_pack_sample(float* dst, const float* latent, int latent_batch, int latent_channels, const float* depth,
                   int depth_batch, int depth_channels, int batch, int pixels, void* stream);
/* Adam step on the null-text embedding (torch defaults; stable_null_inverter.py:143-155) */
int dh_adam_step(float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                 float eps, int step, int n, void* stream);
/* mse(rec, target) and d mse / d rec                        stable_null_inverter.py:152 */
int dh_mse_fwd_bwd(const float* rec, const float* target, int n, float* loss_out, float* d_rec,
                   void* stream);
/* The same loss, and the cotangent that seeds the engine's backward of a null-text inner step
 * (stable_null_inverter.py:150-154: loss.backward() through prev_step and the CFG combine down to eps_uncond):
 *   d_eps = (d mse / d rec) * k * S,   k = d rec / d eps_uncond (a scalar of the timestep),
 * S = the power of two that brings max |d_eps| into (amp / 2, amp] (amp <= 0: S = 1), written to scale_out[0] (device).
 * The 16-bit backward is linear in its cotangent: S only keeps it out of the fp16 subnormal range and is divided out
 * again by dh_adam_step_scaled. */
int dh_mse_cotangent(const float* rec, const float* target, int n, float k, float amp, float* loss_out,
                     float* d_eps, float* scale_out, void* stream);
/* dh_adam_step on g / g_scale[0] (g_scale: device scalar, the S of dh_mse_cotangent) */
int dh_adam_step_scaled(float* p, const float* g, const float* g_scale, float* m, float* v, float lr, float beta1,
                        float beta2, float eps, int step, int n, void* stream);

/* --------------------------------------------------------------------------------------
 * SD AutoencoderKL decoder on the engine's kernels: the decode that ends every edit
 * (guided_stable_diffuser.py:481-483 decode_latent_image, :286; stable_null_inverter.py:105 latent2image).
 * diffusers' AutoencoderKL is [ext]; parameter names are its state-dict names ("decoder.*").
 * z: [B][h][w][latent_channels] f32 channels-last, ALREADY divided by the scaling factor and passed through
 * post_quant_conv (a 4x4 matrix per pixel, host side); image: [B][8h][8w][out_channels] f32.  Forward only.
 * ------------------------------------------------------------------------------------ */
typedef struct dh_vae_decoder dh_vae_decoder;
typedef struct dh_vae_config {
  int latent_channels;        /* 4 */
  int out_channels;           /* 3 */
  int block_out_channels[4];  /* 128 256 512 512 (encoder order, as in the VAE's config.json) */
  int layers_per_block;       /* 2 (the decoder runs layers_per_block + 1 resnets per up block) */
  int norm_groups;            /* 32 */
  int latent_size;            /* 64 (latent H = W; image = 8x) */
  int dtype;                  /* DH_DTYPE_F16 or DH_DTYPE_BF16 */
} dh_vae_config;
int dh_vae_decoder_create(const dh_vae_config* cfg, dh_vae_decoder** out);
void dh_vae_decoder_destroy(dh_vae_decoder* v);
int dh_vae_decoder_num_params(const dh_vae_decoder* v);
int dh_vae_decoder_param_info(const dh_vae_decoder* v, int i, const char** name, int* ndim, int64_t* shape4);
int dh_vae_decoder_load_param(dh_vae_decoder* v, int i, const float* src, void* stream);   /* DEVICE f32, torch layout */
size_t dh_vae_decoder_bytes(const dh_vae_decoder* v);
int dh_vae_decoder_decode(dh_vae_decoder* v, const float* z, int batch, float* image, void* stream);
/* The encoder half on the same kernels (stable_null_inverter.py:72-83 image2latent: `vae.encode(image)['latent_dist'].mean`,
 * once per image).  The handle has the decoder's type and shares its parameter / destroy / bytes entry points; the state-dict
 * names are "encoder.*".  image: [B][8h][8w][out_channels] f32 channels-last in [-1, 1]; moments: [B][h][w][2 * latent_channels]
 * f32 (conv_out of the encoder: the caller applies the 1x1 quant_conv and takes the first latent_channels as the mean). */
int dh_vae_encoder_create(const dh_vae_config* cfg, dh_vae_decoder** out);
int dh_vae_encoder_encode(dh_vae_decoder* v, const float* image, int batch, float* moments, void* stream);

/* --------------------------------------------------------------------------------------
 * CLIP text tower on the engine's kernels (guided_stable_diffuser.py:96-108 `self.text_encoder(ids)[0]`,
 * stable_null_inverter.py:85-103): pre-norm transformer layers with causal attention (heads of 64), erf-GELU MLP and the
 * final LayerNorm.  transformers' CLIPTextModel is [ext]; parameter names are its state-dict names
 * ("text_model.encoder.layers.N.*", "text_model.final_layer_norm.*").  The token + position embedding lookup stays with the
 * caller: embeds [B][tokens][hidden] f32 is their sum; out [B][tokens][hidden] f32 = last_hidden_state.  Forward only.
 * ------------------------------------------------------------------------------------ */
typedef struct dh_text_encoder dh_text_encoder;
typedef struct dh_text_config {
  int hidden;        /* 1024 */
  int heads;         /* 16 (head dim 64) */
  int layers;        /* 23 */
  int intermediate;  /* 4096 */
  int max_tokens;    /* 77 */
  int max_batch;     /* prompts per call */
  float eps;         /* 1e-5 */
  int dtype;         /* DH_DTYPE_F16 or DH_DTYPE_BF16 */
} dh_text_config;
int dh_text_encoder_create(const dh_text_config* cfg, dh_text_encoder** out);
void dh_text_encoder_destroy(dh_text_encoder* e);
int dh_text_encoder_num_params(const dh_text_encoder* e);
int dh_text_encoder_param_info(const dh_text_encoder* e, int i, const char** name, int* ndim, int64_t* shape2);
int dh_text_encoder_load_param(dh_text_encoder* e, int i, const float* src, void* stream);   /* DEVICE f32, torch layout */
size_t dh_text_encoder_bytes(const dh_text_encoder* e);
int dh_text_encoder_encode(dh_text_encoder* e, const float* embeds, int batch, int tokens, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
