// CLIP text tower (the SD-2 text encoder: OpenCLIP ViT-H text transformer, 23 pre-norm layers of width 1024, 16 heads of 64,
// causal attention, erf-GELU MLP, final LayerNorm) on the engine's kernels: LayerNorm, the MFMA GEMM for the fused q|k|v
// projection / out projection / MLP, the flash-attention forward with its causal switch.  Once per prompt
// (guided_stable_diffuser.py:96-108 `self.text_encoder(ids)[0]`; stable_null_inverter.py:85-103); forward only.
// transformers' CLIPTextModel is [ext]; parameter names are its state-dict names ("text_model.encoder.layers.N....").
// The token + position embedding lookup stays with the caller (a table gather): the input is the summed embedding.
#include <string>
#include <vector>

#include "unet_kernels.h"

using namespace dh;

namespace {
struct TParam { std::string name; int ndim; int64_t shape[2]; long w_off; int row_off, K, N; long f32_off; };
struct TLayer { long ln1_g, ln1_b, ln2_g, ln2_b, qkv_b, out_b, fc1_b, fc2_b; size_t qkv_w, out_w, fc1_w, fc2_w; };
}  // namespace

struct dh_text_encoder {
  dh_text_config cfg;
  std::vector<TParam> params;
  std::vector<TLayer> layers;
  long fin_g = 0, fin_b = 0;
  size_t w16_elems = 0, pf_elems = 0;
  unsigned short *w16 = nullptr, *x = nullptr, *t = nullptr, *qkv = nullptr, *att = nullptr, *mlp = nullptr;
  float *pf = nullptr, *stats = nullptr;
  int max_rows = 0;
};

static size_t alloc_w(dh_text_encoder* e, int N, int K) { const size_t o = e->w16_elems; e->w16_elems += align_up((size_t)N * K, 128); return o; }
static long alloc_f(dh_text_encoder* e, int n) { const long o = (long)e->pf_elems; e->pf_elems += align_up((size_t)n, 64); return o; }
static void bind_w(dh_text_encoder* e, const std::string& name, size_t w, int rows, int row_off, int K, int N) {
  e->params.push_back(TParam{name, 2, {rows, K}, (long)w, row_off, K, N, -1});
}
static void bind_f(dh_text_encoder* e, const std::string& name, int n, long off) { e->params.push_back(TParam{name, 1, {n, 0}, -1, 0, 0, 0, off}); }

extern "C" int dh_text_encoder_create(const dh_text_config* cfg, dh_text_encoder** out) {
  DH_REQUIRE(cfg && out, "null pointer");
  DH_REQUIRE(cfg->dtype == DH_DTYPE_F16 || cfg->dtype == DH_DTYPE_BF16, "dtype must be f16 or bf16");
  DH_REQUIRE(cfg->hidden % 64 == 0 && cfg->intermediate % 64 == 0 && cfg->heads * 64 == cfg->hidden, "hidden = 64 * heads, widths multiples of 64");
  DH_REQUIRE(cfg->layers >= 1 && cfg->max_tokens >= 1 && cfg->max_tokens <= 1024 && cfg->max_batch >= 1 && cfg->max_batch <= 64, "bad configuration");
  dh_text_encoder* e = new dh_text_encoder();
  e->cfg = *cfg;
  const int C = cfg->hidden, F = cfg->intermediate;
  for (int i = 0; i < cfg->layers; ++i) {
    const std::string p = "text_model.encoder.layers." + std::to_string(i);
    TLayer l;
    l.ln1_g = alloc_f(e, C); bind_f(e, p + ".layer_norm1.weight", C, l.ln1_g);
    l.ln1_b = alloc_f(e, C); bind_f(e, p + ".layer_norm1.bias", C, l.ln1_b);
    l.qkv_w = alloc_w(e, 3 * C, C);
    bind_w(e, p + ".self_attn.q_proj.weight", l.qkv_w, C, 0, C, 3 * C);
    bind_w(e, p + ".self_attn.k_proj.weight", l.qkv_w, C, C, C, 3 * C);
    bind_w(e, p + ".self_attn.v_proj.weight", l.qkv_w, C, 2 * C, C, 3 * C);
    l.qkv_b = alloc_f(e, 3 * C);
    bind_f(e, p + ".self_attn.q_proj.bias", C, l.qkv_b);
    bind_f(e, p + ".self_attn.k_proj.bias", C, l.qkv_b + C);
    bind_f(e, p + ".self_attn.v_proj.bias", C, l.qkv_b + 2 * C);
    l.out_w = alloc_w(e, C, C); bind_w(e, p + ".self_attn.out_proj.weight", l.out_w, C, 0, C, C);
    l.out_b = alloc_f(e, C); bind_f(e, p + ".self_attn.out_proj.bias", C, l.out_b);
    l.ln2_g = alloc_f(e, C); bind_f(e, p + ".layer_norm2.weight", C, l.ln2_g);
    l.ln2_b = alloc_f(e, C); bind_f(e, p + ".layer_norm2.bias", C, l.ln2_b);
    l.fc1_w = alloc_w(e, F, C); bind_w(e, p + ".mlp.fc1.weight", l.fc1_w, F, 0, C, F);
    l.fc1_b = alloc_f(e, F); bind_f(e, p + ".mlp.fc1.bias", F, l.fc1_b);
    l.fc2_w = alloc_w(e, C, F); bind_w(e, p + ".mlp.fc2.weight", l.fc2_w, C, 0, F, C);
    l.fc2_b = alloc_f(e, C); bind_f(e, p + ".mlp.fc2.bias", C, l.fc2_b);
    e->layers.push_back(l);
  }
  e->fin_g = alloc_f(e, C); bind_f(e, "text_model.final_layer_norm.weight", C, e->fin_g);
  e->fin_b = alloc_f(e, C); bind_f(e, "text_model.final_layer_norm.bias", C, e->fin_b);
  e->max_rows = cfg->max_batch * cfg->max_tokens;
  const size_t R = (size_t)e->max_rows;
  auto fail = [&](hipError_t err, const char* what) {
    set_error(std::string(what) + ": " + hipGetErrorString(err));
    dh_text_encoder_destroy(e);
    return DH_ERR_HIP;
  };
  hipError_t err;
  if ((err = hipMalloc((void**)&e->w16, e->w16_elems * 2 + 256)) != hipSuccess) return fail(err, "hipMalloc weights");
  if ((err = hipMalloc((void**)&e->pf, e->pf_elems * 4 + 256)) != hipSuccess) return fail(err, "hipMalloc f32 params");
  if ((err = hipMalloc((void**)&e->x, R * C * 2 + 256)) != hipSuccess) return fail(err, "hipMalloc activations");
  if ((err = hipMalloc((void**)&e->t, R * C * 2 + 256)) != hipSuccess) return fail(err, "hipMalloc activations");
  if ((err = hipMalloc((void**)&e->qkv, R * 3 * C * 2 + 256)) != hipSuccess) return fail(err, "hipMalloc activations");
  if ((err = hipMalloc((void**)&e->att, R * C * 2 + 256)) != hipSuccess) return fail(err, "hipMalloc activations");
  if ((err = hipMalloc((void**)&e->mlp, R * F * 2 * 2 + 256)) != hipSuccess) return fail(err, "hipMalloc activations");
  if ((err = hipMalloc((void**)&e->stats, R * 2 * 4 + 256)) != hipSuccess) return fail(err, "hipMalloc statistics");
  (void)hipMemset(e->w16, 0, e->w16_elems * 2);
  (void)hipMemset(e->pf, 0, e->pf_elems * 4);
  *out = e;
  return DH_OK;
}

extern "C" void dh_text_encoder_destroy(dh_text_encoder* e) {
  if (!e) return;
  (void)hipFree(e->w16); (void)hipFree(e->pf); (void)hipFree(e->x); (void)hipFree(e->t); (void)hipFree(e->qkv); (void)hipFree(e->att);
  (void)hipFree(e->mlp); (void)hipFree(e->stats);
  delete e;
}

extern "C" int dh_text_encoder_num_params(const dh_text_encoder* e) { return e ? (int)e->params.size() : 0; }

extern "C" int dh_text_encoder_param_info(const dh_text_encoder* e, int i, const char** name, int* ndim, int64_t* shape2) {
  DH_REQUIRE(e && i >= 0 && i < (int)e->params.size() && name && ndim && shape2, "bad arguments");
  const TParam& p = e->params[i];
  *name = p.name.c_str(); *ndim = p.ndim; shape2[0] = p.shape[0]; shape2[1] = p.shape[1];
  return DH_OK;
}

extern "C" int dh_text_encoder_load_param(dh_text_encoder* e, int i, const float* src, void* stream) {
  DH_REQUIRE(e && src && i >= 0 && i < (int)e->params.size(), "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const TParam& p = e->params[i];
  if (p.w_off < 0) {
    DH_CHECK_HIP(hipMemcpyAsync(e->pf + p.f32_off, src, (size_t)p.shape[0] * 4, hipMemcpyDeviceToDevice, st));
    return DH_OK;
  }
  launch_load_weight(e->cfg.dtype, src, (int)p.shape[0], p.K, 1, e->w16 + p.w_off, p.K, p.row_off, nullptr, 0, 0, p.N, 1.f, st);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" size_t dh_text_encoder_bytes(const dh_text_encoder* e) {
  if (!e) return 0;
  const size_t R = (size_t)e->max_rows, C = e->cfg.hidden, F = e->cfg.intermediate;
  return e->w16_elems * 2 + e->pf_elems * 4 + R * (C * 3 + 3 * C + 2 * F) * 2 + R * 8;
}

extern "C" int dh_text_encoder_encode(dh_text_encoder* e, const float* embeds, int batch, int tokens, float* out, void* stream) {
  DH_REQUIRE(e && embeds && out, "null pointer");
  DH_REQUIRE(batch >= 1 && batch <= e->cfg.max_batch && tokens >= 1 && tokens <= e->cfg.max_tokens, "batch / tokens exceed the configured maxima");
  hipStream_t st = (hipStream_t)stream;
  const int dt = e->cfg.dtype, C = e->cfg.hidden, F = e->cfg.intermediate, rows = batch * tokens;
  const size_t n = (size_t)rows * C;
  launch_f32_to_t(dt, embeds, e->x, n, st);
  for (const TLayer& l : e->layers) {
    launch_layernorm_fwd(dt, e->x, e->pf + l.ln1_g, e->pf + l.ln1_b, e->t, e->stats, rows, C, e->cfg.eps, st);
    GemmArgs g;
    g.A = e->t; g.lda = C; g.W = e->w16 + l.qkv_w; g.M = rows; g.N = 3 * C; g.K = C; g.bias = e->pf + l.qkv_b; g.C = e->qkv; g.ldc = 3 * C;
    launch_gemm(dt, g, st);
    launch_attention_fwd(dt, e->qkv, 3 * C, e->qkv + C, e->qkv + 2 * C, 3 * C, e->att, C, nullptr, batch, e->cfg.heads, tokens, tokens, st, 1);
    GemmArgs o;
    o.A = e->att; o.lda = C; o.W = e->w16 + l.out_w; o.M = rows; o.N = C; o.K = C; o.bias = e->pf + l.out_b; o.R = e->x; o.ldr = C;
    o.C = e->t; o.ldc = C;                                 // t = x + out_proj(attention)
    launch_gemm(dt, o, st);
    launch_layernorm_fwd(dt, e->t, e->pf + l.ln2_g, e->pf + l.ln2_b, e->x, e->stats, rows, C, e->cfg.eps, st);
    GemmArgs f1;
    f1.A = e->x; f1.lda = C; f1.W = e->w16 + l.fc1_w; f1.M = rows; f1.N = F; f1.K = C; f1.bias = e->pf + l.fc1_b; f1.C = e->mlp; f1.ldc = F;
    launch_gemm(dt, f1, st);
    launch_gelu(dt, e->mlp, e->mlp + (size_t)e->max_rows * F, (size_t)rows * F, st);
    GemmArgs f2;
    f2.A = e->mlp + (size_t)e->max_rows * F; f2.lda = F; f2.W = e->w16 + l.fc2_w; f2.M = rows; f2.N = C; f2.K = F; f2.bias = e->pf + l.fc2_b;
    f2.R = e->t; f2.ldr = C; f2.C = e->x; f2.ldc = C;      // x = t + fc2(gelu(fc1(LN2(t))))
    launch_gemm(dt, f2, st);
  }
  launch_layernorm_fwd(dt, e->x, e->pf + e->fin_g, e->pf + e->fin_b, e->t, e->stats, rows, C, e->cfg.eps, st);
  launch_t_to_f32(dt, e->t, out, n, 0, st);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
