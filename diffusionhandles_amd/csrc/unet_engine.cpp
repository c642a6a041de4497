// SD-2-depth U-Net engine: builds a static tape of ops from a config, owns weights (forward
// layout + transposed/flipped copies for the input-gradient GEMMs) and activation storage,
// runs forward with activation capture and the backward pass to the sample / text inputs.
//
// Structure follows SURVEY.md section 8 a3 (reference model/unet_2d_condition.py:809-1198,
// unet_2d_blocks.py, transformer_2d.py:242-444, attention.py:219-342,
// attention_processor.py:1178-1262; diffusers-0.23 ResnetBlock2D/Up/Downsample2D [ext]).
// Parameter names are the diffusers state-dict names.
//
// MI355X layout decisions: channels-last 16-bit activations (a pixel row is a GEMM row and
// a token); a forward that is SAVED for a backward pass gives every tensor its own HBM slot (288 GB: nothing is
// recomputed, saved-for-backward is "left in place"), a forward nobody differentiates lets its tensors share the arena by
// liveness (round 5: two layouts of one arena, dh_unet_config.max_diff_batch),
// the 32 text K/V projections and the 22 time-embedding projections are hoisted into one
// GEMM each, q/k/v of self-attention is one GEMM, weights are stored twice ([N][K] and the
// transposed / tap-flipped [K][N]) so forward and input-gradient use the same NT kernel.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <numeric>
#include <vector>

#include "unet_kernels.h"

namespace dh {

enum OpType { OP_CONV_IN, OP_CONV_OUT, OP_GEMM, OP_GN, OP_LN, OP_ATTN, OP_GEGLU, OP_CONCAT, OP_TIMESTEP, OP_T2F };
enum ParamKind { PK_F32, PK_MAT };

struct Ten {
  size_t off = 0, goff = 0;   // element offsets into the activation / gradient arenas
  size_t off_fo = 0;          // ... into the activation arena during a forward nobody differentiates (liveness plan, plan_forward_only)
  int rows = 0, C = 0;        // rows per batch item
  bool req_grad = true;
  bool persistent = false;    // same slot in both layouts, max_batch items: read across passes (text K|V cache) or by the caller (captured activations)
};

struct Wt {
  size_t fwd_off = 0, bwd_off = 0;   // element offsets (16-bit arena, or f32 arena when f32)
  int N = 0, K = 0, taps = 1;
  bool has_bwd = true, f32 = false;
  int glu_F = 0;                     // > 0: the rows are the [2F] value | gate rows of a GEGLU projection, stored in the paired order (glu_col)
};

struct ParamInfo {
  std::string name;
  int ndim = 0;
  int64_t shape[4] = {0, 0, 0, 0};
  int kind = PK_F32;
  size_t f32_off = 0;   // PK_F32
  int wt = -1;          // PK_MAT
  int row_off = 0;      // rows [row_off, row_off + shape[0]) of the (fused) weight
  int glu_F = 0;        // PK_F32: a [2F] GEGLU bias, stored in the paired order
};

struct Op {
  int type = OP_GEMM;
  int in0 = -1, in1 = -1, out = -1, res = -1;
  int wt = -1;
  long bias_off = -1;      // f32 arena
  long gamma_off = -1, beta_off = -1;
  // gemm / conv
  int mode = A_DENSE, Hin = 0, Win = 0, Cin = 0, Hout = 0, Wout = 0, stride = 1, up = 0;
  int in_col = 0, in_cols = 0;      // column window of in0 used as A (dense)
  long rowvec_off = -1; int rowvec_ld = 0;   // f32 arena: per-batch vector (time embedding)
  int act_silu = 0;
  // norms
  float eps = 1e-5f; int silu = 0; int groups = 32; size_t stats_off = 0;
  // attention
  int heads = 0, Nq = 0, Nk = 0, cross = 0, kv_col = 0; size_t lse_off = 0;
  // concat: in0 | in1
  int gn_next = -1;         // index of the GroupNorm op that consumes `out` right after this op (statistics fused here)
  // LayerNorm folded into the GEMM that consumes it (forward only; the backward pass still runs the LayerNorm op):
  // on the GEMM, ln_fold = index of the LayerNorm op and ln_s_off / ln_t_off = the per-column vectors (f32 param arena);
  // on the LayerNorm op, folded = 1 (the forward skips it: its statistics come out of the GEMM)
  int ln_fold = -1, folded = 0;
  long ln_s_off = -1, ln_t_off = -1;
  // GEGLU fused into the GEMMs around it: on the ff.net.0.proj GEMM, glu_op = index of the OP_GEGLU op that consumes its output
  // (the epilogue writes that op's output as well); on the ff.net.2 GEMM, glub_op = the OP_GEGLU op that produced its input (its
  // input-gradient GEMM writes the gradient of the pre-activations directly)
  int glu_op = -1, glub_op = -1;
};

}  // namespace dh

using namespace dh;

struct dh_unet {
  dh_unet_config cfg;
  int dtype = DH_DTYPE_F16;
  std::vector<Ten> tens;
  std::vector<Wt> wts;
  std::vector<ParamInfo> params;
  std::vector<Op> ops;
  std::map<std::string, int> pindex;
  // arena sizes (elements)
  size_t w16_elems = 0, pf_elems = 0, act_elems = 0, grad_elems = 0, f32_elems = 0;
  size_t fo_elems = 0;              // extent of the forward-only layout inside the activation arena
  size_t partial_elems = 0, scratch_elems = 0, small_elems = 0;
  // device arenas
  unsigned short *w16 = nullptr, *act = nullptr, *grad = nullptr, *scratch = nullptr;
  float *pf = nullptr, *f32a = nullptr, *partial = nullptr, *small = nullptr;
  // well-known tensors
  int t_text = -1, t_kv = -1, t_conv_in_out = -1, t_final = -1, act_ids[3] = {-1, -1, -1};
  int act_op_end[3] = {0, 0, 0};   // ops [0, act_op_end[i]) produce captured activation i
  int first_cross_op = 0;           // tape index of the first cross-attention: a backward pass that only wants the text gradient stops there
  int saved_ops = 0;                // number of tape ops the saved forward executed
  // the time-embedding chain (sinusoid -> MLP -> all resnet projections) depends on the timestep only: the 7 passes of
  // a guided step share it.  temb_rows images hold the projections of timestep temb_t (0 rows = nothing cached).
  int temb_ops = 0, temb_rows = 0;
  float temb_t = 0.f;
  hipStream_t temb_stream = nullptr;   // the stream the cached projections were produced on (another stream = miss)
  // the K|V projections of the text (one GEMM over all cross-attention layers) depend on the text embedding only: the
  // caller names the embedding with a key (dh_unet_set_text_key); a forward that finds the projections of the same key,
  // batch and stream in place skips the text conversion and that GEMM (the three optimisation passes of a guided step share
  // the prompt embedding; the CFG pass, whose text differs, overwrites the buffer)
  uint64_t text_key = 0, kv_key = 0;
  int kv_rows = 0;
  hipStream_t kv_stream = nullptr;
  long temb_f32_off = -1;
  int temb_total = 0, kv_total = 0;
  // staging buffers (fixed addresses so a captured graph can be replayed) and graph cache
  float *in_sample = nullptr, *in_text = nullptr, *io_eps = nullptr, *out_dsample = nullptr, *out_dtext = nullptr, *t_dev = nullptr;
  std::map<uint64_t, hipGraphExec_t> graphs;      // 64-bit keys: every field has its own bit range (graph_key_*)
  std::map<uint64_t, double> graph_flops;
  bool use_graphs = true;
  long ones_off = -1, zeros_off = -1;   // f32 vectors of the widest LayerNorm: gamma = 1 / beta = 0 for the unfolded large-batch path
  bool fold_dirty = true;           // a parameter was (re)loaded: W * gamma and the s / t vectors of the folded LayerNorms are stale
  bool owns_weights = true;         // false: w16 / pf belong to the engine this one was shared from (dh_unet_create_shared)
  int n_shared = 0;                 // engines ever shared from this one (its parameters stay frozen once > 0)
  // run state
  int saved_batch = 0;
  const float* saved_sample = nullptr;
  std::vector<char> gready;
  double flops_fwd = 0, flops_bwd = 0;
  int64_t launches = 0;

  bool fo_mode = false;             // the pass being enqueued is a forward nobody differentiates: tensors live at off_fo
  unsigned short* aptr(int t, int B_unused = 0) { return act + (fo_mode && !tens[t].persistent ? tens[t].off_fo : tens[t].off); }
  // gradient buffer of tensor t; a residual input whose gradient is (so far) exactly its consumer's output
  // gradient shares that buffer instead of receiving a copy (galias, reset per backward)
  std::vector<int> galias;
  unsigned short* gptr(int t) {
    while (galias[t] != t) t = galias[t];
    return grad + tens[t].goff;
  }
};

namespace {

struct Builder {
  dh_unet& u;
  int maxB, maxBd;       // largest batch of any forward / of a forward saved for a backward pass
  explicit Builder(dh_unet& uu) : u(uu), maxB(uu.cfg.max_batch), maxBd(uu.cfg.max_diff_batch) {}

  // (offsets are assigned when the tape is complete: layout_tensors)
  int tensor(int rows, int C, bool req_grad = true) {
    Ten t;
    t.rows = rows; t.C = C; t.req_grad = req_grad;
    u.tens.push_back(t);
    return (int)u.tens.size() - 1;
  }
  size_t f32_slot(size_t n) { size_t o = u.f32_elems; u.f32_elems += align_up(n, 64); return o; }

  long param_f32(const std::string& name, int n) {
    ParamInfo p;
    p.name = name; p.ndim = 1; p.shape[0] = n; p.kind = PK_F32;
    p.f32_off = u.pf_elems;
    u.pf_elems += align_up((size_t)n, 64);
    u.pindex[name] = (int)u.params.size();
    u.params.push_back(p);
    return (long)p.f32_off;
  }
  // slice [row_off, row_off+n) of an f32 vector that other params also fill
  void param_f32_at(const std::string& name, int n, size_t off) {
    ParamInfo p;
    p.name = name; p.ndim = 1; p.shape[0] = n; p.kind = PK_F32; p.f32_off = off;
    u.pindex[name] = (int)u.params.size();
    u.params.push_back(p);
  }
  int weight(int N, int K, int taps = 1, bool has_bwd = true, bool f32 = false) {
    Wt w;
    w.N = N; w.K = K; w.taps = taps; w.has_bwd = has_bwd; w.f32 = f32;
    size_t& arena = f32 ? u.pf_elems : u.w16_elems;
    w.fwd_off = arena; arena += align_up((size_t)N * K, 128);
    if (has_bwd) { w.bwd_off = arena; arena += align_up((size_t)N * K, 128); }
    u.wts.push_back(w);
    return (int)u.wts.size() - 1;
  }
  void bind_mat(const std::string& name, int wt, int rows, int row_off, int cin, int taps) {
    ParamInfo p;
    p.name = name; p.kind = PK_MAT; p.wt = wt; p.row_off = row_off;
    if (taps == 9) { p.ndim = 4; p.shape[0] = rows; p.shape[1] = cin; p.shape[2] = 3; p.shape[3] = 3; }
    else { p.ndim = 2; p.shape[0] = rows; p.shape[1] = cin; }
    u.pindex[name] = (int)u.params.size();
    u.params.push_back(p);
  }
  void bind_conv1x1(const std::string& name, int wt, int rows, int cin) {
    bind_mat(name, wt, rows, 0, cin, 1);
    ParamInfo& p = u.params.back();
    p.ndim = 4; p.shape[2] = 1; p.shape[3] = 1;
  }

  // ---- op builders ----------------------------------------------------------------------
  int gn(int x, const std::string& pre, float eps, bool silu) {
    const Ten tx = u.tens[x];   // copy: tensor() grows the vector
    Op o;
    o.type = OP_GN; o.in0 = x; o.out = tensor(tx.rows, tx.C);
    o.gamma_off = param_f32(pre + ".weight", tx.C);
    o.beta_off = param_f32(pre + ".bias", tx.C);
    o.eps = eps; o.silu = silu; o.groups = u.cfg.norm_groups;
    o.stats_off = f32_slot((size_t)maxB * o.groups * 2);
    u.ops.push_back(o);
    return o.out;
  }
  int ln(int x, const std::string& pre) {
    const Ten tx = u.tens[x];
    Op o;
    o.type = OP_LN; o.in0 = x; o.out = tensor(tx.rows, tx.C);
    o.gamma_off = param_f32(pre + ".weight", tx.C);
    o.beta_off = param_f32(pre + ".bias", tx.C);
    o.eps = 1e-5f;
    o.stats_off = f32_slot((size_t)maxB * tx.rows * 2);
    u.ops.push_back(o);
    return o.out;
  }
  // 3x3 conv (pad 1): Hs = source spatial size, stride 1/2, up = source is upsampled 2x first
  int conv3(int x, const std::string& pre, int Cout, int Hs, int stride, int up, long temb_off, int res) {
    const Ten& tx = u.tens[x];
    const int Cin = tx.C;
    const int Hv = Hs << up, Ho = stride == 2 ? Hv / 2 : Hv;
    Op o;
    o.type = OP_GEMM; o.mode = A_CONV3; o.in0 = x; o.res = res;
    o.Hin = Hs; o.Win = Hs; o.Cin = Cin; o.Hout = Ho; o.Wout = Ho; o.stride = stride; o.up = up;
    o.wt = weight(Cout, 9 * Cin, 9);
    bind_mat(pre + ".weight", o.wt, Cout, 0, Cin, 9);
    o.bias_off = param_f32(pre + ".bias", Cout);
    if (temb_off >= 0) { o.rowvec_off = u.temb_f32_off + temb_off; o.rowvec_ld = u.temb_total; }
    o.out = tensor(Ho * Ho, Cout);
    u.ops.push_back(o);
    return o.out;
  }
  // dense GEMM over (a column window of) x
  int linear_w(int x, int wt, long bias_off, int res, int in_col = 0, int in_cols = 0, bool silu = false,
               bool req_grad = true) {
    const Ten tx = u.tens[x];
    Op o;
    o.type = OP_GEMM; o.mode = A_DENSE; o.in0 = x; o.res = res; o.wt = wt; o.bias_off = bias_off;
    o.in_col = in_col; o.in_cols = in_cols ? in_cols : tx.C; o.act_silu = silu;
    o.out = tensor(tx.rows, u.wts[wt].N, req_grad);
    u.ops.push_back(o);
    return o.out;
  }
  int linear(int x, const std::string& pre, int N, bool bias, int res, bool as_conv1x1 = false) {
    const int K = u.tens[x].C;
    const int wt = weight(N, K);
    if (as_conv1x1) bind_conv1x1(pre + ".weight", wt, N, K); else bind_mat(pre + ".weight", wt, N, 0, K, 1);
    const long b = bias ? param_f32(pre + ".bias", N) : -1;
    return linear_w(x, wt, b, res);
  }
  int concat(int a, int b) {
    Op o;
    o.type = OP_CONCAT; o.in0 = a; o.in1 = b;
    o.out = tensor(u.tens[a].rows, u.tens[a].C + u.tens[b].C);
    u.ops.push_back(o);
    return o.out;
  }

  // the GEMM op just pushed consumes the LayerNorm op pushed right before it: fold that LayerNorm into it
  void fold_ln_into_last_gemm() {
    const int gi = (int)u.ops.size() - 1, li = gi - 1;
    Op& g = u.ops[gi];
    Op& l = u.ops[li];
    if (g.type != OP_GEMM || l.type != OP_LN || g.in0 != l.out || g.mode != A_DENSE || g.in_col != 0) return;
    const int N = u.wts[g.wt].N;
    g.ln_fold = li; l.folded = 1;
    g.ln_s_off = (long)u.pf_elems; u.pf_elems += align_up((size_t)N, 64);
    g.ln_t_off = (long)u.pf_elems; u.pf_elems += align_up((size_t)N, 64);
  }

  int temb_cursor = 0, kv_cursor = 0;
  int wt_temb = -1, wt_kv = -1;
  size_t temb_bias_off = 0;

  int resnet(int x, const std::string& pre, int Cout, int H) {
    const int Cin = u.tens[x].C;
    int h = gn(x, pre + ".norm1", 1e-5f, true);
    // time embedding projection rows of the fused weight
    bind_mat(pre + ".time_emb_proj.weight", wt_temb, Cout, temb_cursor, u.cfg.block_out_channels[0] * 4, 1);
    param_f32_at(pre + ".time_emb_proj.bias", Cout, temb_bias_off + temb_cursor);
    const long toff = temb_cursor;
    temb_cursor += Cout;
    h = conv3(h, pre + ".conv1", Cout, H, 1, 0, toff, -1);
    h = gn(h, pre + ".norm2", 1e-5f, true);
    int sc = x;
    if (Cin != Cout) sc = linear(x, pre + ".conv_shortcut", Cout, true, -1, true);
    return conv3(h, pre + ".conv2", Cout, H, 1, 0, -1, sc);
  }

  int transformer(int x, const std::string& pre, int heads, int H) {
    const int C = u.tens[x].C, N = H * H;
    int h = gn(x, pre + ".norm", 1e-6f, false);
    int t0 = linear(h, pre + ".proj_in", C, true, -1);
    const std::string b = pre + ".transformer_blocks.0";
    // self attention: one fused q|k|v GEMM
    int n1 = ln(t0, b + ".norm1");
    const int wqkv = weight(3 * C, C);
    bind_mat(b + ".attn1.to_q.weight", wqkv, C, 0, C, 1);
    bind_mat(b + ".attn1.to_k.weight", wqkv, C, C, C, 1);
    bind_mat(b + ".attn1.to_v.weight", wqkv, C, 2 * C, C, 1);
    int qkv = linear_w(n1, wqkv, -1, -1);
    fold_ln_into_last_gemm();
    Op a;
    a.type = OP_ATTN; a.in0 = qkv; a.heads = heads; a.Nq = N; a.Nk = N; a.cross = 0;
    a.out = tensor(N, C);
    a.lse_off = f32_slot((size_t)maxB * heads * N);
    u.ops.push_back(a);
    int t1 = linear(a.out, b + ".attn1.to_out.0", C, true, t0);
    // cross attention: q GEMM; k|v come from the hoisted text projection
    int n2 = ln(t1, b + ".norm2");
    int q2 = linear(n2, b + ".attn2.to_q", C, false, -1);
    fold_ln_into_last_gemm();
    bind_mat(b + ".attn2.to_k.weight", wt_kv, C, kv_cursor, u.cfg.cross_attention_dim, 1);
    bind_mat(b + ".attn2.to_v.weight", wt_kv, C, kv_cursor + C, u.cfg.cross_attention_dim, 1);
    Op c;
    c.type = OP_ATTN; c.in0 = q2; c.in1 = u.t_kv; c.heads = heads; c.Nq = N; c.Nk = u.cfg.text_len; c.cross = 1;
    c.kv_col = kv_cursor;
    kv_cursor += 2 * C;
    c.out = tensor(N, C);
    c.lse_off = f32_slot((size_t)maxB * heads * N);
    u.ops.push_back(c);
    int t2 = linear(c.out, b + ".attn2.to_out.0", C, true, t1);
    // feed forward (GEGLU)
    int n3 = ln(t2, b + ".norm3");
    int gg = linear(n3, b + ".ff.net.0.proj", 8 * C, true, -1);
    fold_ln_into_last_gemm();
    // value | gate rows in the paired order of the GEMM's lane ownership (unet_kernels.h glu_col): weight rows, bias and the
    // columns of `gg` / its gradient all use it, nothing outside this block sees that tensor
    const int ff1 = (int)u.ops.size() - 1;
    u.wts[u.ops[ff1].wt].glu_F = 4 * C;
    u.params[u.pindex[b + ".ff.net.0.proj.weight"]].glu_F = 4 * C;
    u.params[u.pindex[b + ".ff.net.0.proj.bias"]].glu_F = 4 * C;
    Op g;
    g.type = OP_GEGLU; g.in0 = gg; g.out = tensor(N, 4 * C);
    u.ops.push_back(g);
    u.ops[ff1].glu_op = (int)u.ops.size() - 1;
    int t3 = linear(g.out, b + ".ff.net.2", C, true, t2);
    u.ops.back().glub_op = u.ops[ff1].glu_op;
    return linear(t3, pre + ".proj_out", C, true, x);
  }
};

void count_fused(const dh_unet_config& c, int& temb_total, int& kv_total) {
  const int L = c.n_levels, n = c.layers_per_block;
  temb_total = 0; kv_total = 0;
  for (int i = 0; i < L; ++i) {
    temb_total += n * c.block_out_channels[i];
    if (i < L - 1) kv_total += n * 2 * c.block_out_channels[i];
  }
  temb_total += 2 * c.block_out_channels[L - 1];
  kv_total += 2 * c.block_out_channels[L - 1];
  for (int i = 0; i < L; ++i) {
    const int ch = c.block_out_channels[L - 1 - i];
    temb_total += (n + 1) * ch;
    if (i > 0) kv_total += (n + 1) * 2 * ch;
  }
}

static void layout_tensors(dh_unet& u);

int build(dh_unet& u) {
  const dh_unet_config& c = u.cfg;
  Builder b(u);
  const int L = c.n_levels, n = c.layers_per_block, S = c.sample_size;
  const int* ch = c.block_out_channels;
  const int temb_dim = ch[0] * 4;
  count_fused(c, u.temb_total, u.kv_total);

  // ---- time embedding: sinusoid -> linear+silu -> linear+silu -> fused projections --------
  int t_sin = b.tensor(1, ch[0], false);
  { Op o; o.type = OP_TIMESTEP; o.out = t_sin; u.ops.push_back(o); }
  const int w1 = b.weight(temb_dim, ch[0], 1, false);
  b.bind_mat("time_embedding.linear_1.weight", w1, temb_dim, 0, ch[0], 1);
  int e1 = b.linear_w(t_sin, w1, b.param_f32("time_embedding.linear_1.bias", temb_dim), -1, 0, 0, true, false);
  const int w2 = b.weight(temb_dim, temb_dim, 1, false);
  b.bind_mat("time_embedding.linear_2.weight", w2, temb_dim, 0, temb_dim, 1);
  int e2 = b.linear_w(e1, w2, b.param_f32("time_embedding.linear_2.bias", temb_dim), -1, 0, 0, true, false);
  b.wt_temb = b.weight(u.temb_total, temb_dim, 1, false);
  b.temb_bias_off = u.pf_elems;
  u.pf_elems += align_up((size_t)u.temb_total, 64);
  int t_temb = b.linear_w(e2, b.wt_temb, (long)b.temb_bias_off, -1, 0, 0, false, false);
  u.temb_f32_off = (long)b.f32_slot((size_t)c.max_batch * u.temb_total);
  { Op o; o.type = OP_T2F; o.in0 = t_temb; u.ops.push_back(o); }
  u.temb_ops = (int)u.ops.size();      // ops [0, temb_ops) depend on the timestep only

  // ---- text: hoisted K|V projections of every cross-attention ------------------------------
  u.t_text = b.tensor(c.text_len, c.cross_attention_dim, true);
  b.wt_kv = b.weight(u.kv_total, c.cross_attention_dim);
  u.t_kv = b.linear_w(u.t_text, b.wt_kv, -1, -1);

  // ---- conv_in -------------------------------------------------------------------------------
  { Op o;
    o.type = OP_CONV_IN;
    o.wt = b.weight(ch[0], 9 * c.in_channels, 9, true, true);
    b.bind_mat("conv_in.weight", o.wt, ch[0], 0, c.in_channels, 9);
    o.bias_off = b.param_f32("conv_in.bias", ch[0]);
    o.Hin = S; o.Cin = c.in_channels;
    o.out = b.tensor(S * S, ch[0]);
    u.t_conv_in_out = o.out;
    u.ops.push_back(o); }
  int x = u.t_conv_in_out;
  std::vector<int> skips{x};
  int H = S;
  for (int i = 0; i < L; ++i) {
    const std::string pre = "down_blocks." + std::to_string(i);
    const bool attn = i < L - 1, down = i < L - 1;
    for (int j = 0; j < n; ++j) {
      x = b.resnet(x, pre + ".resnets." + std::to_string(j), ch[i], H);
      if (attn) x = b.transformer(x, pre + ".attentions." + std::to_string(j), c.heads[i], H);
      skips.push_back(x);
    }
    if (down) {
      x = b.conv3(x, pre + ".downsamplers.0.conv", ch[i], H, 2, 0, -1, -1);
      H /= 2;
      skips.push_back(x);
    }
  }
  x = b.resnet(x, "mid_block.resnets.0", ch[L - 1], H);
  x = b.transformer(x, "mid_block.attentions.0", c.heads[L - 1], H);
  x = b.resnet(x, "mid_block.resnets.1", ch[L - 1], H);
  int n_act = 0;
  for (int i = 0; i < L; ++i) {
    const std::string pre = "up_blocks." + std::to_string(i);
    const int co = ch[L - 1 - i];
    const bool attn = i > 0, up = i < L - 1;
    for (int j = 0; j < n + 1; ++j) {
      const int sk = skips.back();
      skips.pop_back();
      x = b.concat(x, sk);
      x = b.resnet(x, pre + ".resnets." + std::to_string(j), co, H);
      if (attn) x = b.transformer(x, pre + ".attentions." + std::to_string(j), c.heads[L - 1 - i], H);
    }
    if (up) {
      x = b.conv3(x, pre + ".upsamplers.0.conv", co, H, 1, 1, -1, -1);
      H *= 2;
    }
    if (attn && n_act < 3) {
      u.act_op_end[n_act] = (int)u.ops.size();
      u.act_ids[n_act++] = x;
    }
  }
  x = b.gn(x, "conv_norm_out", 1e-5f, true);
  u.t_final = x;
  { Op o;
    o.type = OP_CONV_OUT; o.in0 = x;
    o.wt = b.weight(c.out_channels, 9 * ch[0], 9, true, true);
    b.bind_mat("conv_out.weight", o.wt, c.out_channels, 0, ch[0], 9);
    o.bias_off = b.param_f32("conv_out.bias", c.out_channels);
    o.Hin = S; o.Cin = ch[0];
    u.ops.push_back(o); }
  if (b.temb_cursor != u.temb_total || b.kv_cursor != u.kv_total) {
    set_error("internal: fused projection sizes do not match");
    return DH_ERR_STATE;
  }
  u.first_cross_op = 0;
  for (size_t i = 0; i < u.ops.size(); ++i)
    if (u.ops[i].type == OP_ATTN && u.ops[i].cross) { u.first_cross_op = (int)i; break; }
  // a GroupNorm directly after the producer of its input takes its slice statistics from that producer
  // (split-K reduce epilogue / concat copy) instead of a statistics pass of its own
  for (size_t i = 0; i + 1 < u.ops.size(); ++i) {
    Op& o = u.ops[i];
    const Op& nx = u.ops[i + 1];
    if (nx.type == OP_GN && nx.in0 == o.out && (o.type == OP_GEMM || o.type == OP_CONCAT)) o.gn_next = (int)(i + 1);
  }
  { int cmax = 0;
    for (const Op& o : u.ops) if (o.type == OP_LN) cmax = std::max(cmax, u.tens[o.in0].C);
    u.ones_off = (long)u.pf_elems; u.pf_elems += align_up((size_t)cmax, 64);
    u.zeros_off = (long)u.pf_elems; u.pf_elems += align_up((size_t)cmax, 64); }
  layout_tensors(u);
  // scratch: upsample-backward temporary (backward pass only: the saved batch); split-K partial slabs (GEMMs of few output
  // tiles: never the big batches, and the dispatch clamps the splits to what fits); small f32 vectors
  size_t biggest = 0;
  for (const Ten& t : u.tens) biggest = std::max(biggest, (size_t)t.rows * t.C * c.max_diff_batch);
  u.scratch_elems = biggest * 4 + 1024;
  u.partial_elems = std::max<size_t>((size_t)48 << 20, biggest * 2);
  u.small_elems = (size_t)c.max_batch * 64 * 4096 + (size_t)c.max_batch * c.norm_groups * 4 + 4096;
  return DH_OK;
}


// Two layouts of ONE activation arena.
//   saved layout (off): a slot per tensor, max_diff_batch items -- a forward that a backward pass follows leaves everything in
//     place; the gradient arena mirrors it (goff).
//   forward-only layout (off_fo): max_batch items per tensor, tensors share the arena by LIVENESS over the tape: tensor t is
//     live from the first op that writes it to the last op that reads it (inputs of an op: in0, in1, res, the input of a
//     LayerNorm folded into it; outputs: out, and the GEGLU output a fused in-projection writes).  First-fit over the free gaps,
//     in tape order.  The batched CFG pass (B = 2 K) runs in a few hundred MB this way instead of needing every slot doubled.
//   persistent tensors (text, text K|V: cached across passes; the three captured activations: read by the caller after the pass)
//     keep ONE slot of max_batch items in both layouts and are never shared.
static void layout_tensors(dh_unet& u) {
  const size_t maxB = (size_t)u.cfg.max_batch, maxBd = (size_t)u.cfg.max_diff_batch;
  const int nt = (int)u.tens.size(), nops = (int)u.ops.size();
  u.tens[u.t_text].persistent = true;
  u.tens[u.t_kv].persistent = true;
  for (int i = 0; i < 3; ++i)
    if (u.act_ids[i] >= 0) u.tens[u.act_ids[i]].persistent = true;
  std::vector<int> first(nt, nops), last(nt, -1);
  auto rd = [&](int t, int oi) { if (t >= 0) { last[t] = std::max(last[t], oi); first[t] = std::min(first[t], oi); } };
  auto wr = [&](int t, int oi) { if (t >= 0) { first[t] = std::min(first[t], oi); last[t] = std::max(last[t], oi); } };
  for (int oi = 0; oi < nops; ++oi) {
    const Op& o = u.ops[oi];
    rd(o.in0, oi); rd(o.in1, oi); rd(o.res, oi); wr(o.out, oi);
    if (o.ln_fold >= 0) rd(u.ops[o.ln_fold].in0, oi);
    if (o.glu_op >= 0) wr(u.ops[o.glu_op].out, oi);
  }
  // a tensor nothing in the tape writes (an input) or that the caller reads afterwards stays for the whole pass
  std::vector<char> written(nt, 0);
  for (const Op& o : u.ops) { if (o.out >= 0) written[o.out] = 1; if (o.glu_op >= 0) written[u.ops[o.glu_op].out] = 1; }
  for (int t = 0; t < nt; ++t)
    if (!written[t] || t == u.t_final) { first[t] = 0; last[t] = nops; u.tens[t].persistent = u.tens[t].persistent || !written[t]; }
  // saved layout + persistent slots
  u.act_elems = 0; u.grad_elems = 0;
  for (Ten& t : u.tens) {
    t.off = u.act_elems;
    u.act_elems += align_up((size_t)t.rows * t.C * (t.persistent ? maxB : maxBd), 128);
    if (t.req_grad) { t.goff = u.grad_elems; u.grad_elems += align_up((size_t)t.rows * t.C * maxBd, 128); }
  }
  // forward-only layout: first fit among the tensors alive at the definition point
  struct Live { size_t off, end; int last; };
  std::vector<Live> live;
  for (const Ten& t : u.tens)
    if (t.persistent) live.push_back({t.off, t.off + align_up((size_t)t.rows * t.C * maxB, 128), nops + 1});
  std::vector<int> order(nt);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
  size_t fo_end = 0;
  for (const Live& l : live) fo_end = std::max(fo_end, l.end);
  for (int t : order) {
    Ten& tt = u.tens[t];
    if (tt.persistent) { tt.off_fo = tt.off; continue; }
    if (last[t] < 0) { tt.off_fo = 0; continue; }                    // never touched by the tape
    const size_t size = align_up((size_t)tt.rows * tt.C * maxB, 128);
    // tensors whose last reader ran BEFORE this one's first writer are dead (an op may read and write in the same launch, so
    // "before" is strict: the inputs of op `first[t]` are still alive)
    live.erase(std::remove_if(live.begin(), live.end(), [&](const Live& l) { return l.last < first[t]; }), live.end());
    std::sort(live.begin(), live.end(), [](const Live& a, const Live& b) { return a.off < b.off; });
    size_t pos = 0;
    for (const Live& l : live) {
      if (pos + size <= l.off) break;
      pos = std::max(pos, l.end);
    }
    tt.off_fo = pos;
    live.push_back({pos, pos + size, last[t]});
    fo_end = std::max(fo_end, pos + size);
  }
  u.fo_elems = fo_end;
  u.act_elems = std::max(u.act_elems, fo_end);
}

// torch layout [N][C][taps] f32 -> forward matrix rows [row_off, row_off+N) x K = taps*C (k = tap*C + c; tiled 3x3:
// conv_k_index(tap, c)) and the input-gradient matrix rows c x K' = bwd_K (k' = (taps-1-tap)*Nb + col_off + n, likewise).  TILED = the swizzled
// 64x64-tile layout the GEMM streams (wt_index); otherwise plain row-major (the two tiny f32 convolutions).
template <class D, bool TILED>
__global__ void k_load_weight(const float* src, int N, int C, int taps, D* fwd, long fwd_K, long row_off, D* bwd,
                              long bwd_K, long col_off, int Nb, float scale = 1.f, int glu_F = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * C * taps) return;
  const int tap = (int)(idx % taps);
  const int cc = (int)((idx / taps) % C);
  int nn = (int)(idx / ((size_t)taps * C));
  if (glu_F) nn = glu_col(nn >= glu_F ? nn - glu_F : nn, nn >= glu_F ? 1 : 0);      // GEGLU projection: paired row order
  const D v = from_f32<D>(src[idx] * scale);
  int fk = tap * C + cc, bk = (taps - 1 - tap) * Nb + (int)col_off + nn;
  if (TILED && taps == 9) { fk = conv_k_index(tap, cc); bk = conv_k_index(taps - 1 - tap, (int)col_off + nn); }
  const int fr = (int)row_off + nn;
  if (TILED) {
    fwd[wt_index(fr, fk, (int)fwd_K)] = v;
    if (bwd) bwd[wt_index(cc, bk, (int)bwd_K)] = v;
  } else {
    fwd[(size_t)fr * fwd_K + fk] = v;
    if (bwd) bwd[(size_t)cc * bwd_K + bk] = v;
  }
}

}  // namespace
namespace dh {
// torch-layout f32 parameter -> engine weight storage (shared with the VAE decoder engine): rows [row_off, row_off + N) of
// the forward matrix [.][taps * C] (and optionally the transposed / tap-flipped input-gradient matrix), tiled 16-bit when
// dtype is DH_DTYPE_F16 / BF16, plain row-major f32 when dtype is DH_DTYPE_F32
void launch_load_weight(int dtype, const float* src, int N, int C, int taps, void* fwd, long fwd_K, long row_off, void* bwd,
                        long bwd_K, long col_off, int Nb, float scale, hipStream_t st) {
  const size_t total = (size_t)N * C * taps;
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (dtype == DH_DTYPE_F32)
    hipLaunchKernelGGL((k_load_weight<float, false>), dim3(nb), dim3(256), 0, st, src, N, C, taps, (float*)fwd, fwd_K, row_off, (float*)bwd, bwd_K, col_off, Nb, scale);
  else if (dtype == DH_DTYPE_F16)
    hipLaunchKernelGGL((k_load_weight<f16, true>), dim3(nb), dim3(256), 0, st, src, N, C, taps, (f16*)fwd, fwd_K, row_off, (f16*)bwd, bwd_K, col_off, Nb, scale);
  else
    hipLaunchKernelGGL((k_load_weight<bf16, true>), dim3(nb), dim3(256), 0, st, src, N, C, taps, (bf16*)fwd, fwd_K, row_off, (bf16*)bwd, bwd_K, col_off, Nb, scale);
}
}  // namespace dh
namespace {

// LayerNorm fold (one workgroup per output column n of a dense weight [N][K]): reads the unfolded 16-bit W[n][k] from the
// input-gradient copy (rows k, columns n: it is never folded), writes W'[n][k] = round16(W[n][k] * gamma[k]) into the
// forward copy, and leaves s[n] = sum_k W'[n][k] (of the ROUNDED values the MFMA will multiply) and
// t[n] = sum_k beta[k] W[n][k] (+ bias[n]).
template <class D>
__global__ void __launch_bounds__(256) k_fold_ln(const D* bwd, int bwd_K, const float* gamma, const float* beta, const float* bias,
                                                 D* fwd, int K, float* s_out, float* t_out) {
  __shared__ float sm[8];
  const int n = blockIdx.x;
  float s = 0.f, t = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float w = to_f32<D>(bwd[wt_index(k, n, bwd_K)]);
    const D wf = from_f32<D>(w * gamma[k]);
    fwd[wt_index(n, k, K)] = wf;
    s += to_f32<D>(wf);
    t += beta[k] * w;
  }
  s = block_sum(s, sm);
  __syncthreads();
  t = block_sum(t, sm);
  if (threadIdx.x == 0) { s_out[n] = s; t_out[n] = t + (bias ? bias[n] : 0.f); }
}

// [2F] GEGLU bias (value | gate) -> the paired order of the stored weight rows
__global__ void k_load_glu_bias(const float* src, float* dst, int F) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < 2 * F) dst[n] = src[glu_src_row(n, F)];
}

// (re)compute the folded weights and vectors of every LayerNorm-consuming GEMM; runs on `st` before the pass that needs them
__global__ void k_fill_f32(float* p, int n, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// The fold pays where a pass is launch-bound (same-box A/B of the forward pass at 64x64 latents: B = 1 and 2 -2 %, B = 4
// +0.5 %, B = 8 +-0): with many rows the LayerNorm kernel is cheap next to what the row sums cost inside the GEMM's K loop.
// Above this many rows of the widest level the pass runs the LayerNorm as a kernel again -- with gamma = 1, beta = 0,
// because gamma and beta live in the folded weights and in ln_t.
static bool use_ln_fold(const dh_unet* u, int B) {
#ifdef DH_TUNING
  static const long lim = getenv("DH_LN_FOLD_MAXROWS") ? atol(getenv("DH_LN_FOLD_MAXROWS")) : 12288;
#else
  constexpr long lim = 12288;
#endif
  return (long)B * u->cfg.sample_size * u->cfg.sample_size <= lim;
}

static void fold_layernorms(dh_unet* u, hipStream_t st) {
  { int cmax = 0;
    for (const Op& o : u->ops) if (o.type == OP_LN) cmax = std::max(cmax, u->tens[o.in0].C);
    if (cmax > 0) {
      hipLaunchKernelGGL(k_fill_f32, dim3(cdiv(cmax, 256)), dim3(256), 0, st, u->pf + u->ones_off, cmax, 1.f);
      hipLaunchKernelGGL(k_fill_f32, dim3(cdiv(cmax, 256)), dim3(256), 0, st, u->pf + u->zeros_off, cmax, 0.f);
    } }
  for (const Op& g : u->ops) {
    if (g.type != OP_GEMM || g.ln_fold < 0) continue;
    const Op& l = u->ops[g.ln_fold];
    const Wt& w = u->wts[g.wt];
    const float* bias = g.bias_off >= 0 ? u->pf + g.bias_off : nullptr;
    if (u->dtype == DH_DTYPE_F16)
      hipLaunchKernelGGL((k_fold_ln<f16>), dim3(w.N), dim3(256), 0, st, (const f16*)(u->w16 + w.bwd_off), w.N, u->pf + l.gamma_off,
                         u->pf + l.beta_off, bias, (f16*)(u->w16 + w.fwd_off), w.K, u->pf + g.ln_s_off, u->pf + g.ln_t_off);
    else
      hipLaunchKernelGGL((k_fold_ln<bf16>), dim3(w.N), dim3(256), 0, st, (const bf16*)(u->w16 + w.bwd_off), w.N, u->pf + l.gamma_off,
                         u->pf + l.beta_off, bias, (bf16*)(u->w16 + w.fwd_off), w.K, u->pf + g.ln_s_off, u->pf + g.ln_t_off);
  }
  u->fold_dirty = false;
}

}  // namespace

// =============================================================================================
static int create_engine(const dh_unet_config* cfg, dh_unet* parent, dh_unet** out);
extern "C" int dh_unet_create(const dh_unet_config* cfg, dh_unet** out) {
  DH_REQUIRE(cfg && out, "null pointer");
  DH_REQUIRE(cfg->n_levels == 4, "n_levels must be 4");
  DH_REQUIRE(cfg->dtype == DH_DTYPE_F16 || cfg->dtype == DH_DTYPE_BF16, "dtype must be f16 or bf16");
  DH_REQUIRE(cfg->max_batch >= 1 && cfg->max_batch < 4096 && cfg->sample_size % 8 == 0 && cfg->sample_size >= 8, "bad batch / sample size");
  // the kernels index rows with float-reciprocal divisions that are exact below 2^21 rows (common.h div_small)
  DH_REQUIRE((long)cfg->max_batch * cfg->sample_size * cfg->sample_size < (1L << 21), "max_batch * sample_size^2 must stay below 2^21 rows");
  DH_REQUIRE(cfg->norm_groups >= 1 && cfg->norm_groups <= 32, "1..32 GroupNorm groups");
  DH_REQUIRE(cfg->in_channels <= 8 && cfg->out_channels <= 8, "in/out channels must be <= 8");
  DH_REQUIRE(cfg->block_out_channels[0] <= 2048, "block_out_channels[0] must be <= 2048 (few-channel convolutions)");
  DH_REQUIRE(cfg->cross_attention_dim % 64 == 0, "cross_attention_dim must be a multiple of 64");
  for (int i = 0; i < 4; ++i) {
    DH_REQUIRE(cfg->block_out_channels[i] % 64 == 0, "block_out_channels must be multiples of 64");
    DH_REQUIRE(cfg->block_out_channels[i] == cfg->heads[i] * 64, "head dim must be 64");
    DH_REQUIRE(cfg->block_out_channels[i] % cfg->norm_groups == 0, "channels must divide into norm groups");
  }
  return create_engine(cfg, nullptr, out);
}

static int create_engine(const dh_unet_config* cfg, dh_unet* parent, dh_unet** out) {
  dh_unet* u = new dh_unet();
  u->cfg = *cfg;
  if (u->cfg.max_diff_batch <= 0 || u->cfg.max_diff_batch > u->cfg.max_batch) u->cfg.max_diff_batch = u->cfg.max_batch;
  u->dtype = cfg->dtype;
  int rc = build(*u);
  if (rc != DH_OK) { delete u; return rc; }
  auto fail = [&](hipError_t e, const char* what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    dh_unet_destroy(u);
    return DH_ERR_HIP;
  };
  hipError_t e;
  if (parent) {
    // the parameter arenas are laid out by the architecture alone (build() never sizes them by max_batch)
    if (u->w16_elems != parent->w16_elems || u->pf_elems != parent->pf_elems) {
      set_error("internal: shared engine lays its parameters out differently");
      delete u;
      return DH_ERR_STATE;
    }
    u->owns_weights = false;
    u->w16 = parent->w16; u->pf = parent->pf;
    u->fold_dirty = false;
  } else {
    if ((e = hipMalloc((void**)&u->w16, u->w16_elems * 2 + 256)) != hipSuccess) return fail(e, "hipMalloc weights");
    if ((e = hipMalloc((void**)&u->pf, u->pf_elems * 4 + 256)) != hipSuccess) return fail(e, "hipMalloc f32 params");
  }
  if ((e = hipMalloc((void**)&u->act, u->act_elems * 2 + 256)) != hipSuccess) return fail(e, "hipMalloc activations");
  if ((e = hipMalloc((void**)&u->grad, u->grad_elems * 2 + 256)) != hipSuccess) return fail(e, "hipMalloc gradients");
  if ((e = hipMalloc((void**)&u->f32a, u->f32_elems * 4 + 256)) != hipSuccess) return fail(e, "hipMalloc f32 arena");
  if ((e = hipMalloc((void**)&u->partial, u->partial_elems * 4)) != hipSuccess) return fail(e, "hipMalloc split-K");
  if ((e = hipMalloc((void**)&u->scratch, u->scratch_elems * 2)) != hipSuccess) return fail(e, "hipMalloc scratch");
  if ((e = hipMalloc((void**)&u->small, u->small_elems * 4)) != hipSuccess) return fail(e, "hipMalloc small");
  {
    const dh_unet_config& c = u->cfg;
    const size_t ns = (size_t)c.max_batch * c.sample_size * c.sample_size;
    const size_t nt = (size_t)c.max_batch * c.text_len * c.cross_attention_dim;
    if ((e = hipMalloc((void**)&u->in_sample, ns * c.in_channels * 4)) != hipSuccess) return fail(e, "hipMalloc staging");
    if ((e = hipMalloc((void**)&u->out_dsample, ns * c.in_channels * 4)) != hipSuccess) return fail(e, "hipMalloc staging");
    if ((e = hipMalloc((void**)&u->io_eps, ns * c.out_channels * 4)) != hipSuccess) return fail(e, "hipMalloc staging");
    if ((e = hipMalloc((void**)&u->in_text, nt * 4)) != hipSuccess) return fail(e, "hipMalloc staging");
    if ((e = hipMalloc((void**)&u->out_dtext, nt * 4)) != hipSuccess) return fail(e, "hipMalloc staging");
    if ((e = hipMalloc((void**)&u->t_dev, 64)) != hipSuccess) return fail(e, "hipMalloc staging");
    u->use_graphs = !(getenv("DH_GRAPH") && atoi(getenv("DH_GRAPH")) == 0);
  }
  if (!parent) {
    (void)hipMemset(u->w16, 0, u->w16_elems * 2);
    (void)hipMemset(u->pf, 0, u->pf_elems * 4);
  }
  u->gready.assign(u->tens.size(), 0);
  u->galias.resize(u->tens.size());
  std::iota(u->galias.begin(), u->galias.end(), 0);
  *out = u;
  return DH_OK;
}

extern "C" int dh_unet_create_shared(dh_unet* parent, int max_batch, void* stream, dh_unet** out) {
  DH_REQUIRE(parent && out, "null pointer");
  DH_REQUIRE(parent->owns_weights, "share from the engine that owns the weights");
  dh_unet_config cfg = parent->cfg;
  if (max_batch > 0 && max_batch != cfg.max_batch) { cfg.max_batch = max_batch; cfg.max_diff_batch = 0; }      // (another size: every batch may be saved)
  DH_REQUIRE(cfg.max_batch < 4096 && (long)cfg.max_batch * cfg.sample_size * cfg.sample_size < (1L << 21), "bad max_batch");
  // W * gamma and the s / t vectors of the folded LayerNorms live in the shared arenas: finalise them before anyone reads
  if (parent->fold_dirty) {
    fold_layernorms(parent, (hipStream_t)stream);
    DH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  }
  int rc = create_engine(&cfg, parent, out);
  if (rc == DH_OK) ++parent->n_shared;
  return rc;
}

extern "C" void dh_unet_destroy(dh_unet* u) {
  if (!u) return;
  if (u->owns_weights) { (void)hipFree(u->w16); (void)hipFree(u->pf); }      // (a shared engine never touches its parent here: destruction order at process exit is not ours)
  (void)hipFree(u->act); (void)hipFree(u->grad);
  (void)hipFree(u->f32a); (void)hipFree(u->partial); (void)hipFree(u->scratch); (void)hipFree(u->small);
  (void)hipFree(u->in_sample); (void)hipFree(u->in_text); (void)hipFree(u->io_eps); (void)hipFree(u->out_dsample);
  (void)hipFree(u->out_dtext); (void)hipFree(u->t_dev);
  for (auto& kv : u->graphs) (void)hipGraphExecDestroy(kv.second);
  delete u;
}

extern "C" int dh_unet_num_params(const dh_unet* u) { return u ? (int)u->params.size() : 0; }

extern "C" int dh_unet_param_info(const dh_unet* u, int i, const char** name, int* ndim, int64_t* shape4) {
  DH_REQUIRE(u && i >= 0 && i < (int)u->params.size() && name && ndim && shape4, "bad arguments");
  const ParamInfo& p = u->params[i];
  *name = p.name.c_str();
  *ndim = p.ndim;
  for (int k = 0; k < 4; ++k) shape4[k] = p.shape[k];
  return DH_OK;
}

extern "C" int dh_unet_load_param(dh_unet* u, int i, const float* src, void* stream) {
  DH_REQUIRE(u && src && i >= 0 && i < (int)u->params.size(), "bad arguments");
  if (!u->owns_weights || u->n_shared > 0) {
    set_error("dh_unet_load_param: the weights are shared (dh_unet_create_shared): load every parameter before sharing");
    return DH_ERR_STATE;
  }
  u->temb_rows = 0; u->fold_dirty = true; u->kv_key = 0;     // cached time-embedding projections / folded LayerNorm weights belong to the old parameters
  hipStream_t st = (hipStream_t)stream;
  const ParamInfo& p = u->params[i];
  if (p.kind == PK_F32) {
    if (p.glu_F) {
      hipLaunchKernelGGL(k_load_glu_bias, dim3(cdiv((int)p.shape[0], 256)), dim3(256), 0, st, src, u->pf + p.f32_off, p.glu_F);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
    DH_CHECK_HIP(hipMemcpyAsync(u->pf + p.f32_off, src, (size_t)p.shape[0] * 4, hipMemcpyDeviceToDevice, st));
    return DH_OK;
  }
  const Wt& w = u->wts[p.wt];
  const int N = (int)p.shape[0], C = (int)p.shape[1], taps = w.taps;
  const size_t total = (size_t)N * C * taps;
  const unsigned nb = (unsigned)((total + 255) / 256);
  const long fwd_ld = w.K;                       // = taps * C
  const long bwd_ld = (long)taps * w.N;          // fused linear: total N; conv: 9 * N
  if (w.f32) {
    float* f = u->pf + w.fwd_off;
    float* b = w.has_bwd ? u->pf + w.bwd_off : nullptr;
    hipLaunchKernelGGL((k_load_weight<float, false>), dim3(nb), dim3(256), 0, st, src, N, C, taps, f, fwd_ld, (long)p.row_off,
                       b, bwd_ld, (long)p.row_off, w.N);
  } else if (u->dtype == DH_DTYPE_F16) {
    f16* f = (f16*)(u->w16 + w.fwd_off);
    f16* b = w.has_bwd ? (f16*)(u->w16 + w.bwd_off) : nullptr;
    hipLaunchKernelGGL((k_load_weight<f16, true>), dim3(nb), dim3(256), 0, st, src, N, C, taps, f, fwd_ld, (long)p.row_off, b,
                       bwd_ld, (long)p.row_off, w.N, 1.f, w.glu_F);
  } else {
    bf16* f = (bf16*)(u->w16 + w.fwd_off);
    bf16* b = w.has_bwd ? (bf16*)(u->w16 + w.bwd_off) : nullptr;
    hipLaunchKernelGGL((k_load_weight<bf16, true>), dim3(nb), dim3(256), 0, st, src, N, C, taps, f, fwd_ld, (long)p.row_off, b,
                       bwd_ld, (long)p.row_off, w.N, 1.f, w.glu_F);
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" size_t dh_unet_weight_bytes(const dh_unet* u) { return u && u->owns_weights ? u->w16_elems * 2 + u->pf_elems * 4 : 0; }   // (a shared engine holds none)
extern "C" size_t dh_unet_workspace_bytes(const dh_unet* u) {
  return u ? (u->act_elems + u->grad_elems + u->scratch_elems) * 2 + (u->f32_elems + u->partial_elems + u->small_elems) * 4 : 0;
}

// ---------------------------------------------------------------------------------- forward
static void fill_gemm(dh_unet* u, const Op& o, int B, GemmArgs& g) {
  const Wt& w = u->wts[o.wt];
  const Ten& ti = u->tens[o.in0];
  const Ten& to = u->tens[o.out];
  g.A = u->aptr(o.in0) + o.in_col; g.lda = ti.C;
  g.W = u->w16 + w.fwd_off;
  g.M = B * to.rows; g.N = w.N; g.K = w.K;
  g.mode = o.mode; g.Hin = o.Hin; g.Win = o.Win; g.Cin = o.Cin; g.Hout = o.Hout; g.Wout = o.Wout;
  g.stride = o.stride; g.up = o.up;
  g.bias = o.bias_off >= 0 ? u->pf + o.bias_off : nullptr;
  if (o.rowvec_off >= 0) { g.rowvec = u->f32a + o.rowvec_off; g.rowvec_ld = o.rowvec_ld; g.rows_per_batch = to.rows; }
  if (o.res >= 0) { g.R = u->aptr(o.res); g.ldr = u->tens[o.res].C; }
  g.C = u->aptr(o.out); g.ldc = to.C;
  g.act_silu = o.act_silu;
  g.partial = u->partial; g.partial_elems = u->partial_elems;
  if (o.ln_fold >= 0) {
    const Op& l = u->ops[o.ln_fold];
    if (use_ln_fold(u, B)) {     // A = the LayerNorm's INPUT; the normalisation happens in the epilogue (bias is inside ln_t)
      g.A = u->aptr(l.in0); g.lda = u->tens[l.in0].C;
      g.bias = nullptr;
      g.ln_s = u->pf + o.ln_s_off; g.ln_t = u->pf + o.ln_t_off; g.ln_stats = u->f32a + l.stats_off; g.ln_eps = l.eps;
    } else {
      g.bias = u->pf + o.ln_t_off;      // A = the plain-normalised tensor (OP_LN ran with gamma 1, beta 0); W * gamma and t hold the affine part
    }
  }
}

static void forward_ops(dh_unet* u, int B, int n_ops, int first_op, bool kv_hit, bool save, hipStream_t st) {
  const int dt = u->dtype;
  const dh_unet_config& c = u->cfg;
  u->flops_fwd = 0;
  int gn_have = 0;       // the op just executed left the GroupNorm slice statistics of its output in u->small
  if (!kv_hit) launch_f32_to_t(dt, u->in_text, u->aptr(u->t_text), (size_t)B * c.text_len * c.cross_attention_dim, st);
  for (int oi = first_op; oi < n_ops; ++oi) {
    const Op& o = u->ops[oi];
    if (kv_hit && oi == u->temb_ops) continue;        // the hoisted text K|V projection (the op right after the time-embedding chain)
    switch (o.type) {
      case OP_TIMESTEP:
        launch_timestep_embedding(dt, u->t_dev, c.block_out_channels[0], B, u->aptr(o.out), st);
        break;
      case OP_T2F:
        launch_t_to_f32(dt, u->aptr(o.in0), u->f32a + u->temb_f32_off, (size_t)B * u->temb_total, 0, st);
        break;
      case OP_CONV_IN: {
        const Wt& w = u->wts[o.wt];
        launch_conv_small_fwd(dt, u->in_sample, 1, u->pf + w.fwd_off, u->pf + o.bias_off, u->aptr(o.out), 0, B, o.Hin, o.Hin,
                              o.Cin, w.N, st);
        u->flops_fwd += 2.0 * B * o.Hin * o.Hin * w.N * w.K;
        break;
      }
      case OP_CONV_OUT: {
        const Wt& w = u->wts[o.wt];
        launch_conv_small_fwd(dt, u->aptr(o.in0), 0, u->pf + w.fwd_off, u->pf + o.bias_off, u->io_eps, 1, B, o.Hin, o.Hin,
                              o.Cin, w.N, st);
        u->flops_fwd += 2.0 * B * o.Hin * o.Hin * w.N * w.K;
        break;
      }
      case OP_GEMM: {
        GemmArgs g;
        fill_gemm(u, o, B, g);
        gn_have = 0;
        if (o.gn_next >= 0 && oi + 1 < n_ops) {        // the next op normalises this output: statistics ride along
          const Ten& to = u->tens[o.out];
          g.gn_part = u->small; g.gn_HW = to.rows; g.gn_G = u->ops[o.gn_next].groups; g.gn_done = &gn_have;
        }
        if (o.glu_op >= 0 && o.glu_op < n_ops) {
          // the GEGLU that follows runs in this GEMM's epilogue; the pre-activations go to memory only for a backward pass
          const Op& ge = u->ops[o.glu_op];
          g.glu_y = u->aptr(ge.out); g.glu_ldy = u->tens[ge.out].C;
          if (!save) g.C = nullptr;
        }
        u->flops_fwd += launch_gemm(dt, g, st);
        break;
      }
      case OP_GN: {
        const Ten& t = u->tens[o.in0];
        const int have = (oi > 0 && u->ops[oi - 1].gn_next == oi) ? gn_have : 0;
        launch_groupnorm_fwd(dt, u->aptr(o.in0), u->pf + o.gamma_off, u->pf + o.beta_off, u->aptr(o.out),
                             u->f32a + o.stats_off, u->small, B, t.rows, t.C, o.groups, o.eps, o.silu, st, have);
        gn_have = 0;
        break;
      }
      case OP_LN: {
        if (o.folded && use_ln_fold(u, B)) break;           // normalised inside the GEMM that follows (fill_gemm)
        const Ten& t = u->tens[o.in0];
        // (a folded LayerNorm on the large-batch path normalises without the affine part: it sits in the GEMM's weights)
        launch_layernorm_fwd(dt, u->aptr(o.in0), u->pf + (o.folded ? u->ones_off : o.gamma_off),
                             u->pf + (o.folded ? u->zeros_off : o.beta_off), u->aptr(o.out),
                             u->f32a + o.stats_off, B * t.rows, t.C, o.eps, st);
        break;
      }
      case OP_ATTN: {
        const Ten& tq = u->tens[o.in0];
        const int C = u->tens[o.out].C;
        if (!o.cross) {
          const unsigned short* qkv = u->aptr(o.in0);
          launch_attention_fwd(dt, qkv, tq.C, qkv + C, qkv + 2 * C, tq.C, u->aptr(o.out), C, u->f32a + o.lse_off, B,
                               o.heads, o.Nq, o.Nk, st);
        } else {
          const Ten& tkv = u->tens[o.in1];
          const unsigned short* kv = u->aptr(o.in1) + o.kv_col;
          launch_attention_fwd(dt, u->aptr(o.in0), tq.C, kv, kv + C, tkv.C, u->aptr(o.out), C, u->f32a + o.lse_off, B,
                               o.heads, o.Nq, o.Nk, st);
        }
        u->flops_fwd += 4.0 * B * o.heads * (double)o.Nq * o.Nk * 64;
        break;
      }
      case OP_GEGLU: {
        if (oi > 0 && u->ops[oi - 1].glu_op == oi) break;      // done in the epilogue of the GEMM in front of it
        const Ten& t = u->tens[o.out];
        launch_geglu_fwd(dt, u->aptr(o.in0), u->aptr(o.out), B * t.rows, t.C, st);
        break;
      }
      case OP_CONCAT: {
        const Ten &a = u->tens[o.in0], &b2 = u->tens[o.in1], &t = u->tens[o.out];
        gn_have = 0;
        const int G = o.gn_next >= 0 ? u->ops[o.gn_next].groups : 0;
        if (o.gn_next >= 0 && oi + 1 < n_ops && a.C % 8 == 0 && b2.C % 8 == 0 && t.C % G == 0 &&
            (GN_GB * (t.C / G)) % 8 == 0 && GN_GB * (t.C / G) <= 2048) {
          launch_concat_gn(dt, u->aptr(o.in0), a.C, u->aptr(o.in1), b2.C, u->aptr(o.out), u->small, B, t.rows, G, st);
          gn_have = 1;
        } else {
          launch_copy_cols(dt, u->aptr(o.in0), a.C, u->aptr(o.out), t.C, B * t.rows, a.C, 0, st);
          launch_copy_cols(dt, u->aptr(o.in1), b2.C, u->aptr(o.out) + a.C, t.C, B * t.rows, b2.C, 0, st);
        }
        break;
      }
    }
  }
}

// run `body` through a cached hipGraph (captured on first use of `key`) or eagerly
// graph cache keys: batch in bits 0-11 (max_batch < 2^12 by dh_unet_create), the tape length in 12-27, the
// backward's activation mask in 28-30, its flags in 31-33, time-embedding hit in 34, forward / backward in 35, text K|V hit in 36,
// forward saved for a backward pass in 37 (a forward nobody differentiates does not write the GEGLU pre-activations)
static uint64_t graph_key_fwd(int B, int n_ops, bool temb_hit, bool kv_hit, bool save) {
  return (uint64_t)B | ((uint64_t)n_ops << 12) | ((uint64_t)(temb_hit ? 1 : 0) << 34) | ((uint64_t)(kv_hit ? 1 : 0) << 36) |
         ((uint64_t)(save ? 1 : 0) << 37);
}
static uint64_t graph_key_bwd(int B, unsigned mask, bool eps, bool sample, bool text) {
  return (uint64_t)B | ((uint64_t)(mask & 7u) << 28) | ((uint64_t)(eps ? 1 : 0) << 31) | ((uint64_t)(sample ? 1 : 0) << 32) |
         ((uint64_t)(text ? 1 : 0) << 33) | (1ull << 35);
}

template <class F>
static int run_graphed(dh_unet* u, uint64_t key, hipStream_t st, double* flops_slot, F body) {
  // the legacy default stream (0) cannot be captured: callers that want graph replay run on a created stream
  if (!u->use_graphs || st == nullptr || gemm_profiling_on()) { body(); return DH_OK; }
  auto it = u->graphs.find(key);
  if (it == u->graphs.end()) {
    hipGraph_t graph = nullptr;
    DH_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    body();
    DH_CHECK_HIP(hipStreamEndCapture(st, &graph));
    hipGraphExec_t exec = nullptr;
    DH_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    u->graphs[key] = exec;
    u->graph_flops[key] = *flops_slot;
    it = u->graphs.find(key);
  }
  *flops_slot = u->graph_flops[key];
  DH_CHECK_HIP(hipGraphLaunch(it->second, st));
  return DH_OK;
}

extern "C" int dh_unet_forward(dh_unet* u, const float* sample, float timestep, const float* text, int batch,
                               int save_for_backward, float* eps_out, void* const* act_out, void* stream) {
  DH_REQUIRE(u && sample && text, "null pointer");
  DH_REQUIRE(batch >= 1 && batch <= u->cfg.max_batch, "batch exceeds max_batch");
  DH_REQUIRE(!save_for_backward || batch <= u->cfg.max_diff_batch, "a forward saved for a backward pass exceeds max_diff_batch");
  hipStream_t st = (hipStream_t)stream;
  const int B = batch;
  const dh_unet_config& c = u->cfg;
  const size_t ns = (size_t)B * c.sample_size * c.sample_size;
  // (a caller that filled / reads the engine's own buffers -- dh_unet_io_ptr -- passes those pointers: nothing to copy)
  if (sample != u->in_sample) DH_CHECK_HIP(hipMemcpyAsync(u->in_sample, sample, ns * c.in_channels * 4, hipMemcpyDeviceToDevice, st));
  // without eps_out the tape stops after the last requested activation (the tail's only consumer is eps)
  int n_ops = (int)u->ops.size();
  if (!eps_out) {
    n_ops = 0;
    for (int i = 0; i < 3; ++i)
      if (act_out && act_out[i]) n_ops = std::max(n_ops, u->act_op_end[i]);
    DH_REQUIRE(n_ops > 0, "nothing requested: eps_out and every act_out are null");
  }
  if (u->fold_dirty) fold_layernorms(u, st);
  const bool temb_hit = u->temb_rows >= B && u->temb_t == timestep && u->temb_stream == st;
  const bool kv_hit = u->text_key != 0 && u->text_key == u->kv_key && u->kv_rows == B && u->kv_stream == st;
  const int first_op = temb_hit ? u->temb_ops : 0;
  // the text and the timestep are only read by the ops a cache hit skips (K|V projection of the text; time embedding)
  if (!kv_hit && text != u->in_text)
    DH_CHECK_HIP(hipMemcpyAsync(u->in_text, text, (size_t)B * c.text_len * c.cross_attention_dim * 4, hipMemcpyDeviceToDevice, st));
  if (!temb_hit) launch_set_scalar(u->t_dev, timestep, st);
  const bool save = save_for_backward != 0;
  // a forward nobody differentiates runs in the forward-only layout of the arena (layout_tensors); it overwrites what a saved
  // forward left there, which is why it also clears saved_batch below
  u->fo_mode = !save;
  int rc = run_graphed(u, graph_key_fwd(B, n_ops, temb_hit, kv_hit, save), st, &u->flops_fwd,
                       [&]() { forward_ops(u, B, n_ops, first_op, kv_hit, save, st); });
  u->fo_mode = false;
  if (rc != DH_OK) { u->temb_rows = 0; u->kv_key = 0; return rc; }      // nothing cached after a failed capture / launch
  if (!temb_hit) { u->temb_t = timestep; u->temb_rows = B; u->temb_stream = st; }
  if (!kv_hit) { u->kv_key = u->text_key; u->kv_rows = B; u->kv_stream = st; }   // key 0: the buffer now holds an unnamed text
  if (eps_out && eps_out != u->io_eps)
    DH_CHECK_HIP(hipMemcpyAsync(eps_out, u->io_eps, ns * c.out_channels * 4, hipMemcpyDeviceToDevice, st));
  if (act_out) {
    for (int i = 0; i < 3; ++i)
      if (act_out[i]) {
        DH_REQUIRE(u->act_op_end[i] <= n_ops, "activation requested beyond the truncated tape");
        const Ten& t = u->tens[u->act_ids[i]];
        if (act_out[i] != (void*)u->aptr(u->act_ids[i]))
          DH_CHECK_HIP(hipMemcpyAsync(act_out[i], u->aptr(u->act_ids[i]), (size_t)B * t.rows * t.C * 2,
                                      hipMemcpyDeviceToDevice, st));
      }
  }
  u->saved_batch = save_for_backward ? B : 0;
  u->saved_ops = n_ops;
  DH_LAUNCH_CHECK();
  return DH_OK;
}

// --------------------------------------------------------------------------------- backward
namespace {
struct Bwd {
  dh_unet* u;
  int B, dt;
  hipStream_t st;
  bool need_text;
  // dst grad (+)= src (column window copy)
  void add_into(int t, const unsigned short* src, long lds_, int cols, int col_off = 0) {
    const Ten& tt = u->tens[t];
    launch_copy_cols(dt, src, lds_, u->gptr(t) + col_off, tt.C, B * tt.rows, cols, u->gready[t] ? 1 : 0, st);
  }
};
}  // namespace

static void backward_ops(dh_unet* u, int B, unsigned act_mask, bool has_eps, bool want_sample, bool want_text,
                         hipStream_t st) {
  const int dt = u->dtype;
  const dh_unet_config& c = u->cfg;
  const float* d_eps = has_eps ? u->io_eps : nullptr;
  float* d_sample = want_sample ? u->out_dsample : nullptr;
  float* d_text = want_text ? u->out_dtext : nullptr;
  u->flops_bwd = 0;
  std::fill(u->gready.begin(), u->gready.end(), 0);
  for (int i = 0; i < 3; ++i)
    if (act_mask & (1u << i)) u->gready[u->act_ids[i]] = 1;     // seeded by the caller's copy
  Bwd bw{u, B, dt, st, d_text != nullptr};
  bool text_grad_written = false;
  int gnb_have = 0, gnb_for = -1;       // the split-K reduce just run left the backward statistics of GroupNorm op gnb_for
  int lnb_have = 0, lnb_for = -1;       // the split-K reduce just run applied the backward of LayerNorm op lnb_for
  int cat_done = -1;                    // the GroupNorm backward just run wrote the gradient of concatenation op cat_done to its sources
  int glub_done = -1;                   // the input-gradient GEMM just run wrote the pre-activation gradient of GEGLU op glub_done
  for (int oi = (int)u->ops.size() - 1; oi >= 0; --oi) {
    const Op& o = u->ops[oi];
    // only the text gradient is wanted (null-text optimisation): nothing below the first cross-attention contributes to it
    // (its self-attention, proj_in, the first resnet, conv_in) -- except the hoisted K|V projection of the text itself
    if (!want_sample && oi < u->first_cross_op && o.out != u->t_kv) continue;
    switch (o.type) {
      case OP_CONV_OUT: {
        if (!d_eps) break;
        const Wt& w = u->wts[o.wt];
        // dX[C] <- d_eps[4]: few-in kernel with the flipped/transposed weights (write)
        launch_conv_small_bwd(dt, d_eps, 1, u->pf + w.bwd_off, u->gptr(o.in0), 0, 0, B, o.Hin, o.Hin, w.N, o.Cin, st);
        u->gready[o.in0] = 1;
        u->flops_bwd += 2.0 * B * o.Hin * o.Hin * w.N * w.K;
        break;
      }
      case OP_CONV_IN: {
        if (!d_sample || !u->gready[o.out]) break;
        const Wt& w = u->wts[o.wt];
        launch_conv_small_bwd(dt, u->gptr(o.out), 0, u->pf + w.bwd_off, d_sample, 1, 0, B, o.Hin, o.Hin, w.N, o.Cin, st);
        u->flops_bwd += 2.0 * B * o.Hin * o.Hin * w.N * w.K;
        break;
      }
      case OP_GEMM: {
        if (!u->tens[o.out].req_grad || !u->gready[o.out]) break;
        const Ten& to = u->tens[o.out];
        const Ten& ti = u->tens[o.in0];
        const Wt& w = u->wts[o.wt];
        if (o.res >= 0 && u->tens[o.res].req_grad) {
          if (u->gready[o.res]) bw.add_into(o.res, u->gptr(o.out), to.C, to.C);
          else u->galias[o.res] = o.out;      // first contribution: share dOut (nobody reads it after this op)
          u->gready[o.res] = 1;
        }
        if (!ti.req_grad || !w.has_bwd) break;
        if (o.in0 == u->t_text && !bw.need_text) break;
        GemmArgs g;
        g.A = u->gptr(o.out); g.lda = to.C;
        g.W = u->w16 + w.bwd_off;
        g.partial = u->partial; g.partial_elems = u->partial_elems;
        if (o.mode == A_DENSE) {
          g.mode = A_DENSE;
          g.M = B * to.rows; g.N = w.K; g.K = w.N;
          g.C = u->gptr(o.in0) + o.in_col; g.ldc = ti.C;
          if (u->gready[o.in0]) { g.R = g.C; g.ldr = ti.C; }
          lnb_have = 0;
          if (oi > 0 && u->ops[oi - 1].type == OP_LN && u->ops[oi - 1].out == o.in0 && !u->gready[o.in0] && o.in_col == 0 &&
              w.K == ti.C) {
            // the gradient being written is dy of the LayerNorm that is processed next and this GEMM is its only consumer:
            // a split-K reduce applies that LayerNorm's backward to the summed rows directly
            const Op& ln = u->ops[oi - 1];
            g.lnb_x = u->aptr(ln.in0); g.lnb_gamma = u->pf + ln.gamma_off; g.lnb_stats = u->f32a + ln.stats_off;
            g.lnb_add = u->gready[ln.in0] ? u->gptr(ln.in0) : nullptr; g.lnb_dx = u->gptr(ln.in0);
            g.lnb_done = &lnb_have;
            lnb_for = oi - 1;
          }
          if (oi > 0 && u->ops[oi - 1].type == OP_GN && u->ops[oi - 1].out == o.in0 && !u->gready[o.in0] && o.in_col == 0 &&
              w.K == ti.C) {
            // ... or dy of the GroupNorm processed next (a transformer's norm in front of proj_in): the split-K reduce or the
            // GEMM's own epilogue leaves that GroupNorm's backward slice statistics, as for the convolutions below
            const Op& gn = u->ops[oi - 1];
            const Ten& tx = u->tens[gn.in0];
            g.gnb_x = u->aptr(gn.in0); g.gnb_ldx = tx.C;
            g.gnb_gamma = u->pf + gn.gamma_off; g.gnb_beta = u->pf + gn.beta_off; g.gnb_stats = u->f32a + gn.stats_off;
            g.gnb_silu = gn.silu;
            g.gn_part = u->small; g.gn_HW = tx.rows; g.gn_G = gn.groups; g.gn_done = &gnb_have;
            gnb_have = 0;
            gnb_for = oi - 1;
          }
          if (o.glub_op >= 0 && o.glub_op == oi - 1 && !u->gready[o.in0] && o.in_col == 0 && w.K == ti.C) {
            // the gradient being written is dy of the GEGLU processed next and this GEMM is its only consumer: the epilogue
            // applies the GEGLU backward to its tile and writes the gradient of the pre-activations (dy never goes to memory)
            const Op& ge = u->ops[o.glub_op];
            g.glub_x = u->aptr(ge.in0); g.glub_dx = u->gptr(ge.in0); g.C = nullptr;
            u->gready[ge.in0] = 1;
            glub_done = o.glub_op;
          }
          u->flops_bwd += launch_gemm(dt, g, st);
          u->gready[o.in0] = 1;
        } else if (o.up) {
          // gradient w.r.t. the upsampled image, then 2x2 sum back to the source resolution
          g.mode = A_CONV3; g.stride = 1; g.up = 0;
          g.Hin = o.Hout; g.Win = o.Wout; g.Cin = w.N; g.Hout = o.Hout; g.Wout = o.Wout;
          g.M = B * o.Hout * o.Wout; g.N = o.Cin; g.K = 9 * w.N;
          g.C = u->scratch; g.ldc = o.Cin;
          u->flops_bwd += launch_gemm(dt, g, st);
          launch_pool2x2_sum(dt, u->scratch, u->gptr(o.in0), B, o.Hin, o.Win, o.Cin, u->gready[o.in0] ? 1 : 0, st);
          u->gready[o.in0] = 1;
        } else {
          g.mode = o.stride == 2 ? A_CONVT2 : A_CONV3;
          g.stride = 1; g.up = 0;
          g.Hin = o.Hout; g.Win = o.Wout; g.Cin = w.N;          // source = dY
          g.Hout = o.Hin; g.Wout = o.Win;                        // output = dX
          g.M = B * o.Hin * o.Win; g.N = o.Cin; g.K = 9 * w.N;
          g.C = u->gptr(o.in0); g.ldc = ti.C;
          if (u->gready[o.in0]) { g.R = g.C; g.ldr = ti.C; }
          gnb_have = 0;
          // the GroupNorm whose output this convolution reads: the op in front of it, or -- a resnet that changes the channel count
          // runs its 1x1 shortcut between norm2 and conv2 -- the one before that (the shortcut's own backward, a dense GEMM on
          // the resnet's input, touches neither the statistics in u->small nor gnb_have)
          int gi = -1;
          if (oi > 0 && u->ops[oi - 1].type == OP_GN && u->ops[oi - 1].out == o.in0) gi = oi - 1;
          else if (oi > 1 && u->ops[oi - 1].type == OP_GEMM && u->ops[oi - 1].mode == A_DENSE && u->ops[oi - 1].out == o.res &&
                   u->ops[oi - 1].in0 != o.in0 && u->ops[oi - 2].type == OP_GN && u->ops[oi - 2].out == o.in0)
            gi = oi - 2;
          if (gi >= 0) {
            // the gradient being written is dy of the GroupNorm that is processed next: a split-K reduce (or this launch's own
            // epilogue) also leaves that GroupNorm's backward slice statistics
            const Op& gn = u->ops[gi];
            const Ten& tx = u->tens[gn.in0];
            g.gnb_x = u->aptr(gn.in0); g.gnb_ldx = tx.C;
            g.gnb_gamma = u->pf + gn.gamma_off; g.gnb_beta = u->pf + gn.beta_off; g.gnb_stats = u->f32a + gn.stats_off;
            g.gnb_silu = gn.silu;
            g.gn_part = u->small; g.gn_HW = tx.rows; g.gn_G = gn.groups; g.gn_done = &gnb_have;
            gnb_for = gi;
          }
          u->flops_bwd += launch_gemm(dt, g, st);
          u->gready[o.in0] = 1;
        }
        if (o.in0 == u->t_text) text_grad_written = true;
        break;
      }
      case OP_GN: {
        if (!u->gready[o.out]) break;
        const Ten& t = u->tens[o.in0];
        GnBwdSplit split;
        if (oi > 0 && u->ops[oi - 1].type == OP_CONCAT && u->ops[oi - 1].out == o.in0) {
          // x is the concatenation made by the op in front: this GroupNorm is the last writer of its gradient, so the two
          // halves go straight to the concatenated tensors' gradients (no k_split_cols pass) when nothing was accumulated
          // there yet
          const Op& cat = u->ops[oi - 1];
          const int ca = u->tens[cat.in0].C;
          if (!u->gready[cat.in0] && !u->gready[cat.in1] && ca % 8 == 0 && ca + u->tens[cat.in1].C == t.C) {
            split.out0 = u->gptr(cat.in0); split.out1 = u->gptr(cat.in1); split.split_c = ca;
            cat_done = oi - 1;
          }
        }
        launch_groupnorm_bwd(dt, u->aptr(o.in0), u->gptr(o.out), u->pf + o.gamma_off, u->pf + o.beta_off,
                             u->f32a + o.stats_off, u->gptr(o.in0), u->small, B, t.rows, t.C, o.groups, o.silu,
                             u->gready[o.in0] ? 1 : 0, st, gnb_for == oi ? gnb_have : 0, split);
        gnb_have = 0;
        u->gready[o.in0] = 1;
        if (split.out0) { u->gready[u->ops[oi - 1].in0] = 1; u->gready[u->ops[oi - 1].in1] = 1; }
        break;
      }
      case OP_LN: {
        if (!u->gready[o.out]) break;
        if (lnb_have && lnb_for == oi) {      // done by the split-K reduce of the GEMM in front of it
          lnb_have = 0;
          u->gready[o.in0] = 1;
          break;
        }
        const Ten& t = u->tens[o.in0];
        launch_layernorm_bwd(dt, u->aptr(o.in0), u->gptr(o.out), u->pf + o.gamma_off, u->f32a + o.stats_off,
                             u->gready[o.in0] ? u->gptr(o.in0) : nullptr, u->gptr(o.in0), B * t.rows, t.C, st);
        u->gready[o.in0] = 1;
        break;
      }
      case OP_GEGLU: {
        if (!u->gready[o.out]) break;
        if (glub_done == oi) { glub_done = -1; break; }
        const Ten& t = u->tens[o.out];
        launch_geglu_bwd(dt, u->aptr(o.in0), u->gptr(o.out), u->gptr(o.in0), B * t.rows, t.C, st);
        u->gready[o.in0] = 1;
        break;
      }
      case OP_ATTN: {
        if (!u->gready[o.out]) break;
        const Ten& tq = u->tens[o.in0];
        const int C = u->tens[o.out].C;
        float* delta = u->small;
        if (!o.cross) {
          const unsigned short* qkv = u->aptr(o.in0);
          unsigned short* dqkv = u->gptr(o.in0);
          launch_attention_bwd_dq(dt, qkv, tq.C, qkv + C, qkv + 2 * C, tq.C, u->aptr(o.out), C, u->gptr(o.out), C,
                                  u->f32a + o.lse_off, delta, dqkv, tq.C, B, o.heads, o.Nq, o.Nk, st);
          launch_attention_bwd_dkv(dt, qkv, tq.C, qkv + C, qkv + 2 * C, tq.C, u->gptr(o.out), C, u->f32a + o.lse_off,
                                   delta, dqkv + C, dqkv + 2 * C, tq.C, B, o.heads, o.Nq, o.Nk, st);
          u->flops_bwd += 14.0 * B * o.heads * (double)o.Nq * o.Nk * 64;
        } else {
          const Ten& tkv = u->tens[o.in1];
          const unsigned short* kv = u->aptr(o.in1) + o.kv_col;
          launch_attention_bwd_dq(dt, u->aptr(o.in0), tq.C, kv, kv + C, tkv.C, u->aptr(o.out), C, u->gptr(o.out), C,
                                  u->f32a + o.lse_off, delta, u->gptr(o.in0), tq.C, B, o.heads, o.Nq, o.Nk, st);
          u->flops_bwd += 6.0 * B * o.heads * (double)o.Nq * o.Nk * 64;
          if (bw.need_text) {
            unsigned short* dkv = u->gptr(o.in1) + o.kv_col;
            launch_attention_bwd_dkv(dt, u->aptr(o.in0), tq.C, kv, kv + C, tkv.C, u->gptr(o.out), C,
                                     u->f32a + o.lse_off, delta, dkv, dkv + C, tkv.C, B, o.heads, o.Nq, o.Nk, st, u->partial,
                                     u->partial_elems);
            u->gready[o.in1] = 1;
            u->flops_bwd += 8.0 * B * o.heads * (double)o.Nq * o.Nk * 64;
          }
        }
        u->gready[o.in0] = 1;
        break;
      }
      case OP_CONCAT: {
        if (!u->gready[o.out]) break;
        if (cat_done == oi) { cat_done = -1; break; }     // written by the GroupNorm backward behind it
        const Ten &a = u->tens[o.in0], &b2 = u->tens[o.in1], &t = u->tens[o.out];
        launch_split_cols(dt, u->gptr(o.out), t.C, u->gptr(o.in0), a.C, a.C, u->gready[o.in0] ? 1 : 0, u->gptr(o.in1), b2.C,
                          b2.C, u->gready[o.in1] ? 1 : 0, B * t.rows, st);
        u->gready[o.in0] = 1;
        u->gready[o.in1] = 1;
        break;
      }
      default: break;
    }
  }
  if (d_text) {
    const size_t n = (size_t)B * c.text_len * c.cross_attention_dim;
    if (text_grad_written) launch_t_to_f32(dt, u->gptr(u->t_text), d_text, n, 0, st);
    else (void)hipMemsetAsync(d_text, 0, n * 4, st);
  }
}

extern "C" int dh_unet_backward(dh_unet* u, void* const* d_act, const float* d_eps, float* d_sample, float* d_text,
                                void* stream) {
  DH_REQUIRE(u, "null engine");
  DH_REQUIRE(u->saved_batch > 0, "no saved forward: call dh_unet_forward(save_for_backward=1) first");
  hipStream_t st = (hipStream_t)stream;
  const int B = u->saved_batch;
  const dh_unet_config& c = u->cfg;
  const size_t ns = (size_t)B * c.sample_size * c.sample_size;
  const size_t nt = (size_t)B * c.text_len * c.cross_attention_dim;
  unsigned mask = 0;
  std::iota(u->galias.begin(), u->galias.end(), 0);
  DH_REQUIRE(!d_eps || u->saved_ops == (int)u->ops.size(), "d_eps given but the saved forward did not compute eps");
  if (d_act)
    for (int i = 0; i < 3; ++i)
      if (d_act[i]) {
        DH_REQUIRE(u->act_op_end[i] <= u->saved_ops, "d_act given for an activation the saved forward did not compute");
        const Ten& t = u->tens[u->act_ids[i]];
        if (d_act[i] != (void*)u->gptr(u->act_ids[i]))
          DH_CHECK_HIP(hipMemcpyAsync(u->gptr(u->act_ids[i]), d_act[i], (size_t)B * t.rows * t.C * 2,
                                      hipMemcpyDeviceToDevice, st));
        mask |= 1u << i;
      }
  if (d_eps && d_eps != u->io_eps) DH_CHECK_HIP(hipMemcpyAsync(u->io_eps, d_eps, ns * c.out_channels * 4, hipMemcpyDeviceToDevice, st));
  const uint64_t key = graph_key_bwd(B, mask, d_eps != nullptr, d_sample != nullptr, d_text != nullptr);
  int rc = run_graphed(u, key, st, &u->flops_bwd,
                       [&]() { backward_ops(u, B, mask, d_eps != nullptr, d_sample != nullptr, d_text != nullptr, st); });
  if (rc != DH_OK) return rc;
  if (d_sample && d_sample != u->out_dsample)
    DH_CHECK_HIP(hipMemcpyAsync(d_sample, u->out_dsample, ns * c.in_channels * 4, hipMemcpyDeviceToDevice, st));
  if (d_text && d_text != u->out_dtext) DH_CHECK_HIP(hipMemcpyAsync(d_text, u->out_dtext, nt * 4, hipMemcpyDeviceToDevice, st));
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_unet_io_ptr(dh_unet* u, int which, int index, void** ptr, size_t* bytes) {
  DH_REQUIRE(u && ptr, "null pointer");
  const dh_unet_config& c = u->cfg;
  const size_t ns = (size_t)c.max_batch * c.sample_size * c.sample_size;
  const size_t nt = (size_t)c.max_batch * c.text_len * c.cross_attention_dim;
  size_t n = 0;
  switch (which) {
    case DH_IO_SAMPLE: *ptr = u->in_sample; n = ns * c.in_channels * 4; break;
    case DH_IO_TEXT: *ptr = u->in_text; n = nt * 4; break;
    case DH_IO_EPS: *ptr = u->io_eps; n = ns * c.out_channels * 4; break;
    case DH_IO_DSAMPLE: *ptr = u->out_dsample; n = ns * c.in_channels * 4; break;
    case DH_IO_DTEXT: *ptr = u->out_dtext; n = nt * 4; break;
    case DH_IO_ACT:
    case DH_IO_ACT_GRAD: {
      DH_REQUIRE(index >= 0 && index < 3 && u->act_ids[index] >= 0, "activation index out of range");
      const Ten& t = u->tens[u->act_ids[index]];
      *ptr = which == DH_IO_ACT ? (void*)(u->act + t.off) : (void*)(u->grad + t.goff);
      n = (size_t)(which == DH_IO_ACT ? c.max_batch : c.max_diff_batch) * t.rows * t.C * 2;
      break;
    }
    default: DH_REQUIRE(false, "unknown buffer");
  }
  if (bytes) *bytes = n;
  return DH_OK;
}

extern "C" int dh_unet_set_text_key(dh_unet* u, unsigned long long key) {
  DH_REQUIRE(u, "null engine");
  u->text_key = key;
  return DH_OK;
}

extern "C" int dh_unet_stats(const dh_unet* u, double* flops_fwd, double* flops_bwd, int64_t* launches) {
  DH_REQUIRE(u, "null engine");
  if (flops_fwd) *flops_fwd = u->flops_fwd;
  if (flops_bwd) *flops_bwd = u->flops_bwd;
  if (launches) *launches = (int64_t)u->ops.size();
  return DH_OK;
}
