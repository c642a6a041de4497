// SD AutoencoderKL DECODER on the U-Net engine's kernels (widening row f3 of SURVEY.md section 8: the decode at the
// end of every edit, reference guided_stable_diffuser.py:481-483 `decode_latent_image`, :286; stable_null_inverter.py:105
// `latent2image`; diffusers AutoencoderKL [ext], restated in diffusionhandles_amd/vae.py with diffusers' parameter names).
//
//   z [B,h,w,4] f32 (already divided by the scaling factor and passed through post_quant_conv on the host: a 4x4 matrix
//   per pixel) -> conv_in 4->C3 -> mid (resnet, single-head attention of head dim C3, resnet) -> 4 up blocks of 3 resnets
//   (C3, C3, C2, C1; nearest-2x + conv after the first three) -> GroupNorm + SiLU -> conv_out C1->3 -> image [B,8h,8w,3] f32.
//
// Forward only.  Channels-last 16-bit activations; every 3x3 convolution / linear is the MFMA implicit GEMM of gemm.hip
// (the nearest-2x upsample is fused into the consumer's A gather), GroupNorm(+SiLU) the engine's two-stage kernels; the
// attention has ONE head of dim 512, so it runs as two GEMMs around a row softmax (S = Q K^T with K tiled as the "weight",
// O = P V with V^T tiled) instead of the head-dim-64 flash kernel.  Images are processed one at a time (512 x 512 x 128
// channels is 262144 GEMM rows: one image fills the chip, and the row index arithmetic stays below its 2^21-row limit).
#include <algorithm>
#include <string>
#include <vector>

#include "unet_kernels.h"

namespace dh {

template <class T>
__global__ void k_tile_rows(const T* src, long ld, int col0, T* dst, int N, int K, int transpose) {
  // dst (tiled [N][K]) <- transpose ? src[k][col0 + n] : src[n][col0 + k]
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * K) return;
  int n, k;
  if (transpose) { k = (int)(idx / N); n = (int)(idx - (size_t)k * N); }      // consecutive threads read consecutive columns
  else { n = (int)(idx / K); k = (int)(idx - (size_t)n * K); }
  dst[wt_index(n, k, K)] = transpose ? src[(size_t)k * ld + col0 + n] : src[(size_t)n * ld + col0 + k];
}

// in-place softmax of every row of S [rows][cols] (16-bit storage, f32 math); one wave per row, three passes over the row
template <class T>
__global__ void __launch_bounds__(256) k_softmax_rows(T* s, int rows, int cols) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  T* p = s + (size_t)row * cols;
  float mx = -INFINITY;
  for (int c = lane * 8; c < cols; c += 512) {
    const uint4 raw = *reinterpret_cast<const uint4*>(p + c);
    const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) mx = fmaxf(mx, to_f32<T>(v[i]));
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    const uint4 raw = *reinterpret_cast<const uint4*>(p + c);
    const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += __expf(to_f32<T>(v[i]) - mx);
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int c = lane * 8; c < cols; c += 512) {
    uint4 raw = *reinterpret_cast<const uint4*>(p + c);
    T* v = reinterpret_cast<T*>(&raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = from_f32<T>(__expf(to_f32<T>(v[i]) - mx) * inv);
    *reinterpret_cast<uint4*>(p + c) = raw;
  }
}

enum VOp { V_CONV_IN, V_CONV_OUT, V_CONV, V_LINEAR, V_GN, V_ATTN };

struct VTen { size_t off; int H, C; };     // [H*H][C] per image
struct VWt { size_t off; int N, K, taps; bool f32; };
struct VParam { std::string name; int ndim; int64_t shape[4]; int wt; int row_off; long f32_off; float scale; };
struct VOpRec {
  int type = 0, in0 = -1, out = -1, res = -1, wt = -1;
  long bias = -1, gamma = -1, beta = -1;
  int Hin = 0, Cin = 0, Hout = 0, up = 0, silu = 0, stride = 1, pad = 1;
  float eps = 1e-6f;
  size_t stats = 0;
};

}  // namespace dh
using namespace dh;

struct dh_vae_decoder {
  dh_vae_config cfg;
  int dtype = DH_DTYPE_F16;
  std::vector<VTen> tens;
  std::vector<VWt> wts;
  std::vector<VParam> params;
  std::vector<VOpRec> ops;
  size_t w16_elems = 0, pf_elems = 0, act_elems = 0, f32_elems = 0, small_elems = 0, scores_elems = 0, tiled_elems = 0;
  unsigned short *w16 = nullptr, *act = nullptr, *scores = nullptr, *tiled = nullptr;
  float *pf = nullptr, *f32a = nullptr, *small = nullptr, *partial = nullptr;
  size_t partial_elems = 0;
  int t_out = -1;
  bool encoder = false;       // the tape is AutoencoderKL.encoder (image in, 2 * latent_channels moments out)
};

namespace {
struct VB {
  dh_vae_decoder& v;
  explicit VB(dh_vae_decoder& d) : v(d) {}
  int tensor(int H, int C) {
    VTen t{v.act_elems, H, C};
    v.act_elems += align_up((size_t)H * H * C, 128);
    v.tens.push_back(t);
    return (int)v.tens.size() - 1;
  }
  long pf32(const std::string& name, int n, float scale = 1.f) {
    VParam p{name, 1, {n, 0, 0, 0}, -1, 0, (long)v.pf_elems, scale};
    v.pf_elems += align_up((size_t)n, 64);
    v.params.push_back(p);
    return p.f32_off;
  }
  void pf32_at(const std::string& name, int n, long off, float scale) {
    VParam p{name, 1, {n, 0, 0, 0}, -1, 0, off, scale};
    v.params.push_back(p);
  }
  int weight(int N, int K, int taps, bool f32) {
    VWt w{f32 ? v.pf_elems : v.w16_elems, N, K, taps, f32};
    (f32 ? v.pf_elems : v.w16_elems) += align_up((size_t)N * K, 128);
    v.wts.push_back(w);
    return (int)v.wts.size() - 1;
  }
  void bind(const std::string& name, int wt, int rows, int row_off, int cin, int taps, float scale = 1.f) {
    VParam p{name, taps == 9 ? 4 : 2, {rows, cin, taps == 9 ? 3 : 0, taps == 9 ? 3 : 0}, wt, row_off, -1, scale};
    v.params.push_back(p);
  }
  void bind1x1(const std::string& name, int wt, int rows, int cin) {
    VParam p{name, 4, {rows, cin, 1, 1}, wt, 0, -1, 1.f};
    v.params.push_back(p);
  }
  int gn(int x, const std::string& pre, bool silu) {
    const VTen t = v.tens[x];
    VOpRec o;
    o.type = V_GN; o.in0 = x; o.out = tensor(t.H, t.C); o.silu = silu;
    o.gamma = pf32(pre + ".weight", t.C); o.beta = pf32(pre + ".bias", t.C);
    o.stats = v.f32_elems; v.f32_elems += 64 * 2;
    v.ops.push_back(o);
    return o.out;
  }
  int conv3(int x, const std::string& pre, int Cout, int up, int res, int stride = 1, int pad = 1) {
    const VTen t = v.tens[x];
    VOpRec o;
    o.type = V_CONV; o.in0 = x; o.res = res; o.Hin = t.H; o.Cin = t.C; o.up = up; o.stride = stride; o.pad = pad;
    o.Hout = stride == 2 ? t.H / 2 : t.H << up;
    o.wt = weight(Cout, 9 * t.C, 9, false);
    bind(pre + ".weight", o.wt, Cout, 0, t.C, 9);
    o.bias = pf32(pre + ".bias", Cout);
    o.out = tensor(o.Hout, Cout);
    v.ops.push_back(o);
    return o.out;
  }
  int resnet(int x, const std::string& pre, int Cout) {
    const int Cin = v.tens[x].C;
    int h = gn(x, pre + ".norm1", true);
    h = conv3(h, pre + ".conv1", Cout, 0, -1);
    h = gn(h, pre + ".norm2", true);
    int sc = x;
    if (Cin != Cout) {
      VOpRec o;
      o.type = V_LINEAR; o.in0 = x; o.wt = weight(Cout, Cin, 1, false);
      bind1x1(pre + ".conv_shortcut.weight", o.wt, Cout, Cin);
      o.bias = pf32(pre + ".conv_shortcut.bias", Cout);
      o.out = tensor(v.tens[x].H, Cout);
      v.ops.push_back(o);
      sc = o.out;
    }
    return conv3(h, pre + ".conv2", Cout, 0, sc);
  }
  // single-head self-attention over the H*H tokens with a residual connection (the VAE mid block)
  int attention(int x, const std::string& pre) {
    const int Ctop = v.tens[x].C, S = v.tens[x].H;
    int t = gn(x, pre + ".group_norm", false);
    VOpRec q;
    q.type = V_LINEAR; q.in0 = t; q.wt = weight(3 * Ctop, Ctop, 1, false);
    const float sc = 1.f / sqrtf((float)Ctop);            // the softmax scale is folded into the q projection
    bind(pre + ".to_q.weight", q.wt, Ctop, 0, Ctop, 1, sc);
    bind(pre + ".to_k.weight", q.wt, Ctop, Ctop, Ctop, 1);
    bind(pre + ".to_v.weight", q.wt, Ctop, 2 * Ctop, Ctop, 1);
    q.bias = (long)v.pf_elems; v.pf_elems += align_up((size_t)3 * Ctop, 64);
    pf32_at(pre + ".to_q.bias", Ctop, q.bias, sc);
    pf32_at(pre + ".to_k.bias", Ctop, q.bias + Ctop, 1.f);
    pf32_at(pre + ".to_v.bias", Ctop, q.bias + 2 * Ctop, 1.f);
    q.out = tensor(S, 3 * Ctop);
    v.ops.push_back(q);
    VOpRec a;
    a.type = V_ATTN; a.in0 = q.out; a.out = tensor(S, Ctop);
    v.ops.push_back(a);
    VOpRec o;
    o.type = V_LINEAR; o.in0 = a.out; o.res = x; o.wt = weight(Ctop, Ctop, 1, false);
    bind(pre + ".to_out.0.weight", o.wt, Ctop, 0, Ctop, 1);
    o.bias = pf32(pre + ".to_out.0.bias", Ctop);
    o.out = tensor(S, Ctop);
    v.ops.push_back(o);
    v.scores_elems = (size_t)S * S * S * S;
    v.tiled_elems = (size_t)S * S * Ctop;
    return o.out;
  }
};

// Activation arena with liveness reuse (the tape is straight-line and runs one image at a time): a tensor's region is
// taken when the op that writes it is reached -- before that op's inputs are released, so an output never overlaps what
// its op reads -- and returned to a first-fit free list after its last reader.  64x64 latents: 2.0 GB -> ~0.3 GB (decoder).
void plan_arena(dh_vae_decoder& v) {
  const int nt = (int)v.tens.size(), no = (int)v.ops.size();
  std::vector<int> last(nt, -1);
  for (int i = 0; i < no; ++i) {
    const VOpRec& o = v.ops[i];
    if (o.in0 >= 0) last[o.in0] = i;
    if (o.res >= 0) last[o.res] = i;
    if (o.out >= 0) last[o.out] = std::max(last[o.out], i);
  }
  std::vector<std::pair<size_t, size_t>> free_list;      // (offset, elements), sorted by offset
  std::vector<char> placed(nt, 0);
  size_t top = 0;
  auto size_of = [&](int t) { return align_up((size_t)v.tens[t].H * v.tens[t].H * v.tens[t].C, 128); };
  auto take = [&](int t) {
    const size_t need = size_of(t);
    for (size_t k = 0; k < free_list.size(); ++k)
      if (free_list[k].second >= need) {
        v.tens[t].off = free_list[k].first;
        free_list[k].first += need; free_list[k].second -= need;
        if (free_list[k].second == 0) free_list.erase(free_list.begin() + (long)k);
        return;
      }
    if (!free_list.empty() && free_list.back().first + free_list.back().second == top) {      // grow the trailing hole
      v.tens[t].off = free_list.back().first;
      top = free_list.back().first + need;
      free_list.pop_back();
      return;
    }
    v.tens[t].off = top; top += need;
  };
  auto give = [&](int t) {
    std::pair<size_t, size_t> blk(v.tens[t].off, size_of(t));
    auto it = std::lower_bound(free_list.begin(), free_list.end(), blk);
    it = free_list.insert(it, blk);
    if (it + 1 != free_list.end() && it->first + it->second == (it + 1)->first) { it->second += (it + 1)->second; free_list.erase(it + 1); }
    if (it != free_list.begin() && (it - 1)->first + (it - 1)->second == it->first) { (it - 1)->second += it->second; free_list.erase(it); }
  };
  for (int i = 0; i < no; ++i) {
    const VOpRec& o = v.ops[i];
    if (o.out >= 0 && !placed[o.out]) { take(o.out); placed[o.out] = 1; }
    for (int t : {o.in0, o.res, o.out})
      if (t >= 0 && placed[t] == 1 && last[t] == i) { give(t); placed[t] = 2; }
  }
  v.act_elems = top;
}

int build(dh_vae_decoder& v) {
  const dh_vae_config& c = v.cfg;
  VB b(v);
  const int* ch = c.block_out_channels;        // 128 256 512 512
  const int L = 4, Ctop = ch[L - 1], S = c.latent_size;
  // conv_in (few-in direct convolution, f32 weights)
  int x = b.tensor(S, Ctop);
  { VOpRec o;
    o.type = V_CONV_IN; o.out = x; o.Hin = S; o.Cin = c.latent_channels;
    o.wt = b.weight(Ctop, 9 * c.latent_channels, 9, true);
    b.bind("decoder.conv_in.weight", o.wt, Ctop, 0, c.latent_channels, 9);
    o.bias = b.pf32("decoder.conv_in.bias", Ctop);
    v.ops.push_back(o); }
  x = b.resnet(x, "decoder.mid_block.resnets.0", Ctop);
  x = b.attention(x, "decoder.mid_block.attentions.0");
  x = b.resnet(x, "decoder.mid_block.resnets.1", Ctop);
  for (int i = 0; i < L; ++i) {
    const int co = ch[L - 1 - i];
    const std::string pre = "decoder.up_blocks." + std::to_string(i);
    for (int j = 0; j < c.layers_per_block + 1; ++j) x = b.resnet(x, pre + ".resnets." + std::to_string(j), co);
    if (i < L - 1) x = b.conv3(x, pre + ".upsamplers.0.conv", co, 1, -1);
  }
  x = b.gn(x, "decoder.conv_norm_out", true);
  { VOpRec o;
    o.type = V_CONV_OUT; o.in0 = x; o.Hin = v.tens[x].H; o.Cin = v.tens[x].C;
    o.wt = b.weight(c.out_channels, 9 * o.Cin, 9, true);
    b.bind("decoder.conv_out.weight", o.wt, c.out_channels, 0, o.Cin, 9);
    o.bias = b.pf32("decoder.conv_out.bias", c.out_channels);
    v.ops.push_back(o); }
  v.t_out = x;
  size_t biggest = 0;
  for (const VTen& t : v.tens) biggest = std::max(biggest, (size_t)t.H * t.H * t.C);
  v.partial_elems = std::max<size_t>((size_t)16 << 20, biggest);
  v.small_elems = (size_t)64 * 4096 + 4096;
  plan_arena(v);
  return DH_OK;
}

// AutoencoderKL.encoder: conv_in, four down blocks (two ResNets each, a stride-2 convolution padded bottom / right behind
// the first three), the mid block, GroupNorm + SiLU + conv_out to 2 * latent_channels moments
int build_encoder(dh_vae_decoder& v) {
  const dh_vae_config& c = v.cfg;
  VB b(v);
  const int* ch = c.block_out_channels;        // 128 256 512 512
  const int L = 4, R = 8 * c.latent_size;
  v.encoder = true;
  int x = b.tensor(R, ch[0]);
  { VOpRec o;
    o.type = V_CONV_IN; o.out = x; o.Hin = R; o.Cin = c.out_channels;
    o.wt = b.weight(ch[0], 9 * c.out_channels, 9, true);
    b.bind("encoder.conv_in.weight", o.wt, ch[0], 0, c.out_channels, 9);
    o.bias = b.pf32("encoder.conv_in.bias", ch[0]);
    v.ops.push_back(o); }
  for (int i = 0; i < L; ++i) {
    const std::string pre = "encoder.down_blocks." + std::to_string(i);
    for (int j = 0; j < c.layers_per_block; ++j) x = b.resnet(x, pre + ".resnets." + std::to_string(j), ch[i]);
    if (i < L - 1) x = b.conv3(x, pre + ".downsamplers.0.conv", ch[i], 0, -1, 2, 0);
  }
  x = b.resnet(x, "encoder.mid_block.resnets.0", ch[L - 1]);
  x = b.attention(x, "encoder.mid_block.attentions.0");
  x = b.resnet(x, "encoder.mid_block.resnets.1", ch[L - 1]);
  x = b.gn(x, "encoder.conv_norm_out", true);
  { VOpRec o;
    o.type = V_CONV_OUT; o.in0 = x; o.Hin = v.tens[x].H; o.Cin = v.tens[x].C;
    o.wt = b.weight(2 * c.latent_channels, 9 * o.Cin, 9, true);
    b.bind("encoder.conv_out.weight", o.wt, 2 * c.latent_channels, 0, o.Cin, 9);
    o.bias = b.pf32("encoder.conv_out.bias", 2 * c.latent_channels);
    v.ops.push_back(o); }
  v.t_out = x;
  size_t biggest = 0;
  for (const VTen& t : v.tens) biggest = std::max(biggest, (size_t)t.H * t.H * t.C);
  v.partial_elems = std::max<size_t>((size_t)16 << 20, biggest);
  v.small_elems = (size_t)64 * 4096 + 4096;
  plan_arena(v);
  return DH_OK;
}
}  // namespace

static int vae_create(const dh_vae_config* cfg, dh_vae_decoder** out, bool encoder) {
  DH_REQUIRE(cfg && out, "null pointer");
  DH_REQUIRE(cfg->dtype == DH_DTYPE_F16 || cfg->dtype == DH_DTYPE_BF16, "dtype must be f16 or bf16");
  DH_REQUIRE(cfg->latent_size >= 8 && cfg->latent_size % 8 == 0, "latent size must be a multiple of 8");
  DH_REQUIRE((long)cfg->latent_size * cfg->latent_size * 64 < (1L << 21), "image too large for the row index arithmetic (8 * latent_size)^2 < 2^21");
  DH_REQUIRE(cfg->latent_channels >= 1 && cfg->latent_channels <= 8 && cfg->out_channels >= 1 && cfg->out_channels <= 8, "1..8 latent / image channels");
  DH_REQUIRE(cfg->norm_groups >= 1 && cfg->norm_groups <= 32 && cfg->layers_per_block >= 1, "bad configuration");
  for (int i = 0; i < 4; ++i)
    DH_REQUIRE(cfg->block_out_channels[i] % 64 == 0 && cfg->block_out_channels[i] % cfg->norm_groups == 0, "block_out_channels must be multiples of 64 and of the group count");
  DH_REQUIRE(((long)cfg->latent_size * cfg->latent_size) % 64 == 0, "token count must be a multiple of 64 (attention as GEMMs)");
  dh_vae_decoder* v = new dh_vae_decoder();
  v->cfg = *cfg;
  v->dtype = cfg->dtype;
  int rc = encoder ? build_encoder(*v) : build(*v);
  if (rc != DH_OK) { delete v; return rc; }
  auto fail = [&](hipError_t e, const char* what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    dh_vae_decoder_destroy(v);
    return DH_ERR_HIP;
  };
  hipError_t e;
  if ((e = hipMalloc((void**)&v->w16, v->w16_elems * 2 + 256)) != hipSuccess) return fail(e, "hipMalloc weights");
  if ((e = hipMalloc((void**)&v->pf, v->pf_elems * 4 + 256)) != hipSuccess) return fail(e, "hipMalloc f32 params");
  if ((e = hipMalloc((void**)&v->act, v->act_elems * 2 + 256)) != hipSuccess) return fail(e, "hipMalloc activations");
  if ((e = hipMalloc((void**)&v->scores, v->scores_elems * 2 + 256)) != hipSuccess) return fail(e, "hipMalloc attention scores");
  if ((e = hipMalloc((void**)&v->tiled, v->tiled_elems * 2 + 256)) != hipSuccess) return fail(e, "hipMalloc tiled operand");
  if ((e = hipMalloc((void**)&v->f32a, v->f32_elems * 4 + 256)) != hipSuccess) return fail(e, "hipMalloc statistics");
  if ((e = hipMalloc((void**)&v->small, v->small_elems * 4)) != hipSuccess) return fail(e, "hipMalloc scratch");
  if ((e = hipMalloc((void**)&v->partial, v->partial_elems * 4)) != hipSuccess) return fail(e, "hipMalloc split-K");
  (void)hipMemset(v->w16, 0, v->w16_elems * 2);
  (void)hipMemset(v->pf, 0, v->pf_elems * 4);
  *out = v;
  return DH_OK;
}

extern "C" int dh_vae_decoder_create(const dh_vae_config* cfg, dh_vae_decoder** out) { return vae_create(cfg, out, false); }
extern "C" int dh_vae_encoder_create(const dh_vae_config* cfg, dh_vae_decoder** out) {
  DH_REQUIRE(cfg && 2 * cfg->latent_channels <= 8, "the moments (2 * latent_channels) must fit the few-output convolution (<= 8)");
  return vae_create(cfg, out, true);
}

extern "C" void dh_vae_decoder_destroy(dh_vae_decoder* v) {
  if (!v) return;
  (void)hipFree(v->w16); (void)hipFree(v->pf); (void)hipFree(v->act); (void)hipFree(v->scores); (void)hipFree(v->tiled);
  (void)hipFree(v->f32a); (void)hipFree(v->small); (void)hipFree(v->partial);
  delete v;
}

extern "C" int dh_vae_decoder_num_params(const dh_vae_decoder* v) { return v ? (int)v->params.size() : 0; }

extern "C" int dh_vae_decoder_param_info(const dh_vae_decoder* v, int i, const char** name, int* ndim, int64_t* shape4) {
  DH_REQUIRE(v && i >= 0 && i < (int)v->params.size() && name && ndim && shape4, "bad arguments");
  const VParam& p = v->params[i];
  *name = p.name.c_str();
  *ndim = p.ndim;
  for (int k = 0; k < 4; ++k) shape4[k] = p.shape[k];
  return DH_OK;
}

__global__ void k_scale_copy(const float* src, float* dst, int n, float s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] * s;
}

extern "C" int dh_vae_decoder_load_param(dh_vae_decoder* v, int i, const float* src, void* stream) {
  DH_REQUIRE(v && src && i >= 0 && i < (int)v->params.size(), "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const VParam& p = v->params[i];
  if (p.wt < 0) {
    hipLaunchKernelGGL(k_scale_copy, dim3(cdiv((int)p.shape[0], 256)), dim3(256), 0, st, src, v->pf + p.f32_off, (int)p.shape[0], p.scale);
    DH_LAUNCH_CHECK();
    return DH_OK;
  }
  const VWt& w = v->wts[p.wt];
  void* dst = w.f32 ? (void*)(v->pf + w.off) : (void*)(v->w16 + w.off);
  launch_load_weight(w.f32 ? DH_DTYPE_F32 : v->dtype, src, (int)p.shape[0], (int)p.shape[1], w.taps, dst, w.K, p.row_off, nullptr, 0, 0, w.N,
                     p.scale, st);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" size_t dh_vae_decoder_bytes(const dh_vae_decoder* v) {
  return v ? (v->w16_elems + v->act_elems + v->scores_elems + v->tiled_elems) * 2 + (v->pf_elems + v->f32_elems + v->small_elems + v->partial_elems) * 4 : 0;
}

// one image at a time through the tape: `z` is the tape's input (decoder: latents [S][S][latent_channels]; encoder: the image
// [8S][8S][out_channels]) and `image` its output (decoder: the image; encoder: the moments [S][S][2 * latent_channels]), f32 NHWC
static int vae_run(dh_vae_decoder* v, const float* z, int batch, float* image, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int dt = v->dtype;
  const dh_vae_config& c = v->cfg;
  const int S = c.latent_size, R = 8 * S;
  const size_t in_elems = v->encoder ? (size_t)R * R * c.out_channels : (size_t)S * S * c.latent_channels;
  const size_t out_elems = v->encoder ? (size_t)S * S * 2 * c.latent_channels : (size_t)R * R * c.out_channels;
  auto aptr = [&](int t) { return v->act + v->tens[t].off; };
  for (int b = 0; b < batch; ++b) {
    const float* zb = z + (size_t)b * in_elems;
    float* ib = image + (size_t)b * out_elems;
    for (const VOpRec& o : v->ops) {
      switch (o.type) {
        case V_CONV_IN: {
          const VWt& w = v->wts[o.wt];
          launch_conv_small_fwd(dt, zb, 1, v->pf + w.off, v->pf + o.bias, aptr(o.out), 0, 1, o.Hin, o.Hin, o.Cin, w.N, st);
          break;
        }
        case V_CONV_OUT: {
          const VWt& w = v->wts[o.wt];
          launch_conv_small_fwd(dt, aptr(o.in0), 0, v->pf + w.off, v->pf + o.bias, ib, 1, 1, o.Hin, o.Hin, o.Cin, w.N, st);
          break;
        }
        case V_CONV: case V_LINEAR: {
          const VWt& w = v->wts[o.wt];
          const VTen &ti = v->tens[o.in0], &to = v->tens[o.out];
          GemmArgs g;
          g.A = aptr(o.in0); g.lda = ti.C; g.W = v->w16 + w.off;
          g.M = to.H * to.H; g.N = w.N; g.K = w.K;
          if (o.type == V_CONV) {
            g.mode = A_CONV3; g.Hin = o.Hin; g.Win = o.Hin; g.Cin = o.Cin; g.Hout = o.Hout; g.Wout = o.Hout; g.stride = o.stride; g.up = o.up;
            g.pad = o.pad;
          }
          g.bias = v->pf + o.bias;
          if (o.res >= 0) { g.R = aptr(o.res); g.ldr = v->tens[o.res].C; }
          g.C = aptr(o.out); g.ldc = to.C;
          g.partial = v->partial; g.partial_elems = v->partial_elems;
          launch_gemm(dt, g, st);
          break;
        }
        case V_GN: {
          const VTen& t = v->tens[o.in0];
          launch_groupnorm_fwd(dt, aptr(o.in0), v->pf + o.gamma, v->pf + o.beta, aptr(o.out), v->f32a + o.stats, v->small, 1, t.H * t.H,
                               t.C, c.norm_groups, o.eps, o.silu, st, 0);
          break;
        }
        case V_ATTN: {
          const VTen& t = v->tens[o.in0];
          const int N = t.H * t.H, C = t.C / 3;
          const unsigned short* qkv = aptr(o.in0);
          const unsigned nb = (unsigned)(((size_t)N * C + 255) / 256);
          // S = Q K^T (the scale sits in the q projection): K rows tiled as the GEMM's weight operand
          hipLaunchKernelGGL((k_tile_rows<unsigned short>), dim3(nb), dim3(256), 0, st, qkv, (long)t.C, C, v->tiled, N, C, 0);
          GemmArgs g;
          g.A = qkv; g.lda = t.C; g.W = v->tiled; g.M = N; g.N = N; g.K = C; g.C = v->scores; g.ldc = N;
          launch_gemm(dt, g, st);
          if (dt == DH_DTYPE_F16) hipLaunchKernelGGL((k_softmax_rows<f16>), dim3(cdiv(N, 4)), dim3(256), 0, st, (f16*)v->scores, N, N);
          else hipLaunchKernelGGL((k_softmax_rows<bf16>), dim3(cdiv(N, 4)), dim3(256), 0, st, (bf16*)v->scores, N, N);
          // O = P V: V^T tiled as the weight operand ([C][N])
          hipLaunchKernelGGL((k_tile_rows<unsigned short>), dim3(nb), dim3(256), 0, st, qkv, (long)t.C, 2 * C, v->tiled, C, N, 1);
          GemmArgs h;
          h.A = v->scores; h.lda = N; h.W = v->tiled; h.M = N; h.N = C; h.K = N; h.C = aptr(o.out); h.ldc = C;
          h.partial = v->partial; h.partial_elems = v->partial_elems;
          launch_gemm(dt, h, st);
          break;
        }
      }
    }
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_vae_decoder_decode(dh_vae_decoder* v, const float* z, int batch, float* image, void* stream) {
  DH_REQUIRE(v && z && image && batch >= 1, "bad arguments");
  DH_REQUIRE(!v->encoder, "this handle is an encoder (dh_vae_encoder_create)");
  return vae_run(v, z, batch, image, stream);
}

extern "C" int dh_vae_encoder_encode(dh_vae_decoder* v, const float* image, int batch, float* moments, void* stream) {
  DH_REQUIRE(v && image && moments && batch >= 1, "bad arguments");
  DH_REQUIRE(v->encoder, "this handle is a decoder (dh_vae_decoder_create)");
  return vae_run(v, image, batch, moments, stream);
}
