// Depth re-projection kernels for gfx950: unproject -> SE(3) -> project -> two-pass
// 64-bit atomic z-buffer -> mask morphology -> ordered compaction of correspondences ->
// harmonic in-fill (single-workgroup f64 CG) -> normalised disparity.
//
// Arithmetic contract (bit-exact integer outputs vs oracle/depth_ref.py, which is pinned
// to the reference depth_transform.py:198-747): float32 where NumPy/torch compute in
// float32, float64 where they compute in float64, no FMA contraction (this TU is built
// with -ffp-contract=off), divisions done in f64 and rounded once (innocuous double
// rounding: 53 >= 2*24+2).
#include "common.h"
#include "compact.h"

namespace dh {

// ---------------------------------------------------------------------- point geometry
__device__ __forceinline__ void unproject_px(const float* depth, int p, int res, const float* gx, const float* gy,
                                             float ifx, float ify, float& X, float& Y, float& Z) {
  int row = p / res, col = p - row * res;
  float d = depth[p];
  float ax = d * ifx;
  float ay = d * ify;
  X = -(ax * gx[col]);
  Y = -(ay * gy[row]);
  Z = d;
}

__global__ void k_unproject(const float* depth, int res, const float* gx, const float* gy, float ifx, float ify,
                            float* pts) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= res * res) return;
  float X, Y, Z;
  unproject_px(depth, p, res, gx, gy, ifx, ify, X, Y, Z);
  pts[3 * p + 0] = X;
  pts[3 * p + 1] = Y;
  pts[3 * p + 2] = Z;
}

// NumPy's mean over an [N,3] float32 array along axis 0: sequential float32 accumulation in
// row order, then / float32(N).  The additions are inherently serial (bit-exact order); the unprojection is
// not: the whole workgroup stages the next 1024 points into LDS while lanes 0..2 (one coordinate each)
// fold the previous 1024 in order.
constexpr int CEN_CHUNK = 1024;
__global__ void __launch_bounds__(CEN_CHUNK) k_centroid(const float* depth, const int* fg_pix, int n, int res, const float* gx,
                                                         const float* gy, float ifx, float ify, float* cen) {
  __shared__ float sv[2][3][CEN_CHUNK];
  const int t = threadIdx.x;
  float acc = 0.f;
  const int nchunks = (n + CEN_CHUNK - 1) / CEN_CHUNK;
  for (int c = 0; c <= nchunks; ++c) {
    if (c < nchunks) {
      const int i = c * CEN_CHUNK + t;
      if (i < n) {
        float X, Y, Z;
        unproject_px(depth, fg_pix[i], res, gx, gy, ifx, ify, X, Y, Z);
        sv[c & 1][0][t] = X; sv[c & 1][1][t] = Y; sv[c & 1][2][t] = Z;
      }
    }
    if (c > 0 && t < 3) {
      const int base = (c - 1) * CEN_CHUNK;
      const int m = n - base < CEN_CHUNK ? n - base : CEN_CHUNK;
      const float* v = sv[(c - 1) & 1][t];
      // the adds are one dependent chain (the order IS the result); what can overlap them is the LDS read of the NEXT
      // batch: two register batches of 16, the loads of one issued before the adds of the other
      int k = 0;
      float u[16], w[16];
      if (m >= 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) u[j] = v[j];
      }
      for (; k + 32 <= m; k += 32) {
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = v[k + 16 + j];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = acc + u[j];
        if (k + 48 <= m) {
#pragma unroll
          for (int j = 0; j < 16; ++j) u[j] = v[k + 32 + j];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = acc + w[j];
      }
      if (k + 16 <= m) {          // (u holds v[k .. k+16) here: loaded by the prologue or by the last loop pass)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = acc + u[j];
        k += 16;
      }
      for (; k < m; ++k) acc = acc + v[k];
    }
    __syncthreads();
  }
  if (t < 3) cen[t] = (float)((double)acc / (double)(float)n);
}

__device__ __forceinline__ unsigned long long sortable_f64(double z) {
  unsigned long long b = (unsigned long long)__double_as_longlong(z);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double unsort_f64(unsigned long long k) {
  unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

struct Xf {
  double ax, ay, az, c, s, tx, ty, tz;
};

// one thread per (edit, point): points [0,R2) are background pixels, [R2,R2+n_fg) foreground.
__global__ void k_points(const float* depth, const float* bg_depth, const int* fg_pix, int n_fg, int res,
                         const float* gx, const float* gy, float ifx, float ify, double fx, double fy,
                         const Xf* xf, const float* cen, unsigned long long* zbuf, int* pix_out,
                         unsigned long long* key_out) {
  const int R2 = res * res, P = R2 + n_fg;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int e = blockIdx.y;
  if (i >= P) return;
  double X, Y, Z;
  if (i < R2) {
    float x, y, z;
    unproject_px(bg_depth, i, res, gx, gy, ifx, ify, x, y, z);
    X = x; Y = y; Z = z;
  } else {
    float x, y, z;
    unproject_px(depth, fg_pix[i - R2], res, gx, gy, ifx, ify, x, y, z);
    const Xf t = xf[e];
    const float c0 = cen[0], c1 = cen[1], c2 = cen[2];
    const float q0 = x - c0, q1 = y - c1, q2 = z - c2;
    const float a0 = (float)t.ax, a1 = (float)t.ay, a2 = (float)t.az;
    const float cr0 = a1 * q2 - a2 * q1;
    const float cr1 = a2 * q0 - a0 * q2;
    const float cr2 = a0 * q1 - a1 * q0;
    const float dt = (q0 * a0 + q1 * a1) + q2 * a2;
    const double omc = 1.0 - t.c;
    const double r0 = ((double)q0 * t.c + (double)cr0 * t.s) + (double)(a0 * dt) * omc;
    const double r1 = ((double)q1 * t.c + (double)cr1 * t.s) + (double)(a1 * dt) * omc;
    const double r2 = ((double)q2 * t.c + (double)cr2 * t.s) + (double)(a2 * dt) * omc;
    X = (r0 + (double)c0) + t.tx;
    Y = (r1 + (double)c1) + t.ty;
    Z = (r2 + (double)c2) + t.tz;
  }
  const double m = (double)(res - 1);
  double u = (fx * (-X)) / Z;
  double v = (fy * (-Y)) / Z;
  u = (u * 0.5 + 0.5) * m;
  v = (v * 0.5 + 0.5) * m;
  u = fmin(fmax(u, 0.0), m);
  v = fmin(fmax(v, 0.0), m);
  const int ui = (int)rint(u), vi = (int)rint(v);
  const int pix = vi * res + ui;
  const unsigned long long key = sortable_f64(Z);
  pix_out[(size_t)e * P + i] = pix;
  key_out[(size_t)e * P + i] = key;
  atomicMin(&zbuf[(size_t)e * R2 + pix], key);
}

__global__ void k_resolve(int P, int R2, const unsigned long long* zbuf, const int* pix_arr,
                          const unsigned long long* key_arr, int* owner) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int e = blockIdx.y;
  if (i >= P) return;
  int pix = pix_arr[(size_t)e * P + i];
  if (key_arr[(size_t)e * P + i] == zbuf[(size_t)e * R2 + pix]) atomicMin(&owner[(size_t)e * R2 + pix], i);
}

__device__ __forceinline__ unsigned int sortable_f32(float x) {
  unsigned int b = __float_as_uint(x);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unsort_f32(unsigned int k) {
  unsigned int b = (k >> 31) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

// per pixel: depth map, raw fg mask, raw disparity 1/z and its min/max keys
constexpr int PIX_PER_THREAD = 16;
__global__ void k_pixels(int R2, const unsigned long long* zbuf, const int* owner, float* zmap, uint8_t* raw_mask,
                         float* disp, unsigned int* minmax) {
  __shared__ unsigned int smin[4], smax[4];
  int e = blockIdx.y;
  unsigned int kmin = 0xffffffffu, kmax = 0u;
  // PIX_PER_THREAD pixels per thread: 16x fewer workgroups contend on the two min / max words of the edit
#pragma unroll 4
  for (int it = 0; it < PIX_PER_THREAD; ++it) {
    int p = (blockIdx.x * PIX_PER_THREAD + it) * blockDim.x + threadIdx.x;
    if (p >= R2) break;
    size_t o = (size_t)e * R2 + p;
    unsigned long long k = zbuf[o];
    float z = (k == ~0ull) ? __uint_as_float(0x7f800000u) : (float)unsort_f64(k);
    zmap[o] = z;
    int w = owner[o];
    raw_mask[o] = (k != ~0ull && w >= R2) ? 1 : 0;
    float d = (float)(1.0 / (double)z);
    disp[o] = d;
    const unsigned int kd = sortable_f32(d);
    kmin = kd < kmin ? kd : kmin;
    kmax = kd > kmax ? kd : kmax;
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    unsigned int a = __shfl_xor(kmin, s, 64), b = __shfl_xor(kmax, s, 64);
    kmin = a < kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
  }
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { smin[wv] = kmin; smax[wv] = kmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
      kmin = smin[i] < kmin ? smin[i] : kmin;
      kmax = smax[i] > kmax ? smax[i] : kmax;
    }
    atomicMin(&minmax[2 * e], kmin);
    atomicMax(&minmax[2 * e + 1], kmax);
  }
}

__global__ void k_fg_vis(int n_fg, int R2, int P, int res, const int* pix_arr, const int* owner, uint8_t* vis,
                         int* target_xy) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int e = blockIdx.y;
  if (j >= n_fg) return;
  int i = R2 + j;
  int pix = pix_arr[(size_t)e * P + i];
  vis[(size_t)e * n_fg + j] = owner[(size_t)e * R2 + pix] == i ? 1 : 0;
  target_xy[((size_t)e * n_fg + j) * 2 + 0] = pix % res;
  target_xy[((size_t)e * n_fg + j) * 2 + 1] = pix / res;
}

// binary morphology with an explicit offset list; pixels outside the image are ignored.
__global__ void k_morph(const uint8_t* src, uint8_t* dst, int res, const int2* offs, int n_off, int dilate) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int e = blockIdx.y;
  if (p >= res * res) return;
  const uint8_t* s = src + (size_t)e * res * res;
  int y = p / res, x = p - y * res;
  int acc = dilate ? 0 : 1;
  for (int k = 0; k < n_off; ++k) {
    int yy = y + offs[k].y, xx = x + offs[k].x;
    if (yy < 0 || yy >= res || xx < 0 || xx >= res) continue;
    int v = s[yy * res + xx] ? 1 : 0;
    acc = dilate ? (acc | v) : (acc & v);
    if (acc == dilate) break;          // decided: a set pixel found (dilate) / a clear one (erode); most waves lie in uniform regions
  }
  dst[(size_t)e * res * res + p] = (uint8_t)acc;
}

// The same morphology on BIT rows (res % 32 == 0): a thread owns one 32-pixel word of the output; a tap (dx, dy) is a funnel
// shift of three source words of row y + dy, so the 76 taps of the 10 x 10 closing ellipse cost 76 shifts per 32 pixels instead
// of 76 byte loads per pixel (57 us per pass at K = 8, 512 x 512: 75 GB/s of useful traffic).  Erosion = NOT dilate(NOT x) with
// the same tap list: taps outside the image are ignored either way (clear rows / words of the complemented image).
//   out = inv_out ^ dilate(inv_in ^ src)
__global__ void k_pack_bits(const uint8_t* src, unsigned* dst, int n_words) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const uint4 a = *reinterpret_cast<const uint4*>(src + (size_t)w * 32), b = *reinterpret_cast<const uint4*>(src + (size_t)w * 32 + 16);
  const unsigned v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned bits = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) bits |= ((v[i] >> (8 * j)) & 0xffu ? 1u : 0u) << (4 * i + j);
  dst[w] = bits;
}
__global__ void k_unpack_bits(const unsigned* src, uint8_t* dst, int n_words) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const unsigned bits = src[w];
  unsigned v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i] |= ((bits >> (4 * i + j)) & 1u) << (8 * j);
  }
  *reinterpret_cast<uint4*>(dst + (size_t)w * 32) = make_uint4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<uint4*>(dst + (size_t)w * 32 + 16) = make_uint4(v[4], v[5], v[6], v[7]);
}
__global__ void k_morph_bits(const unsigned* src, unsigned* dst, int res, const int2* offs, int n_off, int inv_in, int inv_out) {
  const int wpr = res >> 5;                                  // words per row
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // word of this edit's image
  if (idx >= wpr * res) return;
  const unsigned* s = src + (size_t)blockIdx.y * wpr * res;
  const int y = idx / wpr, wx = idx - y * wpr;
  const unsigned flip = inv_in ? 0xffffffffu : 0u;
  unsigned acc = 0;
  for (int k = 0; k < n_off; ++k) {
    const int yy = y + offs[k].y, dx = offs[k].x;            // out(x) |= in(x + dx)
    if (yy < 0 || yy >= res) continue;
    const unsigned* row = s + (size_t)yy * wpr;
    const unsigned c = row[wx] ^ flip;
    const unsigned l = wx > 0 ? row[wx - 1] ^ flip : 0u;     // pixels left of this word (outside the image: clear)
    const unsigned r = wx + 1 < wpr ? row[wx + 1] ^ flip : 0u;
    unsigned v;
    if (dx == 0) v = c;
    else if (dx > 0) v = dx >= 32 ? r : (c >> dx) | (r << (32 - dx));
    else v = dx <= -32 ? l : (c << -dx) | (l >> (32 + dx));
    acc |= v;
  }
  dst[(size_t)blockIdx.y * wpr * res + idx] = inv_out ? ~acc : acc;
}

__global__ void k_keep_flags(int n_fg, int R2, int P, const int* pix_arr, const uint8_t* vis, const uint8_t* clean,
                             uint8_t* keep) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int e = blockIdx.y;
  if (j >= n_fg) return;
  int pix = pix_arr[(size_t)e * P + R2 + j];
  keep[(size_t)e * n_fg + j] = (vis[(size_t)e * n_fg + j] && clean[(size_t)e * R2 + pix]) ? 1 : 0;
}

__global__ void k_write_corr(int n_fg, int res, const int* fg_pix, const int* target_xy, const int* keep_idx,
                             const int* counts, int count_stride, long long* corr) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  int e = blockIdx.y;
  if (r >= counts[e * count_stride]) return;
  int j = keep_idx[(size_t)e * n_fg + r];
  int p = fg_pix[j];
  long long* c = corr + ((size_t)e * n_fg + r) * 4;
  c[0] = p % res;
  c[1] = p / res;
  c[2] = target_xy[((size_t)e * n_fg + j) * 2 + 0];
  c[3] = target_xy[((size_t)e * n_fg + j) * 2 + 1];
}

__global__ void k_count_flags(const uint8_t* f, int n, int* out, int out_stride, int slot) {
  __shared__ int sm[4];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int e = blockIdx.y;
  int c = (i < n && f[(size_t)e * n + i]) ? 1 : 0;
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sm[k];
    if (t) atomicAdd(&out[e * out_stride + slot], t);
  }
}

// normalise disparity in place and flag the pixels to in-fill (cleaned xor raw)
__global__ void k_normalize(int R2, float* disp, const unsigned int* minmax, const float* bounds,
                            const uint8_t* raw, const uint8_t* clean, uint8_t* inpaint) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int e = blockIdx.y;
  if (p >= R2) return;
  size_t o = (size_t)e * R2 + p;
  float lo = bounds ? bounds[0] : unsort_f32(minmax[2 * e]);
  float hi = bounds ? bounds[1] : unsort_f32(minmax[2 * e + 1]);
  float t = disp[o] - lo;
  t = 255.0f * t;
  float den = hi - lo;
  disp[o] = (float)((double)t / (double)den);
  inpaint[o] = (clean[o] != 0) != (raw[o] != 0) ? 1 : 0;
}

// Harmonic in-fill: A x = b with A = 4 I - adjacency(masked), solved by CG in float64 by ONE
// workgroup per edit (deterministic fixed-tree reductions).  The vectors are indexed by UNKNOWN (compact, coalesced);
// the four neighbour unknowns of every unknown are resolved once through a pixel -> unknown map, so an iteration is
// three streaming passes plus four gathers per unknown.  (Used for holes of more than CG_SLOTS * 1024 pixels; smaller
// ones stay on chip in cg_fill_lds, which visits the unknowns in the same order: identical iterates.)
__global__ void __launch_bounds__(1024) k_cg_fill(int res, float* disp, const uint8_t* inpaint, const int* unk,
                                                  const int* counts, int count_stride, int slot_n, int slot_it,
                                                  double* __restrict__ vx, double* __restrict__ vr, double* __restrict__ vp,
                                                  double* __restrict__ vq, int max_iter,
                                                  double tol2, int* counts_out, const float* rhs_extra, int min_n,
                                                  int* pixmap, int2* nb_ud, size_t nb_ud_stride, int2* nb_lr,
                                                  size_t nb_lr_stride) {
  __shared__ double sm[16];
  const int e = blockIdx.x, R2 = res * res;
  const int n = counts[e * count_stride + slot_n];
  if (n <= min_n) return;                                // solved on chip by cg_fill_lds (inside k_cg_fill_multi)
  float* d = disp + (size_t)e * R2;
  const uint8_t* mk = inpaint + (size_t)e * R2;
  const int* U = unk + (size_t)e * R2;
  // (__restrict__: without it every store to q / x / r / p orders the loads of the next element behind it and an iteration
  // is ~20 dependent L2 round trips per pass; with it the loads of an unrolled batch are issued together)
  double* __restrict__ x = vx + (size_t)e * R2;
  double* __restrict__ r = vr + (size_t)e * R2;
  double* __restrict__ p = vp + (size_t)e * R2;
  double* __restrict__ q = vq + (size_t)e * R2;
  int* map = pixmap + (size_t)e * R2;
  int2* __restrict__ nud = nb_ud + (size_t)e * nb_ud_stride;
  int2* __restrict__ nlr = nb_lr + (size_t)e * nb_lr_stride;
  for (int i = threadIdx.x; i < n; i += blockDim.x) map[U[i]] = i;
  __syncthreads();
  double part = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int pix = U[i], y = pix / res, xx = pix - y * res;
    double b = 0.0;
    int2 ud = make_int2(-1, -1), lr = make_int2(-1, -1);
    if (y > 0) { if (!mk[pix - res]) b += (double)d[pix - res]; else ud.x = map[pix - res]; }
    if (y < res - 1) { if (!mk[pix + res]) b += (double)d[pix + res]; else ud.y = map[pix + res]; }
    if (xx > 0) { if (!mk[pix - 1]) b += (double)d[pix - 1]; else lr.x = map[pix - 1]; }
    if (xx < res - 1) { if (!mk[pix + 1]) b += (double)d[pix + 1]; else lr.y = map[pix + 1]; }
    if (rhs_extra) b -= (double)rhs_extra[(size_t)e * R2 + pix];
    nud[i] = ud; nlr[i] = lr;
    x[i] = 0.0;
    r[i] = b;
    p[i] = b;
    part += b * b;
  }
  double rs = block_sum(part, sm);
  const double bnorm = rs;
  int it = 0;
  for (; it < max_iter; ++it) {
    if (!(rs > tol2 * bnorm)) break;
    __syncthreads();
    part = 0.0;
#pragma unroll 8
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int2 ud = nud[i], lr = nlr[i];
      const double pi = p[i];
      double a = 4.0 * pi;
      if (ud.x >= 0) a -= p[ud.x];
      if (ud.y >= 0) a -= p[ud.y];
      if (lr.x >= 0) a -= p[lr.x];
      if (lr.y >= 0) a -= p[lr.y];
      q[i] = a;
      part += pi * a;
    }
    const double pq = block_sum(part, sm);
    const double alpha = rs / pq;
    part = 0.0;
#pragma unroll 8
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      x[i] += alpha * p[i];
      double rr = r[i] - alpha * q[i];
      r[i] = rr;
      part += rr * rr;
    }
    const double rsn = block_sum(part, sm);
    const double beta = rsn / rs;
#pragma unroll 8
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = r[i] + beta * p[i];
    rs = rsn;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) d[U[i]] = (float)x[i];
  if (threadIdx.x == 0) counts_out[e * count_stride + slot_it] = it;
}


// The iteration of k_cg_fill with the vectors on chip: thread t owns unknowns t, t + 1024, ... (the order that kernel visits
// them in, so every partial sum and therefore every iterate is bit-identical to it), x / r / q live in registers, p in LDS where
// the four neighbours are read through a pixel -> unknown index map.  One iteration is three workgroup barriers and two LDS
// reductions instead of five passes over global memory.  Run by workgroup 0 of a system inside k_cg_fill_multi (round 6: the
// small systems of a batch are solved WHILE the sixteen-workgroup ones run, not in a launch of their own in front of them).
constexpr int CG_SLOTS = 8;      // unknowns per thread: n <= 8192 takes this path
__device__ __forceinline__ void cg_fill_lds(int e, int res, float* disp, const uint8_t* inpaint, const int* unk, const int* counts,
                                            int count_stride, int slot_n, int slot_it, int* pixmap, int max_iter, double tol2,
                                            int* counts_out, const float* rhs_extra) {
  __shared__ double sm[16];
  __shared__ double sp[CG_SLOTS * 1024];
  const int R2 = res * res;
  const int n = counts[e * count_stride + slot_n];
  if (n > CG_SLOTS * 1024) return;
  float* d = disp + (size_t)e * R2;
  const uint8_t* mk = inpaint + (size_t)e * R2;
  const int* U = unk + (size_t)e * R2;
  int* map = pixmap + (size_t)e * R2;
  if (n == 0) {
    if (threadIdx.x == 0) counts_out[e * count_stride + slot_it] = 0;
    return;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) map[U[i]] = i;
  __syncthreads();
  int nb[CG_SLOTS][4];
  double x[CG_SLOTS], r[CG_SLOTS], q[CG_SLOTS];
  double part = 0.0;
#pragma unroll
  for (int sl = 0; sl < CG_SLOTS; ++sl) {
    const int i = threadIdx.x + sl * 1024;
    x[sl] = 0.0; r[sl] = 0.0; q[sl] = 0.0;
    nb[sl][0] = nb[sl][1] = nb[sl][2] = nb[sl][3] = -1;
    if (i < n) {
      const int pix = U[i], y = pix / res, xx = pix - y * res;
      double b = 0.0;
      if (y > 0) { if (!mk[pix - res]) b += (double)d[pix - res]; else nb[sl][0] = map[pix - res]; }
      if (y < res - 1) { if (!mk[pix + res]) b += (double)d[pix + res]; else nb[sl][1] = map[pix + res]; }
      if (xx > 0) { if (!mk[pix - 1]) b += (double)d[pix - 1]; else nb[sl][2] = map[pix - 1]; }
      if (xx < res - 1) { if (!mk[pix + 1]) b += (double)d[pix + 1]; else nb[sl][3] = map[pix + 1]; }
      if (rhs_extra) b -= (double)rhs_extra[(size_t)e * R2 + pix];
      r[sl] = b;
      sp[i] = b;
      part += b * b;
    }
  }
  double rs = block_sum(part, sm);
  const double bnorm = rs;
  int it = 0;
  for (; it < max_iter; ++it) {
    if (!(rs > tol2 * bnorm)) break;
    __syncthreads();
    part = 0.0;
#pragma unroll
    for (int sl = 0; sl < CG_SLOTS; ++sl) {
      const int i = threadIdx.x + sl * 1024;
      if (i < n) {
        const double pi = sp[i];
        double a = 4.0 * pi;
        if (nb[sl][0] >= 0) a -= sp[nb[sl][0]];
        if (nb[sl][1] >= 0) a -= sp[nb[sl][1]];
        if (nb[sl][2] >= 0) a -= sp[nb[sl][2]];
        if (nb[sl][3] >= 0) a -= sp[nb[sl][3]];
        q[sl] = a;
        part += pi * a;
      }
    }
    const double pq = block_sum(part, sm);
    const double alpha = rs / pq;
    part = 0.0;
#pragma unroll
    for (int sl = 0; sl < CG_SLOTS; ++sl) {
      const int i = threadIdx.x + sl * 1024;
      if (i < n) {
        x[sl] += alpha * sp[i];
        const double rr = r[sl] - alpha * q[sl];
        r[sl] = rr;
        part += rr * rr;
      }
    }
    const double rsn = block_sum(part, sm);       // its barriers also order the reads of p above before the update below
    const double beta = rsn / rs;
#pragma unroll
    for (int sl = 0; sl < CG_SLOTS; ++sl) {
      const int i = threadIdx.x + sl * 1024;
      if (i < n) sp[i] = r[sl] + beta * sp[i];
    }
    rs = rsn;
  }
#pragma unroll
  for (int sl = 0; sl < CG_SLOTS; ++sl) {
    const int i = threadIdx.x + sl * 1024;
    if (i < n) d[U[i]] = (float)x[sl];
  }
  if (threadIdx.x == 0) counts_out[e * count_stride + slot_it] = it;
}

// Holes too large for one CU's LDS: CGM_WGS workgroups per system.  One workgroup is bound by its CU's memory pipe (15 eight-byte
// accesses per unknown and iteration through one L1: 37 us per iteration at 20 000 unknowns, rocprofv3); here every workgroup owns a
// contiguous 1 / CGM_WGS of the unknowns and the workgroups of a system meet at grid-wide SEAMS.  Rounds 3-5 ran the classic
// recurrence with two seams per iteration (one per dot product); round 6 runs the pipelined form with ONE (see the loop).
// A seam = agent-scope release, one arrival counter per system (monotonic: epoch * CGM_WGS), relaxed polling by one lane, one
// agent-scope acquire, __syncthreads (cdna_hip_programming.md Guideline 16); the dot products are per-workgroup partials written
// before the seam and added by everyone in workgroup order afterwards: identical in every workgroup (the convergence test must
// agree: it decides whether the next seam is entered) and deterministic.
constexpr int CGM_WGS = 16, CGM_EPT = 4;       // unknowns per thread <= 4: n <= 16 * 4 * 1024 (r, w, z, s of an unknown: 8 registers; 8 unknowns spilled)
struct CgSync { unsigned arrive; unsigned fail; unsigned pad[30]; double red[2][CGM_WGS]; double red2[2][CGM_WGS]; };
// Forward progress of the seams needs the CGM_WGS workgroups of a system co-resident.  In-order dispatch gives that on an
// otherwise idle chip (a system's workgroups are consecutive blocks, and the chip holds >= 256 of them), but nothing enforces
// it: CU masks, other streams or processes holding the slots.  So every spin is BOUNDED: after CG_SPIN_CAP polls (seconds; a
// normal seam waits microseconds) the poller raises `fail`, every workgroup of the system leaves at its next seam, and the
// iteration count comes back as -1, which the host turns into an error instead of a hung GPU (advisor, round 3).
constexpr unsigned CG_SPIN_CAP = 1u << 21;
typedef __attribute__((address_space(1))) unsigned cg_gu32;
// FENCED: the payload went through plain stores (release: L2 write-back) and is read with plain loads (acquire: stale lines
// dropped).  Otherwise everything shared was stored write-through and is loaded past the caches (agent-scope relaxed atomics on
// 8-byte words, the guide's R1 form): every storing wave drains its stores, one lane counts the arrival -- no fences.
typedef __attribute__((address_space(1))) unsigned long long cg_gu64;
__device__ __forceinline__ void cg_put(double* p, double v) {
  __hip_atomic_store((cg_gu64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double cg_get(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((cg_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// returns false when the system has failed (a bounded spin ran out here or in another workgroup): the caller leaves the kernel
template <bool FENCED>
__device__ __forceinline__ bool cg_seam(CgSync* s, unsigned epoch) {
  __shared__ unsigned seam_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (FENCED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cg_gu32* ctr = (cg_gu32*)&s->arrive;
    cg_gu32* fail = (cg_gu32*)&s->fail;
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0, ok = 1;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * CGM_WGS) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
      if (spins > CG_SPIN_CAP) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
    }
    if (ok && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0;
    if (FENCED) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    seam_ok = ok;
  }
  __syncthreads();
  return seam_ok != 0;
}
__global__ void __launch_bounds__(1024) k_cg_fill_multi(int res, float* disp, const uint8_t* inpaint, const int* unk,
                                                        const int* counts, int count_stride, int slot_n, int slot_it, double* vx,
                                                        double* vr, double* vp0, double* vp1, int max_iter, double tol2,
                                                        int* counts_out, const float* rhs_extra, int min_n, int* pixmap,
                                                        int2* nb_ud, size_t nb_ud_stride, int2* nb_lr, size_t nb_lr_stride,
                                                        CgSync* sync) {
  __shared__ double sm[16];
  const int e = blockIdx.y, wg = blockIdx.x, R2 = res * res;
  const int n = counts[e * count_stride + slot_n];
  if (n <= min_n) {                                                 // small enough for one CU's LDS: workgroup 0 solves it on chip
    if (wg == 0 && min_n > 0)
      cg_fill_lds(e, res, disp, inpaint, unk, counts, count_stride, slot_n, slot_it, pixmap, max_iter, tol2, counts_out, rhs_extra);
    return;
  }
  if (n > CGM_WGS * CGM_EPT * 1024) return;                        // (uniform over the system's workgroups)
  CgSync* sy = sync + e;
  float* d = disp + (size_t)e * R2;
  const uint8_t* mk = inpaint + (size_t)e * R2;
  const int* U = unk + (size_t)e * R2;
  double* x = vx + (size_t)e * R2;
  double* r = vr + (size_t)e * R2;
  double* P[2] = {vp0 + (size_t)e * R2, vp1 + (size_t)e * R2};
  int* map = pixmap + (size_t)e * R2;
  int2* nud = nb_ud + (size_t)e * nb_ud_stride;
  int2* nlr = nb_lr + (size_t)e * nb_lr_stride;
  const int per = (n + CGM_WGS - 1) / CGM_WGS, i0 = wg * per, i1 = i0 + per < n ? i0 + per : n;
  unsigned epoch = 0;
  // The iteration count is published with atomicMin over a slot that workgroup 0 arms here, BEFORE its first seam: a workgroup
  // whose bounded spin runs out on the LAST seam (the others have read fail == 0, pass it and write their slices) still leaves
  // -1 in the slot whatever the order of the two updates, so the host never sees a count >= 0 beside a partly written
  // disparity.  (A failure before workgroup 0 even started is seen by workgroup 0 at its first seam: the flag is sticky.)
  if (wg == 0 && threadIdx.x == 0) atomicExch(&counts_out[e * count_stride + slot_it], 0x7fffffff);
  for (int i = i0 + (int)threadIdx.x; i < i1; i += 1024) map[U[i]] = i;
  bool alive = cg_seam<true>(sy, ++epoch);                           // every workgroup's part of the pixel -> unknown map (plain stores / loads)
  if (!alive) { if (threadIdx.x == 0) atomicMin(&counts_out[e * count_stride + slot_it], -1); return; }
  // PIPELINED conjugate gradients (Ghysels & Vanroose; round 6): with w = A r carried by recurrence, the two dot products of an
  // iteration -- gamma = (r, r), delta = (w, r) -- are taken from the SAME state and the matrix product of the iteration, q = A w,
  // needs nothing from them, so an iteration has ONE grid-wide seam (the partial sums and the new w are published before it)
  // instead of two; the classic recurrence above pays a seam per dot product, 92 x 2 x 6.5 us on the largest hole of the K = 8
  // benchmark call.  Per iteration:   beta = gamma / gamma', alpha = gamma / (delta - beta gamma / alpha')   (first: 0, gamma / delta)
  //     z = q + beta z,  s = w + beta s,  p = r + beta p,   x += alpha p,  r -= alpha s,  w -= alpha z.
  // r, w, z, s of a thread's unknowns live in registers; x and p are private global arrays (p takes over the array the
  // right-hand side was exchanged through); only w is shared, double-buffered (neighbours read the old one while the new one is
  // written).  Same fixed point as cg_fill_lds / k_cg_fill (the iterates differ in rounding: f64, relative residual 1e-12).
  double* pv = r;                                // b is exchanged through this array during the set-up; afterwards it holds p
  double rl[CGM_EPT], wl[CGM_EPT], zl[CGM_EPT], sl[CGM_EPT];
  double part = 0.0;
#pragma unroll
  for (int k = 0; k < CGM_EPT; ++k) {
    const int i = i0 + (int)threadIdx.x + k * 1024;
    rl[k] = 0.0; wl[k] = 0.0; zl[k] = 0.0; sl[k] = 0.0;
    if (i < i1) {
      const int pix = U[i], y = pix / res, xx = pix - y * res;
      double b = 0.0;
      int2 ud = make_int2(-1, -1), lr = make_int2(-1, -1);
      if (y > 0) { if (!mk[pix - res]) b += (double)d[pix - res]; else ud.x = map[pix - res]; }
      if (y < res - 1) { if (!mk[pix + res]) b += (double)d[pix + res]; else ud.y = map[pix + res]; }
      if (xx > 0) { if (!mk[pix - 1]) b += (double)d[pix - 1]; else lr.x = map[pix - 1]; }
      if (xx < res - 1) { if (!mk[pix + 1]) b += (double)d[pix + 1]; else lr.y = map[pix + 1]; }
      if (rhs_extra) b -= (double)rhs_extra[(size_t)e * R2 + pix];
      nud[i] = ud; nlr[i] = lr;                  // (read back by this thread only)
      x[i] = 0.0;
      cg_put(r + i, b);
      rl[k] = b;
      part += b * b;
    }
  }
  // partials of this workgroup -> seam -> sums in workgroup order (identical in every workgroup: the convergence test must agree)
  auto all_sum2 = [&](double v0, double v1, int slot, double& t0, double& t1) {
    v0 = block_sum(v0, sm);
    v1 = block_sum(v1, sm);
    if (threadIdx.x == 0) { cg_put(&sy->red[slot][wg], v0); cg_put(&sy->red2[slot][wg], v1); }
    if (!cg_seam<false>(sy, ++epoch)) alive = false;
    t0 = 0.0; t1 = 0.0;
    for (int k = 0; k < CGM_WGS; ++k) { t0 += cg_get(&sy->red[slot][k]); t1 += cg_get(&sy->red2[slot][k]); }
  };
  double bnorm, unused;
  all_sum2(part, 0.0, 0, bnorm, unused);        // (the seam also makes every workgroup's b visible)
  if (!alive) { if (threadIdx.x == 0) atomicMin(&counts_out[e * count_stride + slot_it], -1); return; }
#pragma unroll
  for (int k = 0; k < CGM_EPT; ++k) {           // w = A r (r = b: x starts at zero)
    const int i = i0 + (int)threadIdx.x + k * 1024;
    if (i < i1) {
      const int2 ud = nud[i], lr = nlr[i];
      double a = 4.0 * rl[k];
      if (ud.x >= 0) a -= cg_get(r + ud.x);
      if (ud.y >= 0) a -= cg_get(r + ud.y);
      if (lr.x >= 0) a -= cg_get(r + lr.x);
      if (lr.y >= 0) a -= cg_get(r + lr.y);
      wl[k] = a;
      cg_put(P[0] + i, a);
    }
  }
  double gamma_old = 1.0, alpha_old = 1.0;
  int cur = 0, it = 0;
  for (; it < max_iter; ++it) {
    double pg = 0.0, pd = 0.0;
#pragma unroll
    for (int k = 0; k < CGM_EPT; ++k) { pg += rl[k] * rl[k]; pd += wl[k] * rl[k]; }
    double gamma, delta;
    all_sum2(pg, pd, (it + 1) & 1, gamma, delta);     // (slots alternate: a workgroup still adding one cannot be overtaken by the next write to it)
    if (!alive) break;
    if (!(gamma > tol2 * bnorm)) break;
    const double beta = it ? gamma / gamma_old : 0.0;
    const double alpha = it ? gamma / (delta - beta * gamma / alpha_old) : gamma / delta;
    const double* wo = P[cur];
    double* wn = P[cur ^ 1];
#pragma unroll
    for (int k = 0; k < CGM_EPT; ++k) {
      const int i = i0 + (int)threadIdx.x + k * 1024;
      if (i < i1) {
        const int2 ud = nud[i], lr = nlr[i];
        double q = 4.0 * wl[k];
        if (ud.x >= 0) q -= cg_get(wo + ud.x);
        if (ud.y >= 0) q -= cg_get(wo + ud.y);
        if (lr.x >= 0) q -= cg_get(wo + lr.x);
        if (lr.y >= 0) q -= cg_get(wo + lr.y);
        zl[k] = q + beta * zl[k];
        sl[k] = wl[k] + beta * sl[k];
        const double pk = it ? rl[k] + beta * pv[i] : rl[k];
        pv[i] = pk;
        x[i] += alpha * pk;
        rl[k] -= alpha * sl[k];
        wl[k] -= alpha * zl[k];
        cg_put(wn + i, wl[k]);
      }
    }
    gamma_old = gamma; alpha_old = alpha;
    cur ^= 1;
  }
  if (!alive) { if (threadIdx.x == 0) atomicMin(&counts_out[e * count_stride + slot_it], -1); return; }
  for (int i = i0 + (int)threadIdx.x; i < i1; i += 1024) d[U[i]] = (float)x[i];
  if (threadIdx.x == 0 && wg == 0) atomicMin(&counts_out[e * count_stride + slot_it], it);      // (a -1 of another workgroup stays)
}

// cross-element binary dilation (scipy.ndimage.binary_dilation default structure, border 0)
__global__ void k_dilate_cross(const uint8_t* src, uint8_t* dst, int res) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= res * res) return;
  int y = p / res, x = p - y * res;
  int v = src[p];
  if (y > 0) v |= src[p - res];
  if (y < res - 1) v |= src[p + res];
  if (x > 0) v |= src[p - 1];
  if (x < res - 1) v |= src[p + 1];
  dst[p] = v ? 1 : 0;
}
// 5-point Laplacian with zero padding (scipy.ndimage.convolve(mode='constant')), f64 sum rounded to f32
__global__ void k_laplacian(const float* img, float* out, int res) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= res * res) return;
  int y = p / res, x = p - y * res;
  double a = -4.0 * (double)img[p];
  if (y > 0) a += (double)img[p - res];
  if (y < res - 1) a += (double)img[p + res];
  if (x > 0) a += (double)img[p - 1];
  if (x < res - 1) a += (double)img[p + 1];
  out[p] = (float)a;
}

// OpenCV's classic MORPH_ELLIPSE rows, as (dx, dy) offsets from the anchor
static int ellipse_offsets(int k, int2* out) {
  int n = 0;
  if (k <= 1) {
    out[n++] = make_int2(0, 0);
    return n;
  }
  const int r = k / 2, c = k / 2;
  const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
  for (int i = 0; i < k; ++i) {
    int dy = i - r;
    if (abs(dy) > r) continue;
    int dx = (int)nearbyint(c * sqrt((r * r - dy * dy) * inv_r2));
    int j1 = c - dx > 0 ? c - dx : 0, j2 = c + dx + 1 < k ? c + dx + 1 : k;
    for (int j = j1; j < j2; ++j) out[n++] = make_int2(j - c, i - r);
  }
  return n;
}

constexpr int MAX_OFFS = 4096;

}  // namespace dh
// test hook (host only, no device call): the (dx, dy) offsets of the k x k elliptical structuring element the mask
// clean-up uses, so the CPU suite can pin them to OpenCV's published tables
extern "C" int dh_dbg_ellipse_offsets(int k, int32_t* xy, int cap, int* n_out) {
  DH_REQUIRE(k >= 1 && (long)k * k <= dh::MAX_OFFS && xy && n_out && cap >= k * k, "bad arguments");
  static int2 tmp[dh::MAX_OFFS];
  const int n = dh::ellipse_offsets(k, tmp);
  for (int i = 0; i < n; ++i) { xy[2 * i] = tmp[i].x; xy[2 * i + 1] = tmp[i].y; }
  *n_out = n;
  return DH_OK;
}
namespace dh {

struct ReprojectWs {
  Xf* xf;
  float* cen;
  unsigned long long* zbuf;
  int* owner;
  int* pix;
  unsigned long long* key;
  unsigned int* minmax;
  uint8_t* tmp_a;
  uint8_t* tmp_b;
  uint8_t* keep;
  uint8_t* inpaint;
  int* keep_idx;
  int* unk;
  int* block_counts;
  int2* offs_close;
  int2* offs_open;
  double *vx, *vr, *vp, *vq;
  CgSync* cgsync;
};

static bool carve(Arena& a, int res, int n_fg, int K, ReprojectWs& w) {
  const size_t R2 = (size_t)res * res, P = R2 + n_fg;
  w.xf = a.take<Xf>(K);
  w.cen = a.take<float>(4);
  w.zbuf = a.take<unsigned long long>(K * R2);
  w.owner = a.take<int>(K * R2);
  w.pix = a.take<int>(K * P);
  w.key = a.take<unsigned long long>(K * P);
  w.minmax = a.take<unsigned int>(2 * K);
  w.tmp_a = a.take<uint8_t>(K * R2);
  w.tmp_b = a.take<uint8_t>(K * R2);
  w.keep = a.take<uint8_t>((size_t)K * (n_fg > 0 ? n_fg : 1));
  w.inpaint = a.take<uint8_t>(K * R2);
  w.keep_idx = a.take<int>((size_t)K * (n_fg > 0 ? n_fg : 1));
  w.unk = a.take<int>(K * R2);
  w.block_counts = a.take<int>((size_t)K * (cdiv((int)P, CP_TILE) + 1));
  w.offs_close = a.take<int2>(MAX_OFFS);
  w.offs_open = a.take<int2>(MAX_OFFS);
  w.vx = a.take<double>(K * R2);
  w.vr = a.take<double>(K * R2);
  w.vp = a.take<double>(K * R2);
  w.vq = a.take<double>(K * R2);
  w.cgsync = a.take<CgSync>(K);
  return a.ok();
}

}  // namespace dh

using namespace dh;

extern "C" int dh_reproject_workspace_bytes(int res, int n_fg, int n_edits, size_t* bytes) {
  DH_REQUIRE(res >= 2 && n_fg >= 0 && n_edits >= 1 && bytes, "bad arguments");
  Arena a(nullptr, (size_t)-1);
  ReprojectWs w;
  carve(a, res, n_fg, n_edits, w);
  *bytes = a.off + 256;
  return DH_OK;
}

extern "C" int dh_fg_pixel_list(const uint8_t* fg_mask, int res, int32_t* fg_pix, int32_t* n_fg_dev, void* workspace,
                                size_t workspace_bytes, void* stream) {
  DH_REQUIRE(fg_mask && fg_pix && n_fg_dev && workspace, "null pointer");
  int n = res * res;
  DH_REQUIRE(workspace_bytes >= (size_t)(cdiv(n, CP_TILE) + 1) * sizeof(int), "workspace too small");
  compact(fg_mask, n, 1, 0, fg_pix, 0, n_fg_dev, 1, (int*)workspace, (hipStream_t)stream);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_unproject(const float* depth, int res, const float* grid_x, const float* grid_y, float inv_fx,
                            float inv_fy, float* points, void* stream) {
  DH_REQUIRE(depth && grid_x && grid_y && points && res >= 2, "bad arguments");
  hipLaunchKernelGGL(k_unproject, dim3(cdiv(res * res, 256)), dim3(256), 0, (hipStream_t)stream, depth, res, grid_x,
                     grid_y, inv_fx, inv_fy, points);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_masked_centroid(const float* depth, const int32_t* fg_pix, int n_fg, int res, const float* grid_x,
                                  const float* grid_y, float inv_fx, float inv_fy, float* centroid, void* stream) {
  DH_REQUIRE(depth && fg_pix && centroid && n_fg > 0, "bad arguments");
  hipLaunchKernelGGL(k_centroid, dim3(1), dim3(CEN_CHUNK), 0, (hipStream_t)stream, depth, fg_pix, n_fg, res, grid_x, grid_y,
                     inv_fx, inv_fy, centroid);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_reproject_edits(const float* depth, const float* bg_depth, const int32_t* fg_pix, int n_fg, int res,
                                  const float* grid_x, const float* grid_y, float inv_fx, float inv_fy, double fx,
                                  double fy, int n_edits, const double* xforms_host, const float* bounds, float* zmap,
                                  uint8_t* raw_mask, uint8_t* clean_mask, float* disparity, uint8_t* vis,
                                  int32_t* target_xy, int64_t* corr, int32_t* counts, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  DH_REQUIRE(depth && bg_depth && fg_pix && grid_x && grid_y && xforms_host, "null input");
  DH_REQUIRE(zmap && raw_mask && clean_mask && disparity && vis && target_xy && corr && counts, "null output");
  DH_REQUIRE(res >= 2 && n_fg > 0 && n_edits >= 1, "bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int K = n_edits, R2 = res * res, P = R2 + n_fg;
  Arena a(workspace, workspace_bytes);
  ReprojectWs w;
  DH_REQUIRE(carve(a, res, n_fg, K, w), "workspace too small");

  int2 hc[MAX_OFFS], ho[MAX_OFFS];
  const int kc = res / 50, ko = res / 250;
  DH_REQUIRE(kc * kc <= MAX_OFFS && kc >= 1, "unsupported resolution for the close kernel");
  const int nc = ellipse_offsets(kc, hc);
  const int no = ellipse_offsets(ko < 1 ? 1 : ko, ho);
  DH_CHECK_HIP(hipMemcpyAsync(w.offs_close, hc, nc * sizeof(int2), hipMemcpyHostToDevice, st));
  DH_CHECK_HIP(hipMemcpyAsync(w.offs_open, ho, no * sizeof(int2), hipMemcpyHostToDevice, st));
  DH_CHECK_HIP(hipMemcpyAsync(w.xf, xforms_host, K * sizeof(Xf), hipMemcpyHostToDevice, st));
  DH_CHECK_HIP(hipMemsetAsync(w.zbuf, 0xff, (size_t)K * R2 * sizeof(unsigned long long), st));
  DH_CHECK_HIP(hipMemsetAsync(w.owner, 0x7f, (size_t)K * R2 * sizeof(int), st));
  DH_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)K * 4 * sizeof(int), st));
  // minmax init: min slot = 0xffffffff, max slot = 0
  DH_CHECK_HIP(hipMemsetAsync(w.minmax, 0, (size_t)2 * K * sizeof(unsigned int), st));
  for (int e = 0; e < K; ++e) DH_CHECK_HIP(hipMemsetAsync(w.minmax + 2 * e, 0xff, sizeof(unsigned int), st));

  hipLaunchKernelGGL(k_centroid, dim3(1), dim3(CEN_CHUNK), 0, st, depth, fg_pix, n_fg, res, grid_x, grid_y, inv_fx, inv_fy,
                     w.cen);
  hipLaunchKernelGGL(k_points, dim3(cdiv(P, 256), K), dim3(256), 0, st, depth, bg_depth, fg_pix, n_fg, res, grid_x,
                     grid_y, inv_fx, inv_fy, fx, fy, w.xf, w.cen, w.zbuf, w.pix, w.key);
  hipLaunchKernelGGL(k_resolve, dim3(cdiv(P, 256), K), dim3(256), 0, st, P, R2, w.zbuf, w.pix, w.key, w.owner);
  hipLaunchKernelGGL(k_pixels, dim3(cdiv(R2, 256 * PIX_PER_THREAD), K), dim3(256), 0, st, R2, w.zbuf, w.owner, zmap, raw_mask,
                     disparity, w.minmax);
  hipLaunchKernelGGL(k_fg_vis, dim3(cdiv(n_fg, 256), K), dim3(256), 0, st, n_fg, R2, P, res, w.pix, w.owner, vis,
                     target_xy);
  // CLOSE = dilate, erode ; OPEN = erode, dilate
  if (res % 32 == 0 && kc < 32 && ko < 32) {        // bit rows: pack, four passes (erode = NOT dilate NOT), unpack
    const int nw = K * R2 / 32, wpe = R2 / 32;
    unsigned* ba = reinterpret_cast<unsigned*>(w.tmp_a);
    unsigned* bb = reinterpret_cast<unsigned*>(w.tmp_b);
    hipLaunchKernelGGL(k_pack_bits, dim3(cdiv(nw, 256)), dim3(256), 0, st, raw_mask, ba, nw);
    hipLaunchKernelGGL(k_morph_bits, dim3(cdiv(wpe, 256), K), dim3(256), 0, st, ba, bb, res, w.offs_close, nc, 0, 0);
    hipLaunchKernelGGL(k_morph_bits, dim3(cdiv(wpe, 256), K), dim3(256), 0, st, bb, ba, res, w.offs_close, nc, 1, 1);
    hipLaunchKernelGGL(k_morph_bits, dim3(cdiv(wpe, 256), K), dim3(256), 0, st, ba, bb, res, w.offs_open, no, 1, 1);
    hipLaunchKernelGGL(k_morph_bits, dim3(cdiv(wpe, 256), K), dim3(256), 0, st, bb, ba, res, w.offs_open, no, 0, 0);
    hipLaunchKernelGGL(k_unpack_bits, dim3(cdiv(nw, 256)), dim3(256), 0, st, ba, clean_mask, nw);
  } else {
    hipLaunchKernelGGL(k_morph, dim3(cdiv(R2, 256), K), dim3(256), 0, st, raw_mask, w.tmp_a, res, w.offs_close, nc, 1);
    hipLaunchKernelGGL(k_morph, dim3(cdiv(R2, 256), K), dim3(256), 0, st, w.tmp_a, w.tmp_b, res, w.offs_close, nc, 0);
    hipLaunchKernelGGL(k_morph, dim3(cdiv(R2, 256), K), dim3(256), 0, st, w.tmp_b, w.tmp_a, res, w.offs_open, no, 0);
    hipLaunchKernelGGL(k_morph, dim3(cdiv(R2, 256), K), dim3(256), 0, st, w.tmp_a, clean_mask, res, w.offs_open, no, 1);
  }
  hipLaunchKernelGGL(k_keep_flags, dim3(cdiv(n_fg, 256), K), dim3(256), 0, st, n_fg, R2, P, w.pix, vis, clean_mask,
                     w.keep);
  compact(w.keep, n_fg, K, n_fg, w.keep_idx, n_fg, counts + 0, 4, w.block_counts, st);
  hipLaunchKernelGGL(k_write_corr, dim3(cdiv(n_fg, 256), K), dim3(256), 0, st, n_fg, res, fg_pix, target_xy,
                     w.keep_idx, counts, 4, (long long*)corr);
  hipLaunchKernelGGL(k_count_flags, dim3(cdiv(n_fg, 256), K), dim3(256), 0, st, vis, n_fg, counts, 4, 1);
  hipLaunchKernelGGL(k_normalize, dim3(cdiv(R2, 256), K), dim3(256), 0, st, R2, disparity, w.minmax, bounds, raw_mask,
                     clean_mask, w.inpaint);
  compact(w.inpaint, R2, K, R2, w.unk, R2, counts + 2, 4, w.block_counts, st);
#ifdef DH_TUNING
  static const bool cg_lds = !(getenv("DH_CG_LDS") && atoi(getenv("DH_CG_LDS")) == 0);
#else
  constexpr bool cg_lds = true;
#endif
  // scratch of the solve: the z-buffer, owner map and key list are dead by now (last read by k_pixels / k_write_corr)
  // holes beyond one CU's LDS: CGM_WGS workgroups per edit (every polled word zeroed per call); beyond that kernel's
  // register budget (65 536 unknowns) the single-workgroup kernel
  DH_CHECK_HIP(hipMemsetAsync(w.cgsync, 0, (size_t)K * sizeof(CgSync), st));
  hipLaunchKernelGGL(k_cg_fill_multi, dim3(CGM_WGS, K), dim3(1024), 0, st, res, disparity, w.inpaint, w.unk, counts, 4, 2, 3, w.vx,
                     w.vr, w.vp, w.vq, 20000, 1e-24, counts, (const float*)nullptr, cg_lds ? CG_SLOTS * 1024 : 0, w.owner,
                     reinterpret_cast<int2*>(w.zbuf), (size_t)R2, reinterpret_cast<int2*>(w.key), (size_t)P, w.cgsync);
  hipLaunchKernelGGL(k_cg_fill, dim3(K), dim3(1024), 0, st, res, disparity, w.inpaint, w.unk, counts, 4, 2, 3, w.vx,
                     w.vr, w.vp, w.vq, 20000, 1e-24, counts, (const float*)nullptr, CGM_WGS * CGM_EPT * 1024, w.owner,
                     reinterpret_cast<int2*>(w.zbuf), (size_t)R2, reinterpret_cast<int2*>(w.key), (size_t)P);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_laplacian_blend_workspace_bytes(int res, size_t* bytes) {
  DH_REQUIRE(res >= 2 && bytes, "bad arguments");
  const size_t R2 = (size_t)res * res;
  *bytes = 4 * align_up(R2 * 8, 256) + 3 * align_up(R2, 256) + 3 * align_up(R2 * 4, 256) + 2 * align_up(R2 * 8, 256) +
           (size_t)(cdiv((int)R2, CP_TILE) + 2) * 4 + 4096 + align_up(sizeof(CgSync), 256);
  return DH_OK;
}

// DiffusionHandles.set_foreground (diffusion_handles.py:90-111) = utils.solve_laplacian_depth (utils.py:49-102)
// over binary_dilation(fg_mask, iterations): out = depth outside the dilated mask, inside the solution of
// 4 x - sum(masked nbrs) = sum(known depth nbrs) - laplacian(bg_depth).
extern "C" int dh_laplacian_blend(const float* depth, const float* bg_depth, const uint8_t* fg_mask, int res,
                                  int dilate_iters, float* out, int32_t* counts, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  DH_REQUIRE(depth && bg_depth && fg_mask && out && counts && workspace && res >= 2 && dilate_iters >= 0, "bad arguments");
  size_t need;
  dh_laplacian_blend_workspace_bytes(res, &need);
  DH_REQUIRE(workspace_bytes >= need, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int R2 = res * res;
  Arena a(workspace, workspace_bytes);
  double* vx = a.take<double>(R2); double* vr = a.take<double>(R2); double* vp = a.take<double>(R2); double* vq = a.take<double>(R2);
  uint8_t* m0 = a.take<uint8_t>(R2); uint8_t* m1 = a.take<uint8_t>(R2);
  float* lap = a.take<float>(R2);
  int* unk = a.take<int>(R2);
  int* pixmap = a.take<int>(R2);
  int2* nb_ud = a.take<int2>(R2);
  int2* nb_lr = a.take<int2>(R2);
  int* bc = a.take<int>(cdiv(R2, CP_TILE) + 2);
  CgSync* cgs = a.take<CgSync>(1);
  DH_CHECK_HIP(hipMemcpyAsync(m0, fg_mask, R2, hipMemcpyDeviceToDevice, st));
  uint8_t *src = m0, *dst = m1;
  for (int i = 0; i < dilate_iters; ++i) {
    hipLaunchKernelGGL(k_dilate_cross, dim3(cdiv(R2, 256)), dim3(256), 0, st, src, dst, res);
    uint8_t* t = src; src = dst; dst = t;
  }
  hipLaunchKernelGGL(k_laplacian, dim3(cdiv(R2, 256)), dim3(256), 0, st, bg_depth, lap, res);
  DH_CHECK_HIP(hipMemcpyAsync(out, depth, (size_t)R2 * 4, hipMemcpyDeviceToDevice, st));
  DH_CHECK_HIP(hipMemsetAsync(counts, 0, 4 * sizeof(int), st));
  compact(src, R2, 1, 0, unk, 0, counts + 2, 1, bc, st);
  DH_CHECK_HIP(hipMemsetAsync(cgs, 0, sizeof(CgSync), st));
  hipLaunchKernelGGL(k_cg_fill_multi, dim3(CGM_WGS, 1), dim3(1024), 0, st, res, out, src, unk, counts, 4, 2, 3, vx, vr, vp, vq,
                     50000, 1e-24, counts, (const float*)lap, CG_SLOTS * 1024, pixmap, nb_ud, (size_t)R2, nb_lr, (size_t)R2, cgs);
  hipLaunchKernelGGL(k_cg_fill, dim3(1), dim3(1024), 0, st, res, out, src, unk, counts, 4, 2, 3, vx, vr, vp, vq, 50000,
                     1e-24, counts, (const float*)lap, CGM_WGS * CGM_EPT * 1024, pixmap, nb_ud, (size_t)R2, nb_lr, (size_t)R2);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
