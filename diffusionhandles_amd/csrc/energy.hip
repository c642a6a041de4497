// Guidance energy + its gradient w.r.t. the current activations (losses.py:4-84), on
// channels-last maps [cell][C] so that a cell gather is one coalesced row read.
//
//   fg term   = mean_c mean_n | A1[c, o_n] - A2[c, t_n] |          (pairs, duplicates kept)
//   bg global = mean_c | mean_{BGo} F1[c] - mean_{BGt} F2[c] |
//   bg local  = fg-style term over the identity pairs of BG_both
//   A = pool_p(w F) / (pool_p(w) + 1e-10), w = indicator of the cells named by the index lists
//
// The gradient is accumulated per target cell from a CSR (target cell -> source cells)
// built by a counting sort: every output element has exactly one writer, sign sums are
// integers, so the gradient is bit-deterministic with no float atomics.  Loss values are
// accumulated in float64 (segment order is the only order dependence).
#include "common.h"

namespace dh {

// ---- map load / store (with optional bilinear resize, align_corners = False) ----------
__device__ __forceinline__ void bilinear_src(int dst, int n_in, float scale, int& i0, int& i1, float& l0, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

template <class T>
__global__ void k_load_map(const T* src, float* dst, int C, int hin, int win, int grid) {
  const int cell = blockIdx.x;
  const int y = cell / grid, x = cell - y * grid;
  if (hin == grid && win == grid) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) dst[(size_t)cell * C + c] = to_f32<T>(src[(size_t)cell * C + c]);
    return;
  }
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  bilinear_src(y, hin, (float)hin / (float)grid, y0, y1, ly0, ly1);
  bilinear_src(x, win, (float)win / (float)grid, x0, x1, lx0, lx1);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v00 = to_f32<T>(src[((size_t)y0 * win + x0) * C + c]);
    float v01 = to_f32<T>(src[((size_t)y0 * win + x1) * C + c]);
    float v10 = to_f32<T>(src[((size_t)y1 * win + x0) * C + c]);
    float v11 = to_f32<T>(src[((size_t)y1 * win + x1) * C + c]);
    dst[(size_t)cell * C + c] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  }
}

// transpose of k_load_map: one block per INPUT cell, gathers from the grid cells that read it
template <class T>
__global__ void k_store_grad(const float* g, T* out, int C, int hin, int win, int grid, float scale) {
  const int cell = blockIdx.x;
  if (hin == grid && win == grid) {
    for (int c = threadIdx.x; c < C; c += blockDim.x)
      out[(size_t)cell * C + c] = from_f32<T>(g[(size_t)cell * C + c] * scale);
    return;
  }
  const int yi = cell / win, xi = cell - yi * win;
  const float sy = (float)hin / (float)grid, sx = (float)win / (float)grid;
  int ylo = (int)floorf(((float)yi - 1.f) / sy) - 1, yhi = (int)ceilf(((float)yi + 1.f) / sy) + 1;
  int xlo = (int)floorf(((float)xi - 1.f) / sx) - 1, xhi = (int)ceilf(((float)xi + 1.f) / sx) + 1;
  ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
  yhi = yhi > grid - 1 ? grid - 1 : yhi; xhi = xhi > grid - 1 ? grid - 1 : xhi;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
      int y0, y1; float ly0, ly1;
      bilinear_src(y, hin, sy, y0, y1, ly0, ly1);
      float wy = (y0 == yi ? ly0 : 0.f) + (y1 == yi ? ly1 : 0.f);
      if (wy == 0.f) continue;
      for (int x = xlo; x <= xhi; ++x) {
        int x0, x1; float lx0, lx1;
        bilinear_src(x, win, sx, x0, x1, lx0, lx1);
        float wx = (x0 == xi ? lx0 : 0.f) + (x1 == xi ? lx1 : 0.f);
        if (wx == 0.f) continue;
        acc += wy * wx * g[((size_t)y * grid + x) * C + c];
      }
    }
    out[(size_t)cell * C + c] = from_f32<T>(acc * scale);
  }
}

// ---- CSR: target cell -> list of source cells -----------------------------------------
__global__ void k_hist(const int* pairs, int n, int* cnt, uint8_t* w1, uint8_t* w2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int o = pairs[2 * (size_t)i], t = pairs[2 * (size_t)i + 1];
  atomicAdd(&cnt[t], 1);
  w1[o] = 1;
  w2[t] = 1;
}

__global__ void k_identity_pairs(const int* list, int n, int* pairs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pairs[2 * (size_t)i] = list[i];
  pairs[2 * (size_t)i + 1] = list[i];
}

// exclusive scan of cnt[0..n) into off[0..n], cursor copy; single workgroup of 1024
__global__ void __launch_bounds__(1024) k_scan_cells(const int* cnt, int n, int* off, int* cursor) {
  __shared__ int sm[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + threadIdx.x;
    int v = i < n ? cnt[i] : 0;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int t = threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
      __syncthreads();
      sm[threadIdx.x] += t;
      __syncthreads();
    }
    int excl = sm[threadIdx.x] - v + carry;
    if (i < n) { off[i] = excl; cursor[i] = excl; }
    __syncthreads();
    if (threadIdx.x == 1023) carry += sm[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) off[n] = carry;
}

__global__ void k_fill(const int* pairs, int n, int* cursor, int* src) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int o = pairs[2 * (size_t)i], t = pairs[2 * (size_t)i + 1];
  src[atomicAdd(&cursor[t], 1)] = o;
}

// ---- pooling --------------------------------------------------------------------------
// out[cell][c] = sum_window(w * X) / (sum_window(w) + p*p*1e-10); den[cell] = that denominator
__global__ void k_pool(const float* X, const uint8_t* w, int C, int grid, int p, float* out, float* den) {
  const int cell = blockIdx.x, y = cell / grid, x = cell - y * grid, r = p / 2;
  float wn = 0.f;
  for (int dy = -r; dy <= r; ++dy)
    for (int dx = -r; dx <= r; ++dx) {
      int yy = y + dy, xx = x + dx;
      if (yy >= 0 && yy < grid && xx >= 0 && xx < grid && w[yy * grid + xx]) wn += 1.f;
    }
  const float d = wn + (float)(p * p) * 1e-10f;
  if (threadIdx.x == 0) den[cell] = d;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int dy = -r; dy <= r; ++dy)
      for (int dx = -r; dx <= r; ++dx) {
        int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < grid && xx >= 0 && xx < grid && w[yy * grid + xx]) s += X[((size_t)yy * grid + xx) * C + c];
      }
    out[(size_t)cell * C + c] = s / d;
  }
}

// acc[cell][c] += w[cell] * sum_window(G / den)
__global__ void k_spread(const float* G, const uint8_t* w, const float* den, int C, int grid, int p, float* acc) {
  const int cell = blockIdx.x, y = cell / grid, x = cell - y * grid, r = p / 2;
  if (!w[cell]) return;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int dy = -r; dy <= r; ++dy)
      for (int dx = -r; dx <= r; ++dx) {
        int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < grid && xx >= 0 && xx < grid) s += G[((size_t)yy * grid + xx) * C + c] / den[yy * grid + xx];
      }
    acc[(size_t)cell * C + c] += s;
  }
}

// ---- pair term --------------------------------------------------------------------------
// one block per target cell t: G[t][c] (+)= -coef * sum_{k in seg(t)} sign(X1[src_k][c] - X2[t][c])
__global__ void k_pair_term(const float* X1, const float* X2, const int* off, const int* src, int C, float coef,
                            float* G, int accumulate, double* loss_part) {
  __shared__ double sm[4];
  const int t = blockIdx.x;
  const int b = off[t], e = off[t + 1];
  double l = 0.0;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float a = X2[(size_t)t * C + c];
    int s = 0;
    double la = 0.0;
    for (int k = b; k < e; ++k) {
      float d = X1[(size_t)src[k] * C + c] - a;
      la += (double)fabsf(d);
      s += (d > 0.f) - (d < 0.f);
    }
    float g = -coef * (float)s;
    if (accumulate) G[(size_t)t * C + c] += g; else G[(size_t)t * C + c] = g;
    l += la;
  }
  l = block_sum(l, sm);
  if (threadIdx.x == 0) loss_part[t] = l;
}

// ---- global-average term ----------------------------------------------------------------
// part[s][c] = sum over the s-th slice of `list` of X[cell][c]
__global__ void k_colsum(const float* X, const int* list, int n, int C, int S, float* part) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
  if (c >= C) return;
  const int per = (n + S - 1) / S, b = s * per, e = b + per < n ? b + per : n;
  float acc = 0.f;
  for (int k = b; k < e; ++k) acc += X[(size_t)list[k] * C + c];
  part[(size_t)s * C + c] = acc;
}

// sgn[c] = sign(m1 - m2), loss_c = |m1 - m2|.  Workgroup = 64 channels x 4 waves: wave q adds the slice partials of
// quarter q in slice order, the four quarter sums are added in quarter order (fixed tree: deterministic)
constexpr int GD_BLOCK = 64;
__global__ void __launch_bounds__(4 * GD_BLOCK) k_global_diff(const float* p1, const float* p2, int S, int C, int n1, int n2,
                                                              float* sgn, double* loss_part) {
  __shared__ float sq[2][4][GD_BLOCK];
  const int cl = threadIdx.x & (GD_BLOCK - 1), q = threadIdx.x / GD_BLOCK;
  const int c = blockIdx.x * GD_BLOCK + cl;
  const int per = (S + 3) / 4, s0 = q * per, s1 = s0 + per < S ? s0 + per : S;
  float a = 0.f, b = 0.f;
  if (c < C) {
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
      float va[8], vb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { va[j] = p1[(size_t)(s + j) * C + c]; vb[j] = p2[(size_t)(s + j) * C + c]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { a += va[j]; b += vb[j]; }
    }
    for (; s < s1; ++s) { a += p1[(size_t)s * C + c]; b += p2[(size_t)s * C + c]; }
  }
  sq[0][q][cl] = a; sq[1][q][cl] = b;
  __syncthreads();
  if (q != 0) return;
  double l = 0.0;
  if (c < C) {
    a = ((sq[0][0][cl] + sq[0][1][cl]) + sq[0][2][cl]) + sq[0][3][cl];
    b = ((sq[1][0][cl] + sq[1][1][cl]) + sq[1][2][cl]) + sq[1][3][cl];
    const float d = a / (float)n1 - b / (float)n2;
    sgn[c] = (float)((d > 0.f) - (d < 0.f));
    l = (double)fabsf(d);
  }
  l = wave_sum(l);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = l;
}

__global__ void k_global_apply(const int* list, int n, const float* sgn, int C, float coef, float* acc) {
  const int cell = list[blockIdx.x];
  for (int c = threadIdx.x; c < C; c += blockDim.x) acc[(size_t)cell * C + c] += -coef * sgn[c];
}

__global__ void k_final_loss(const double* fg_part, int n_fg_part, float fg_norm, const double* bg_part, int n_bg_part,
                             float bg_norm, float fg_w, float bg_w, float* out) {
  __shared__ double sm[4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n_fg_part; i += blockDim.x) a += fg_part[i];
  for (int i = threadIdx.x; i < n_bg_part; i += blockDim.x) b += bg_part[i];
  a = block_sum(a, sm);
  b = block_sum(b, sm);
  if (threadIdx.x == 0) {
    float fg = (float)(a * (double)fg_norm), bg = (float)(b * (double)bg_norm);
    out[0] = fg_w * fg + bg_w * bg;
    out[1] = fg;
    out[2] = bg;
  }
}

struct EnergyWs {
  float *cur, *orig, *acc, *A1, *A2, *G, *den1, *den2, *part1, *part2, *sgn;
  int *cnt, *off, *cursor, *src, *idpairs;
  uint8_t *w1, *w2;
  double *fg_part, *bg_part;
};
constexpr int COLSUM_S = 128;     // slices of a background column sum (more slices = shorter serial gather chains)

static bool carve_energy(Arena& a, int C, int grid, int n_pairs, EnergyWs& w) {
  const size_t G2 = (size_t)grid * grid, M = G2 * C;
  const size_t np = (size_t)(n_pairs > (int)G2 ? n_pairs : (int)G2);
  w.cur = a.take<float>(M); w.orig = a.take<float>(M); w.acc = a.take<float>(M);
  w.A1 = a.take<float>(M); w.A2 = a.take<float>(M); w.G = a.take<float>(M);
  w.den1 = a.take<float>(G2); w.den2 = a.take<float>(G2);
  w.part1 = a.take<float>((size_t)COLSUM_S * C); w.part2 = a.take<float>((size_t)COLSUM_S * C);
  w.sgn = a.take<float>(C);
  w.cnt = a.take<int>(G2 + 1); w.off = a.take<int>(G2 + 1); w.cursor = a.take<int>(G2 + 1);
  w.src = a.take<int>(np); w.idpairs = a.take<int>(2 * G2);
  w.w1 = a.take<uint8_t>(G2); w.w2 = a.take<uint8_t>(G2);
  w.fg_part = a.take<double>(G2); w.bg_part = a.take<double>(G2);
  return a.ok();
}

template <class T>
static void launch_load(const void* src, float* dst, int C, int hin, int win, int grid, hipStream_t st) {
  hipLaunchKernelGGL((k_load_map<T>), dim3(grid * grid), dim3(256), 0, st, (const T*)src, dst, C, hin, win, grid);
}
template <class T>
static void launch_store(const float* g, void* out, int C, int hin, int win, int grid, float scale, hipStream_t st) {
  hipLaunchKernelGGL((k_store_grad<T>), dim3(hin * win), dim3(256), 0, st, g, (T*)out, C, hin, win, grid, scale);
}

// one pair-type term: CSR build, optional pooling, gradient into acc, loss partials into part
static void pair_term(const EnergyWs& w, const int* pairs, int n, int C, int grid, int patch, float coef,
                      double* part, hipStream_t st) {
  const int G2 = grid * grid;
  (void)hipMemsetAsync(w.cnt, 0, (G2 + 1) * sizeof(int), st);
  (void)hipMemsetAsync(w.w1, 0, G2, st);
  (void)hipMemsetAsync(w.w2, 0, G2, st);
  hipLaunchKernelGGL(k_hist, dim3(cdiv(n, 256)), dim3(256), 0, st, pairs, n, w.cnt, w.w1, w.w2);
  hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, st, w.cnt, G2, w.off, w.cursor);
  hipLaunchKernelGGL(k_fill, dim3(cdiv(n, 256)), dim3(256), 0, st, pairs, n, w.cursor, w.src);
  if (patch <= 1) {
    hipLaunchKernelGGL(k_pair_term, dim3(G2), dim3(256), 0, st, w.orig, w.cur, w.off, w.src, C, coef, w.acc, 1, part);
  } else {
    hipLaunchKernelGGL(k_pool, dim3(G2), dim3(256), 0, st, w.orig, w.w1, C, grid, patch, w.A1, w.den1);
    hipLaunchKernelGGL(k_pool, dim3(G2), dim3(256), 0, st, w.cur, w.w2, C, grid, patch, w.A2, w.den2);
    hipLaunchKernelGGL(k_pair_term, dim3(G2), dim3(256), 0, st, w.A1, w.A2, w.off, w.src, C, coef, w.G, 0, part);
    hipLaunchKernelGGL(k_spread, dim3(G2), dim3(256), 0, st, w.G, w.w2, w.den2, C, grid, patch, w.acc);
  }
}

// ---- planned fast path ------------------------------------------------------------------
// The default configuration (maps already at the cell grid, fg_patch 1, 'global_avg' background) needs no f32
// staging, no pooling and -- because the correspondences are fixed for an edit -- no CSR rebuild per evaluation:
//   plan (once per edit)  : CSR target cell -> source cells, flag of the transformed-background cells
//   per evaluation        : k_colsum_q (both background column sums in the slice order of k_colsum, combined per quarter of
//                                       the slices as k_global_diff combines them)
//                           k_energy_grad (prologue: sign of the mean difference per channel from the quarter sums -- the rest of
//                                          k_global_diff's arithmetic; then pair term + background term -> one 16-byte gradient store per lane)
//                           k_final_loss  (only when the caller wants the loss values)
// Same arithmetic, in the same order per element, as the general path above: the gradient is bit-identical.
struct EnergyPlan {
  int *off, *src, *cnt, *cursor, *mult;      // after the build: cnt[t] = number of DISTINCT sources of target t,
  uint8_t *bgflag, *w1, *w2;                 // src / mult [off[t], off[t] + cnt[t]) = their cells (ascending) / multiplicities
};
static bool carve_plan(Arena& a, int grid, int n_pairs, EnergyPlan& p) {
  const size_t G2 = (size_t)grid * grid;
  p.off = a.take<int>(G2 + 1); p.cnt = a.take<int>(G2 + 1); p.cursor = a.take<int>(G2 + 1);
  p.src = a.take<int>(n_pairs > 0 ? n_pairs : 1);
  p.mult = a.take<int>(n_pairs > 0 ? n_pairs : 1);
  p.bgflag = a.take<uint8_t>(G2); p.w1 = a.take<uint8_t>(G2); p.w2 = a.take<uint8_t>(G2);
  return a.ok();
}
struct PlannedWs {
  float* partq;                  // [2 lists][4 quarters][C] quarter sums of the background column sums (k_colsum_q)
  double *fg_part, *bg_part;
};
static bool carve_planned(Arena& a, int C, int grid, PlannedWs& w) {
  const size_t G2 = (size_t)grid * grid;
  w.partq = a.take<float>((size_t)8 * C);
  w.fg_part = a.take<double>(G2); w.bg_part = a.take<double>(2);
  return a.ok();
}

// one workgroup per target cell: its source list (one entry per pixel pair: ~8x8 pixels share a cell pair) becomes
// (distinct source cell, multiplicity) in ascending cell order, through an LDS histogram over the grid's cells
__global__ void __launch_bounds__(256) k_dedupe_sources(const int* off, int* src, int* mult, int* ucnt, int G2) {
  extern __shared__ int hist[];          // [G2] counts, then [256] scan scratch
  int* scan = hist + G2;
  const int t = blockIdx.x, b = off[t], e = off[t + 1];
  if (b == e) { if (threadIdx.x == 0) ucnt[t] = 0; return; }
  for (int c = threadIdx.x; c < G2; c += blockDim.x) hist[c] = 0;
  __syncthreads();
  for (int k = b + threadIdx.x; k < e; k += blockDim.x) atomicAdd(&hist[src[k]], 1);
  __syncthreads();
  const int per = (G2 + blockDim.x - 1) / blockDim.x, c0 = threadIdx.x * per, c1 = min(c0 + per, G2);
  int mine = 0;
  for (int c = c0; c < c1; ++c) mine += hist[c] != 0;
  scan[threadIdx.x] = mine;
  __syncthreads();
  for (int o = 1; o < (int)blockDim.x; o <<= 1) {
    const int v = (int)threadIdx.x >= o ? scan[threadIdx.x - o] : 0;
    __syncthreads();
    scan[threadIdx.x] += v;
    __syncthreads();
  }
  int pos = b + scan[threadIdx.x] - mine;
  for (int c = c0; c < c1; ++c)
    if (hist[c]) { src[pos] = c; mult[pos] = hist[c]; ++pos; }
  if (threadIdx.x == blockDim.x - 1) ucnt[t] = scan[threadIdx.x];
}

__global__ void k_flag_cells(const int* list, int n, uint8_t* flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[list[i]] = 1;
}

// Background (global_avg) term, first half (round 5: the evaluation is two launches, k_colsum_q -> k_energy_grad; it used to be
// k_colsum16 -> k_global_diff -> k_energy_grad).  Workgroup = (64 channels, quarter q of the COLSUM_S = 128 slices, list z); thread
// = (slice of the quarter, 8-channel chunk): the eight lanes of a slice read 128 contiguous bytes of a row.  A thread adds the rows
// of its slice in list order (16 gathers in flight), the 32 slice sums of the quarter are added in slice order through LDS:
// partq[z][q][c] -- exactly k_colsum's slices combined the way k_global_diff combines the slices of a quarter, so that
// ((q0 + q1) + q2) + q3 in k_energy_grad's prologue reproduces sgn[c] bit for bit.
constexpr int CQ_SL = COLSUM_S / 4;          // slices per quarter
template <class T>
__global__ void __launch_bounds__(8 * CQ_SL) k_colsum_q(const T* X1, const int* list1, int n1, const T* X2, const int* list2, int n2,
                                                       int C, float* partq) {
  __shared__ float sp[CQ_SL][8][8];
  const int z = blockIdx.z, q = blockIdx.y, sloc = threadIdx.x >> 3, cl = threadIdx.x & 7;
  const int ch = blockIdx.x * 8 + cl;                       // 8-channel chunk
  const T* X = z ? X2 : X1;
  const int* list = z ? list2 : list1;
  const int n = z ? n2 : n1;
  const int sl = q * CQ_SL + sloc;
  const int per = (n + COLSUM_S - 1) / COLSUM_S, b = sl * per, e = b + per < n ? b + per : n;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (ch * 8 < C) {
    int k = b;
    for (; k + 16 <= e; k += 16) {          // 16 gathers in flight; the adds keep the list order
      int id[16];
      uint4 raw[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) id[j] = list[k + j];
#pragma unroll
      for (int j = 0; j < 16; ++j) raw[j] = *reinterpret_cast<const uint4*>(X + (size_t)id[j] * C + ch * 8);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const T* v = reinterpret_cast<const T*>(&raw[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += to_f32<T>(v[i]);
      }
    }
    if (k + 8 <= e) {
      int id[8];
      uint4 raw[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) id[j] = list[k + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[j] = *reinterpret_cast<const uint4*>(X + (size_t)id[j] * C + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const T* v = reinterpret_cast<const T*>(&raw[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += to_f32<T>(v[i]);
      }
      k += 8;
    }
    for (; k < e; ++k) {
      const uint4 raw = *reinterpret_cast<const uint4*>(X + (size_t)list[k] * C + ch * 8);
      const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += to_f32<T>(v[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) sp[sloc][cl][i] = acc[i];
  __syncthreads();
  if (threadIdx.x < 64) {                   // (chunk, channel): the quarter's slices in slice order
    const int c2 = threadIdx.x >> 3, i = threadIdx.x & 7;
    float a = 0.f;
    for (int s = 0; s < CQ_SL; ++s) a += sp[s][c2][i];
    const int c = (blockIdx.x * 8 + c2) * 8 + i;
    if (c < C) partq[((size_t)z * 4 + q) * C + c] = a;
  }
}

// thread = (target cell, 8-channel chunk); block = 256 / (C/8) cells
template <class T, class TG>
__global__ void __launch_bounds__(256) k_energy_grad(const T* orig, const T* cur, const int* off, const int* ucnt,
                                                     const int* src, const int* mult, const uint8_t* bgflag, const float* partq, int n1, int n2,
                                                     int C, int G2, float coef_fg,
                                                     float coef_bg, int use_bg, float scale, TG* grad, double* loss_part, double* bg_loss) {
  __shared__ double sm[4];
  const int nch = C / 8, cpb = (int)blockDim.x / nch;
  const int lc = threadIdx.x / nch, ch = threadIdx.x - lc * nch;
  const int cell = blockIdx.x * cpb + lc;
  double la = 0.0, lb = 0.0;
  // prologue: sign of the difference of the two background means of this thread's 8 channels from the quarter sums of k_colsum_q
  // (a 2 x 4 x C f32 table, L2-resident; every lane of a chunk reads the same 16 sectors) -- the arithmetic of k_global_diff
  float sgn[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sgn[i] = 0.f;
  if (use_bg && lc < cpb) {
    float qa[4][8], qb[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<float4*>(&qa[q][0]) = *reinterpret_cast<const float4*>(partq + (size_t)q * C + ch * 8);
      *reinterpret_cast<float4*>(&qa[q][4]) = *reinterpret_cast<const float4*>(partq + (size_t)q * C + ch * 8 + 4);
      *reinterpret_cast<float4*>(&qb[q][0]) = *reinterpret_cast<const float4*>(partq + (size_t)(4 + q) * C + ch * 8);
      *reinterpret_cast<float4*>(&qb[q][4]) = *reinterpret_cast<const float4*>(partq + (size_t)(4 + q) * C + ch * 8 + 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float a = ((qa[0][i] + qa[1][i]) + qa[2][i]) + qa[3][i];
      const float b = ((qb[0][i] + qb[1][i]) + qb[2][i]) + qb[3][i];
      const float d = a / (float)n1 - b / (float)n2;
      sgn[i] = (float)((d > 0.f) - (d < 0.f));
      if (blockIdx.x == 0 && lc == 0) lb += (double)fabsf(d);          // the loss of the term: once, by the first cell's lanes of block 0
    }
  }
  if (lc < cpb && cell < G2) {
    const uint4 ra = *reinterpret_cast<const uint4*>(cur + (size_t)cell * C + ch * 8);
    const T* av = reinterpret_cast<const T*>(&ra);
    float a[8];
    int sg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = to_f32<T>(av[i]); sg[i] = 0; }
    const int b = off[cell], e = b + ucnt[cell];
    int k = b;
    for (; k + 8 <= e; k += 8) {        // 8 distinct source rows in flight
      int id[8], mu[8];
      uint4 ro[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { id[j] = src[k + j]; mu[j] = mult[k + j]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) ro[j] = *reinterpret_cast<const uint4*>(orig + (size_t)id[j] * C + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const T* ov = reinterpret_cast<const T*>(&ro[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = to_f32<T>(ov[i]) - a[i];
          la += (double)mu[j] * (double)fabsf(d);
          sg[i] += mu[j] * ((d > 0.f) - (d < 0.f));
        }
      }
    }
    for (; k < e; ++k) {
      const uint4 ro = *reinterpret_cast<const uint4*>(orig + (size_t)src[k] * C + ch * 8);
      const int mu = mult[k];
      const T* ov = reinterpret_cast<const T*>(&ro);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = to_f32<T>(ov[i]) - a[i];
        la += (double)mu * (double)fabsf(d);
        sg[i] += mu * ((d > 0.f) - (d < 0.f));
      }
    }
    const bool bg = use_bg && bgflag[cell];
    TG o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float g = 0.f;
      if (b < e) g += -coef_fg * (float)sg[i];
      if (bg) g += -coef_bg * sgn[i];
      o[i] = from_f32<TG>(g * scale);
    }
    if (sizeof(TG) == 2) {
      *reinterpret_cast<uint4*>(grad + (size_t)cell * C + ch * 8) = *reinterpret_cast<uint4*>(o);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) grad[(size_t)cell * C + ch * 8 + i] = o[i];
    }
  }
  la = block_sum(la, sm);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = la;
  if (blockIdx.x == 0 && use_bg) {
    lb = block_sum(lb, sm);
    if (threadIdx.x == 0) bg_loss[0] = lb;
  }
}

}  // namespace dh

using namespace dh;

extern "C" int dh_energy_workspace_bytes(int C, int grid, int n_pairs, size_t* bytes) {
  DH_REQUIRE(C >= 1 && grid >= 1 && n_pairs >= 0 && bytes, "bad arguments");
  Arena a(nullptr, (size_t)-1);
  EnergyWs w;
  carve_energy(a, C, grid, n_pairs, w);
  *bytes = a.off + 256;
  return DH_OK;
}

extern "C" int dh_energy_fwd_bwd(const void* cur, const void* orig, int dtype, int C, int h_in, int w_in, int grid,
                                 const int32_t* pairs, int n_pairs, const int32_t* bg_both, int n_bg_both,
                                 const int32_t* bg_orig, int n_bg_orig, const int32_t* bg_trans, int n_bg_trans,
                                 float fg_w, float bg_w, int fg_patch, int bg_patch, int bg_mode, float grad_scale,
                                 float* loss_out, void* grad, int grad_dtype, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  DH_REQUIRE(cur && orig && loss_out && grad && workspace, "null pointer");
  DH_REQUIRE(C >= 1 && h_in >= 1 && w_in >= 1 && grid >= 1, "bad sizes");
  DH_REQUIRE(fg_patch >= 1 && (fg_patch & 1) && bg_patch >= 1 && (bg_patch & 1), "patch sizes must be odd");
  DH_REQUIRE(bg_mode == 0 || bg_mode == 1, "unknown background loss type");
  DH_REQUIRE(dtype >= 0 && dtype <= 2 && grad_dtype >= 0 && grad_dtype <= 2, "bad dtype");
  hipStream_t st = (hipStream_t)stream;
  const int G2 = grid * grid;
  Arena a(workspace, workspace_bytes);
  EnergyWs w;
  DH_REQUIRE(carve_energy(a, C, grid, n_pairs, w), "workspace too small");

  switch (dtype) {
    case DH_DTYPE_F16: launch_load<f16>(cur, w.cur, C, h_in, w_in, grid, st); launch_load<f16>(orig, w.orig, C, h_in, w_in, grid, st); break;
    case DH_DTYPE_BF16: launch_load<bf16>(cur, w.cur, C, h_in, w_in, grid, st); launch_load<bf16>(orig, w.orig, C, h_in, w_in, grid, st); break;
    default: launch_load<float>(cur, w.cur, C, h_in, w_in, grid, st); launch_load<float>(orig, w.orig, C, h_in, w_in, grid, st); break;
  }
  DH_CHECK_HIP(hipMemsetAsync(w.acc, 0, (size_t)G2 * C * sizeof(float), st));
  DH_CHECK_HIP(hipMemsetAsync(w.fg_part, 0, G2 * sizeof(double), st));
  DH_CHECK_HIP(hipMemsetAsync(w.bg_part, 0, G2 * sizeof(double), st));

  float fg_norm = 0.f, bg_norm = 0.f;
  int n_fg_part = 0, n_bg_part = 0;
  if (n_pairs > 0) {   // n == 0 would be NaN in the reference (mean over an empty axis): skipped here
    DH_REQUIRE(pairs, "null pairs");
    fg_norm = 1.f / ((float)C * (float)n_pairs);
    pair_term(w, pairs, n_pairs, C, grid, fg_patch, fg_w * fg_norm, w.fg_part, st);
    n_fg_part = G2;
  }
  if (bg_mode == 0) {
    if (n_bg_orig > 0 && n_bg_trans > 0) {
      DH_REQUIRE(bg_orig && bg_trans, "null bg list");
      hipLaunchKernelGGL(k_colsum, dim3(cdiv(C, 256), COLSUM_S), dim3(256), 0, st, w.orig, bg_orig, n_bg_orig, C,
                         COLSUM_S, w.part1);
      hipLaunchKernelGGL(k_colsum, dim3(cdiv(C, 256), COLSUM_S), dim3(256), 0, st, w.cur, bg_trans, n_bg_trans, C,
                         COLSUM_S, w.part2);
      hipLaunchKernelGGL(k_global_diff, dim3(cdiv(C, GD_BLOCK)), dim3(4 * GD_BLOCK), 0, st, w.part1, w.part2, COLSUM_S, C,
                         n_bg_orig, n_bg_trans, w.sgn, w.bg_part);
      bg_norm = 1.f / (float)C;
      hipLaunchKernelGGL(k_global_apply, dim3(n_bg_trans), dim3(256), 0, st, bg_trans, n_bg_trans, w.sgn, C,
                         bg_w * bg_norm / (float)n_bg_trans, w.acc);
      n_bg_part = cdiv(C, GD_BLOCK);
    }
  } else if (n_bg_both > 0) {
    DH_REQUIRE(bg_both, "null bg list");
    hipLaunchKernelGGL(k_identity_pairs, dim3(cdiv(n_bg_both, 256)), dim3(256), 0, st, bg_both, n_bg_both, w.idpairs);
    bg_norm = 1.f / ((float)C * (float)n_bg_both);
    pair_term(w, w.idpairs, n_bg_both, C, grid, bg_patch, bg_w * bg_norm, w.bg_part, st);
    n_bg_part = G2;
  }
  hipLaunchKernelGGL(k_final_loss, dim3(1), dim3(256), 0, st, w.fg_part, n_fg_part, fg_norm, w.bg_part, n_bg_part,
                     bg_norm, fg_w, bg_w, loss_out);
  switch (grad_dtype) {
    case DH_DTYPE_F16: launch_store<f16>(w.acc, grad, C, h_in, w_in, grid, grad_scale, st); break;
    case DH_DTYPE_BF16: launch_store<bf16>(w.acc, grad, C, h_in, w_in, grid, grad_scale, st); break;
    default: launch_store<float>(w.acc, grad, C, h_in, w_in, grid, grad_scale, st); break;
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_energy_plan_bytes(int grid, int n_pairs, size_t* bytes) {
  DH_REQUIRE(grid >= 1 && n_pairs >= 0 && bytes, "bad arguments");
  Arena a(nullptr, (size_t)-1);
  EnergyPlan p;
  carve_plan(a, grid, n_pairs, p);
  *bytes = a.off + 256;
  return DH_OK;
}

extern "C" int dh_energy_plan_build(const int32_t* pairs, int n_pairs, const int32_t* bg_trans, int n_bg_trans, int grid,
                                    void* plan, size_t plan_bytes, void* stream) {
  DH_REQUIRE(plan && grid >= 1 && n_pairs >= 0 && n_bg_trans >= 0, "bad arguments");
  DH_REQUIRE((n_pairs == 0 || pairs) && (n_bg_trans == 0 || bg_trans), "null list");
  hipStream_t st = (hipStream_t)stream;
  const int G2 = grid * grid;
  Arena a(plan, plan_bytes);
  EnergyPlan p;
  DH_REQUIRE(carve_plan(a, grid, n_pairs, p), "plan buffer too small");
  DH_CHECK_HIP(hipMemsetAsync(p.cnt, 0, (G2 + 1) * sizeof(int), st));
  DH_CHECK_HIP(hipMemsetAsync(p.bgflag, 0, G2, st));
  if (n_pairs > 0) hipLaunchKernelGGL(k_hist, dim3(cdiv(n_pairs, 256)), dim3(256), 0, st, pairs, n_pairs, p.cnt, p.w1, p.w2);
  hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, st, p.cnt, G2, p.off, p.cursor);
  if (n_pairs > 0) hipLaunchKernelGGL(k_fill, dim3(cdiv(n_pairs, 256)), dim3(256), 0, st, pairs, n_pairs, p.cursor, p.src);
  if (n_pairs > 0)
    hipLaunchKernelGGL(k_dedupe_sources, dim3(G2), dim3(256), (size_t)(G2 + 256) * sizeof(int), st, p.off, p.src, p.mult, p.cnt, G2);
  else
    DH_CHECK_HIP(hipMemsetAsync(p.cnt, 0, (G2 + 1) * sizeof(int), st));
  if (n_bg_trans > 0) hipLaunchKernelGGL(k_flag_cells, dim3(cdiv(n_bg_trans, 256)), dim3(256), 0, st, bg_trans, n_bg_trans, p.bgflag);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_energy_planned_workspace_bytes(int C, int grid, size_t* bytes) {
  DH_REQUIRE(C >= 1 && grid >= 1 && bytes, "bad arguments");
  Arena a(nullptr, (size_t)-1);
  PlannedWs w;
  carve_planned(a, C, grid, w);
  *bytes = a.off + 256;
  return DH_OK;
}

template <class T, class TG>
static void launch_energy_grad(const void* orig, const void* cur, const EnergyPlan& p, const PlannedWs& w, int C, int G2,
                               float coef_fg, float coef_bg, int use_bg, float scale, void* grad, int nblocks,
                               hipStream_t st, int n1, int n2) {
  hipLaunchKernelGGL((k_energy_grad<T, TG>), dim3(nblocks), dim3(256), 0, st, (const T*)orig, (const T*)cur, p.off, p.cnt,
                     p.src, p.mult, p.bgflag, w.partq, n1, n2, C, G2, coef_fg, coef_bg, use_bg, scale, (TG*)grad, w.fg_part, w.bg_part);
}

extern "C" int dh_energy_fwd_bwd_planned(const void* cur, const void* orig, int dtype, int C, int grid, const void* plan,
                                         size_t plan_bytes, int n_pairs, const int32_t* bg_orig, int n_bg_orig,
                                         const int32_t* bg_trans, int n_bg_trans, float fg_w, float bg_w, float grad_scale,
                                         float* loss_out, void* grad, int grad_dtype, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  DH_REQUIRE(cur && orig && grad && plan && workspace, "null pointer");
  DH_REQUIRE(dtype == DH_DTYPE_F16 || dtype == DH_DTYPE_BF16, "the planned path takes 16-bit activations");
  DH_REQUIRE(grad_dtype >= 0 && grad_dtype <= 2, "bad dtype");
  DH_REQUIRE(C >= 8 && C % 8 == 0 && C <= 2048 && grid >= 1 && n_pairs >= 0, "bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int G2 = grid * grid;
  Arena ap(const_cast<void*>(plan), plan_bytes);
  EnergyPlan p;
  DH_REQUIRE(carve_plan(ap, grid, n_pairs, p), "plan buffer too small");
  Arena aw(workspace, workspace_bytes);
  PlannedWs w;
  DH_REQUIRE(carve_planned(aw, C, grid, w), "workspace too small");

  const float fg_norm = n_pairs > 0 ? 1.f / ((float)C * (float)n_pairs) : 0.f;
  float bg_norm = 0.f, coef_bg = 0.f;
  int use_bg = 0;
  if (n_bg_orig > 0 && n_bg_trans > 0) {
    DH_REQUIRE(bg_orig && bg_trans, "null bg list");
    if (dtype == DH_DTYPE_F16)
      hipLaunchKernelGGL((k_colsum_q<f16>), dim3(cdiv(C, 64), 4, 2), dim3(8 * CQ_SL), 0, st, (const f16*)orig, bg_orig, n_bg_orig,
                         (const f16*)cur, bg_trans, n_bg_trans, C, w.partq);
    else
      hipLaunchKernelGGL((k_colsum_q<bf16>), dim3(cdiv(C, 64), 4, 2), dim3(8 * CQ_SL), 0, st, (const bf16*)orig, bg_orig, n_bg_orig,
                         (const bf16*)cur, bg_trans, n_bg_trans, C, w.partq);
    bg_norm = 1.f / (float)C;
    coef_bg = bg_w * bg_norm / (float)n_bg_trans;
    use_bg = 1;
  }
  const int cpb = 256 / (C / 8);
  const int nblocks = cdiv(G2, cpb);
  const float coef_fg = fg_w * fg_norm;
#define DH_EG(T_)                                                                                                          \
  do {                                                                                                                     \
    if (grad_dtype == DH_DTYPE_F16) launch_energy_grad<T_, f16>(orig, cur, p, w, C, G2, coef_fg, coef_bg, use_bg, grad_scale, grad, nblocks, st, n_bg_orig, n_bg_trans);        \
    else if (grad_dtype == DH_DTYPE_BF16) launch_energy_grad<T_, bf16>(orig, cur, p, w, C, G2, coef_fg, coef_bg, use_bg, grad_scale, grad, nblocks, st, n_bg_orig, n_bg_trans); \
    else launch_energy_grad<T_, float>(orig, cur, p, w, C, G2, coef_fg, coef_bg, use_bg, grad_scale, grad, nblocks, st, n_bg_orig, n_bg_trans);    \
  } while (0)
  if (dtype == DH_DTYPE_F16) DH_EG(f16);
  else DH_EG(bf16);
#undef DH_EG
  if (loss_out)
    hipLaunchKernelGGL(k_final_loss, dim3(1), dim3(256), 0, st, w.fg_part, n_pairs > 0 ? nblocks : 0, fg_norm, w.bg_part,
                       use_bg ? 1 : 0, bg_norm, fg_w, bg_w, loss_out);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
