// k_gemm_pp: the MFMA implicit GEMM of gemm.hip (same operands, same tiled / swizzled weight storage, same epilogue
// arithmetic) with an eight-wave PING-PONG main loop, for launches whose grid fills the chip (batched edits, the B = 16 CFG
// pass, 96 x 96 latents).  Round 5.
//
//   D[m][n] = sum_k A(m,k) W[n][k] (+ bias, + residual), A dense rows or the im2col view of a channels-last image
//   (reference: /root/reference/diffhandles/model/unet_2d_blocks.py:2216-2393 and the diffusers leaf blocks [ext]).
//
// Why another main loop.  k_gemm_dma runs all of a workgroup's waves in lockstep: every wave reads its fragments, multiplies,
// issues its share of the next tile's LDS-DMA and meets the others at ONE barrier per K tile -- so the two waves of a SIMD want
// the matrix pipe at the same time and the LDS / the texture path at the same time, and each K tile starts with both of them
// waiting for fragments.  Here the eight waves are two GROUPS of four (waves w and w + 4 share a SIMD) that run the same
// code one barrier apart:
//
//      interval      group 0                          group 1
//      4t            LOAD(t, k-step 0)                MFMA(t-1, k-step 1)
//      4t + 1        MFMA(t, 0)                       LOAD(t, 0)
//      4t + 2        LOAD(t, 1)                       MFMA(t, 0)
//      4t + 3        MFMA(t, 1)                       LOAD(t, 1)
//
//   LOAD = ds_read_b128 of the fragments of one 32-deep k-step into registers + this wave's share of the LDS-DMA of tile
//   t + NST - 1; MFMA = the TM x TN v_mfma_f32_16x16x32 of that k-step, at raised priority.  Every interval ends with one
//   s_barrier of all eight waves.  While one wave of a SIMD multiplies, its partner loads: the matrix pipe never waits for a
//   fragment, fragments are single-buffered (the wave that loads does not multiply), and the DMA queue is never drained:
//   each wave waits with a COUNTED vmcnt for its own pieces of tile t + 1 once per tile, NST - 2 tiles stay in flight.
//
// Tiles: 256 x 160 (N = 320, 640, 960: two / four / six column tiles), 256 x 128, 128 x 160, 128 x 128; wave layout 4 (M) x 2 (N),
// the column half = the group; a wave owns (BM / 4) x (BN / 2) outputs = TM x TN blocks of 16 x 16 (operands swapped as in
// k_gemm_dma: a lane owns one output row and four consecutive columns per block).
//
// Staging: `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor) -- the per-lane source offset is one 32-bit
// register per piece, computed once; the K cursor is a scalar offset; rows outside the matrix and taps outside the image
// are an out-of-range offset, which the hardware turns into zeros in LDS (no zero page, no 64-bit address arithmetic in
// the loop).  Stage image, source-side swizzle and weight layout are exactly k_gemm_dma's (unet_kernels.h wt_index).
#include <hip/hip_ext.h>

#include "gemm_k.h"

namespace dh {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned lane_u2p __attribute__((ext_vector_type(2)));

template <class T> struct Mfma16;
template <> struct Mfma16<f16> {
  static __device__ __forceinline__ v4f run(uint4 a, uint4 b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  }
};
template <> struct Mfma16<bf16> {
  static __device__ __forceinline__ v4f run(uint4 a, uint4 b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8b, a), __builtin_bit_cast(v8b, b), c, 0, 0, 0);
  }
};

constexpr unsigned PP_OOB = 0x80000000u;     // an offset no descriptor of < 2 GiB contains: the DMA writes zeros

// one 1-KiB piece: lane i moves 16 bytes from (descriptor base + voff + soff) to LDS byte lds_dst + 16 i
__device__ __forceinline__ void pp_dma(unsigned voff, v4i rsrc, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most `tiles` tiles of P pieces each may stay outstanding (tiles < CAP)
template <int P, int CAP> __device__ __forceinline__ void pp_wait_tiles(int tiles) {
  if constexpr (CAP <= 1) { pp_wait_vm<0>(); }
  else {
    if (tiles >= CAP - 1) pp_wait_vm<P * (CAP - 1)>(); else pp_wait_tiles<P, CAP - 1>(tiles);
  }
}
__device__ __forceinline__ void pp_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

enum { PP_DENSE = 0, PP_CONV_S1 = 1, PP_GENERIC = 2 };

template <class T, int BM, int BN, int MODE, int NST>
__global__ void __launch_bounds__(512) k_gemm_pp(const GemmK p) {
  constexpr int RPW = BM / 4, CPW = BN / 2;                 // rows / columns of the output tile per wave
  constexpr int TM = RPW / 16, TN = CPW / 16;
  static_assert(RPW % 16 == 0 && CPW % 16 == 0, "a wave owns whole 16 x 16 blocks");
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NPA = BM / 64;                              // A pieces per wave and tile (BM / 8 pieces over eight waves)
  constexpr int WPIECES = BN / 8;
  static_assert(WPIECES % 8 == 0 || WPIECES % 8 == 4, "the W pieces split evenly inside each wave group");
  constexpr int NPB0 = (WPIECES + 7) / 8, NPB1 = WPIECES / 8;   // W pieces per wave: waves 0-3 / waves 4-7
  constexpr int NP0 = NPA + NPB0, NP1 = NPA + NPB1;
  static_assert(NST >= 3 && NST * STAGE <= 160 * 1024, "ring does not fit the LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wm = wave & 3;                   // group = column half; waves w and w + 4 share a SIMD
  const int l15 = lane & 15, quad = lane >> 4;

  // ---- which output tile: XCD-aware order (block b runs on XCD b % 8; consecutive work items share an operand panel) ----
  int m0, n0, zsplit;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    int rt, ct;
    if (p.pp_order == 0) { ct = idx % p.pp_tn; const int rest = idx / p.pp_tn; rt = rest % p.pp_tm; zsplit = rest / p.pp_tm; }
    else { rt = idx % p.pp_tm; const int rest = idx / p.pp_tm; ct = rest % p.pp_tn; zsplit = rest / p.pp_tn; }
    m0 = rt * BM; n0 = ct * BN;
  }
  int kbeg = zsplit * p.k_per_split;
  int kend = kbeg + p.k_per_split;
  if (kend > p.K) kend = p.K;
  const int nt = (kend - kbeg) >> 6;

  // ---- descriptors (scalar) ------------------------------------------------------------------------------------------------
  // conv: the base is moved back by one image row + one pixel so that the tap offset (ky Win + kx) lda is never negative
  const unsigned a_bias = MODE == PP_CONV_S1 ? (unsigned)((p.Win + 1) * p.lda * 2) : 0u;
  const size_t a_base = (size_t)p.A - a_bias;
  v4i ra, rw;
  ra[0] = (int)(unsigned)a_base; ra[1] = (int)((a_base >> 32) & 0xffff); ra[2] = (int)(p.pp_a_bytes + a_bias); ra[3] = 0x00020000;
  rw[0] = (int)(unsigned)(size_t)p.W; rw[1] = (int)(((size_t)p.W >> 32) & 0xffff); rw[2] = (int)p.pp_w_bytes; rw[3] = 0x00020000;

  // ---- per-lane source offsets of this wave's pieces -------------------------------------------------------------------------
  const int prow = lane >> 3;                               // row of this lane inside a piece
  const int lchunk = (lane & 7) ^ (4 * (wave & 1) + (prow >> 1));      // source-side swizzle: chunk c of row r at c ^ ((r >> 1) & 7)
  unsigned a_voff[NPA];                                     // dense / conv_s1: byte offset of (row, chunk); generic: batch base pixel
  unsigned a_taps[NPA];
  int a_oy[NPA], a_ox[NPA];
  bool a_ok[NPA];
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
    const int m = m0 + 8 * (wave + 8 * j) + prow;
    a_ok[j] = m < p.M;
    a_taps[j] = 0; a_oy[j] = 0; a_ox[j] = 0;
    if (MODE == PP_DENSE) {
      a_voff[j] = a_ok[j] ? (unsigned)m * (unsigned)(p.lda * 2) + lchunk * 16 : PP_OOB;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = m / hw, r = m - b * hw;
      a_oy[j] = r / p.Wout;
      a_ox[j] = r - a_oy[j] * p.Wout;
      if (MODE == PP_CONV_S1) {
        a_voff[j] = (unsigned)((b * p.Hin + a_oy[j]) * p.Win + a_ox[j]) * (unsigned)(p.lda * 2) + lchunk * 16;
        if (a_ok[j]) {
#pragma unroll
          for (int t = 0; t < 9; ++t)
            if ((unsigned)(a_oy[j] + t / 3 - 1) < (unsigned)p.Hin && (unsigned)(a_ox[j] + t % 3 - 1) < (unsigned)p.Win) a_taps[j] |= 1u << t;
        }
      } else {
        a_voff[j] = (unsigned)(b * p.Hin * p.Win);
      }
    }
  }
  const unsigned w_voff = lane * 16;
  const int KT = p.K >> 6;
  unsigned w_soff[NPB0];                                     // scalar: byte offset of the piece's rows in the tiled weights (K tile 0)
#pragma unroll
  for (int j = 0; j < NPB0; ++j) {
    const int n = n0 + 8 * (wave + 8 * j);
    w_soff[j] = (unsigned)(((n >> 6) * KT) * 8192 + (n & 63) * 128) + (unsigned)(kbeg >> 6) * 8192u;
  }
  int tap = 0, c0 = 0;
  if (MODE != PP_DENSE) { const int kt0 = kbeg >> 6, ch = kt0 / 9; tap = kt0 - ch * 9; c0 = ch * 64; }    // conv_k_index order
  int kt_issue = 0;                                          // next tile to issue

  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  // one piece q (0 .. NPA - 1: A, then W) of the tile at the issue cursor into ring slot `stage`
  auto issue_piece = [&](int stage, int q) {
    const unsigned sbase = lds0 + stage * STAGE + wave * 1024;
    if (q < NPA) {
      const int j = q;
      if (MODE == PP_DENSE) {
        pp_dma(a_voff[j], ra, (unsigned)(kbeg + kt_issue * 64) * 2u, sbase + j * 8192);
      } else if (MODE == PP_CONV_S1) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned soff = (unsigned)((ky * p.Win + kx) * (int)p.lda + c0) * 2u;
        const bool ok = (a_taps[j] >> tap) & 1u;
        pp_dma(ok ? a_voff[j] + a_bias : PP_OOB, ra, soff, sbase + j * 8192);
      } else {
        const int ky = tap / 3, kx = tap - ky * 3;
        bool ok = a_ok[j];
        int sy, sx;
        if (p.mode == A_CONV3) {
          const int iy = a_oy[j] * p.stride + ky - p.pad, ix = a_ox[j] * p.stride + kx - p.pad;
          ok = ok && iy >= 0 && ix >= 0 && iy < (p.Hin << p.up) && ix < (p.Win << p.up);
          sy = iy >> p.up; sx = ix >> p.up;
        } else {
          const int ty = a_oy[j] + ky - 1, tx = a_ox[j] + kx - 1;
          ok = ok && ty >= 0 && tx >= 0 && !(ty & 1) && !(tx & 1) && (ty >> 1) < p.Hin && (tx >> 1) < p.Win;
          sy = ty >> 1; sx = tx >> 1;
        }
        const unsigned off = (a_voff[j] + (unsigned)(sy * p.Win + sx)) * (unsigned)(p.lda * 2) + lchunk * 16;
        pp_dma(ok ? off : PP_OOB, ra, (unsigned)c0 * 2u, sbase + j * 8192);
      }
    } else {
      const int j = q - NPA;
      if (j < NPB1 || g == 0) pp_dma(w_voff, rw, w_soff[j] + (unsigned)kt_issue * 8192u, sbase + BM * 128 + j * 8192);
    }
  };
  auto next_tile = [&]() {
    ++kt_issue;
    if (MODE != PP_DENSE) { if (++tap == 9) { tap = 0; c0 += 64; } }
  };

  v4f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: tiles 0 .. NST - 2 in flight ---------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) {
    if (s < nt) {
#pragma unroll
      for (int q = 0; q < NP0; ++q) issue_piece(s, q);
      next_tile();
    }
  }
  // fragment addresses: per lane (row l15 of a 16-row block, chunk 4 s + quad of the 32-deep k-step s, swizzled); the swizzle
  // term ((row >> 1) & 7) only depends on l15 because every block starts on a multiple of 16 rows
  const unsigned fsw = (unsigned)((l15 >> 1) & 7);
  const unsigned fo0 = ((unsigned)quad ^ fsw) << 4, fo1 = ((unsigned)(4 + quad) ^ fsw) << 4;
  const unsigned fa_base = (unsigned)((wm * RPW + l15) * 128);
  const unsigned fb_base = (unsigned)(BM * 128 + (g * CPW + l15) * 128);

  // tile 0 has landed once at most the later tiles of the prologue are outstanding
  {
    const int later = (nt < NST - 1 ? nt : NST - 1) - 1;
    if (g == 0) pp_wait_tiles<NP0, NST - 1>(later); else pp_wait_tiles<NP1, NST - 1>(later);
  }
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();               // the stagger: group 1 runs one interval behind group 0

  int cur = 0;                                              // ring slot of tile t
  uint4 fw[TN], fx[TM];
  for (int t = 0; t < nt; ++t) {
    const bool more = kt_issue < nt;
    const int nstage = cur == 0 ? NST - 1 : cur - 1;        // slot of tile t + NST - 1 = the slot tile t - 1 has left
    const unsigned char* st = smem + cur * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // ---- LOAD(t, s) ----
      {
        const unsigned char* pb = st + fb_base + (s ? fo1 : fo0);
        const unsigned char* pa = st + fa_base + (s ? fo1 : fo0);
#pragma unroll
        for (int j = 0; j < TN; ++j) fw[j] = *reinterpret_cast<const uint4*>(pb + j * 2048);
#pragma unroll
        for (int i = 0; i < TM; ++i) fx[i] = *reinterpret_cast<const uint4*>(pa + i * 2048);
      }
      if (more) {
#pragma unroll
        for (int q = s * NP0 / 2; q < (s + 1) * NP0 / 2; ++q) issue_piece(nstage, q);
      }
      if (s == 1) {
        if (more) next_tile();
        // this wave's pieces of tile t + 1 have landed when only the tiles after it are outstanding
        const int later = kt_issue - (t + 2);             // tiles t + 2 .. kt_issue - 1
        if (g == 0) pp_wait_tiles<NP0, NST - 1>(later); else pp_wait_tiles<NP1, NST - 1>(later);
      }
      pp_wait_lds();
      __builtin_amdgcn_s_barrier();
      // ---- MFMA(t, s) ----
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mfma16<T>::run(fw[j], fx[i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    cur = cur + 1 == NST ? 0 : cur + 1;
  }
  if (g == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue --------------------------------------------------------------------------------------------------------------
  // acc[i][j][r] = D[m = m0 + wm RPW + 16 i + l15][n = n0 + g CPW + 16 j + 4 quad + r]
  const int nb = n0 + g * CPW;
  if (p.splits > 1) {
    float* part = p.partial + (size_t)zsplit * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * RPW + 16 * i + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j)
        *reinterpret_cast<float4*>(part + (size_t)m * p.N + nb + 16 * j + 4 * quad) =
            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    return;
  }
  typedef T T4 __attribute__((ext_vector_type(4)));
  typedef T T8 __attribute__((ext_vector_type(8)));
  const bool hb = p.bias != nullptr, hr = p.R != nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * RPW + 16 * i + l15;
    const bool mok = m < p.M;
    T* crow = reinterpret_cast<T*>(p.C) + (size_t)m * p.ldc;
    const T* rrow = reinterpret_cast<const T*>(p.R) + (size_t)m * p.ldr;
#pragma unroll
    for (int j = 0; j + 1 < TN; j += 2) {
      // blocks j, j + 1: the lane pairs (l, l ^ 16) trade four columns so that every lane holds EIGHT consecutive columns
      // (one 16-byte store): lanes of an even quad keep block j, lanes of an odd quad get block j + 1
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const lane_u2p x = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, acc[i][j][r]),
                                                           __builtin_bit_cast(unsigned, acc[i][j + 1][r]), false, false);
        v[r] = __builtin_bit_cast(float, x[0]);
        v[4 + r] = __builtin_bit_cast(float, x[1]);
      }
      const int n = nb + 16 * (j + (quad & 1)) + 8 * (quad >> 1);
      if (!mok) continue;
      if (hb) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (hr) {
        const T8 rv = __builtin_bit_cast(T8, *reinterpret_cast<const uint4*>(rrow + n));
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] += to_f32<T>(rv[c]);
      }
      T8 o;
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = from_f32<T>(v[c]);
      *reinterpret_cast<uint4*>(crow + n) = __builtin_bit_cast(uint4, o);
    }
    if constexpr (TN & 1) {
      constexpr int j = TN - 1;
      const int n = nb + 16 * j + 4 * quad;
      if (mok) {
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (hb) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
        if (hr) {
          const T4 rv = __builtin_bit_cast(T4, *reinterpret_cast<const uint2*>(rrow + n));
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += to_f32<T>(rv[c]);
        }
        T4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = from_f32<T>(v[c]);
        *reinterpret_cast<uint2*>(crow + n) = __builtin_bit_cast(uint2, o);
      }
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
static int pp_mode(const GemmK& k) {
  if (k.mode == A_DENSE) return PP_DENSE;
  if (k.mode == A_CONV3 && k.stride == 1 && k.up == 0 && k.pad == 1) return PP_CONV_S1;
  return PP_GENERIC;
}

// bytes the A descriptor must cover
static size_t pp_a_bytes(const GemmK& k) {
  if (k.mode == A_DENSE) return ((size_t)(k.M - 1) * k.lda + k.K) * 2;
  const size_t B = (size_t)k.M / ((size_t)k.Hout * k.Wout);
  return B * k.Hin * k.Win * (size_t)k.lda * 2;
}

bool gemm_pp_plan(const GemmK& k, size_t partial_elems, int force, PpPlan* plan) {
  if (force == 1) return false;
  // what the kernel carries: the plain epilogue (bias, residual), 16-byte aligned rows, descriptors below 2 GiB
  if (k.ln_s || k.glu_y || k.glub_x || k.rowvec || k.act_silu) return false;
  if (k.K % 64 || k.M <= 0 || !k.C) return false;
  if (k.N % 160 && k.N % 128) return false;
  if (k.ldc % 8 || ((size_t)k.C & 15) || k.lda % 8 || ((size_t)k.A & 15) || ((size_t)k.W & 15)) return false;
  if (k.R && (k.ldr % 8 || ((size_t)k.R & 15))) return false;
  if (k.mode != A_DENSE && (k.Hout <= 0 || k.Wout <= 0 || k.M % (k.Hout * k.Wout) || k.Cin % 64 || k.K != 9 * k.Cin)) return false;
  const size_t ab = pp_a_bytes(k) + (size_t)(k.Win + 1) * k.lda * 2, wb = (size_t)k.N * k.K * 2;
  if (ab >= 0x7ff00000ull || wb >= 0x7ff00000ull) return false;
  const int ktiles = k.K / 64;
  // tile: the 160-column tile where it divides N (N = 320, 640, 960, 1280 ...), else 128 columns
  const int bn = k.N % 160 == 0 ? 160 : 128;
  const int tn = k.N / bn;
  const long t256 = (long)cdiv(k.M, 256) * tn, t128 = (long)cdiv(k.M, 128) * tn;
  int bm = 0, splits = 1;
  if (t256 >= 224) bm = 256;                                 // one round or more of 256-row tiles
  else if (t128 >= 224) bm = 128;                            // (M = 8192, N = 640 at batch 8: 64 x 4)
  else if (k.partial && ktiles >= 32 && t256 >= 16 && k.M >= 1024) {
    // long K loops on fewer tiles than CUs: split K over workgroups (f32 slabs + the reduce kernels of gemm.hip)
    bm = 256;
    splits = (int)(256 / t256);
    if (splits > ktiles / 8) splits = ktiles / 8;
    if (splits > 8) splits = 8;
    const size_t fit = partial_elems / ((size_t)k.M * k.N);
    if ((size_t)splits > fit) splits = (int)fit;
    if (splits < 2) bm = 0;
  }
  if (force == 2 && bm == 0) { bm = k.M > 128 ? 256 : 128; splits = 1; }      // test hook: any shape the kernel can carry
  if (bm == 0) return false;
  plan->bm = bm; plan->bn = bn; plan->splits = splits;
  return true;
}

template <class T, int BM, int BN, int NST>
static void pp_launch_tile(int mode, dim3 grid, hipStream_t st, const GemmK& k, hipEvent_t e0, hipEvent_t e1) {
#define DH_PP_LAUNCH(KERNEL)                                                                      \
  do {                                                                                            \
    if (e0) hipExtLaunchKernelGGL(KERNEL, grid, dim3(512), 0, st, e0, e1, 0, k);                  \
    else hipLaunchKernelGGL(KERNEL, grid, dim3(512), 0, st, k);                                   \
  } while (0)
  if (mode == PP_DENSE) DH_PP_LAUNCH((k_gemm_pp<T, BM, BN, PP_DENSE, NST>));
  else if (mode == PP_CONV_S1) DH_PP_LAUNCH((k_gemm_pp<T, BM, BN, PP_CONV_S1, NST>));
  else DH_PP_LAUNCH((k_gemm_pp<T, BM, BN, PP_GENERIC, NST>));
#undef DH_PP_LAUNCH
}

template <class T>
static void pp_launch(const GemmK& k, const PpPlan& plan, int mode, dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  if (plan.bm == 256 && plan.bn == 160) pp_launch_tile<T, 256, 160, 3>(mode, grid, st, k, e0, e1);
  else if (plan.bm == 256) pp_launch_tile<T, 256, 128, 3>(mode, grid, st, k, e0, e1);
  else if (plan.bn == 160) pp_launch_tile<T, 128, 160, 4>(mode, grid, st, k, e0, e1);
  else pp_launch_tile<T, 128, 128, 4>(mode, grid, st, k, e0, e1);
}

void launch_gemm_pp(int dtype, const GemmK& kin, const PpPlan& plan, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  GemmK k = kin;
  const int ktiles = k.K / 64;
  const int tiles_per_split = cdiv(ktiles, plan.splits);
  k.splits = cdiv(ktiles, tiles_per_split);
  k.k_per_split = tiles_per_split * 64;
  k.pp_tm = cdiv(k.M, plan.bm);
  k.pp_tn = k.N / plan.bn;
  k.pp_a_bytes = (unsigned)pp_a_bytes(k);
  k.pp_w_bytes = (unsigned)((size_t)k.N * k.K * 2);
  // which operand panel consecutive work items (= one XCD's L2) share: the column tiles of a row tile share its A rows, the
  // row tiles of a column tile share its W rows; take the order with the smaller traffic estimate (the other operand is
  // then read once per XCD)
  const double a_tot = (double)k.pp_a_bytes, w_tot = (double)k.pp_w_bytes;
  const double col_fastest = a_tot + 8.0 * w_tot;
  const double row_fastest = w_tot + (double)(k.pp_tn * k.splits < 8 ? k.pp_tn * k.splits : 8) * a_tot;
  k.pp_order = row_fastest < col_fastest ? 1 : 0;
  dim3 grid((unsigned)(k.pp_tm * k.pp_tn * k.splits));
  const int mode = pp_mode(k);
  if (dtype == DH_DTYPE_F16) pp_launch<f16>(k, plan, mode, grid, st, e0, e1);
  else pp_launch<bf16>(k, plan, mode, grid, st, e0, e1);
}

}  // namespace dh
