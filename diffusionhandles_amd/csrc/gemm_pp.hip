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
//   (The table shows the 32-deep form, VAR without PPV_K64; the SHIPPED variants run a whole 64-deep K tile per segment --
//   LOAD(t) then MFMA(t), two intervals per K tile -- and issue the last one / three pieces of a tile between the MFMAs.)
//
// Around the loop (all measured in DESIGN.md section 4, "What a short-K launch waits for"): the workgroups are PERSISTENT (at most
// 256, each walks the work items blockIdx.x, + gridDim.x, ... in XCD order) so that a tile's stores drain under the next tile's
// prologue and loop; the residual tile and the GEGLU backward's saved pre-activations are fetched behind the prologue's DMA; the
// per-column vectors are read once in front of the first store; the plain epilogue stages a wave's outputs in its own rows of
// the idle ring and stores whole row segments (in the MFMA layout a store instruction is 64 scattered 16-byte requests).
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

#include <type_traits>
#include <utility>

#include "gemm_k.h"

namespace dh {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned lane_u2p __attribute__((ext_vector_type(2)));

template <class T> struct Mfma16;
// (fragments are loaded AS the MFMA operand type: a uint4 that is bit-cast at the MFMA gets split into dwords and re-packed with
//  v_pk_mov / v_mov pairs between the MFMAs)
template <> struct Mfma16<f16> {
  typedef v8h frag;
  static __device__ __forceinline__ v4f run(frag a, frag b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<bf16> {
  typedef v8b frag;
  static __device__ __forceinline__ v4f run(frag a, frag b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

constexpr unsigned PP_OOB = 0x80000000u;     // an offset no descriptor of < 2 GiB contains: the DMA writes zeros

// LDS-DMA pieces.  M0 holds the LDS byte address of a wave's first piece of a group (its A pieces / its W pieces of a stage);
// piece j of the group lies j KiB further and is addressed by the instruction's immediate offset, which the hardware adds
// to BOTH the LDS and the memory address -- so the per-piece source offset carries -1024 j.  M0 is not saved: nothing else in
// this kernel uses it (LDS instructions need no M0 on gfx9+; checked in the ISA by tools/check_isa.py).
__device__ __forceinline__ void pp_set_m0(unsigned lds_dst) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_dst) : "memory"); }
template <int IMM> __device__ __forceinline__ void pp_dma(unsigned voff, v4i rsrc, unsigned soff) {
  // The scalar offset is copied by an s_mov INSIDE the statement: (1) MUBUF's soffset must be an SGPR and the "s" constraint
  // alone lets a literal or a vector register through; (2) when the compiler produces the value with v_readfirstlane (a VALU
  // write of an SGPR) a VMEM instruction must not read it for five wait states, and hipcc pads nothing for the operands of an
  // asm statement -- measured: the first piece behind such a readfirstlane fetched from the PREVIOUS tile's offset.  An SALU
  // copy has no such hazard towards VMEM.
  unsigned tmp;
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 %0, %3\n\tbuffer_load_dwordx4 %1, %2, %0 offen offset:%4 lds"
               : "=&s"(tmp) : "v"(voff), "s"(rsrc), "s"(soff), "n"(IMM) : "memory");
}
template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most n (0 .. HI; more is clamped, which only waits longer) vector-memory operations of this wave may stay outstanding
template <int LO, int HI> __device__ __forceinline__ void pp_wait_vm_dyn(int n) {
  if constexpr (LO == HI) { pp_wait_vm<LO>(); }
  else {
    constexpr int MID = (LO + HI + 1) / 2;
    if (n >= MID) pp_wait_vm_dyn<MID, HI>(n); else pp_wait_vm_dyn<LO, MID - 1>(n);
  }
}
__device__ __forceinline__ void pp_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

enum { PP_DENSE = 0, PP_CONV_S1 = 1, PP_GENERIC = 2 };
// variants (measurement / A-B: tools/bench_gemm_pp.py; the shipped choice is PP_SHIP)
enum { PPV_STAMP = 1,        // s_memtime stamps of waves 0 and 4 of workgroup 0 into p.pp_ts (timeline of the segments)
       PPV_DMA_IN_MFMA = 2,  // the LDS-DMA pieces are issued between the MFMAs of the multiply segment instead of in the load segment
       PPV_NO_RPRE = 8,      // the residual tile is loaded in the epilogue, chunk by chunk, instead of being prefetched behind the prologue
       PPV_K64 = 4,          // a segment covers a whole 64-deep K tile (two k-steps): half the barriers, twice the fragment registers
       PPV_NDIM_SHIFT = 4,   // bits 4-6: only the LAST n piece slots of a tile (the W pieces first) go out inside the multiply segment
       PPV_SPLIT_ORDER = 128 }; // the odd waves of a group issue their DMA pieces BEFORE they read their fragments (the even ones after):
                             // the LDS and the texture path are busy side by side through the load segment instead of one after the other

// GLU = 1: the tile is the (paired-layout) pre-activation of a GEGLU -- 16-column block 2q holds the VALUE columns of outputs
// 16 q' .. 16 q' + 15 and block 2q + 1 their GATE columns (unet_kernels.h glu_col), so a lane owns value and gate of the same four
// outputs: the epilogue also writes h * gelu(gate) (and the pre-activations only when p.C is set).  GLU = 2: the tile is dy of a
// GEGLU: the epilogue reads the saved pre-activations of its outputs and writes d_value | d_gate instead of dy.  As in k_gemm_dma
// (reference model/attention.py:345-400; diffusers GEGLU [ext]).
template <class T, int BM, int BN, int MODE, int NST, int VAR, int GLU = 0>
__global__ void __launch_bounds__(512) k_gemm_pp(const GemmK p) {
  static_assert(GLU == 0 || (MODE == 0 && (BN / 2) % 32 == 0), "the GEGLU epilogues: dense, whole (value, gate) block pairs per wave");
  constexpr int RPW = BM / 4, CPW = BN / 2;                 // rows / columns of the output tile per wave
  constexpr int TM = RPW / 16, TN = CPW / 16;
  static_assert(RPW % 16 == 0 && CPW % 16 == 0, "a wave owns whole 16 x 16 blocks");
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NPA = BM / 64;                              // A pieces per wave and tile (BM / 8 pieces over eight waves)
  constexpr int WPIECES = BN / 8;
  static_assert(WPIECES % 8 == 0 || WPIECES % 8 == 4, "the W pieces split evenly inside each wave group");
  constexpr int NPB0 = (WPIECES + 7) / 8, NPB1 = WPIECES / 8;   // W pieces per wave: waves 0-3 / waves 4-7
  constexpr int NP0 = NPA + NPB0;                           // piece slots per wave and tile (the last W slot is empty for waves 4-7 when NPB1 < NPB0)
  static_assert(NPA <= 4 && NPB0 <= 4, "immediate offsets reach 3 KiB");
  // (a two-stage ring of the 128-row tiles -- 64 / 73 KB, <= 128 registers, two workgroups per CU -- was measured in round 5 and
  //  dropped: -0..12 % on some short-K shapes, +20 % on others, profiles/r05_ab_pp_two_per_cu.txt; NST >= 3 is also what lets a
  //  group issue pieces of tile t + NST - 1 AFTER its wait for tile t + 1)
  static_assert(NST >= 3 && NST * STAGE <= 160 * 1024, "ring does not fit the LDS");
  constexpr int KS = (VAR & PPV_K64) ? 2 : 1;               // 32-deep k-steps per segment
  constexpr int SEG = 2 / KS;                               // segments per K tile
  constexpr bool STAMP = (VAR & PPV_STAMP) != 0;
  // piece slots [0, XL) of a tile are issued in the load segments, [XL, NP0) between the MFMAs of the multiply segments: the
  // two segments of an interval run side by side on a SIMD, the interval lasts as long as the longer one
  constexpr int NDIM = (VAR & PPV_DMA_IN_MFMA) ? NP0 : ((VAR >> PPV_NDIM_SHIFT) & 7);
  constexpr int XL = NP0 - (NDIM < NP0 ? NDIM : NP0);
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wm = wave & 3;                   // group = column half; waves w and w + 4 share a SIMD
  const int l15 = lane & 15, quad = lane >> 4;
  const int npw = g == 0 ? NP0 : NPA + NPB1;                // DMA pieces this wave issues per tile

  // ---- which output tile: XCD-aware order (block b runs on XCD b % 8; consecutive work items share an operand panel) ----
  unsigned long long* ts = nullptr;
  int tsn = 0;
  if constexpr (STAMP) { if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) ts = p.pp_ts + (wave >> 2) * 512; }
  auto stamp = [&]() { if constexpr (STAMP) { if (ts && tsn < 512) ts[tsn++] = __builtin_amdgcn_s_memtime(); } };
  // PERSISTENT form: the grid is at most one workgroup per slot of the chip (gemm launch code); a workgroup walks the work items
  // vb = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8 whenever it is smaller than the item count, so an item keeps
  // the XCD of its first-round position).  The stores of one tile's epilogue are in flight while the next tile's prologue and K loop
  // run -- with one launch-sized round per tile every CU computed, then every CU stored, and HBM idled in between.
  for (int vb = blockIdx.x; vb < p.pp_nwork; vb += gridDim.x) {
  if (vb != (int)blockIdx.x) __builtin_amdgcn_s_barrier();               // every wave has read its staged outputs back (epilogue)
  stamp();
  int m0, n0, zsplit;
  {
    const int nwg = p.pp_nwork, bid = vb;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    int rt, ct;
    if ((p.pp_order & 1) == 0) { ct = idx % p.pp_tn; const int rest = idx / p.pp_tn; rt = rest % p.pp_tm; zsplit = rest / p.pp_tm; }
    else { rt = idx % p.pp_tm; const int rest = idx / p.pp_tm; ct = rest % p.pp_tn; zsplit = rest / p.pp_tn; }
    m0 = rt * BM; n0 = ct * BN;
  }
  int kbeg = zsplit * p.k_per_split;
  int kend = kbeg + p.k_per_split;
  if (kend > p.K) kend = p.K;
#ifdef DH_PP_VARIANTS
  // timing ablations (measurement build only): 0x200 = one K tile instead of all (prologue + epilogue + launch), 0x100 = no epilogue
  const int nt = (p.pp_order & 0x200) ? 1 : (kend - kbeg) >> 6;
#else
  const int nt = (kend - kbeg) >> 6;
#endif

  // ---- descriptors (scalar) ------------------------------------------------------------------------------------------------
  // The A base is moved back by a_bias bytes and every source offset carries +a_bias: the stride-1 convolution's tap offset
  // (ky Win + kx) lda is then never negative (a_bias = one image row + one pixel), and no offset goes below zero when the
  // immediate's -1024 j is folded in (generic gather: 4 KiB).
  const unsigned a_bias = MODE == PP_CONV_S1 ? (unsigned)((p.Win + 1) * p.lda * 2) : (MODE == PP_GENERIC ? 4096u : 0u);
  const size_t a_base = (size_t)p.A - a_bias;
  v4i ra, rw;
  ra[0] = (int)(unsigned)a_base; ra[1] = (int)((a_base >> 32) & 0xffff); ra[2] = (int)(p.pp_a_bytes + a_bias); ra[3] = 0x00020000;
  rw[0] = (int)(unsigned)(size_t)p.W; rw[1] = (int)(((size_t)p.W >> 32) & 0xffff); rw[2] = (int)p.pp_w_bytes; rw[3] = 0x00020000;

  // ---- per-lane source offsets of this wave's pieces: wave w stages A rows [w BM / 8, (w + 1) BM / 8) and a contiguous run of
  // W rows, so that its pieces of a group are consecutive KiB of the stage ------------------------------------------------------
  const int prow = lane >> 3;                               // row of this lane inside a piece
  const int qa0 = wave * NPA;                               // first A piece of this wave
  const int qb0 = g == 0 ? wave * NPB0 : 4 * NPB0 + (wave - 4) * NPB1;     // first W piece
  unsigned a_voff[NPA];                                     // dense / conv_s1: byte offset of (row, chunk) - 1024 j; generic: batch base pixel
  unsigned a_lch[NPA];                                      // generic: the swizzled chunk offset - 1024 j + a_bias
  unsigned a_taps[NPA];
  int a_oy[NPA], a_ox[NPA];
  bool a_ok[NPA];
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
    const int m = m0 + 8 * (qa0 + j) + prow;
    // source-side swizzle: chunk c of row r is stored at c ^ ((r >> 1) & 7); r = 8 (qa0 + j) + prow inside the tile
    const int lchunk = (lane & 7) ^ ((4 * ((qa0 + j) & 1) + (prow >> 1)) & 7);
    a_ok[j] = m < p.M;
    a_taps[j] = 0; a_oy[j] = 0; a_ox[j] = 0; a_lch[j] = 0;
    if (MODE == PP_DENSE) {
      a_voff[j] = (a_ok[j] ? (unsigned)m * (unsigned)(p.lda * 2) + lchunk * 16 : PP_OOB) - 1024u * j;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = m / hw, r = m - b * hw;
      a_oy[j] = r / p.Wout;
      a_ox[j] = r - a_oy[j] * p.Wout;
      if (MODE == PP_CONV_S1) {
        a_voff[j] = (unsigned)((b * p.Hin + a_oy[j]) * p.Win + a_ox[j]) * (unsigned)(p.lda * 2) + lchunk * 16 - 1024u * j;
        if (a_ok[j]) {
#pragma unroll
          for (int t = 0; t < 9; ++t)
            if ((unsigned)(a_oy[j] + t / 3 - 1) < (unsigned)p.Hin && (unsigned)(a_ox[j] + t % 3 - 1) < (unsigned)p.Win) a_taps[j] |= 1u << t;
        }
      } else {
        a_voff[j] = (unsigned)(b * p.Hin * p.Win);
        a_lch[j] = lchunk * 16 + a_bias - 1024u * j;
      }
    }
  }
  const unsigned w_voff = lane * 16;
  const int KT = p.K >> 6;
  unsigned w_soff[NPB0];                                     // scalar: byte offset of the piece's rows in the tiled weights (K tile 0) - 1024 j
#pragma unroll
  for (int j = 0; j < NPB0; ++j) {
    const int n = n0 + 8 * (qb0 + j);
    w_soff[j] = (unsigned)(((n >> 6) * KT) * 8192 + (n & 63) * 128) + (unsigned)(kbeg >> 6) * 8192u - 1024u * j;
  }
  int tap = 0, c0 = 0;
  if (MODE != PP_DENSE) { const int kt0 = kbeg >> 6, ch = kt0 / 9; tap = kt0 - ch * 9; c0 = ch * 64; }    // conv_k_index order
  int kt_issue = 0;                                          // tile at the issue cursor
  int kti = 0;                                               // ... as the issue code reads it: a readfirstlane'd copy (the compiler
                                                             // otherwise keeps the cursor in a vector register in the dense kernel)
  int issued = 0;                                            // DMA pieces this wave has issued so far

  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  // LDS byte address of this wave's first A / first W piece in ring slot 0 (scalar registers: they feed M0)
  const unsigned m0_a = __builtin_amdgcn_readfirstlane(lds0 + qa0 * 1024), m0_w = __builtin_amdgcn_readfirstlane(lds0 + BM * 128 + qb0 * 1024);
  // piece slot Q (0 .. NPA - 1: A, then W) of the tile at the issue cursor into ring slot `stage`; FIRST: first piece of an
  // issue run (M0 must be set even inside a group)
  auto issue_piece = [&](int stage, auto Qc, bool first) {
    constexpr int Q = decltype(Qc)::value;
    if constexpr (Q < NPA) {
      constexpr int j = Q;
      if (j == 0 || first) pp_set_m0(m0_a + stage * STAGE);
      if (MODE == PP_DENSE) {
        pp_dma<j * 1024>(a_voff[j], ra, (unsigned)(kbeg + kti * 64) * 2u);
      } else if (MODE == PP_CONV_S1) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned soff = (unsigned)((ky * p.Win + kx) * (int)p.lda + c0) * 2u;      // relative to the moved base
        const bool ok = (a_taps[j] >> tap) & 1u;
        pp_dma<j * 1024>(ok ? a_voff[j] : PP_OOB - 1024u * j, ra, soff);
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
        const unsigned off = (a_voff[j] + (unsigned)(sy * p.Win + sx)) * (unsigned)(p.lda * 2) + a_lch[j];
        pp_dma<j * 1024>(ok ? off : PP_OOB - 1024u * j, ra, (unsigned)c0 * 2u);
      }
      ++issued;
    } else {
      constexpr int j = Q - NPA;
      if (j < NPB1 || g == 0) {
        if (j == 0 || first) pp_set_m0(m0_w + stage * STAGE);
        pp_dma<j * 1024>(w_voff, rw, w_soff[j] + (unsigned)kti * 8192u);
        ++issued;
      }
    }
  };
  // pieces [LO, HI) of the tile at the cursor
  auto issue_range = [&](int stage, auto LOc, auto HIc) {
    constexpr int LO = decltype(LOc)::value, HI = decltype(HIc)::value;
    if constexpr (LO < HI) {
      [&]<int... I>(std::integer_sequence<int, I...>) {
        (issue_piece(stage, std::integral_constant<int, LO + I>{}, I == 0), ...);
      }(std::make_integer_sequence<int, HI - LO>{});
    }
  };
  auto next_tile = [&]() {
    ++kt_issue;
    kti = __builtin_amdgcn_readfirstlane(kt_issue);
    if (MODE != PP_DENSE) { if (++tap == 9) { tap = 0; c0 += 64; } }
  };
  // this wave's pieces of tile u have landed once at most (issued - (u + 1) pieces-per-tile) later pieces are outstanding
  // (the residual loads rpre_n sit in the queue behind the prologue's pieces and ahead of everything the loop issues)
  int rpre_n = 0;
  auto wait_tile = [&](int u) {
    int n = issued - (u + 1) * npw;
    if (n < 0) n = 0;
    if (u < NST - 1) n += rpre_n;
    pp_wait_vm_dyn<0, 31>(n);
  };

  v4f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  stamp();
  // ---- prologue: tiles 0 .. NST - 2 in flight ---------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) {
    if (s < nt) {
      issue_range(s, std::integral_constant<int, 0>{}, std::integral_constant<int, NP0>{});
      next_tile();
    }
  }
  // ---- the residual tile is fetched NOW, in the layout the epilogue stores in (8 consecutive columns per lane and block pair,
  // 4 for an odd last block): its first-touch latency (HBM: the tensor was written many kernels ago) flies under the K loop
  // instead of standing, one dependent round trip per chunk, at the end of a kernel whose grid is a single round.  Plain loads:
  // the compiler waits for them where the epilogue first reads them (vmcnt(0): everything has long landed); the counted waits of
  // the loop see them as rpre_n extra operations behind the prologue's pieces.
  constexpr int NRP = TN / 2, NR1 = TN & 1;                  // 16-byte chunks / trailing 8-byte chunk per output row block
  const int nb = n0 + g * CPW;
  const bool has_r = p.R != nullptr && p.splits == 1;
  const bool pre_r = has_r && !(VAR & PPV_NO_RPRE);
  uint4 rp16[TM][NRP > 0 ? NRP : 1];
  uint2 rp8[TM];
  if (pre_r) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * RPW + 16 * i + l15;
      const T* rrow = reinterpret_cast<const T*>(p.R) + (size_t)(m < p.M ? m : p.M - 1) * p.ldr;
#pragma unroll
      for (int jp = 0; jp < NRP; ++jp) rp16[i][jp] = *reinterpret_cast<const uint4*>(rrow + nb + 16 * (2 * jp + (quad & 1)) + 8 * (quad >> 1));
      if constexpr (NR1) rp8[i] = *reinterpret_cast<const uint2*>(rrow + nb + 16 * (TN - 1) + 4 * quad);
    }
    rpre_n = TM * (NRP + NR1);
  }
  // GEGLU backward: the saved pre-activations of this lane's outputs (16 bytes per row block) ride under the K loop the same way --
  // fetched in the epilogue they were four dependent HBM round trips per tile (the tensor is 168 MB: never cached)
  uint4 gx[GLU == 2 ? TM : 1][GLU == 2 ? TN : 1];
  if constexpr (GLU == 2) {
    const T* X = reinterpret_cast<const T*>(p.glub_x);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * RPW + 16 * i + l15;
      const size_t xrow = (size_t)(m < p.M ? m : p.M - 1) * (2 * (size_t)p.N);
#pragma unroll
      for (int j = 0; j < TN; ++j) gx[i][j] = *reinterpret_cast<const uint4*>(X + xrow + 2 * (nb + 16 * j) + 16 * (quad & 1) + 8 * (quad >> 1));
    }
    rpre_n = TM * TN;
  }
  // fragment addresses: per lane (row l15 of a 16-row block, chunk 4 s + quad of the 32-deep k-step s, swizzled); the swizzle
  // term ((row >> 1) & 7) only depends on l15 because every block starts on a multiple of 16 rows
  const unsigned fsw = (unsigned)((l15 >> 1) & 7);
  const unsigned fo0 = ((unsigned)quad ^ fsw) << 4, fo1 = ((unsigned)(4 + quad) ^ fsw) << 4;
  const unsigned fa_base = (unsigned)((wm * RPW + l15) * 128);
  const unsigned fb_base = (unsigned)(BM * 128 + (g * CPW + l15) * 128);

  wait_tile(0);
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();               // the stagger: group 1 runs one interval behind group 0
  stamp();

  int cur = 0;                                              // ring slot of tile t
  typedef typename Mfma16<T>::frag frag_t;
  frag_t fw[KS][TN], fx[KS][TM];
  constexpr int NMF = TM * TN * KS;                         // MFMAs of a segment
  for (int t = 0; t < nt; ++t) {
    const bool more = kt_issue < nt;
    const int nstage = cur == 0 ? NST - 1 : cur - 1;        // slot of tile t + NST - 1 = the slot tile t - 1 has left
    const unsigned char* st = smem + cur * STAGE;
    [&]<int... S>(std::integer_sequence<int, S...>) {
      ([&] {
        constexpr int s = S;
        constexpr int PLO = s * NP0 / SEG, PHI = (s + 1) * NP0 / SEG;     // piece slots of this segment
        // ---- LOAD(t, s) ----
        stamp();
        auto read_frags = [&]() {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const unsigned fo = (s * KS + ks) ? fo1 : fo0;
            const unsigned char* pb = st + fb_base + fo;
            const unsigned char* pa = st + fa_base + fo;
#pragma unroll
            for (int j = 0; j < TN; ++j) fw[ks][j] = *reinterpret_cast<const frag_t*>(pb + j * 2048);
#pragma unroll
            for (int i = 0; i < TM; ++i) fx[ks][i] = *reinterpret_cast<const frag_t*>(pa + i * 2048);
          }
        };
        auto issue_load_part = [&]() {
          constexpr int LLO = PLO < XL ? PLO : XL, LHI = PHI < XL ? PHI : XL;      // this segment's slots that stay in the load segment
          if (more) issue_range(nstage, std::integral_constant<int, LLO>{}, std::integral_constant<int, LHI>{});
          if constexpr (s == SEG - 1 && XL == NP0) { if (more) next_tile(); }
        };
        if constexpr ((VAR & PPV_SPLIT_ORDER) != 0) {
          if (wm & 1) { issue_load_part(); stamp(); read_frags(); }
          else { read_frags(); stamp(); issue_load_part(); }
        } else { read_frags(); stamp(); issue_load_part(); }
        stamp();
        // group 1 is in this segment when group 0 finishes MFMA(t, last) and moves on to read tile t + 1
        if constexpr (s == SEG - 1) { if (g == 1) wait_tile(t + 1); }
        pp_wait_lds();
        stamp();
        __builtin_amdgcn_s_barrier();
        stamp();
        // ---- MFMA(t, s) ----
        __builtin_amdgcn_s_setprio(1);
        {
          // this segment's slots that go out between its MFMAs, evenly spaced (piece k after MFMA (k + 1) NMF / (n + 1))
          constexpr int MLO = PLO > XL ? PLO : XL, MHI = PHI > XL ? PHI : XL;
          constexpr int NPC = MHI - MLO;
          [&]<int... X>(std::integer_sequence<int, X...>) {
            ([&] {
              constexpr int x = X, ks = x / (TM * TN), i = (x / TN) % TM, j = x % TN;
              acc[i][j] = Mfma16<T>::run(fw[ks][j], fx[ks][i], acc[i][j]);
              if constexpr (NPC > 0) {
                [&]<int... Kp>(std::integer_sequence<int, Kp...>) {
                  ([&] {
                    if constexpr ((Kp + 1) * NMF / (NPC + 1) - 1 == x) {
                      if (more) issue_piece(nstage, std::integral_constant<int, MLO + Kp>{}, true);
                    }
                  }(), ...);
                }(std::make_integer_sequence<int, NPC>{});
              }
            }(), ...);
          }(std::make_integer_sequence<int, NMF>{});
          if constexpr (s == SEG - 1 && XL < NP0) { if (more) next_tile(); }
        }
        __builtin_amdgcn_s_setprio(0);
        stamp();
        // group 0 is in this segment right before it reads tile t + 1
        if constexpr (s == SEG - 1) { if (g == 0) wait_tile(t + 1); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }(), ...);
    }(std::make_integer_sequence<int, SEG>{});
    cur = cur + 1 == NST ? 0 : cur + 1;
  }
  if (g == 0) __builtin_amdgcn_s_barrier();
  stamp();

  // ---- epilogue --------------------------------------------------------------------------------------------------------------
#ifdef DH_PP_VARIANTS
  if (p.pp_order & 0x100) { if (acc[0][0][0] == 12345.678f) p.partial[0] = 1.f; continue; }
#endif
  // acc[i][j][r] = D[m = m0 + wm RPW + 16 i + l15][n = n0 + g CPW + 16 j + 4 quad + r]
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
    continue;
  }
  typedef T T4 __attribute__((ext_vector_type(4)));
  typedef T T8 __attribute__((ext_vector_type(8)));
  // two packed 4-column groups (8 bytes each) of the lane pair (l, l ^ 16) -> one 16-byte chunk per lane: lanes of an even quad get
  // {their a, the partner's a}, lanes of an odd quad {the partner's b, their b} (the exchange the plain epilogue does on f32 values)
  auto pair16 = [&](uint2 a, uint2 b) -> uint4 {
    const lane_u2p x = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    const lane_u2p y = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    return make_uint4(x[0], y[0], x[1], y[1]);
  };
  if constexpr (GLU == 1) {
    // GEGLU forward.  Lane (l15, quad) owns, of block pair q = (2q, 2q + 1), the value and the gate of outputs 4 quad .. + 3 of
    // the pair's 16 outputs.  y is computed from the ROUNDED pre-activations (what the backward pass reads back).
    const bool hb1 = p.bias != nullptr;
    T* Y = reinterpret_cast<T*>(p.glu_y);
    // bias of this lane's value / gate columns: loaded once, before the first store (see the plain epilogue)
    float4 gbv[TN / 2], gbg[TN / 2];
#pragma unroll
    for (int q = 0; q < TN / 2; ++q) {
      gbv[q] = hb1 ? *reinterpret_cast<const float4*>(p.bias + nb + 32 * q + 4 * quad) : make_float4(0.f, 0.f, 0.f, 0.f);
      gbg[q] = hb1 ? *reinterpret_cast<const float4*>(p.bias + nb + 32 * q + 16 + 4 * quad) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * RPW + 16 * i + l15;
      const bool mok = m < p.M;
      uint2 yq[TN / 2];
#pragma unroll
      for (int q = 0; q < TN / 2; ++q) {
        const int cv = nb + 32 * q;                       // first value column of the pair (paired layout); gates at + 16
        float v[4], gt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * q][r]; gt[r] = acc[i][2 * q + 1][r]; }
        {
          const float4 bv = gbv[q], bg = gbg[q];
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; gt[0] += bg.x; gt[1] += bg.y; gt[2] += bg.z; gt[3] += bg.w;
        }
        T4 vt, gtt, yt;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          vt[r] = from_f32<T>(v[r]); gtt[r] = from_f32<T>(gt[r]);
          yt[r] = from_f32<T>(to_f32<T>(vt[r]) * gelu_f(to_f32<T>(gtt[r])));
        }
        yq[q] = __builtin_bit_cast(uint2, yt);
        const uint4 pre = pair16(__builtin_bit_cast(uint2, vt), __builtin_bit_cast(uint2, gtt));
        if (p.C && mok)
          *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.C) + (size_t)m * p.ldc + cv + 16 * (quad & 1) + 8 * (quad >> 1)) = pre;
      }
      // y [M][F]: the pair's 16 outputs start at column (nb >> 1) + 16 q; two pairs trade like two blocks
#pragma unroll
      for (int q = 0; q + 1 < TN / 2; q += 2) {
        const uint4 yy = pair16(yq[q], yq[q + 1]);
        if (mok) *reinterpret_cast<uint4*>(Y + (size_t)m * p.glu_ldy + (nb >> 1) + 16 * (q + (quad & 1)) + 8 * (quad >> 1)) = yy;
      }
      if constexpr ((TN / 2) & 1) {
        constexpr int q = TN / 2 - 1;
        if (mok) *reinterpret_cast<uint2*>(Y + (size_t)m * p.glu_ldy + (nb >> 1) + 16 * q + 4 * quad) = yq[q];
      }
    }
    continue;
  }
  if constexpr (GLU == 2) {
    // GEGLU backward.  The tile holds dy (natural output columns o); the saved pre-activations of outputs [ob, ob + 16) are the
    // paired 32-column group at 2 ob: value columns, then gate columns.  The lane pair loads / stores 16-byte chunks of that
    // group and trades halves so that each lane works on the value and the gate of ITS four outputs.
    T* DX = reinterpret_cast<T*>(p.glub_dx);
    const size_t ldx = 2 * (size_t)p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * RPW + 16 * i + l15;
      const bool mok = m < p.M;
      const size_t xrow = (size_t)(mok ? m : p.M - 1) * ldx;
      const uint4 (&raw)[TN] = gx[i];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        // chunk of an even quad = value columns 4 quad .. + 7 (its own four and the partner's), of an odd quad = gate columns of
        // the partner's four and its own: the same exchange gives every lane (value, gate) of its own four outputs
        const uint4 hx = pair16(make_uint2(raw[j].x, raw[j].y), make_uint2(raw[j].z, raw[j].w));
        const T4 hh = __builtin_bit_cast(T4, make_uint2(hx.x, hx.y)), gg = __builtin_bit_cast(T4, make_uint2(hx.z, hx.w));
        T4 oh, og;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dy = acc[i][j][r], gate = to_f32<T>(gg[r]);
          const GeluParts gp = gelu_parts(gate);
          oh[r] = from_f32<T>(dy * gate * gp.Phi);
          og[r] = from_f32<T>(dy * to_f32<T>(hh[r]) * fmaf(gate, gp.pdf, gp.Phi));
        }
        const uint4 out = pair16(__builtin_bit_cast(uint2, oh), __builtin_bit_cast(uint2, og));
        if (mok) *reinterpret_cast<uint4*>(DX + xrow + 2 * (nb + 16 * j) + 16 * (quad & 1) + 8 * (quad >> 1)) = out;
      }
    }
    continue;
  }
  const bool hb = p.bias != nullptr, hv = p.rowvec != nullptr;
  // The per-column vectors are loaded ONCE, before the first store: a load placed behind a store may alias it as far as the
  // compiler knows, so a bias read inside the row loop was re-issued after every store -- ten dependent L2 round trips, 9 000
  // cycles of epilogue on a tile whose K loop (K = 320) takes 13 000 (profiles/r05_pp_tile_timeline.txt).
  // cbv[j] = bias (+ the per-image vector) of the four columns of block j this lane holds BEFORE the lane-pair exchange.
  float cbv[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) cbv[j][c] = 0.f;
  auto add_cols = [&](const float* vec) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float4 b = *reinterpret_cast<const float4*>(vec + nb + 16 * j + 4 * quad);
      cbv[j][0] += b.x; cbv[j][1] += b.y; cbv[j][2] += b.z; cbv[j][3] += b.w;
    }
  };
  if (hb) add_cols(p.bias);
  // per-image vector (the time-embedding projection added to a resnet's first convolution): row m belongs to image
  // m / rows_per_batch; a tile that lies inside one image (every tile when the image's rows are a multiple of BM) adds it here
  bool hv_rows = hv;
  if (hv) {
    const int mlast = (m0 + BM - 1 < p.M ? m0 + BM - 1 : p.M - 1);
    const int im0 = div_small(m0, p.inv_rows_per_batch);
    if (im0 == div_small(mlast, p.inv_rows_per_batch)) { add_cols(p.rowvec + (size_t)im0 * p.rowvec_ld); hv_rows = false; }
  }
  // The row loop exists in four straight-line versions (residual or not, per-row vector or not), chosen ONCE: as uniform branches
  // inside the unrolled loop they were three taken branches per 16-byte store.  Per store now: the bias add lands in fresh
  // registers (no copies in front of the in-place lane exchange), the address is one row pointer + an immediate.
  // THE STORES GO THROUGH LDS.  In the MFMA layout a lane owns a ROW: the 64 lanes of a store instruction write 16 bytes each to
  // 16 rows x 4 chunks with neighbouring lanes in different rows, and the memory pipeline takes such an instruction as 64 separate
  // 16-byte requests -- the tile's 82 KB needed 7 500 cycles per CU, four times what the same bytes need as whole row segments
  // (tools/ubench_store.hip, profiles/r05_ubench_store.txt; profiles/r05_pp_tile_timeline.txt).  Each wave stages its RPW x CPW
  // outputs in its own rows of the (idle) ring, reads them back with lanes running ALONG the rows and stores 16-byte chunks of
  // consecutive addresses: same values, same rounding, a quarter of the requests.  Wave-private rows: no barrier here (one at the
  // top of the next tile, before its prologue overwrites the ring).
  constexpr int SPITCH = CPW * 2 + 16;                                    // staged row: the wave's CPW outputs + 16 bytes (bank spread)
  static_assert(8 * RPW * SPITCH <= NST * STAGE, "staging rows fit the ring");
  unsigned char* stg = smem + wave * (RPW * SPITCH);
  // (lane-constant address terms of the epilogue would be hoisted out of the persistent tile loop and held in registers through
  //  every K loop -- 30 registers, spills on the 256 x 160 tile: an opaque copy of the lane id keeps them here)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int l15e = lane_e & 15, quade = lane_e >> 4;
  stamp();
  auto rows = [&](auto HRc, auto HVc) {
    constexpr bool HR = decltype(HRc)::value, HV = decltype(HVc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * RPW + 16 * i + l15;
      const bool mok = m < p.M;
      // this lane's 16-byte chunk of block pair jp starts 32 jp columns further; the odd last block's 8-byte chunk at ctail
      // (addresses in the wave's staging rows, see below)
      unsigned char* cpair = stg + (16 * i + l15e) * SPITCH + (16 * (quade & 1) + 8 * (quade >> 1)) * 2;
      unsigned char* ctail = stg + (16 * i + l15e) * SPITCH + (16 * (TN - 1) + 4 * quade) * 2;
      const float* vrow = HV ? p.rowvec + (size_t)div_small(mok ? m : 0, p.inv_rows_per_batch) * p.rowvec_ld + nb : nullptr;
      const T* rrow = HR ? reinterpret_cast<const T*>(p.R) + (size_t)(mok ? m : 0) * p.ldr + nb : nullptr;
#pragma unroll
      for (int jp = 0; jp < NRP; ++jp) {
        constexpr int dummy = 0; (void)dummy;
        const int j = 2 * jp;
        // blocks j, j + 1: the lane pairs (l, l ^ 16) trade four columns so that every lane holds EIGHT consecutive columns
        // (one 16-byte store): lanes of an even quad keep block j, lanes of an odd quad get block j + 1
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // (__builtin_bit_cast applied to the vector-ELEMENT lvalue acc[i][j][r] reads element 0 for every r with hipcc of
          //  ROCm 7.2 -- seen in the ISA and on the device: the sums below are plain floats)
          float fa = acc[i][j][r] + cbv[j][r], fb = acc[i][j + 1][r] + cbv[j + 1][r];
          if constexpr (HV) { fa += vrow[16 * j + 4 * quad + r]; fb += vrow[16 * (j + 1) + 4 * quad + r]; }
          const lane_u2p x = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, fa), __builtin_bit_cast(unsigned, fb), false, false);
          const unsigned x0 = x[0], x1 = x[1];
          v[r] = __builtin_bit_cast(float, x0);
          v[4 + r] = __builtin_bit_cast(float, x1);
        }
        if constexpr (HR) {
          uint4 raw = rp16[i][jp];
          if (!pre_r) raw = *reinterpret_cast<const uint4*>(rrow + 16 * (j + (quad & 1)) + 8 * (quad >> 1));
          const T8 rv = __builtin_bit_cast(T8, raw);
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] += to_f32<T>(rv[c]);
        }
        T8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = from_f32<T>(v[c]);
        *reinterpret_cast<uint4*>(cpair + 64 * jp) = __builtin_bit_cast(uint4, o);
      }
      if constexpr (TN & 1) {
        constexpr int j = TN - 1;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] + cbv[j][r];
          if constexpr (HV) v[r] += vrow[16 * j + 4 * quad + r];
        }
        if constexpr (HR) {
          uint2 raw = rp8[i];
          if (!pre_r) raw = *reinterpret_cast<const uint2*>(rrow + 16 * j + 4 * quad);
          const T4 rv = __builtin_bit_cast(T4, raw);
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += to_f32<T>(rv[c]);
        }
        T4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = from_f32<T>(v[c]);
        *reinterpret_cast<uint2*>(ctail) = __builtin_bit_cast(uint2, o);
      }
      // (LDS stores alias nothing the compiler knows of: without this fence it schedules all row blocks side by side and the
      //  256 x 160 tile spills)
      __builtin_amdgcn_sched_barrier(0);
      stamp();
    }
  };
  if (has_r) { if (hv_rows) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
  else       { if (hv_rows) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
  {
    constexpr int CPR = CPW / 8, NCH = RPW * CPR;                         // 16-byte chunks per staged row / per wave
    static_assert(NCH % 64 == 0, "whole store instructions");
    T* cw = reinterpret_cast<T*>(p.C) + (size_t)(m0 + wm * RPW) * p.ldc + nb;
    const int mleft = p.M - (m0 + wm * RPW);                              // rows of this wave inside the matrix
#pragma unroll
    for (int k = 0; k < NCH / 64; ++k) {
      const int c = 64 * k + lane_e, row = c / CPR, cc = c - row * CPR;
      const uint4 v = *reinterpret_cast<const uint4*>(stg + row * SPITCH + cc * 16);
      if (row < mleft) *reinterpret_cast<uint4*>(cw + (size_t)row * p.ldc + cc * 8) = v;
    }
  }
  stamp();
  }      // work items of this workgroup
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

// Compute units of the current device (cached): the persistent grid has one workgroup per CU and the policy asks for 7/8 of a
// round of tiles.  256 on an MI355X in SPX mode (what every measurement of this file was taken on) and the fallback without a device
// (dh_dbg_gemm_pp_plan on the CPU); a partitioned (CPX) or smaller device gets its own count instead of a mistuned 256.
static int pp_device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) cus = n - n % 8;
    else { (void)hipGetLastError(); cus = 256; }
  }
  return cus;
}

static int g_pp_glu = -1;                           // dh_dbg_gemm_pp_glu: the GEGLU-epilogue launches (-1 policy, 0 never, 1 always)
bool gemm_pp_plan(const GemmK& k, size_t partial_elems, int force, PpPlan* plan) {
  if (force == 1) return false;
  // what the kernel carries: the plain epilogue (bias, per-image vector, residual), 16-byte aligned rows, descriptors below 2 GiB
  if (k.ln_s || k.act_silu) return false;
  const int glu = k.glu_y ? 1 : (k.glub_x ? 2 : 0);
  // The GEGLU epilogues (tests/test_gemm_pp_gpu.py).  With the ping-pong loop alone they ran at k_gemm_dma's time
  // (profiles/r05_ab_glu_pp.txt); persistent workgroups, the bias read once and the saved pre-activations prefetched under the K loop
  // changed that (profiles/r05_ab_pp_shortk.txt, M = 32768, F = 1280: forward 133 -> 124 us saving / 110 -> 97 not saving, backward
  // 121 -> 113; M = 8192, F = 2560: forward 93.5 -> 90.9 / 86.6 -> 82.3 but backward 68 -> 73; the other batch sizes in
  // profiles/r05_ab_pp_glu_by_batch.txt): forward from M = 2048 on, backward from M = 32768 on.
  // g_pp_glu: -1 = this policy, 0 = never, 1 = always.
  const bool glu_here = force == 2 || g_pp_glu > 0 || (g_pp_glu < 0 && ((glu == 1 && k.M >= 2048) || k.M >= 32768));
  if (glu && !glu_here) return false;
  if (glu && (k.mode != A_DENSE || k.N % 128 || k.rowvec || k.R || (glu == 1 && (k.glu_ldy % 8 || ((size_t)k.glu_y & 15))) ||
              (glu == 2 && (((size_t)k.glub_x & 15) || ((size_t)k.glub_dx & 15))))) return false;
  if (k.rowvec && (k.rowvec_ld % 4 || ((size_t)k.rowvec & 15))) return false;
  if (k.K % 64 || k.M <= 0 || (!k.C && !glu)) return false;
  if (k.N % 160 && k.N % 128) return false;
  // the W descriptor covers N * K * 2 bytes of [N / 64][K / 64] tiles: a last partial 64-row tile would lie partly beyond it (the
  // LDS-DMA then returns zeros); the bias is read as float4
  if (k.N % 64) return false;
  if (k.bias && ((size_t)k.bias & 15)) return false;
  if ((k.C && (k.ldc % 8 || ((size_t)k.C & 15))) || k.lda % 8 || ((size_t)k.A & 15) || ((size_t)k.W & 15)) return false;
  if (k.R && (k.ldr % 8 || ((size_t)k.R & 15))) return false;
  if (k.mode != A_DENSE && (k.Hout <= 0 || k.Wout <= 0 || k.M % (k.Hout * k.Wout) || k.Cin % 64 || k.K != 9 * k.Cin)) return false;
  const size_t ab = pp_a_bytes(k) + (size_t)(k.Win + 1) * k.lda * 2, wb = (size_t)k.N * k.K * 2;
  if (ab >= 0x7ff00000ull || wb >= 0x7ff00000ull) return false;
  const int ktiles = k.K / 64;
  // tile: the 160-column tile where it divides N (N = 320, 640, 960, 1280 ...), else 128 columns (the GEGLU epilogues: 128, a
  // wave must own whole (value, gate) block pairs)
  const int bn = (k.N % 160 == 0 && !glu) ? 160 : 128;
  const int tn = k.N / bn;
  const long t256 = (long)cdiv(k.M, 256) * tn, t128 = (long)cdiv(k.M, 128) * tn;
  int bm = 0, splits = 1;
  const int cus = pp_device_cus(), fill = cus - cus / 8;      // 256 / 224 on the MI355X
  if (t256 >= fill) bm = 256;                                // one round or more of 256-row tiles
  else if (t128 >= fill) bm = 128;                           // (M = 8192, N = 640 at batch 8: 64 x 4)
  else if (!glu && k.partial && ktiles >= 64 && t256 >= 64 && k.M >= 1024 && k.M <= 4096) {
    // long K loops on fewer tiles than CUs: split K over workgroups (f32 slabs + the reduce kernels of gemm.hip).  Measured for
    // the 16x16-latent level at batch 8 (M = 2048, N = 1280, K = 11520: 64 tiles x 4 splits, 74.7 -> 65.0 us); the B = 1 / B = 2
    // shapes of this kind stay on k_gemm_dma's tiles.  From 64 K tiles on: the B = 1 input gradients with K = 2880 (45 tiles, M = 4096,
    // N = 640 / 960) lose 7 - 18 % here, K = 5760 wins 3 - 4 % (profiles/r05_ab_pp_splitk_branch.txt)
    bm = 256;
    splits = (int)(cus / t256);
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

// the shipped main-loop variants (see the PPV_* flags; A/B in profiles/r05_ab_pp_variants.txt): 64-deep segments everywhere
// (half the barriers: 256x160 conv K = 2880 64.6 -> 61.5 us); of the 7 (256-row tiles) / 5 (128-row tiles) LDS-DMA pieces of a
// wave and tile the last 1 / 3 go out between the MFMAs of the multiply segment, which balances the two segments of an interval
// (-> 59.7 us; 128x160 conv K = 5760 76.2 -> 65.9 us)
constexpr int PP_SHIP_256 = PPV_K64 | (1 << PPV_NDIM_SHIFT), PP_SHIP_128 = PPV_K64 | (3 << PPV_NDIM_SHIFT);
static int g_pp_persist = 1;                        // dh_dbg_gemm_pp_persist: 0 = one workgroup per work item (the round-by-round form)
static int g_pp_ablate = 0;                         // dh_dbg_gemm_pp_ablate (measurement build)
static int g_pp_variant = -1;                       // measurement builds (-DDH_PP_VARIANTS): dh_dbg_gemm_pp_variant; -1 = shipped
static unsigned long long* g_pp_ts = nullptr;

template <class T, int BM, int BN, int NST, int VAR>
static void pp_launch_tile(int mode, dim3 grid, hipStream_t st, const GemmK& k, hipEvent_t e0, hipEvent_t e1) {
#define DH_PP_LAUNCH(KERNEL)                                                                      \
  do {                                                                                            \
    if (e0) hipExtLaunchKernelGGL(KERNEL, grid, dim3(512), 0, st, e0, e1, 0, k);                  \
    else hipLaunchKernelGGL(KERNEL, grid, dim3(512), 0, st, k);                                   \
  } while (0)
  if constexpr (BN == 128 && VAR == (BM == 256 ? PP_SHIP_256 : PP_SHIP_128)) {
    if (k.glu_y) { DH_PP_LAUNCH((k_gemm_pp<T, BM, BN, PP_DENSE, NST, VAR, 1>)); return; }
    if (k.glub_x) { DH_PP_LAUNCH((k_gemm_pp<T, BM, BN, PP_DENSE, NST, VAR, 2>)); return; }
  }
  if (mode == PP_DENSE) DH_PP_LAUNCH((k_gemm_pp<T, BM, BN, PP_DENSE, NST, VAR>));
  else if (mode == PP_CONV_S1) DH_PP_LAUNCH((k_gemm_pp<T, BM, BN, PP_CONV_S1, NST, VAR>));
  else {
    if constexpr (VAR == (BM == 256 ? PP_SHIP_256 : PP_SHIP_128)) DH_PP_LAUNCH((k_gemm_pp<T, BM, BN, PP_GENERIC, NST, VAR>));
  }
#undef DH_PP_LAUNCH
}

// VAR < 0: the shipped variant of the tile
template <class T, int VAR>
static void pp_launch_var(const GemmK& k, const PpPlan& plan, int mode, dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  constexpr int V256 = VAR < 0 ? PP_SHIP_256 : VAR, V128 = VAR < 0 ? PP_SHIP_128 : VAR;
  if (plan.bm == 256 && plan.bn == 160) pp_launch_tile<T, 256, 160, 3, V256>(mode, grid, st, k, e0, e1);
  else if (plan.bm == 256) pp_launch_tile<T, 256, 128, 3, V256>(mode, grid, st, k, e0, e1);
  else if (plan.bn == 160) pp_launch_tile<T, 128, 160, 4, V128>(mode, grid, st, k, e0, e1);
  else pp_launch_tile<T, 128, 128, 4, V128>(mode, grid, st, k, e0, e1);
}

template <class T>
static void pp_launch(const GemmK& k, const PpPlan& plan, int mode, dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
#ifdef DH_PP_VARIANTS
  // measurement build: every variant of the main loop side by side (f16, dense / stride-1 convolution)
  if constexpr (std::is_same<T, f16>::value) {
    if (mode != PP_GENERIC) {
      switch (g_pp_variant) {
#define DH_PP_CASE(V) case V: pp_launch_var<T, V>(k, plan, mode, grid, st, e0, e1); return;
        DH_PP_CASE(0) DH_PP_CASE(1) DH_PP_CASE(2) DH_PP_CASE(4) DH_PP_CASE(5) DH_PP_CASE(6) DH_PP_CASE(7)
        DH_PP_CASE(4 + 16) DH_PP_CASE(4 + 32) DH_PP_CASE(4 + 48) DH_PP_CASE(4 + 64) DH_PP_CASE(4 + 80)
        DH_PP_CASE(5 + 16) DH_PP_CASE(5 + 32) DH_PP_CASE(5 + 48) DH_PP_CASE(5 + 64)
        DH_PP_CASE(32) DH_PP_CASE(48) DH_PP_CASE(20 + 8) DH_PP_CASE(52 + 8)
        DH_PP_CASE(20 + 128) DH_PP_CASE(52 + 128) DH_PP_CASE(21 + 128) DH_PP_CASE(53 + 128) DH_PP_CASE(4 + 128) DH_PP_CASE(36 + 128)
#undef DH_PP_CASE
        default: break;
      }
    }
  }
#endif
  pp_launch_var<T, -1>(k, plan, mode, grid, st, e0, e1);
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
  k.pp_ts = g_pp_ts;
  // which operand panel consecutive work items (= one XCD's L2) share: the column tiles of a row tile share its A rows, the
  // row tiles of a column tile share its W rows; take the order with the smaller traffic estimate (the other operand is
  // then read once per XCD)
  const double a_tot = (double)k.pp_a_bytes, w_tot = (double)k.pp_w_bytes;
  const double col_fastest = a_tot + 8.0 * w_tot;
  const double row_fastest = w_tot + (double)(k.pp_tn * k.splits < 8 ? k.pp_tn * k.splits : 8) * a_tot;
  k.pp_order = (row_fastest < col_fastest ? 1 : 0) | g_pp_ablate;
  k.pp_nwork = k.pp_tm * k.pp_tn * k.splits;
  // persistent grid: one workgroup per CU (256: a multiple of 8, the XCD order of the work items holds in every round)
  int slots = pp_device_cus();
  if (g_pp_persist == 0 || k.pp_nwork <= slots) slots = k.pp_nwork;
  dim3 grid((unsigned)slots);
  const int mode = pp_mode(k);
  if (dtype == DH_DTYPE_F16) pp_launch<f16>(k, plan, mode, grid, st, e0, e1);
  else pp_launch<bf16>(k, plan, mode, grid, st, e0, e1);
}

}  // namespace dh

// measurement hook: main-loop variant of the next k_gemm_pp launches (builds with -DDH_PP_VARIANTS carry them; the product
// library accepts only the shipped one) and the device buffer (2 x 512 u64) the stamping variants write their timeline to
// host-only query of the policy (no launch, no device): which tile / K split gemm_pp_plan gives a launch of this shape, or 0 in *bm when
// it stays on k_gemm_dma.  conv: 0 dense, 1 = 3x3 stride-1 convolution of hw x hw images (K = 9 Cin); glu: 0 plain, 1 GEGLU forward,
// 2 GEGLU backward; partial_elems: f32 elements of split-K workspace.  tests/test_layout_models.py pins the table on CPU.
extern "C" int dh_dbg_gemm_pp_plan(int M, int N, int K, int conv, int hw, int glu, size_t partial_elems, int* bm, int* bn, int* splits) {
  DH_REQUIRE(bm && bn && splits, "null pointer");
  static __attribute__((aligned(16))) unsigned char dummy[16];
  dh::GemmK k{};
  k.A = dummy; k.W = dummy; k.C = dummy; k.M = M; k.N = N; k.K = K; k.lda = conv ? K / 9 : K; k.ldc = N;
  k.mode = conv ? dh::A_CONV3 : dh::A_DENSE;
  if (conv) { k.Hin = k.Win = k.Hout = k.Wout = hw; k.Cin = K / 9; k.stride = 1; k.up = 0; k.pad = 1; }
  k.partial = partial_elems ? reinterpret_cast<float*>(dummy) : nullptr;
  if (glu == 1) { k.glu_y = dummy; k.glu_ldy = N / 2; }
  if (glu == 2) { k.glub_x = dummy; k.glub_dx = dummy; k.C = nullptr; }
  dh::PpPlan plan;
  const bool use = dh::gemm_pp_plan(k, partial_elems, 0, &plan);
  *bm = use ? plan.bm : 0; *bn = use ? plan.bn : 0; *splits = use ? plan.splits : 0;
  return DH_OK;
}
extern "C" int dh_dbg_gemm_pp_glu(int on) {
  dh::g_pp_glu = on;
  return DH_OK;
}
extern "C" int dh_dbg_gemm_pp_persist(int on) {
  dh::g_pp_persist = on;
  return DH_OK;
}
extern "C" int dh_dbg_gemm_pp_ablate(int bits) {
#ifndef DH_PP_VARIANTS
  DH_REQUIRE(bits == 0, "timing ablations exist in the measurement build only (tools/lab.sh build-pp-variants)");
#endif
  dh::g_pp_ablate = bits & 0x300;
  return DH_OK;
}
extern "C" int dh_dbg_gemm_pp_variant(int variant, unsigned long long* ts) {
#ifndef DH_PP_VARIANTS
  DH_REQUIRE(variant < 0, "this build carries the shipped k_gemm_pp variants only (tools/lab.sh build-pp-variants)");
#endif
  DH_REQUIRE(variant >= -1 && variant < 256, "variant: bit 0 stamps, bit 1 all DMA inside the MFMA segment, bit 2 64-deep segments, bits 4-6 pieces inside the MFMA segment");
  dh::g_pp_variant = variant;
  dh::g_pp_ts = ts;
  return DH_OK;
}
