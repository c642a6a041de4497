// Flash attention for head dim 64 on gfx950 MFMA (v_mfma_f32_32x32x16), forward + backward.
//
// Everything is computed TRANSPOSED so that softmax statistics are per-lane scalars:
//   S^T = K Q^T   : A operand = K rows from LDS, B operand = Q fragment in registers
//                   -> accumulator rows = keys (spread over the 16 registers and the two
//                      lane halves), column = query = lane & 31
//   O^T += V^T P^T: the accumulator layout of P^T is, register-for-register, a valid B
//                   operand (k = keys) as long as the A operand (V^T) enumerates the keys in
//                   the same order -- slot j of lane half hi in MFMA step s is key
//                   16 s + 8 (j >> 2) + 4 hi + (j & 3).
// V^T (and K^T, Q^T, dO^T in the backward kernels) is never materialised: tiles are staged
// row-major and the transposed fragments come from ds_read_b64_tr_b16 (each 16-lane group
// reads a [4 rows][16 cols] block and lane t receives column t), so every LDS write is a
// 16-byte store and every tile exists once.
// The backward kernels use the same two product shapes (one with the roles of keys and
// queries exchanged).  dQ and dK/dV are separate kernels (no atomics, deterministic).
#include "unet_kernels.h"

namespace dh {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v4s __attribute__((ext_vector_type(4)));

template <class T> struct Mma;
template <> struct Mma<f16> {
  static __device__ __forceinline__ v16f run(uint4 a, uint4 b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  }
};
template <> struct Mma<bf16> {
  static __device__ __forceinline__ v16f run(uint4 a, uint4 b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8b, a), __builtin_bit_cast(v8b, b), c, 0, 0, 0);
  }
};

constexpr int HD = 64;        // head dim
constexpr int TLD = 72;       // LDS row stride in halves (144 B: conflict-free b128 and tr_b16 reads)
constexpr int TILE = 64 * TLD;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float SCALE = 0.125f;               // 1/sqrt(64)
constexpr float CEXP = SCALE * LOG2E;         // scores are exponentiated as exp2(s * CEXP - m * CEXP)

__device__ __forceinline__ v16f zero16() {
  v16f z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#ifndef DH_ATTN_ABL
#define DH_ATTN_ABL 0      // timing-only ablations of the forward loop (tools/lab.sh ablate-attn): never set in the product build
#endif

// 4 register fragments (k = d) of row `row` of a [rows][ld] matrix at column col0: B-operand layout
template <class T>
__device__ __forceinline__ void load_row_frags(const T* base, long ld, long row, bool ok, int col0, int hi, uint4 (&f)[4]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    f[kk] = ok ? *reinterpret_cast<const uint4*>(base + row * ld + col0 + 16 * kk + 8 * hi) : make_uint4(0, 0, 0, 0);
}

// a 64 x 64 tile travels global -> registers (fetch, issued one tile ahead of its use) -> LDS (commit); the GT threads
// of a wave group share the 512 sixteen-byte chunks of the tile.
// The fetch is INLINE ASM and every load is unconditional (round 4, s_memtime stamps of the dK/dV loop, tools/attn_timeline_dkv.py:
// a third of a 5 950-cycle tile sat between the tile's two barriers).  With compiler-issued loads the prefetch never overlapped the
// MFMAs it was issued under: `ok ? load : 0` made hipcc merge the loaded registers with the zeros right behind the load, it moved
// loaded values to their loop-carried registers at once, and __syncthreads() is a fence (s_waitcnt vmcnt(0)) -- and since vmcnt
// retires in order, any one of these waits drains every load issued before it.  Now: asm loads the compiler does not count (a
// chunk index past the tile reads chunk 511 again and is never committed; a row past the matrix reads its last row again -- finite
// values whose probabilities are zero: keys past Nk are masked, queries past Nq carry lse = +inf), LDS-only barriers
// (lds_barrier), and ONE s_waitcnt vmcnt(0) in front of the commit, pinned to the tile registers (tile_wait) so that no use of
// them can be scheduled in front of it.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int GT> struct TileRegs { u4v v[(512 + GT - 1) / GT]; };
template <class T, int GT>
__device__ __forceinline__ void fetch_tile(const T* base /* + head column */, long ld, long row0, long rows_total,
                                           TileRegs<GT>& t, int tid) {
#pragma unroll
  for (int j = 0; j < (512 + GT - 1) / GT; ++j) {
    int idx = tid + j * GT;
    idx = idx < 512 ? idx : 511;
    long r = row0 + (idx >> 3);
    r = r < rows_total ? r : rows_total - 1;
    const T* ptr = base + r * ld + (idx & 7) * 8;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(t.v[j]) : "v"(ptr) : "memory");
  }
}
// The compiler-issued loads of a kernel's prologue (row fragments, lse) must be COMPLETE, and known to hipcc as complete, before a
// loop that keeps asm loads in flight: otherwise its own s_waitcnt for them (re-evaluated on the loop's back edge) counts the asm
// loads and drains them.  Using the registers as asm operands makes it wait here.
__device__ __forceinline__ void frags_ready(uint4 (&f)[4]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(f[kk].x), "+v"(f[kk].y), "+v"(f[kk].z), "+v"(f[kk].w));
}
// every load this wave has issued has landed; the tile registers are (re)defined here: their uses stay behind the wait
template <int GT>
__device__ __forceinline__ void tile_wait(TileRegs<GT>& a, TileRegs<GT>& b) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < (512 + GT - 1) / GT; ++j) { asm volatile("" : "+v"(a.v[j])); asm volatile("" : "+v"(b.v[j])); }
}
template <int GT>
__device__ __forceinline__ void commit_tile(const TileRegs<GT>& t, unsigned short* rm, int tid) {
#pragma unroll
  for (int j = 0; j < (512 + GT - 1) / GT; ++j) {
    const int idx = tid + j * GT;
    if (512 % GT == 0 || idx < 512) *reinterpret_cast<u4v*>(&rm[(idx >> 3) * TLD + (idx & 7) * 8]) = t.v[j];
  }
}

// acc = sum_kk mfma(A = rows (rowbase + lane&31) of an LDS row-major tile, B = register fragments)
template <class T>
__device__ __forceinline__ v16f tile_times_frags(const unsigned short* tile, int rowbase, int ln, int hi, const uint4 (&f)[4]) {
  v16f acc = zero16();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const uint4 a = *reinterpret_cast<const uint4*>(&tile[(rowbase + ln) * TLD + 16 * kk + 8 * hi]);
    acc = Mma<T>::run(a, f[kk], acc);
  }
  return acc;
}

// A operand = TRANSPOSE of a row-major tile: fragment row = tile column (cbase + lane&31), reduction slots =
// tile rows r0.., in the accumulator order (rows r0 + 4 hi + 0..3 and r0 + 8 + 4 hi + 0..3).
// `tptr` = &tile[(4 hi + (t >> 2)) * TLD + 16 * ((lane >> 4) & 1) + 4 * (t & 3)], t = lane & 15 (per lane, hoisted)
__device__ __forceinline__ uint4 tr_frag(const unsigned short* tptr, int cbase, int r0) {
  typedef __attribute__((address_space(3))) v4s* lp;
  const unsigned short* a = tptr + r0 * TLD + cbase;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a));
  const v4s up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a + 8 * TLD));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), u2 = __builtin_bit_cast(uint2, up);
  return make_uint4(l2.x, l2.y, u2.x, u2.y);
}

// ---- dense tiles filled by LDS-DMA (k_attn_fwd, DB = true) --------------------------------------------------------
// global_load_lds_dwordx4 writes 64 lanes x 16 bytes to CONSECUTIVE LDS addresses, so a staged tile has no row padding:
// [64 rows][128 B], and the bank spread comes from the per-lane SOURCE address instead -- 16-byte chunk c of row r is stored at
// chunk position c ^ dswz(r), dswz(r) = the three bits of (r >> 1) reversed.  Row-fragment reads (ds_read_b128, 16 lanes per
// cycle = rows of all eight (r >> 1) & 7 classes and both parities, as in gemm.hip) hit 16 distinct slots of the 256-byte bank
// row; the transposed reads (ds_read_b64_tr_b16: four consecutive rows x 64 bytes per cycle) need rows r and r + 2 in
// different 64-byte quarters, which is why bit 0 of (r >> 1) lands on bit 2 of the chunk index.  SQ_LDS_BANK_CONFLICT of the
// forward kernel: 0 (profiles/r03_pmc_attention.txt; the padded 144-byte rows of the register-staged tiles: 18 - 20 %).
constexpr int DTILE = 64 * 64;                // halves
__device__ uint4 g_attn_zero_page[8];         // source of the rows past the end of K / V
#ifdef DH_ATTN_STAMP
__device__ unsigned long long g_attn_ts_dkv[16];   // the same for k_attn_bwd_dkv (DKV_STAMP)
#define DKV_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_attn_ts_dkv[i] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_attn_ts[8];   // s_memtime stamps of wave 0 of block (0,0,0) (timing builds only, tools/attn_timeline.py)
#define ATTN_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_attn_ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#define DKV_STAMP(i) do { } while (0)
#endif
#ifdef DH_ATTN_STAMP_DQ      // the same stamps in k_attn_bwd_dq instead (one kernel at a time writes the array)
#undef DKV_STAMP
#define DKV_STAMP(i) do { } while (0)
#define DQ_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_attn_ts_dkv[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DQ_STAMP(i) do { } while (0)
#endif
__device__ __forceinline__ int dswz(int r) {
  const int y = r >> 1;
  return ((y & 1) << 2) | (y & 2) | ((y >> 2) & 1);
}
__device__ __forceinline__ void attn_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// The issue of a tile's DMA pieces is VALU / SALU work of the waves that also run the MFMAs, and on a SIMD the two add up: with the
// address of every piece computed per lane in 64 bits (row * pitch, tail select, readfirstlane of the LDS address) a piece cost
// ~38 instructions, ~150 cycles -- ~900 of a lone wave's 4 300 cycles per dK/dV tile by the s_memtime stamps
// (profiles/r04_ab_attn_dkv_dma.txt).  A tile that lies wholly inside its matrix needs none of it: the per-lane part of a piece's
// address (row-in-tile * pitch + swizzled chunk) is a 32-bit offset computed ONCE per kernel, the tile's base is a scalar, and
// global_load_lds takes exactly that pair (SGPR base + VGPR offset).  Ragged last tiles keep the per-lane form.
template <int QW> struct DmaPieces { unsigned voff[(16 + QW - 1) / QW]; };
template <int QW>
__device__ __forceinline__ void dma_pieces_init(DmaPieces<QW>& d, int wave_u, int lane, long ld_a, long ld_b) {
  const int d_row = lane >> 3;
  const int d_c0 = (lane & 7) ^ ((((lane >> 4) >> 1) & 1) << 1 | ((lane >> 4) & 1) << 2);
#pragma unroll
  for (int j = 0; j < (16 + QW - 1) / QW; ++j) {
    const int p = wave_u + j * QW, pp = p & 7;
    d.voff[j] = (unsigned)(((long)(8 * pp + d_row) * (p < 8 ? ld_a : ld_b) + ((d_c0 ^ (pp & 1)) << 3)) * 2);
  }
}
__device__ __forceinline__ void attn_dma16_s(const void* base, unsigned voff, unsigned lds) {
  // (the operands ARE wave-uniform; the readfirstlanes tell the compiler so -- an "s" constraint alone does not)
  const unsigned long long b64 = (unsigned long long)base;
  const unsigned long long sbase = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64) |
                                   ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32)) << 32);
  const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_dst) : "memory");
}
// the 16 pieces of a tile pair (A rows 0-63 then B rows 0-63) whose 64 rows all exist; base_a / base_b: row 0 of the tile (+ head
// column), lds_tile: byte address of the pair's buffer -- all wave-uniform
template <int QW>
__device__ __forceinline__ void dma_tile_full(const DmaPieces<QW>& d, int wave_u, const void* base_a, const void* base_b,
                                              unsigned lds_tile) {
#pragma unroll
  for (int j = 0; j < (16 + QW - 1) / QW; ++j) {
    const int p = wave_u + j * QW;
    if (p < 16) attn_dma16_s(p < 8 ? base_a : base_b, d.voff[j], lds_tile + (unsigned)(p * 1024));
  }
}
// acc = sum_kk mfma(A = rows (rowbase + lane&31) of a dense tile, B = register fragments); xk = hi ^ dswz(lane & 31)
template <class T>
__device__ __forceinline__ v16f dtile_times_frags(const unsigned short* tile, int rowbase, int ln, int xk, const uint4 (&f)[4]) {
  v16f acc = zero16();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const uint4 a = *reinterpret_cast<const uint4*>(&tile[(rowbase + ln) * 64 + ((xk ^ (2 * kk)) << 3)]);
    acc = Mma<T>::run(a, f[kk], acc);
  }
  return acc;
}
// transposed fragment of a dense tile: `trow` = tile + per-lane row offset, lo_c / up_c = the lane's chunk index of rows +0 / +8
// already XORed with their dswz (the column half dt flips bit 2 of both)
__device__ __forceinline__ uint4 dtr_frag(const unsigned short* trow, int lo_c, int up_c, int within, int dt, int r0) {
  typedef __attribute__((address_space(3))) v4s* lp;
  const unsigned short* a = trow + r0 * 64 + within;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a + ((lo_c ^ (4 * dt)) << 3)));
  const v4s up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a + 8 * 64 + ((up_c ^ (4 * dt)) << 3)));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), u2 = __builtin_bit_cast(uint2, up);
  return make_uint4(l2.x, l2.y, u2.x, u2.y);
}

// registers 8s..8s+7 of an accumulator -> B operand (16-bit)
template <class T>
__device__ __forceinline__ uint4 pack8(const v16f& p, int s);
template <>
__device__ __forceinline__ uint4 pack8<f16>(const v16f& p, int s) {
  uint4 o;
  o.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p[8 * s + 0], p[8 * s + 1]));
  o.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p[8 * s + 2], p[8 * s + 3]));
  o.z = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p[8 * s + 4], p[8 * s + 5]));
  o.w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p[8 * s + 6], p[8 * s + 7]));
  return o;
}
template <>
__device__ __forceinline__ uint4 pack8<bf16>(const v16f& p, int s) {
  bf16 o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)p[8 * s + j];
  return *reinterpret_cast<uint4*>(o);
}

// store a transposed accumulator pair (rows = d, col = lane's row) as row `row` of a [rows][ld] matrix.  The two lanes
// of a row (lane, lane ^ 32) own alternating 4-column groups: they swap two groups and write 16-byte chunks.
// Must be called by both lanes of a pair (the guard `row valid` is the same for both).
template <class T>
__device__ __forceinline__ void store_rows_t(T* base, long ld, long row, int col0, int hi, const v16f (&acc)[2], float mul) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    uint2 w[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      T o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = from_f32<T>(acc[dt][4 * g + i] * mul);
      w[g] = *reinterpret_cast<uint2*>(o);
    }
    const uint4 ca = half_exchange(w[0], w[1]), cb = half_exchange(w[2], w[3]);
    T* out = base + row * ld + col0 + dt * 32 + 8 * hi;
    *reinterpret_cast<uint4*>(out) = ca;
    *reinterpret_cast<uint4*>(out + 16) = cb;
  }
}

__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Workgroup barrier of the register-staged tile loops that orders LDS traffic ONLY.  __syncthreads() is a fence + s_barrier, and
// the fence is `s_waitcnt vmcnt(0) lgkmcnt(0)`: it drains the global loads of the NEXT tile that were issued a moment earlier
// precisely so that they would fly under this tile's MFMAs -- the prefetch never overlapped anything (round 4, s_memtime stamps
// of the dK/dV loop, tools/attn_timeline_dkv.py: 1 900 of a tile's 5 950 cycles sat in front of the second barrier).  The tiles'
// consumers are ordered by data dependence (commit_tile stores registers the compiler waits for by itself).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------- forward
// One block = 128 queries x KS key ranges: wave group ks (4 waves, own K/V tiles) runs the online softmax over
// keys [ks, ks+1) * ceil(tiles / KS) and the groups' (m, l, O) are merged through LDS at the end.  A single
// image has only Nq/128 * H query tiles (160 at 64x64 latents, 5 heads): splitting the keys inside the block
// puts KS waves on every SIMD instead of one, so one wave's softmax VALU work hides under another's MFMAs.
template <class T, int KS, int QW, bool DB>
__global__ void __launch_bounds__(64 * QW * KS) k_attn_fwd(const T* q, long ldq, const T* k, const T* v, long ldk, T* o, long ldo,
                                                       float* lse, int H, int Nq, int Nk, int causal) {
  // DB (key-split blocks of grids that fit the chip once or twice): the K/V tiles are double-buffered per wave group and go
  // global -> LDS directly (global_load_lds_dwordx4 into dense, source-swizzled tiles, see dswz above): tile it+1 is issued
  // when everyone has left tile it-1 and lands under the work on tile it, so ONE block-wide barrier per key tile (behind
  // vmcnt(0)) orders both (N = 4096, B = 1: 49.8 -> 45.3 us against the single buffer; no staging registers, no LDS store
  // pass, no bank conflicts, 128 KB).  Short loops (KS = 1: at most 7 tiles, the 77-key cross-attention) keep the single
  // register-staged buffer: there the second buffer's extra prologue step costs more than the barrier it saves (guided step
  // -0.7 % when every shape was double-buffered); so do big grids (batch 8: 1280 blocks), where one block per CU instead of two
  // loses (310 -> 352 us).
  ATTN_STAMP(0);
  constexpr int NBUF = DB ? 2 : 1;
  constexpr int TSZ = DB ? DTILE : TILE;        // halves per staged tile
  __shared__ __attribute__((aligned(1024))) unsigned short smem[KS * NBUF * 2 * TSZ];
  constexpr int GT = 64 * QW;                 // threads of one wave group (QW waves of 32 rows each)
  const int ks = threadIdx.x / GT, tid = threadIdx.x - ks * GT;
  const int lane = tid & 63, wave = tid >> 6, ln = lane & 31, hi = lane >> 5, t16 = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  unsigned short* sK0 = smem + ks * NBUF * 2 * TSZ;     // buffer p: K at sK0 + 2 p TSZ, V one tile behind it
  const long qrow = (long)blockIdx.x * (32 * QW) + wave * 32 + ln;
  const bool qok = qrow < Nq;
  uint4 qf[4];
  load_row_frags<T>(q + (long)b * Nq * ldq, ldq, qrow, qok, h * HD, hi, qf);
  frags_ready(qf);      // (DB as well: hipcc's own wait for these prologue loads, re-evaluated inside the loop, counted the LDS-DMA
                        //  pieces of the NEXT tile issued just before it -- vmcnt(1), vmcnt(0) -- and drained them on the spot: the
                        //  "double-buffered" forward had been loading synchronously since round 3)
  v16f oacc[2] = {zero16(), zero16()};
  float m_run = -INFINITY, l_run = 0.f;
  const T* kp = k + (long)b * Nk * ldk + h * HD;
  const T* vp = v + (long)b * Nk * ldk + h * HD;
  const int vt_off = (4 * hi + (t16 >> 2)) * TLD + 16 * ((lane >> 4) & 1) + 4 * (t16 & 3);
  const int tiles = (Nk + 63) >> 6, tps = (tiles + KS - 1) / KS;
  const int t_begin = ks * tps, t_end = min(t_begin + tps, tiles);
  TileRegs<GT> rk, rv;                           // (DB: unused)
  // DB: the 16 one-KiB pieces of a K/V tile pair (8 rows each) are issued by the group's waves in turn; lane l of a piece
  // fetches the chunk that belongs at position l & 7 of row l >> 3
  const unsigned lds_grp = DB ? (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) + ks * NBUF * 2 * TSZ * 2 : 0u;
  const int d_row = lane >> 3;
  const int d_c0 = (lane & 7) ^ ((((lane >> 4) >> 1) & 1) << 1 | ((lane >> 4) & 1) << 2);      // chunk for even pieces; odd: ^ 1
  const T* d_zero = reinterpret_cast<const T*>(g_attn_zero_page) + (lane & 7) * 8;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  DmaPieces<QW> dpc;
  if (DB) dma_pieces_init<QW>(dpc, wave_u, lane, ldk, ldk);
  auto dma_tile = [&](int key0, int buf) {
    if (key0 + 64 <= Nk) {                           // (wave-uniform) every row of the tile exists: scalar base + hoisted lane offsets
      dma_tile_full<QW>(dpc, wave_u, kp + (long)key0 * ldk, vp + (long)key0 * ldk, lds_grp + (unsigned)(buf * 2 * TSZ * 2));
      return;
    }
#pragma unroll
    for (int j = 0; j < (16 + QW - 1) / QW; ++j) {
      const int p = wave_u + j * QW;                 // wave-uniform
      if (p < 16) {
        const int pp = p & 7;
        const long r = (long)key0 + 8 * pp + d_row;
        const T* src = (p < 8 ? kp : vp) + r * ldk + ((d_c0 ^ (pp & 1)) << 3);
        attn_dma16(r < Nk ? src : d_zero, __builtin_amdgcn_readfirstlane(lds_grp + (unsigned)(buf * 2 * TSZ * 2 + p * 1024)));
      }
    }
  };
  if (t_begin < t_end) {
    if (DB) {
      dma_tile(t_begin * 64, 0);
    } else {
      fetch_tile<T, GT>(kp, ldk, t_begin * 64, Nk, rk, tid);
      fetch_tile<T, GT>(vp, ldk, t_begin * 64, Nk, rv, tid);
    }
  }
  // dense-tile fragment addressing (per lane, hoisted)
  const int xk = hi ^ dswz(ln);
  const int d_rl = 4 * hi + (t16 >> 2), d_cl = 2 * ((lane >> 4) & 1) + ((t16 & 3) >> 1);
  const int d_lo = d_cl ^ dswz(d_rl), d_up = d_cl ^ dswz(d_rl + 8), d_within = 4 * (t16 & 1);
  ATTN_STAMP(1);
  for (int it = 0; it < tps; ++it) {
    if (it == 1) ATTN_STAMP(2);
    const int k0 = (t_begin + it) * 64;
    const bool act = t_begin + it < t_end;        // uniform per wave group; barriers are block-wide
    if (DB) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of tile `it` have landed ...
      __syncthreads();                                             // ... and so have everyone's; tile it-1's buffer is free
    } else {
      lds_barrier();                                               // (the next tile's global loads stay in flight)
    }
    if (DB && act && t_begin + it + 1 < t_end) dma_tile(k0 + 64, (it + 1) & 1);
    if (!DB) {
      if (act) {
        tile_wait<GT>(rk, rv);
        commit_tile<GT>(rk, sK0, tid);
        commit_tile<GT>(rv, sK0 + TILE, tid);
      }
      lds_barrier();
    }
    if (!act) continue;
    if (!DB && t_begin + it + 1 < t_end) {            // next tile's loads fly under this tile's MFMAs
      fetch_tile<T, GT>(kp, ldk, k0 + 64, Nk, rk, tid);
      fetch_tile<T, GT>(vp, ldk, k0 + 64, Nk, rv, tid);
    }
    const unsigned short* sK = sK0 + (DB ? (it & 1) * 2 * TSZ : 0);
    const unsigned short* vt = sK + TSZ + (DB ? d_rl * 64 : vt_off);
    v16f s[2];
    if (DH_ATTN_ABL == 5) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[0][r] = __uint_as_float(qf[r & 3].x) * (float)(it + r); s[1][r] = __uint_as_float(qf[r & 3].y) * (float)(it - r); }
    } else {
    if (DB) {
      s[0] = dtile_times_frags<T>(sK, 0, ln, xk, qf);
      s[1] = dtile_times_frags<T>(sK, 32, ln, xk, qf);
    } else {
      s[0] = tile_times_frags<T>(sK, 0, ln, hi, qf);
      s[1] = tile_times_frags<T>(sK, 32, ln, hi, qf);
    }
    }
    if (causal) {                  // text tower: key j is visible to query i only for j <= i (forward only)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + t2 * 32 + acc_row(r, hi) > (int)qrow) s[t2][r] = -INFINITY;
    }
    if (k0 + 64 > Nk) {            // ragged last tile: mask the keys past Nk
      // (the empty asm keeps this a real branch: flattened into selects it cost 32 compares + 32 selects + the key
      // index arithmetic in EVERY tile of a loop that is bound by its softmax VALU work)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + t2 * 32 + acc_row(r, hi) >= Nk) s[t2][r] = -INFINITY;
    }
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    mx = xor32_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = fast_exp2((m_run - m_new) * CEXP);
    const float mc = m_new * CEXP;
    float rs = 0.f;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = DH_ATTN_ABL == 1 ? __builtin_fmaf(s[t2][r], CEXP, -mc) : fast_exp2(__builtin_fmaf(s[t2][r], CEXP, -mc));
        s[t2][r] = p;
        rs += p;
      }
    rs = xor32_sum(rs);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const uint4 pf = pack8<T>(s[t2], st);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          if (DH_ATTN_ABL == 4) { oacc[dt][st] += __uint_as_float(pf.x) + __uint_as_float(pf.y) + __uint_as_float(pf.z) + __uint_as_float(pf.w); continue; }
          oacc[dt] = Mma<T>::run(DB ? dtr_frag(vt, d_lo, d_up, d_within, dt, t2 * 32 + 16 * st) : tr_frag(vt, dt * 32, t2 * 32 + 16 * st), pf,
                                 oacc[dt]);
        }
      }
  }
  ATTN_STAMP(3);
  if (KS > 1) {
    // pairwise merge of the key ranges: group ks + step hands (O, m, l) to group ks through LDS (f32,
    // [34 values][64 lanes] per wave: conflict-free), halving the number of live groups per round
    float* cb = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int step = KS / 2; step >= 1; step >>= 1) {
      __syncthreads();
      if (ks >= step && ks < 2 * step) {
        float* slot = cb + ((ks - step) * QW + wave) * (34 * 64) + lane;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) slot[(dt * 16 + r) * 64] = oacc[dt][r];
        slot[32 * 64] = m_run;
        slot[33 * 64] = l_run;
      }
      __syncthreads();
      if (ks < step) {
        const float* slot = cb + (ks * QW + wave) * (34 * 64) + lane;
        const float m2 = slot[32 * 64], l2 = slot[33 * 64];
        const float m_new = fmaxf(m_run, m2);
        const float a1 = fast_exp2((m_run - m_new) * CEXP), a2 = fast_exp2((m2 - m_new) * CEXP);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[dt][r] = oacc[dt][r] * a1 + slot[(dt * 16 + r) * 64] * a2;
        l_run = l_run * a1 + l2 * a2;
        m_run = m_new;
      }
    }
    if (ks != 0) return;
  }
  ATTN_STAMP(4);
  if (qok) {
    store_rows_t<T>(o + (long)b * Nq * ldo, ldo, qrow, h * HD, hi, oacc, 1.f / l_run);
    if (lse && hi == 0) lse[((long)b * H + h) * Nq + qrow] = (m_run * CEXP + log2f(l_run)) * LN2;
  }
#ifdef DH_ATTN_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATTN_STAMP(5);
#endif
}

template <class T>
__global__ void k_attn_delta(const T* o, long ldo, const T* d_o, long lddo, float* delta, int H, int Nq, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*H*Nq
  if (idx >= total) return;
  const long qrow = idx % Nq;
  const int h = (int)((idx / Nq) % H);
  const long b = idx / ((long)Nq * H);
  const T* po = o + (b * Nq + qrow) * ldo + h * HD;
  const T* pd = d_o + (b * Nq + qrow) * lddo + h * HD;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    uint4 a = *reinterpret_cast<const uint4*>(po + c * 8), d = *reinterpret_cast<const uint4*>(pd + c * 8);
    const T* av = reinterpret_cast<const T*>(&a);
    const T* dv = reinterpret_cast<const T*>(&d);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += to_f32<T>(av[i]) * to_f32<T>(dv[i]);
  }
  delta[idx] = s;
}

// ---------------------------------------------------------------------------- backward dQ
// Same block shape as the forward (128 queries x KS key ranges, dQ partial sums merged through LDS).
// delta = rowsum(dO * O) is computed here from the row fragments and written for the dK/dV kernel.
// DB (round 4, grids that fit the chip once or twice, like the forward): the K / V tiles go global -> LDS by LDS-DMA into dense
// source-swizzled tiles, double-buffered per key group -- no staging registers, no commit, ONE barrier per key tile.  The s_memtime
// stamps of the register-staged loop (tools/attn_timeline_dkv.py, DH_TIMELINE_KERNEL=dq) put 1 700 of a key tile's 6 000 cycles into
// wait-for-prefetch + commit + second barrier + fetch issue, phases in which no MFMA runs on any of the block's twelve waves.
template <class T, int KS, int QW, bool DB>
__global__ void __launch_bounds__(64 * QW * KS) k_attn_bwd_dq(const T* q, long ldq, const T* k, const T* v, long ldk, const T* o,
                                                          long ldo, const T* d_o, long lddo, const float* lse, float* delta,
                                                          T* dq, long lddq, int H, int Nq, int Nk) {
  constexpr int NBUF = DB ? 2 : 1;
  constexpr int TSZ = DB ? DTILE : TILE;        // halves per staged tile
  __shared__ __attribute__((aligned(1024))) unsigned short smem[KS * NBUF * 2 * TSZ];
  constexpr int GT = 64 * QW;                 // threads of one wave group (QW waves of 32 rows each)
  const int ks = threadIdx.x / GT, tid = threadIdx.x - ks * GT;
  const int lane = tid & 63, wave = tid >> 6, ln = lane & 31, hi = lane >> 5, t16 = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  unsigned short* sK0 = smem + ks * NBUF * 2 * TSZ;     // buffer p: K at sK0 + 2 p TSZ, V one tile behind it
  const long qrow = (long)blockIdx.x * (32 * QW) + wave * 32 + ln;
  const bool qok = qrow < Nq;
  uint4 qf[4], dof[4];
  load_row_frags<T>(q + (long)b * Nq * ldq, ldq, qrow, qok, h * HD, hi, qf);
  load_row_frags<T>(d_o + (long)b * Nq * lddo, lddo, qrow, qok, h * HD, hi, dof);
  float del_q = 0.f;
  {
    uint4 of[4];
    load_row_frags<T>(o + (long)b * Nq * ldo, ldo, qrow, qok, h * HD, hi, of);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const T* ov = reinterpret_cast<const T*>(&of[kk]);
      const T* dv = reinterpret_cast<const T*>(&dof[kk]);
#pragma unroll
      for (int i = 0; i < 8; ++i) del_q += to_f32<T>(ov[i]) * to_f32<T>(dv[i]);
    }
    del_q = xor32_sum(del_q);
    if (delta && qok && ks == 0 && hi == 0) delta[((long)b * H + h) * Nq + qrow] = del_q;      // NULL: already there
  }
  float lse_q = qok ? lse[((long)b * H + h) * Nq + qrow] * LOG2E : INFINITY;
  frags_ready(qf); frags_ready(dof);
  asm volatile("" : "+v"(lse_q), "+v"(del_q));
  v16f dqacc[2] = {zero16(), zero16()};
  const T* kp = k + (long)b * Nk * ldk + h * HD;
  const T* vp = v + (long)b * Nk * ldk + h * HD;
  const int kt_off = (4 * hi + (t16 >> 2)) * TLD + 16 * ((lane >> 4) & 1) + 4 * (t16 & 3);
  const int tiles = (Nk + 63) >> 6, tps = (tiles + KS - 1) / KS;
  const int t_begin = ks * tps, t_end = min(t_begin + tps, tiles);
  TileRegs<GT> rk, rv;                          // (DB: unused)
  // DB: the 16 one-KiB pieces of a K/V tile pair (8 rows each) are issued by the group's waves in turn (k_attn_fwd has the layout)
  const unsigned lds_grp = DB ? (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) + ks * NBUF * 2 * TSZ * 2 : 0u;
  const int d_row = lane >> 3;
  const int d_c0 = (lane & 7) ^ ((((lane >> 4) >> 1) & 1) << 1 | ((lane >> 4) & 1) << 2);      // chunk for even pieces; odd: ^ 1
  const T* d_zero = reinterpret_cast<const T*>(g_attn_zero_page) + (lane & 7) * 8;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  DmaPieces<QW> dpc;
  if (DB) dma_pieces_init<QW>(dpc, wave_u, lane, ldk, ldk);
  auto dma_tile = [&](int key0, int buf) {
    if (key0 + 64 <= Nk) {                           // (wave-uniform) every row of the tile exists: scalar base + hoisted lane offsets
      dma_tile_full<QW>(dpc, wave_u, kp + (long)key0 * ldk, vp + (long)key0 * ldk, lds_grp + (unsigned)(buf * 2 * TSZ * 2));
      return;
    }
#pragma unroll
    for (int j = 0; j < (16 + QW - 1) / QW; ++j) {
      const int p = wave_u + j * QW;                 // wave-uniform
      if (p < 16) {
        const int pp = p & 7;
        const long r = (long)key0 + 8 * pp + d_row;
        const T* src = (p < 8 ? kp : vp) + r * ldk + ((d_c0 ^ (pp & 1)) << 3);
        attn_dma16(r < Nk ? src : d_zero, __builtin_amdgcn_readfirstlane(lds_grp + (unsigned)(buf * 2 * TSZ * 2 + p * 1024)));
      }
    }
  };
  if (t_begin < t_end) {
    if (DB) {
      dma_tile(t_begin * 64, 0);
    } else {
      fetch_tile<T, GT>(kp, ldk, t_begin * 64, Nk, rk, tid);
      fetch_tile<T, GT>(vp, ldk, t_begin * 64, Nk, rv, tid);
    }
  }
  // dense-tile fragment addressing (per lane, hoisted)
  const int xk = hi ^ dswz(ln);
  const int d_rl = 4 * hi + (t16 >> 2), d_cl = 2 * ((lane >> 4) & 1) + ((t16 & 3) >> 1);
  const int d_lo = d_cl ^ dswz(d_rl), d_up = d_cl ^ dswz(d_rl + 8), d_within = 4 * (t16 & 1);
  DQ_STAMP(0);
  for (int it = 0; it < tps; ++it) {
    const int k0 = (t_begin + it) * 64;
    const bool act = t_begin + it < t_end;
    if (it == 2) DQ_STAMP(1);
    if (it == 3) DQ_STAMP(6);
    if (DB) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of tile `it` have landed ...
      __syncthreads();                                             // ... and so have everyone's; tile it-1's buffer is free
      if (it == 2) { DQ_STAMP(2); DQ_STAMP(8); DQ_STAMP(3); DQ_STAMP(4); }
      if (act && t_begin + it + 1 < t_end) dma_tile(k0 + 64, (it + 1) & 1);
      if (!act) continue;
    } else {
      lds_barrier();
      if (it == 2) DQ_STAMP(2);
      if (act) {
        tile_wait<GT>(rk, rv);
        if (it == 2) DQ_STAMP(8);
        commit_tile<GT>(rk, sK0, tid);
        commit_tile<GT>(rv, sK0 + TILE, tid);
      }
      if (it == 2) DQ_STAMP(3);
      lds_barrier();
      if (it == 2) DQ_STAMP(4);
      if (!act) continue;
      if (t_begin + it + 1 < t_end) {
        fetch_tile<T, GT>(kp, ldk, k0 + 64, Nk, rk, tid);
        fetch_tile<T, GT>(vp, ldk, k0 + 64, Nk, rv, tid);
      }
    }
    if (it == 2) DQ_STAMP(9);
    const unsigned short* sK = sK0 + (DB ? (it & 1) * 2 * TSZ : 0);
    const unsigned short* sV = sK + TSZ;
    const unsigned short* kt = sK + (DB ? d_rl * 64 : kt_off);
    const bool ragged = k0 + 64 > Nk;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      if (it == 2 && t2 == 1) DQ_STAMP(5);
      v16f s = DB ? dtile_times_frags<T>(sK, t2 * 32, ln, xk, qf) : tile_times_frags<T>(sK, t2 * 32, ln, hi, qf);
      const v16f dp = DB ? dtile_times_frags<T>(sV, t2 * 32, ln, xk, dof) : tile_times_frags<T>(sV, t2 * 32, ln, hi, dof);
      if (ragged) {                // keys past Nk (a real branch, see k_attn_fwd): their probabilities are zero
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + t2 * 32 + acc_row(r, hi) >= Nk) s[r] = -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(__builtin_fmaf(s[r], CEXP, -lse_q));
        s[r] = p * (dp[r] - del_q);            // the 1/sqrt(d) of dS (a power of two) is applied once, to the finished dQ
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const uint4 dsf = pack8<T>(s, st);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dqacc[dt] = Mma<T>::run(DB ? dtr_frag(kt, d_lo, d_up, d_within, dt, t2 * 32 + 16 * st) : tr_frag(kt, dt * 32, t2 * 32 + 16 * st), dsf,
                                  dqacc[dt]);
      }
    }
  }
  DQ_STAMP(7);
  if (KS > 1) {
    float* cb = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int step = KS / 2; step >= 1; step >>= 1) {
      __syncthreads();
      if (ks >= step && ks < 2 * step) {
        float* slot = cb + ((ks - step) * QW + wave) * (32 * 64) + lane;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) slot[(dt * 16 + r) * 64] = dqacc[dt][r];
      }
      __syncthreads();
      if (ks < step) {
        const float* slot = cb + (ks * QW + wave) * (32 * 64) + lane;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dqacc[dt][r] += slot[(dt * 16 + r) * 64];
      }
    }
    if (ks != 0) return;
  }
  if (qok) store_rows_t<T>(dq + (long)b * Nq * lddq, lddq, qrow, h * HD, hi, dqacc, SCALE);
}

// ------------------------------------------------------------------------- backward dK, dV
// 128 keys x KS query ranges per block; the groups' dK / dV partial sums are merged through LDS.
// (capping this kernel at 168 VGPRs -- three waves per SIMD -- spills 216 bytes per lane into the loop: guided step -5 %,
// profiles/r03_ab_dkv_occ.txt; a four-group variant needs the register diet first)
// DB (round 4, like k_attn_bwd_dq): the Q / dO tiles are dense source-swizzled tiles filled by LDS-DMA, double-buffered per query
// group -- no staging registers (24 - 32 VGPRs), no commit phase (1 150 of a tile's 4 800 cycles by the s_memtime stamps,
// profiles/r04_dkv_timeline_after_prefetch_fix.txt), ONE barrier per query tile, and none of the bank conflicts of the padded
// register-staged tiles (SQ_LDS_BANK_CONFLICT 0.21 of this kernel's LDS-active cycles, profiles/r04_pmc_gemm_sq.txt).  The tile's
// 64 lse / delta values still travel through a register of the group's first wave, into a statistics buffer per tile buffer.
template <class T, int KS, int QW, bool DB>
__global__ void __launch_bounds__(64 * QW * KS) k_attn_bwd_dkv(const T* q, long ldq, const T* k, const T* v, long ldk, const T* d_o,
                                                           long lddo, const float* lse, const float* delta, T* dk, T* dv,
                                                           long lddk, int H, int Nq, int Nk, float* qpart, int qchunks) {
  // qchunks > 1 (cross-attention: one key block per head, thousands of queries): blockIdx.x is a QUERY chunk and the
  // block leaves f32 partial dK / dV in qpart [chunk][b][h][dK|dV][32*QW keys][64]; k_attn_dkv_reduce sums the chunks
  constexpr int NBUF = DB ? 2 : 1;
  constexpr int TSZ = DB ? DTILE : TILE;        // halves per staged tile
  constexpr int GRP = NBUF * (2 * TSZ + 256);   // shorts per group: per buffer a Q tile, a dO tile, 64 lse + 64 delta (f32)
  __shared__ __attribute__((aligned(1024))) unsigned short smem[KS * GRP];
  constexpr int GT = 64 * QW;                 // threads of one wave group (QW waves of 32 rows each)
  const int ks = threadIdx.x / GT, tid = threadIdx.x - ks * GT;
  const int lane = tid & 63, wave = tid >> 6, ln = lane & 31, hi = lane >> 5, t16 = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  unsigned short* sQ = smem + ks * GRP;        // buffer p: Q at sQ + 2 p TSZ, dO one tile behind it; statistics behind all tiles
  unsigned short* sdO = sQ + TSZ;
  float* sLse = reinterpret_cast<float*>(sQ + NBUF * 2 * TSZ);      // buffer p: + 128 p
  float* sDel = sLse + 64;
  const int chunk = qchunks > 1 ? blockIdx.x : 0;
  const long krow = (long)(qchunks > 1 ? 0 : blockIdx.x) * (32 * QW) + wave * 32 + ln;
  const bool kok = krow < Nk;
  uint4 kf[4], vf[4];
  load_row_frags<T>(k + (long)b * Nk * ldk, ldk, krow, kok, h * HD, hi, kf);
  load_row_frags<T>(v + (long)b * Nk * ldk, ldk, krow, kok, h * HD, hi, vf);
  frags_ready(kf); frags_ready(vf);
  v16f dkacc[2] = {zero16(), zero16()}, dvacc[2] = {zero16(), zero16()};
  const T* qp = q + (long)b * Nq * ldq + h * HD;
  const T* dop = d_o + (long)b * Nq * lddo + h * HD;
  const int toff = (4 * hi + (t16 >> 2)) * TLD + 16 * ((lane >> 4) & 1) + 4 * (t16 & 3);
  const unsigned short* qt = sQ + toff;
  const unsigned short* dot = sdO + toff;
  const int tiles_all = (Nq + 63) >> 6;
  const int tpc = (tiles_all + qchunks - 1) / qchunks;                 // q tiles of this block's chunk
  const int c_end = min((chunk + 1) * tpc, tiles_all);
  const int tps = (tpc + KS - 1) / KS;
  const int t_begin = chunk * tpc + ks * tps, t_end = min(t_begin + tps, c_end);
  TileRegs<GT> rq, rdo;
  // the tile's 64 lse / delta values travel like the tile itself: fetched one tile ahead into a register of the group's first
  // wave, committed to LDS with the tile.  (Round 4: they used to be loaded between the two barriers of the tile -- a dependent
  // global load of ~1 900 cycles in front of the second barrier, a third of the 5 950-cycle tile by the s_memtime stamps of
  // tools/attn_timeline_dkv.py, profiles/r04_dkv_timeline.txt.)
  float lse_n = 0.f, del_n = 0.f;
  auto fetch_stats = [&](long q0n) {       // raw, unconditional loads (clamped row): nothing here may need the value yet
    if (tid < 64) {
      long qr = q0n + tid;
      qr = qr < Nq ? qr : Nq - 1;
      const float* pl = lse + ((long)b * H + h) * Nq + qr;
      const float* pd = delta + ((long)b * H + h) * Nq + qr;
      asm volatile("global_load_dword %0, %1, off" : "=&v"(lse_n) : "v"(pl) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=&v"(del_n) : "v"(pd) : "memory");
    }
  };
  // DB: the 16 one-KiB pieces of a Q / dO tile pair (8 rows each) are issued by the group's waves in turn (k_attn_fwd has the layout)
  const unsigned lds_grp = DB ? (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) + ks * GRP * 2 : 0u;
  const int d_row = lane >> 3;
  const int d_c0 = (lane & 7) ^ ((((lane >> 4) >> 1) & 1) << 1 | ((lane >> 4) & 1) << 2);      // chunk for even pieces; odd: ^ 1
  const T* d_zero = reinterpret_cast<const T*>(g_attn_zero_page) + (lane & 7) * 8;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  DmaPieces<QW> dpc;
  if (DB) dma_pieces_init<QW>(dpc, wave_u, lane, ldq, lddo);
  auto dma_tile = [&](int q0n, int buf) {
    if (q0n + 64 <= Nq) {                            // (wave-uniform) every row of the tile exists: scalar base + hoisted lane offsets
      dma_tile_full<QW>(dpc, wave_u, qp + (long)q0n * ldq, dop + (long)q0n * lddo, lds_grp + (unsigned)(buf * 2 * TSZ * 2));
      return;
    }
#pragma unroll
    for (int j = 0; j < (16 + QW - 1) / QW; ++j) {
      const int p = wave_u + j * QW;                 // wave-uniform
      if (p < 16) {
        const int pp = p & 7;
        const long r = (long)q0n + 8 * pp + d_row;
        const T* src = p < 8 ? qp + r * ldq + ((d_c0 ^ (pp & 1)) << 3) : dop + r * lddo + ((d_c0 ^ (pp & 1)) << 3);
        attn_dma16(r < Nq ? src : d_zero, __builtin_amdgcn_readfirstlane(lds_grp + (unsigned)(buf * 2 * TSZ * 2 + p * 1024)));
      }
    }
  };
  if (t_begin < t_end) {
    if (DB) {
      dma_tile(t_begin * 64, 0);
    } else {
      fetch_tile<T, GT>(qp, ldq, t_begin * 64, Nq, rq, tid);
      fetch_tile<T, GT>(dop, lddo, t_begin * 64, Nq, rdo, tid);
    }
    fetch_stats((long)t_begin * 64);
  }
  // dense-tile fragment addressing (per lane, hoisted)
  const int xk = hi ^ dswz(ln);
  const int d_rl = 4 * hi + (t16 >> 2), d_cl = 2 * ((lane >> 4) & 1) + ((t16 & 3) >> 1);
  const int d_lo = d_cl ^ dswz(d_rl), d_up = d_cl ^ dswz(d_rl + 8), d_within = 4 * (t16 & 1);
  DKV_STAMP(0);
  for (int it = 0; it < tps; ++it) {
    const int q0 = (t_begin + it) * 64;
    const bool act = t_begin + it < t_end;
    const int buf = DB ? (it & 1) : 0;
    if (it == 2) DKV_STAMP(1);
    if (it == 3) DKV_STAMP(6);
    if (DB) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of tile `it` (and its statistics) have landed ...
      asm volatile("" : "+v"(lse_n), "+v"(del_n));
      if (act && tid < 64) {
        const bool qin = q0 + tid < Nq;
        sLse[buf * 128 + tid] = qin ? lse_n * LOG2E : INFINITY;
        sDel[buf * 128 + tid] = qin ? del_n : 0.f;
      }
      __syncthreads();                                             // ... and so have everyone's; tile it-1's buffer is free
      if (it == 2) { DKV_STAMP(2); DKV_STAMP(8); DKV_STAMP(9); DKV_STAMP(3); DKV_STAMP(4); }
      if (act && t_begin + it + 1 < t_end) {
        dma_tile(q0 + 64, (it + 1) & 1);
        fetch_stats((long)q0 + 64);
      }
      if (!act) continue;
    } else {
      lds_barrier();
      if (it == 2) DKV_STAMP(2);
      if (act) {
        tile_wait<GT>(rq, rdo);
        if (it == 2) DKV_STAMP(8);
        asm volatile("" : "+v"(lse_n), "+v"(del_n));
        commit_tile<GT>(rq, sQ, tid);
        commit_tile<GT>(rdo, sdO, tid);
        if (it == 2) DKV_STAMP(9);
        if (tid < 64) {
          const bool qin = q0 + tid < Nq;
          sLse[tid] = qin ? lse_n * LOG2E : INFINITY;
          sDel[tid] = qin ? del_n : 0.f;
        }
        if (t_begin + it + 1 < t_end) {
          fetch_tile<T, GT>(qp, ldq, q0 + 64, Nq, rq, tid);
          fetch_tile<T, GT>(dop, lddo, q0 + 64, Nq, rdo, tid);
          fetch_stats((long)q0 + 64);
        }
      }
      if (it == 2) DKV_STAMP(3);
      lds_barrier();
      if (it == 2) DKV_STAMP(4);
      if (!act) continue;
    }
    const unsigned short* tQ = sQ + buf * 2 * TSZ;
    const unsigned short* tdO = tQ + TSZ;
    const unsigned short* qtr = DB ? tQ + d_rl * 64 : qt;          // transposed-read bases of this lane
    const unsigned short* dotr = DB ? tdO + d_rl * 64 : dot;
    const float* tLse = sLse + buf * 128;
    const float* tDel = sDel + buf * 128;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      if (it == 2 && t2 == 1) DKV_STAMP(5);
      v16f s = DB ? dtile_times_frags<T>(tQ, t2 * 32, ln, xk, kf) : tile_times_frags<T>(tQ, t2 * 32, ln, hi, kf);      // rows = queries, col = key
      const v16f dp = DB ? dtile_times_frags<T>(tdO, t2 * 32, ln, xk, vf) : tile_times_frags<T>(tdO, t2 * 32, ln, hi, vf);
      v16f ds;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 l4 = *reinterpret_cast<const float4*>(&tLse[t2 * 32 + 8 * g + 4 * hi]);
        const float4 d4 = *reinterpret_cast<const float4*>(&tDel[t2 * 32 + 8 * g + 4 * hi]);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * g + i;
          const float p = fast_exp2(__builtin_fmaf(s[r], CEXP, -lv[i]));
          s[r] = p;
          ds[r] = p * (dp[r] - dv4[i]);        // the 1/sqrt(d) is applied once, to the finished dK
        }
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const uint4 pf = pack8<T>(s, st), dsf = pack8<T>(ds, st);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dvacc[dt] = Mma<T>::run(DB ? dtr_frag(dotr, d_lo, d_up, d_within, dt, t2 * 32 + 16 * st) : tr_frag(dotr, dt * 32, t2 * 32 + 16 * st), pf,
                                  dvacc[dt]);
          dkacc[dt] = Mma<T>::run(DB ? dtr_frag(qtr, d_lo, d_up, d_within, dt, t2 * 32 + 16 * st) : tr_frag(qtr, dt * 32, t2 * 32 + 16 * st), dsf,
                                  dkacc[dt]);
        }
      }
    }
  }
  DKV_STAMP(7);
  if (KS > 1) {
    static_assert(KS <= 2, "the dK/dV merge buffer holds one pair of groups");
    float* cb = reinterpret_cast<float*>(smem);       // 4 waves x 32 x 64 f32 = 32 KiB <= KS * GRP * 2 bytes
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {            // dK, then dV
      v16f (&acc)[2] = pass == 0 ? dkacc : dvacc;
      __syncthreads();
      if (ks == 1) {
        float* slot = cb + wave * (32 * 64) + lane;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) slot[(dt * 16 + r) * 64] = acc[dt][r];
      }
      __syncthreads();
      if (ks == 0) {
        const float* slot = cb + wave * (32 * 64) + lane;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[dt][r] += slot[(dt * 16 + r) * 64];
      }
    }
    if (ks != 0) return;
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dkacc[dt][r] *= SCALE;
  if (qchunks > 1) {
    float* base = qpart + ((((size_t)chunk * gridDim.z + b) * H + h) * 2) * (32 * QW * 64) + (size_t)(wave * 32 + ln) * 64;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const v16f (&acc)[2] = which == 0 ? dkacc : dvacc;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(base + which * (32 * QW * 64) + dt * 32 + 8 * g + 4 * hi) =
              make_float4(acc[dt][4 * g], acc[dt][4 * g + 1], acc[dt][4 * g + 2], acc[dt][4 * g + 3]);
    }
    return;
  }
  if (kok) {
    store_rows_t<T>(dk + (long)b * Nk * lddk, lddk, krow, h * HD, hi, dkacc, 1.f);
    store_rows_t<T>(dv + (long)b * Nk * lddk, lddk, krow, h * HD, hi, dvacc, 1.f);
  }
}

// ------------------------------------------------------------------------------ launchers
// dk / dv [b][key][h*64 + d] = sum over the query chunks of the f32 partials (chunk order: deterministic)
template <class T>
__global__ void k_attn_dkv_reduce(const float* qpart, int qchunks, int rows_pad, T* dk, T* dv, long lddk, int B, int H, int Nk) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;        // over B*H*2*Nk*16 (4 d values each)
  const long total = (long)B * H * 2 * Nk * 16;
  if (idx >= total) return;
  const int d4 = (int)(idx & 15);
  long r = idx >> 4;
  const int key = (int)(r % Nk); r /= Nk;
  const int which = (int)(r & 1); r >>= 1;
  const int h = (int)(r % H);
  const int b = (int)(r / H);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = 0; c < qchunks; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(
        qpart + (((((size_t)c * B + b) * H + h) * 2 + which) * rows_pad + key) * 64 + d4 * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  T o[4] = {from_f32<T>(s.x), from_f32<T>(s.y), from_f32<T>(s.z), from_f32<T>(s.w)};
  T* out = (which ? dv : dk) + ((long)b * Nk + key) * lddk + h * HD + d4 * 4;
  *reinterpret_cast<uint2*>(out) = *reinterpret_cast<uint2*>(o);
}

// ---- launch shapes ------------------------------------------------------------------------------------------
// KS: key-range (query-range for dK/dV) split over wave groups inside a block (measured on MI355X, fp16, B=1 H=5
//     N=4096: fwd 97 -> 57 us at KS=4, dq+dkv 244 -> 160 us at KS=2; still ahead at 9216 keys and B=2, so the choice
//     depends on the loop length only)
// QW: waves of 32 rows per wave group, i.e. the block owns 32*QW rows.  One image has only N/32 * H row-waves
//     (640 at N=4096, H=5): the row tile is chosen so that the blocks cover the 256 CUs as evenly as possible
//     (128-row tiles = 160 blocks leave 96 CUs idle, 96-row tiles = 215 blocks).
static int attn_key_split(int tiles, int max_ks) {
  // (two groups from 4 tiles on -- the 16x16-latent self-attention, 256 keys: forward + backward 30.6 -> 21.7 us, guided step
  //  +0.5 %, profiles/r03_ab_attn_ks_short.txt; four groups from 4 or 8 tiles, two from 2: no further gain)
  int ks = tiles >= 16 ? 4 : tiles >= 4 ? 2 : 1;
#ifdef DH_TUNING
  static const int force = getenv("DH_ATTN_KS") ? atoi(getenv("DH_ATTN_KS")) : 0;
  if (force == 1 || force == 2 || force == 4) ks = force;
#endif
  return ks < max_ks ? ks : max_ks;
}
static int attn_row_waves(int rows, int hb, int loop_rows) {
#ifdef DH_TUNING
  static const int force = getenv("DH_ATTN_QW") ? atoi(getenv("DH_ATTN_QW")) : 0;
  if (force >= 1 && force <= 4) return force;
#endif
  const int min_qw = loop_rows > 1024 ? 2 : 1;        // a 32-row block would stream the whole K/V per 32 rows
  // grids that fill the chip twice over with 128-row blocks (batched edits): the big block, whatever the rounding of the last
  // round -- it streams K / V once per 128 rows instead of once per 64 (batch 8, N = 1024, H = 10: forward 52.9 -> 42.8 us with
  // one key group, forward + backward 222 -> 169 us; profiles/r05_attn_b8_sweep.txt)
  if ((long)cdiv(rows, 128) * hb >= 512) return 4;
  int best = 4;
  long best_cost = -1;
  for (int qw = 4; qw >= min_qw; --qw) {
    const long blocks = (long)cdiv(rows, 32 * qw) * hb;
    const long cost = ((blocks + 255) / 256) * qw;    // rounds over the CUs x work per block
    if (best_cost < 0 || cost < best_cost) { best = qw; best_cost = cost; }
  }
  return best;
}

template <class T, int KS, int QW>
static void attn_fwd_launch(int B, hipStream_t st, const void* q, long ldq, const void* k, const void* v, long ldk, void* o,
                            long ldo, float* lse, int H, int Nq, int Nk, int causal) {
  const dim3 grid(cdiv(Nq, 32 * QW), H, B);
  if constexpr (KS >= 2 && KS * QW < 16) {      // (the 16-wave block has no registers to spare for the second fetch)
    if ((long)grid.x * grid.y * grid.z <= 512) {
      hipLaunchKernelGGL((k_attn_fwd<T, KS, QW, true>), grid, dim3(64 * QW * KS), 0, st, (const T*)q, ldq, (const T*)k, (const T*)v,
                         ldk, (T*)o, ldo, lse, H, Nq, Nk, causal);
      return;
    }
  }
  hipLaunchKernelGGL((k_attn_fwd<T, KS, QW, false>), grid, dim3(64 * QW * KS), 0, st, (const T*)q, ldq, (const T*)k, (const T*)v,
                     ldk, (T*)o, ldo, lse, H, Nq, Nk, causal);
}
template <class T, int KS, int QW>
static void attn_dq_launch(int B, hipStream_t st, const void* q, long ldq, const void* k, const void* v, long ldk,
                           const void* o, long ldo, const void* d_o, long lddo, const float* lse, float* delta, void* dq,
                           long lddq, int H, int Nq, int Nk) {
  // (the 16-wave shape is never chosen, launch_attention_bwd_dq: at 128 registers per lane it would spill; not instantiated)
  if constexpr (KS * QW < 16) {
    const dim3 grid(cdiv(Nq, 32 * QW), H, B);
    if constexpr (KS >= 2) {      // key-split blocks of grids that fit the chip once or twice: LDS-DMA double buffer (as the forward)
      if ((long)grid.x * grid.y * grid.z <= 512) {
        hipLaunchKernelGGL((k_attn_bwd_dq<T, KS, QW, true>), grid, dim3(64 * QW * KS), 0, st, (const T*)q, ldq, (const T*)k, (const T*)v,
                           ldk, (const T*)o, ldo, (const T*)d_o, lddo, lse, delta, (T*)dq, lddq, H, Nq, Nk);
        return;
      }
    }
    hipLaunchKernelGGL((k_attn_bwd_dq<T, KS, QW, false>), grid, dim3(64 * QW * KS), 0, st, (const T*)q, ldq,
                       (const T*)k, (const T*)v, ldk, (const T*)o, ldo, (const T*)d_o, lddo, lse, delta, (T*)dq, lddq, H, Nq, Nk);
  }
}
#ifndef DH_DKV_DB
#define DH_DKV_DB 1       // 0: the register-staged tiles (same-box A/B builds, tools/lab.sh build-tuning with DH_DEFS=-DDH_DKV_DB=0)
#endif
template <class T, int KS, int QW>
static void attn_dkv_launch(int B, hipStream_t st, const void* q, long ldq, const void* k, const void* v, long ldk,
                            const void* d_o, long lddo, const float* lse, const float* delta, void* dk, void* dv, long lddk,
                            int H, int Nq, int Nk, float* qpart, int qchunks) {
  // (the kernel's registers hold it to one workgroup per CU whatever the LDS: the double-buffered form serves every grid)
  hipLaunchKernelGGL((k_attn_bwd_dkv<T, KS, QW, DH_DKV_DB != 0>), dim3(qchunks > 1 ? qchunks : cdiv(Nk, 32 * QW), H, B), dim3(64 * QW * KS), 0,
                     st, (const T*)q, ldq, (const T*)k, (const T*)v, ldk, (const T*)d_o, lddo, lse, delta, (T*)dk, (T*)dv, lddk,
                     H, Nq, Nk, qpart, qchunks);
  if (qchunks > 1) {
    const long total = (long)B * H * 2 * Nk * 16;
    hipLaunchKernelGGL((k_attn_dkv_reduce<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, qpart, qchunks, 32 * QW,
                       (T*)dk, (T*)dv, lddk, B, H, Nk);
  }
}

// (ks, qw) -> instantiation
#define DH_ATTN_QW(FN, T_, KS_, ...)                                   \
  do {                                                                  \
    if (qw == 4) FN<T_, KS_, 4>(__VA_ARGS__);                           \
    else if (qw == 3) FN<T_, KS_, 3>(__VA_ARGS__);                      \
    else if (qw == 2) FN<T_, KS_, 2>(__VA_ARGS__);                      \
    else FN<T_, KS_, 1>(__VA_ARGS__);                                   \
  } while (0)

void launch_attention_fwd(int dtype, const void* q, long ldq, const void* k, const void* v, long ldk, void* o, long ldo,
                          float* lse, int B, int H, int Nq, int Nk, hipStream_t st, int causal) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return;      // (the tile fetch clamps rows to rows_total - 1: never with zero rows)
  // causal: no key split.  A wave group whose whole key range lies after a query row would carry m = -inf into the merge
  // (exp2(-inf - -inf) = NaN); with one group the first tile always holds key 0, visible to every row
  int ks = attn_key_split((Nk + 63) / 64, causal ? 1 : 4);
  const int qw = attn_row_waves(Nq, H * B, Nk);
  // ... and on such grids a short key loop (< 64 tiles) is not split over wave groups: there are blocks enough, the merge only costs
  if ((long)cdiv(Nq, 128) * H * B >= 512 && (Nk + 63) / 64 < 64) ks = 1;
#define DH_ATTN_FWD(T_)                                                                         \
  do {                                                                                          \
    if (ks == 4) DH_ATTN_QW(attn_fwd_launch, T_, 4, B, st, q, ldq, k, v, ldk, o, ldo, lse, H, Nq, Nk, causal);       \
    else if (ks == 2) DH_ATTN_QW(attn_fwd_launch, T_, 2, B, st, q, ldq, k, v, ldk, o, ldo, lse, H, Nq, Nk, causal);  \
    else DH_ATTN_QW(attn_fwd_launch, T_, 1, B, st, q, ldq, k, v, ldk, o, ldo, lse, H, Nq, Nk, causal);               \
  } while (0)
  if (dtype == DH_DTYPE_F16) DH_ATTN_FWD(f16);
  else DH_ATTN_FWD(bf16);
#undef DH_ATTN_FWD
}

void launch_attention_delta(int dtype, const void* o, long ldo, const void* d_o, long lddo, float* delta, int B, int H,
                            int Nq, hipStream_t st) {
  const long total = (long)B * H * Nq;
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (dtype == DH_DTYPE_F16)
    hipLaunchKernelGGL((k_attn_delta<f16>), dim3(nb), dim3(256), 0, st, (const f16*)o, ldo, (const f16*)d_o, lddo, delta, H, Nq, total);
  else
    hipLaunchKernelGGL((k_attn_delta<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)o, ldo, (const bf16*)d_o, lddo, delta, H, Nq, total);
}

void launch_attention_bwd_dq(int dtype, const void* q, long ldq, const void* k, const void* v, long ldk, const void* o,
                             long ldo, const void* d_o, long lddo, const float* lse, float* delta, void* dq, long lddq,
                             int B, int H, int Nq, int Nk, hipStream_t st) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return;
  const int qw = attn_row_waves(Nq, H * B, Nk);
  // four key groups like the forward (12 waves per CU instead of 6 at N = 4096: the loop is VALU-bound and one to two waves
  // per SIMD leave the pipe idle across every LDS wait; guided step +0.45 %, profiles/r03_ab_dq_ks4.txt); the 16-wave block
  // of the 128-row tile would spill, it keeps two
  int ks = attn_key_split((Nk + 63) / 64, 4);
  if (ks == 4 && qw == 4) ks = 2;
#define DH_ATTN_DQ(T_)                                                                                                          \
  do {                                                                                                                           \
    if (ks == 4) DH_ATTN_QW(attn_dq_launch, T_, 4, B, st, q, ldq, k, v, ldk, o, ldo, d_o, lddo, lse, delta, dq, lddq, H, Nq, Nk); \
    else if (ks == 2) DH_ATTN_QW(attn_dq_launch, T_, 2, B, st, q, ldq, k, v, ldk, o, ldo, d_o, lddo, lse, delta, dq, lddq, H, Nq, Nk); \
    else DH_ATTN_QW(attn_dq_launch, T_, 1, B, st, q, ldq, k, v, ldk, o, ldo, d_o, lddo, lse, delta, dq, lddq, H, Nq, Nk);          \
  } while (0)
  if (dtype == DH_DTYPE_F16) DH_ATTN_DQ(f16);
  else DH_ATTN_DQ(bf16);
#undef DH_ATTN_DQ
}

void launch_attention_bwd_dkv(int dtype, const void* q, long ldq, const void* k, const void* v, long ldk, const void* d_o,
                              long lddo, const float* lse, const float* delta, void* dk, void* dv, long lddk, int B,
                              int H, int Nq, int Nk, hipStream_t st, float* scratch, size_t scratch_elems) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return;
  int ks = attn_key_split((Nq + 63) / 64, 2);
  int qw = attn_row_waves(Nk, H * B, Nq);
#ifdef DH_TUNING
  { static const int fq = getenv("DH_ATTN_DKV_QW") ? atoi(getenv("DH_ATTN_DKV_QW")) : 0;
    static const int fk = getenv("DH_ATTN_DKV_KS") ? atoi(getenv("DH_ATTN_DKV_KS")) : 0;
    if (fq >= 1 && fq <= 4 && Nk > 128) qw = fq;
    if ((fk == 1 || fk == 2) && Nk > 128) ks = fk; }
#endif
  // cross-attention shape (77 keys, thousands of queries): the key blocks alone are H*B workgroups; split the queries
  // over workgroups too and sum f32 partials.  (All keys in ONE block for that: the even-coverage rule above would pick two
  // 64-key blocks for 77 keys and leave 10 workgroups streaming 4096 queries each -- 72 us per layer in the null-text
  // backward, profiles/r03_invert_kernel_types.txt.)
  int qchunks = 1;
  float* qpart = nullptr;
  if (scratch && Nk <= 128 && Nq >= 512) qw = cdiv(Nk, 32);
  if (scratch && Nk <= 32 * qw && Nq >= 512) {
    const int tiles_all = (Nq + 63) / 64;
    qchunks = 256 / (H * B);
    if (qchunks > tiles_all / 2) qchunks = tiles_all / 2;
    while (qchunks > 1 && (size_t)qchunks * B * H * 2 * 32 * qw * 64 > scratch_elems) --qchunks;
    if (qchunks > 1) { qpart = scratch; ks = attn_key_split(cdiv(tiles_all, qchunks), 2); } else qchunks = 1;
  }
#define DH_ATTN_DKV(T_)                                                                                                         \
  do {                                                                                                                           \
    if (ks == 2) DH_ATTN_QW(attn_dkv_launch, T_, 2, B, st, q, ldq, k, v, ldk, d_o, lddo, lse, delta, dk, dv, lddk, H, Nq, Nk, qpart, qchunks);    \
    else DH_ATTN_QW(attn_dkv_launch, T_, 1, B, st, q, ldq, k, v, ldk, d_o, lddo, lse, delta, dk, dv, lddk, H, Nq, Nk, qpart, qchunks);             \
  } while (0)
  if (dtype == DH_DTYPE_F16) DH_ATTN_DKV(f16);
  else DH_ATTN_DKV(bf16);
#undef DH_ATTN_DKV
}
#undef DH_ATTN_QW

}  // namespace dh

#ifdef DH_ATTN_STAMP
extern "C" int dh_dbg_attn_stamps_dkv(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dh::g_attn_ts_dkv), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int dh_dbg_attn_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dh::g_attn_ts), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
