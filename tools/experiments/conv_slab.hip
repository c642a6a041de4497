// k_conv_slab: the stride-1 3x3 convolution of the B = 1 / B = 2 passes as an implicit GEMM whose A operand is staged ONCE per
// 64-channel chunk instead of once per tap, with LOADER waves beside three groups of multiplying waves.  Round 6.
//
//   D[m][n] = sum_{tap, c} X[pixel(m) + tap][c] W[n][(c / 64, tap, c % 64)]   (+ bias, + per-image vector, + residual)
//   (reference: the 3x3 convolutions of /root/reference/diffhandles/model/unet_2d_blocks.py:2216-2393 resnets / samplers and their
//    input gradients; same operands, same tiled / swizzled weight storage, same epilogue arithmetic as k_gemm_dma in gemm.hip)
//
// What the measurements behind it say (profiles/r06_gemm_tile_timeline.txt, r06_conv_slab_timeline.txt): in a B = 1 convolution
// launch a multiplying wave spends its K loop neither waiting for data (21-23 %) nor saturating the matrix pipe (256 of ~900 cycles
// per tile): an in-order wave pays ~44 cycles per MFMA, ~34 per ds_read_b128 and 60-100 per LDS-DMA piece it issues, whatever the
// order (burst, double-buffered or one read per MFMA: the same 1 260-1 340 cycles for 16 + 16), and the SIMD only fills up when
// SEVERAL waves run that chain side by side.  So:
//   * SLAB: for a 128-row output tile (a run of 128 consecutive pixels of one image) and a 64-channel chunk, the source pixels of all
//     nine taps are the 128 + 2 (W + 1) consecutive pixels around the run: staged once (33 KB at W = 64), read by the nine taps at nine
//     row offsets (pixel r, tap (ky, kx) -> slab row r + ky W + kx; left / right image border = a per-lane select of a zero row,
//     top / bottom = zero-filled by the DMA).  Per chunk 33 KB + 9 x 8 KB of weights instead of 9 x 24 KB through the texture path.
//   * LOADERS: four waves only issue DMA (the next chunk's slab, the weight tiles two steps ahead) and wait for it.
//   * THREE TAP GROUPS of four multiplying waves (2 x 2 wave tiles of 64 x 32): step j of a chunk multiplies taps 3 j, 3 j + 1,
//     3 j + 2, one per group -- three steps per chunk, three multiplying waves on every SIMD; the groups' partial sums are added
//     in group order through LDS at the end.
// One s_barrier of all sixteen waves per step; a loader waits with a counted vmcnt for what it issued BEFORE the step.
//
// Grid (M / 128, N / 64, K splits over chunks); split K writes the f32 slabs of gemm.hip's reduce kernels.
#include <hip/hip_ext.h>

#include "gemm_k.h"

namespace dh {

namespace {

constexpr int CS_WT = 8192;                             // one 64 x 64 weight tile
constexpr int CS_NG = 3;                                // tap groups = weight tiles per step = steps per chunk
constexpr int CS_RING = 3;                              // steps whose weight tiles are resident / in flight
// LDS: two slab buffers of SLABR rows (>= 128 + 2 W + 2, a multiple of 8: pieces of 8 rows; 128 bytes per pixel = 64 channels), a
// ring of CS_RING steps x three weight tiles, the zero row
template <int SLABR> struct CsLds {
  static constexpr int SLAB = SLABR * 128, RING_OFF = 2 * SLAB, ZERO_OFF = RING_OFF + CS_RING * CS_NG * CS_WT, TOTAL = ZERO_OFF + 256;
  static constexpr int NPL = (SLABR / 8 + 3) / 4;       // slab pieces per loader wave
  static_assert(SLABR % 8 == 0 && TOTAL <= 160 * 1024, "LDS budget");
};
constexpr unsigned CS_OOB = 0x80000000u;

typedef int cs_v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void cs_dma(unsigned voff, cs_v4i rsrc, unsigned soff, unsigned lds_dst) {
  unsigned tmp;
  soff = __builtin_amdgcn_readfirstlane(soff);
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 m0, %4\n\ts_mov_b32 %0, %3\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds"
               : "=&s"(tmp) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void cs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most n (0 .. HI; more is clamped, which only waits longer) vector-memory operations of this wave may stay outstanding
template <int LO, int HI> __device__ __forceinline__ void cs_wait_vm_dyn(int n) {
  if constexpr (LO == HI) { cs_wait_vm<LO>(); }
  else {
    constexpr int MID = (LO + HI + 1) / 2;
    if (n >= MID) cs_wait_vm_dyn<MID, HI>(n); else cs_wait_vm_dyn<LO, MID - 1>(n);
  }
}

}  // namespace

// measurement builds (-DDH_CS_TIMELINE, tools/conv_slab_timeline.py): s_memtime stamps of workgroup (0,0,0): wave 0 (a multiplying wave)
// into ts[0 .. 255], wave 12 (a loader) into ts[256 .. 511]
#ifdef DH_CS_TIMELINE
#define CS_STAMP(base, idx) do { if (p.pp_ts && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && (idx) < 256) p.pp_ts[(base) + (idx)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CS_STAMP(base, idx) do { } while (0)
#endif
static unsigned long long* g_cs_ts = nullptr;

template <class T, int SLABR>
__global__ void __launch_bounds__(1024) k_conv_slab(const GemmK p) {
  typedef CsLds<SLABR> LD;
  constexpr int CS_SLAB = LD::SLAB, CS_RING_OFF = LD::RING_OFF, CS_ZERO_OFF = LD::ZERO_OFF, CS_NPL = LD::NPL;
  constexpr int LOOK = CS_RING - 1;                      // a step's weight tiles are issued LOOK steps ahead
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LD::TOTAL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 64;
  const int Wimg = p.Win, HW = p.Hin * p.Win;
  const int b = m0 / HW, p0 = m0 - b * HW;                 // image of the tile, first pixel of its run inside the image
  const int chunks = p.Cin >> 6, cps = p.k_per_split / 576;
  const int cbeg = blockIdx.z * cps;
  int cend = cbeg + cps;
  if (cend > chunks) cend = chunks;
  const int nchunks = cend - cbeg, nsteps = nchunks * CS_NG;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);

  if (tid < 64) reinterpret_cast<unsigned*>(smem + CS_ZERO_OFF)[tid] = 0u;      // the zero row (left / right image border)

  if (wave >= 12) {
    // ------------------------------------------------------------------------------------------------------------ loaders
    const int lw = wave - 12;
    cs_v4i ra, rw;
    ra[0] = (int)(unsigned)(size_t)p.A; ra[1] = (int)(((size_t)p.A >> 32) & 0xffff); ra[2] = (int)p.pp_a_bytes; ra[3] = 0x00020000;
    rw[0] = (int)(unsigned)(size_t)p.W; rw[1] = (int)(((size_t)p.W >> 32) & 0xffff); rw[2] = (int)p.pp_w_bytes; rw[3] = 0x00020000;
    const int rows_needed = 128 + 2 * Wimg + 2;
    const int npieces = (rows_needed + 7) >> 3;
    // slab piece q = lw + 4 k of this wave: 8 slab rows, lane -> (row 8 q + lane / 8, 16-byte chunk lane % 8, swizzled on the source side)
    unsigned sv[CS_NPL];
#pragma unroll
    for (int k = 0; k < CS_NPL; ++k) {
      const int q = lw + 4 * k, s = q * 8 + (lane >> 3);
      const int qpix = p0 - Wimg - 1 + s;
      const bool ok = q < npieces && s < rows_needed && qpix >= 0 && qpix < HW;
      const int lchunk = (lane & 7) ^ ((s >> 1) & 7);
      sv[k] = ok ? (unsigned)(b * HW + qpix) * (unsigned)((int)p.lda * 2) + (unsigned)lchunk * 16u : CS_OOB;
    }
    const unsigned wv = lane * 16;
    const int KT = p.K >> 6;
    const unsigned wbase = (unsigned)(blockIdx.y * KT) * 8192u;
    int issued = 0;
    // slab pieces k in [klo, khi) of chunk c (absolute) into slab buffer sb
    auto issue_slab = [&](int c, int sb, int klo, int khi) {
#pragma unroll
      for (int k = 0; k < CS_NPL; ++k) {
        if (k >= klo && k < khi && lw + 4 * k < npieces) {
          cs_dma(sv[k], ra, (unsigned)c * 128u, lds0 + sb * CS_SLAB + (lw + 4 * k) * 1024);
          ++issued;
        }
      }
    };
    // the three weight tiles of step S (taps 3 j .. 3 j + 2 of its chunk) into ring slot S % 3: 24 pieces, six per loader wave
    auto issue_w = [&](int S) {
      const int cr = S / CS_NG, j = S - cr * CS_NG;
      const unsigned so = wbase + (unsigned)((cbeg + cr) * 9 + 3 * j) * 8192u;       // the step's tiles are consecutive K tiles
      const unsigned dst = lds0 + CS_RING_OFF + (S % CS_RING) * (CS_NG * CS_WT);
#pragma unroll
      for (int h = 0; h < 6; ++h) {
        const int pc = 6 * lw + h;                        // piece 0 .. 23 of the 24-KB run
        cs_dma(wv, rw, so + pc * 1024, dst + pc * 1024);
        ++issued;
      }
    };
    issue_slab(cbeg, 0, 0, CS_NPL);
#pragma unroll
    for (int S = 0; S < LOOK; ++S)
      if (S < nsteps) issue_w(S);
    if (lw == 0) CS_STAMP(256, 0);
    cs_wait_vm<0>();
    if (lw == 0) CS_STAMP(256, 1);
    __builtin_amdgcn_s_barrier();
    // The tiles of step S + 1 went out during step S - 1 and must have landed at the end of step S: what may stay outstanding
    // there is what this wave issued DURING step S.
    for (int S = 0; S < nsteps; ++S) {
      const int before = issued;
      if (S + LOOK < nsteps) issue_w(S + LOOK);
      const int cr = S / CS_NG, j = S - cr * CS_NG;
      // the next chunk's slab goes out over the first two steps of this one (the counted wait at the end of its third step then
      // covers it); buffer (cr + 1) & 1 was last read in the previous chunk
      constexpr int PER = (CS_NPL + 1) / 2;
      if (cr + 1 < nchunks && j < 2) issue_slab(cbeg + cr + 1, (cr + 1) & 1, j * PER, (j + 1) * PER);
      if (lw == 0) CS_STAMP(256, 2 + 3 * S);          // pieces of this step issued
      cs_wait_vm_dyn<0, 15>(issued - before);
      if (lw == 0) CS_STAMP(256, 3 + 3 * S);          // counted wait passed
      __builtin_amdgcn_s_barrier();
      if (lw == 0) CS_STAMP(256, 4 + 3 * S);          // barrier passed
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------------------------- consumers
  const int grp = wave >> 2, wrow = (wave >> 1) & 1, wcol = wave & 1, ln = lane & 31, hi = lane >> 5;
  // rows of this lane: r_i = 64 wrow + 32 i + ln (i = 0, 1); left / right border flags of their pixels
  int r_[2];
  bool xl[2], xr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    r_[i] = wrow * 64 + i * 32 + ln;
    const int pix = p0 + r_[i], x = pix - (pix / Wimg) * Wimg;
    xl[i] = x == 0; xr[i] = x == Wimg - 1;
  }
  unsigned fb[4];                                         // W fragment offsets inside a weight tile (columns 32 wcol + ln), per k-step
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fb[kk] = (unsigned)((wcol * 32 + ln) * 128) + (unsigned)(((2 * kk + hi) ^ ((ln >> 1) & 7)) << 4);
  v16f acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (wave == 0) CS_STAMP(0, 0);
  __syncthreads();                                        // (the zero row is an LDS store of this side: a fenced barrier)
  if (wave == 0) CS_STAMP(0, 1);
  for (int S = 0; S < nsteps; ++S) {
    const int cr = S / CS_NG, j3 = S - cr * CS_NG;
    const int tap = 3 * j3 + grp;
    const int ky = j3, kx = grp;                            // tap = 3 ky + kx
    const unsigned wt = (unsigned)(CS_RING_OFF + ((S % CS_RING) * CS_NG + grp) * CS_WT);
    const unsigned slab_o = (unsigned)((cr & 1) * CS_SLAB);
    unsigned ab[2], sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int s2 = r_[i] + ky * Wimg + kx;
      const bool z = (kx == 0 && xl[i]) || (kx == 2 && xr[i]);
      ab[i] = z ? (unsigned)CS_ZERO_OFF : slab_o + (unsigned)s2 * 128u;
      sw[i] = z ? 0u : (unsigned)((s2 >> 1) & 7);
    }
    (void)tap;
    // half sets of fragments: P = k-steps 0, 1 (burst at the top of the step), Q = k-steps 2, 3 (one read between the MFMAs on P)
    uint4 pw[2], px[2][2], qw[2], qx[2][2];
    auto read_half = [&](int kk0, int y, uint4 (&fw)[2], uint4 (&fx)[2][2]) {      // y = 0 .. 5: k-step kk0 + y / 3, then W, A0, A1
      const int kq = y / 3, q = y - 3 * kq, kk = kk0 + kq;
      if (q == 0) fw[kq] = *reinterpret_cast<const uint4*>(smem + wt + fb[kk]);
      else fx[kq][q - 1] = *reinterpret_cast<const uint4*>(smem + ab[q - 1] + ((((unsigned)(2 * kk + hi)) ^ sw[q - 1]) << 4));
    };
#if !(defined(DH_CS_ABL) && DH_CS_ABL == 1)
#pragma unroll
    for (int y = 0; y < 6; ++y) read_half(0, y, pw, px);
#else
    pw[0] = pw[1] = qw[0] = qw[1] = make_uint4(S, lane, 1, 2);
    px[0][0] = px[0][1] = px[1][0] = px[1][1] = qx[0][0] = qx[0][1] = qx[1][0] = qx[1][1] = make_uint4(S, lane, 3, 4);
#endif
    __builtin_amdgcn_sched_barrier(0);
    // MFMAs on P (4) with the six reads of Q between them, then the MFMAs on Q
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[i] = Mfma<T>::run(pw[kq], px[kq][i], acc[i]);
#if !(defined(DH_CS_ABL) && DH_CS_ABL == 1)
        const int x = kq * 2 + i;
        read_half(2, x, qw, qx);
        if (x >= 2) read_half(2, x + 2, qw, qx);
#endif
      }
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = Mfma<T>::run(qw[kq], qx[kq][i], acc[i]);
    if (wave == 0) CS_STAMP(0, 2 + 2 * S);             // reads and MFMAs of the step issued
    __builtin_amdgcn_s_barrier();
    if (wave == 0) CS_STAMP(0, 3 + 2 * S);             // barrier passed
  }

  // ---- add the tap groups (1, 2 -> 0, in that order) through the idle slabs: [group - 1][wave in group][value][lane] f32 ----
  {
    float* cb = reinterpret_cast<float*>(smem);
    const int wig = wave & 3;
    if (grp > 0) {
      float* slot = cb + ((grp - 1) * 4 + wig) * (32 * 64) + lane;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[(i * 16 + r) * 64] = acc[i][r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int og = 0; og < 2; ++og) {
      const float* slot = cb + (og * 4 + wig) * (32 * 64) + lane;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] += slot[(i * 16 + r) * 64];
    }
  }

  // ---- epilogue (waves 0 .. 3: rows 64 wrow + 32 i + ln, columns n0 + 32 wcol + 8 g + 4 hi .. + 3) ----
  typedef T T4 __attribute__((ext_vector_type(4)));
  const int nb = n0 + wcol * 32;
  if (p.splits > 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wrow * 64 + i * 32 + ln;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(p.partial + ((size_t)blockIdx.z * p.M + m) * p.N + nb + 8 * g + 4 * hi) =
            make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
    }
    return;
  }
  // per-column vectors once, before the first store (bias + the per-image vector: the tile lies inside one image)
  float4 cv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = nb + 8 * g + 4 * hi;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) v = *reinterpret_cast<const float4*>(p.bias + n);
    if (p.rowvec) {
      const float4 rv = *reinterpret_cast<const float4*>(p.rowvec + (size_t)b * p.rowvec_ld + n);
      v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
    }
    cv[g] = v;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wrow * 64 + i * 32 + ln;
    uint2 rres[4];
    if (p.R) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        rres[g] = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(p.R) + (size_t)m * p.ldr + nb + 8 * g + 4 * hi);
    }
    uint2 w[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v0 = acc[i][4 * g] + cv[g].x, v1 = acc[i][4 * g + 1] + cv[g].y;
      float v2 = acc[i][4 * g + 2] + cv[g].z, v3 = acc[i][4 * g + 3] + cv[g].w;
      if (p.R) {
        const T4 rv = __builtin_bit_cast(T4, rres[g]);
        v0 += to_f32<T>(rv[0]); v1 += to_f32<T>(rv[1]); v2 += to_f32<T>(rv[2]); v3 += to_f32<T>(rv[3]);
      }
      T4 o;
      o[0] = from_f32<T>(v0); o[1] = from_f32<T>(v1); o[2] = from_f32<T>(v2); o[3] = from_f32<T>(v3);
      w[g] = __builtin_bit_cast(uint2, o);
    }
    const uint4 ca = half_exchange(w[0], w[1]), cbv = half_exchange(w[2], w[3]);
    T* out = reinterpret_cast<T*>(p.C) + (size_t)m * p.ldc + nb + 8 * hi;
    *reinterpret_cast<uint4*>(out) = ca;
    *reinterpret_cast<uint4*>(out + 16) = cbv;
  }
}

// Can this launch run on k_conv_slab?  Stride-1 3x3, 64-channel chunks, output tiles of whole 128-pixel runs of one image, 64-column
// tiles, the plain epilogue (bias, per-image vector, residual) with 16-byte stores, descriptors below 2 GiB, image width <= 96.
bool conv_slab_eligible(const GemmK& k) {
  if (!(k.mode == A_CONV3 && k.stride == 1 && k.up == 0 && k.pad == 1)) return false;
  if (k.ln_s || k.act_silu || k.glu_y || k.glub_x) return false;
  if (k.Hin != k.Hout || k.Win != k.Wout || k.Hin <= 0 || k.Win <= 0 || k.Win > 96) return false;
  const int HW = k.Hin * k.Win;
  if (HW % 128 || k.M % HW || k.N % 64 || k.Cin % 64 || k.K != 9 * k.Cin) return false;
  if (!k.C || k.ldc % 8 || ((size_t)k.C & 15) || k.lda % 8 || ((size_t)k.A & 15) || ((size_t)k.W & 15)) return false;
  if (k.R && (k.ldr % 4 || ((size_t)k.R & 7))) return false;
  if (k.bias && ((size_t)k.bias & 15)) return false;
  if (k.rowvec && (k.rowvec_ld % 4 || ((size_t)k.rowvec & 15) || k.rows_per_batch != HW)) return false;
  const size_t ab = (size_t)(k.M / HW) * HW * (size_t)k.lda * 2, wb = (size_t)k.N * k.K * 2;
  if (ab >= 0x7ff00000ull || wb >= 0x7ff00000ull) return false;
  return true;
}

// splits: K splits over whole 64-channel chunks (the caller's choice is rounded to that); k.splits / k.k_per_split are set here
// (in the caller's block too: the reduce kernels read them)
void launch_conv_slab(int dtype, GemmK& k, int splits, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const int chunks = k.Cin / 64;
  if (splits < 1) splits = 1;
  if (splits > chunks) splits = chunks;
  const int cps = cdiv(chunks, splits);
  splits = cdiv(chunks, cps);
  k.splits = splits;
  k.k_per_split = cps * 576;
  const int HW = k.Hin * k.Win;
  k.pp_a_bytes = (unsigned)((size_t)(k.M / HW) * HW * (size_t)k.lda * 2);
  k.pp_w_bytes = (unsigned)((size_t)k.N * k.K * 2);
  k.pp_ts = g_cs_ts;
  dim3 grid(k.M / 128, k.N / 64, splits);
#define DH_CS_LAUNCH(KERNEL)                                                          \
  do {                                                                                \
    if (e0) hipExtLaunchKernelGGL(KERNEL, grid, dim3(1024), 0, st, e0, e1, 0, k);      \
    else hipLaunchKernelGGL(KERNEL, grid, dim3(1024), 0, st, k);                       \
  } while (0)
  // image width <= 64: 264-row slabs; up to 96 (768 x 768 images): 328-row slabs
  if (dtype == DH_DTYPE_F16) { if (k.Win <= 64) DH_CS_LAUNCH((k_conv_slab<f16, 264>)); else DH_CS_LAUNCH((k_conv_slab<f16, 328>)); }
  else { if (k.Win <= 64) DH_CS_LAUNCH((k_conv_slab<bf16, 264>)); else DH_CS_LAUNCH((k_conv_slab<bf16, 328>)); }
#undef DH_CS_LAUNCH
}

}  // namespace dh

// measurement hook: device buffer (512 x u64) the next k_conv_slab launches of a -DDH_CS_TIMELINE build stamp their timeline into
extern "C" int dh_dbg_conv_slab_timeline(unsigned long long* ts) {
  dh::g_cs_ts = ts;
  return DH_OK;
}
