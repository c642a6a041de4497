"""CPU: bank arithmetic of the attention forward's dense LDS-DMA tiles (csrc/attention.hip, `dswz`): the claims in the kernel's
comments checked against the LDS rules of the MI355X guide (64 banks of 4 bytes; ds_read_b128 serves the lane groups
{0-3,12-15,20-27} / {4-11,16-19,28-31} / the same + 32, one LDS cycle each; ds_read_b64_tr_b16 serves lanes 0-31, then 32-63).
The measured counterpart is SQ_LDS_BANK_CONFLICT = 0 in profiles/r03_pmc_attention.txt."""


def dswz(r):
    y = r >> 1
    return ((y & 1) << 2) | (y & 2) | ((y >> 2) & 1)


def banks(byte_addr, nbytes):
    return {((byte_addr + 4 * i) // 4) % 64 for i in range(nbytes // 4)}


B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def test_dswz_is_a_permutation_of_the_row_classes():
    assert sorted(dswz(2 * c) for c in range(8)) == list(range(8))
    assert all(dswz(r) == dswz(r ^ 1) for r in range(64))                    # rows 2c, 2c+1 share a class (they sit in different bank halves)
    assert all(dswz(r) == dswz(r + 16) for r in range(48))                   # the swizzle has period 16 rows


def test_row_fragment_reads_are_conflict_free():
    """dtile_times_frags: lane (ln = lane & 31, hi = lane >> 5) reads 16 bytes of row rowbase + ln at chunk (2 kk + hi) ^ dswz(row)."""
    for rowbase in (0, 32):
        for kk in range(4):
            for grp in B128_GROUPS:
                seen = set()
                for lane in grp:
                    ln, hi = lane & 31, lane >> 5
                    row = rowbase + ln
                    b = banks(row * 128 + (((2 * kk + hi) ^ dswz(row)) << 4), 16)
                    assert not (seen & b), (rowbase, kk, lane)
                    seen |= b
                assert len(seen) == 64


def test_transposed_reads_are_conflict_free():
    """dtr_frag: lane l (t = l & 15, g = (l >> 4) & 1, hi = l >> 5) reads 8 bytes of row r0 + 4 hi + (t >> 2) [+ 8] at column
    32 dt + 16 g + 4 (t & 3) halves; 32 lanes per LDS cycle."""
    for r0 in (0, 16, 32, 48):
        for dt in (0, 1):
            for up in (0, 8):
                for half in (range(0, 32), range(32, 64)):
                    seen = set()
                    for lane in half:
                        t, g, hi = lane & 15, (lane >> 4) & 1, lane >> 5
                        row = r0 + 4 * hi + (t >> 2) + up
                        col = 32 * dt + 16 * g + 4 * (t & 3)
                        b = banks(row * 128 + (((col >> 3) ^ dswz(row)) << 4) + 2 * (col & 7), 8)
                        assert not (seen & b), (r0, dt, up, lane)
                        seen |= b
                    assert len(seen) == 64


def test_dma_pieces_cover_a_tile_exactly_once():
    """dma_tile: lane l of one-KiB piece pp lands at LDS row 8 pp + (l >> 3), position l & 7, and fetches chunk
    (l & 7) ^ dswz(row) -- every (row, chunk) of the 64 x 8 tile once; the per-lane chunk formula of the kernel
    (d_c0 ^ (pp & 1)) equals it."""
    got = set()
    for pp in range(8):
        for lane in range(64):
            row = 8 * pp + (lane >> 3)
            chunk = (lane & 7) ^ dswz(row)
            l4 = lane >> 4
            d_c0 = (lane & 7) ^ ((((l4 >> 1) & 1) << 1) | ((l4 & 1) << 2))
            assert chunk == d_c0 ^ (pp & 1), (pp, lane)
            got.add((row, chunk))
    assert len(got) == 512


def test_gemm_fragment_reads_are_conflict_free():
    """csrc/gemm.hip: staged rows are 128 bytes, chunk c of row r at c ^ ((r >> 1) & 7) (also the tiled weight layout, wt_index);
    a fragment read is 16 bytes of row base + (lane & 31) at chunk 2 kk + (lane >> 5)."""
    for base in (0, 32, 64, 96):
        for kk in range(4):
            for grp in B128_GROUPS:
                seen = set()
                for lane in grp:
                    ln, hi = lane & 31, lane >> 5
                    row = base + ln
                    b = banks(row * 128 + (((2 * kk + hi) ^ ((row >> 1) & 7)) << 4), 16)
                    assert not (seen & b), (base, kk, lane)
                    seen |= b
                assert len(seen) == 64


def test_conv_k_index_is_a_bijection_with_taps_adjacent():
    """csrc/unet_kernels.h conv_k_index: (tap, channel) -> K index, 64-channel chunks outermost, the nine taps of a chunk on
    consecutive 64-wide K tiles."""
    C = 320
    idx = {(t, c): (c >> 6) * 576 + t * 64 + (c & 63) for t in range(9) for c in range(C)}
    assert sorted(idx.values()) == list(range(9 * C))
    for t in range(9):
        for c0 in range(0, C, 64):
            tiles = {idx[(t, c)] >> 6 for c in range(c0, c0 + 64)}
            assert tiles == {(c0 >> 6) * 9 + t}


def _wt_index(n, k, K):
    r, c = n & 63, (k & 63) >> 3
    return ((n >> 6) * (K >> 6) + (k >> 6)) * 4096 + r * 64 + ((c ^ ((r >> 1) & 7)) << 3) + (k & 7)


def test_gemm_128x160_tile_pieces_and_wave_blocks():
    """csrc/gemm.hip, the 128x160 tile (one wave column: BN is not a multiple of 64).  (i) W pieces: wave w fetches the 1-KiB
    pieces of rows 8 (w + 4 j) .. + 7 of the 160-row column tile, j = 0..4, from the tiled weight layout at
    ((n0 + row) >> 6) * KT * 4096 + ((n0 + row) & 63) * 64 + lane * 8 -- n0 = 160 * tile is 32 mod 64 on odd tiles -- and the DMA
    puts lane l of a piece at LDS row (row + (l >> 3)), 16-byte position l & 7: the staged tile must hold element (n, k) of the
    column tile at row n - n0, chunk ((k & 63) >> 3) ^ ((row >> 1) & 7), which is where the fragment reads look for it.
    (ii) the four waves' 1 x 5 blocks of 32 x 32 tile the 128 x 160 outputs exactly once."""
    K, kt = 256, 2                                  # K tile 2 of a 4-tile K
    KT = K >> 6
    for tile in range(4):                           # N = 640
        n0 = 160 * tile
        staged = {}
        for w in range(4):
            for j in range(5):
                row = 8 * (w + 4 * j)
                for lane in range(64):
                    src = ((n0 + row) >> 6) * KT * 4096 + ((n0 + row) & 63) * 64 + lane * 8 + kt * 4096
                    lds_row, pos = row + (lane >> 3), lane & 7
                    assert (lds_row, pos) not in staged
                    staged[(lds_row, pos)] = src
        assert len(staged) == 160 * 8
        for r in range(160):
            for c in range(8):
                k = 64 * kt + 8 * c
                pos = c ^ ((r >> 1) & 7)            # fragment reads: chunk c of row r sits at c ^ ((r >> 1) & 7)
                assert staged[(r, pos)] == _wt_index(n0 + r, k, K), (tile, r, c)
    owned = set()
    for w in range(4):                              # wm = w, wn = 0: rows 32 w .. + 31, five column blocks
        for j in range(5):
            for ln in range(32):
                for col in range(32):
                    key = (32 * w + ln, 32 * j + col)
                    assert key not in owned
                    owned.add(key)
    assert len(owned) == 128 * 160


def test_gemm_pp_policy_table():
    """gemm_pp_plan (csrc/gemm_pp.hip), queried on the host (dh_dbg_gemm_pp_plan: no device): which launches of the SD-2-depth passes
    run on the eight-wave ping-pong loop and with which tile -- pinned so that a threshold edit shows up as a diff of THIS table.
    (bm, bn, K splits); (0, 0, 0) = stays on k_gemm_dma."""
    import ctypes
    import pytest
    from diffusionhandles_amd import _lib
    L = _lib.lib()
    if "dh_dbg_gemm_pp_plan" in L.dh_missing_symbols:
        pytest.skip("library predates dh_dbg_gemm_pp_plan")

    def plan(M, N, K, conv=0, hw=0, glu=0, part=64 << 20):
        bm, bn, sp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(L.dh_dbg_gemm_pp_plan(M, N, K, conv, hw, glu, part, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp)))
        return bm.value, bn.value, sp.value

    # batch 8, 64 x 64 latents: convolutions of the three upper levels, dense projections
    assert plan(32768, 320, 2880, 1, 64) == (256, 160, 1)
    assert plan(8192, 640, 5760, 1, 32) == (128, 160, 1)          # 32 x 4 tiles of 256 rows would leave half the chip idle
    assert plan(2048, 1280, 11520, 1, 16) == (256, 160, 4)        # 64 tiles x 4 K splits
    assert plan(2048, 1280, 11520, 1, 16, part=0) == (0, 0, 0)    # ... only with a split-K workspace
    assert plan(32768, 320, 320) == (256, 160, 1) and plan(32768, 960, 320) == (256, 160, 1)
    assert plan(8192, 640, 640) == (128, 160, 1)
    assert plan(512, 1280, 11520, 1, 8) == (0, 0, 0)
    # the B = 16 CFG pass of batched edits, the 96 x 96 level at B = 2
    assert plan(65536, 320, 2880, 1, 64) == (256, 160, 1)
    assert plan(18432, 320, 2880, 1, 96) == (128, 160, 1)
    # a single edit stays on k_gemm_dma's tiles (40 - 160 workgroups) ...
    assert plan(4096, 320, 2880, 1, 64) == (0, 0, 0) and plan(4096, 320, 320) == (0, 0, 0) and plan(9216, 320, 2880, 1, 96) == (0, 0, 0)
    assert plan(4096, 640, 2880, 1, 64) == (0, 0, 0)              # (45 K tiles: the split-K branch starts at 64)
    assert plan(4096, 640, 5760, 1, 64) == (256, 160, 4)
    # ... except the GEGLU forward from M = 2048 on; the GEGLU backward from M = 32768 on
    assert plan(32768, 2560, 320, glu=1) == (256, 128, 1) and plan(4096, 2560, 320, glu=1) == (256, 128, 1)
    assert plan(1024, 5120, 640, glu=1) == (0, 0, 0)
    assert plan(32768, 1280, 320, glu=2) == (256, 128, 1) and plan(8192, 2560, 640, glu=2) == (0, 0, 0)
    # what the kernel cannot carry: K not a multiple of 64, N neither a multiple of 160 nor of 128
    assert plan(32768, 320, 352) == (0, 0, 0) and plan(32768, 192, 320) == (0, 0, 0)
