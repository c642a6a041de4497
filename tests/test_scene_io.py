"""Scene-directory readers (PNG / OpenEXR) and the real-scene golden vectors, CPU only.

The scene under tests/golden/scene_banana_fruits is a copy of DATA files of the reference's own test set
(test/data/photogen/banana_fruits); g11_scene.npz holds what the reference's depth_transform produced on it
(tools/make_golden_scene.py)."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

from diffusionhandles_amd import scene_io as S

SCENE = os.path.join(os.path.dirname(__file__), "golden", "scene_banana_fruits")


def _png_bytes(a, depth, ctype, filters):
    """Encode with a chosen filter per row (test-side forward filters)."""
    h = a.shape[0]
    rows = a.reshape(h, -1)
    if depth == 16:
        rows = a.astype(">u2").view(np.uint8).reshape(h, -1)
    bpp = max(1, {0: 1, 2: 3, 4: 2, 6: 4}[ctype] * depth // 8)
    raw, prev = b"", np.zeros(rows.shape[1], np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        ft = filters[y % len(filters)]
        if ft == 0:
            enc = cur
        elif ft == 1:
            enc = cur - left
        elif ft == 2:
            enc = cur - prev
        elif ft == 3:
            enc = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            enc = cur - pred
        raw += bytes([ft]) + (enc & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    w = a.shape[1]
    comp = zlib.compress(raw)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
            chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("depth,ctype,shape", [(8, 0, (13, 17)), (8, 2, (13, 17, 3)), (8, 6, (9, 11, 4)), (16, 0, (7, 9)),
                                               (16, 2, (7, 9, 3)), (8, 4, (6, 5, 2))])
def test_png_all_filters(tmp_path, depth, ctype, shape):
    rng = np.random.default_rng(depth + ctype)
    a = rng.integers(0, 1 << depth, shape).astype(np.uint16 if depth == 16 else np.uint8)
    p = tmp_path / "a.png"
    p.write_bytes(_png_bytes(a, depth, ctype, [0, 1, 2, 3, 4]))
    b = S.read_png(str(p))
    assert b.dtype == a.dtype and np.array_equal(a, b)


def test_png_write_read_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.random((20, 31, 3)).astype(np.float32)
    S.write_png(str(tmp_path / "x.png"), img)
    assert np.array_equal(S.read_png(str(tmp_path / "x.png")), (img * 255.0).astype(np.uint8))   # truncation, like save_image
    g = rng.integers(0, 256, (5, 4)).astype(np.uint8)
    S.write_png(str(tmp_path / "g.png"), g)
    assert np.array_equal(S.read_png(str(tmp_path / "g.png")), g)
    with pytest.raises(ValueError):
        (tmp_path / "bad.png").write_bytes(b"not a png")
        S.read_png(str(tmp_path / "bad.png"))


def _exr_bytes(chans, comp, decreasing=False):
    """chans: list of (name, array [H,W] of f16 / f32 / u32).  Scanline file, compression 0 (none), 2 (zips) or 3 (zip)."""
    names = sorted(chans)
    h, w = chans[names[0]].shape
    ptype = {np.dtype("uint32"): 0, np.dtype("float16"): 1, np.dtype("float32"): 2}

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(val)) + val

    chl = b"".join(n.encode() + b"\0" + struct.pack("<IB3xii", ptype[chans[n].dtype], 0, 1, 1) for n in names) + b"\0"
    hdr = struct.pack("<II", 20000630, 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp])) + \
        attr("dataWindow", "box2i", struct.pack("<4i", 0, 0, w - 1, h - 1)) + \
        attr("displayWindow", "box2i", struct.pack("<4i", 0, 0, w - 1, h - 1)) + \
        attr("lineOrder", "lineOrder", bytes([1 if decreasing else 0])) + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + \
        attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = {0: 1, 2: 1, 3: 16}[comp]
    chunks = []
    for y in range(0, h, lines):
        raw = b"".join(chans[n][r].astype(chans[n].dtype.newbyteorder("<")).tobytes()
                       for r in range(y, min(y + lines, h)) for n in names)
        data = raw
        if comp:
            t = np.frombuffer(raw, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]]).astype(np.int32)
            d = t.copy()
            d[1:] = (t[1:] - t[:-1] + 128 + 256) & 255
            z = zlib.compress(d.astype(np.uint8).tobytes())
            data = z if len(z) < len(raw) else raw
        chunks.append((y, data))
    order = chunks[::-1] if decreasing else chunks
    off0 = len(hdr) + 8 * len(chunks)
    offs, body = {}, b""
    for y, data in order:
        offs[y] = off0 + len(body)
        body += struct.pack("<ii", y, len(data)) + data
    return hdr + b"".join(struct.pack("<Q", offs[y]) for y, _ in chunks) + body


@pytest.mark.parametrize("comp", [0, 2, 3])
def test_exr_none_and_zip(tmp_path, comp):
    rng = np.random.default_rng(comp)
    yy, xx = np.mgrid[0:37, 0:29]
    ch = {"Y": (1.0 + 0.01 * yy + 0.02 * xx).astype(np.float16), "Z": rng.random((37, 29)).astype(np.float32),
          "id": rng.integers(0, 1 << 30, (37, 29)).astype(np.uint32)}
    p = tmp_path / "a.exr"
    p.write_bytes(_exr_bytes(ch, comp, decreasing=(comp == 3)))
    out = S.read_exr(str(p))
    assert np.array_equal(out["Y"], ch["Y"].astype(np.float32))
    assert np.array_equal(out["Z"], ch["Z"]) and np.array_equal(out["id"], ch["id"])
    assert np.array_equal(S.read_depth_exr(str(p)), ch["Y"].astype(np.float32))


def test_exr_piz_reference_depth_maps(golden):
    """The reference's own depth maps: one HALF channel, PIZ, decreasing line order."""
    g = golden("g11_scene.npz")
    d = S.read_depth_exr(os.path.join(SCENE, "depth.exr"))
    b = S.read_depth_exr(os.path.join(SCENE, "bg_depth.exr"))
    assert d.shape == (512, 512) and d.dtype == np.float32 and np.isfinite(d).all() and np.isfinite(b).all()
    assert 0.5 < d.min() and d.max() < 10.0
    assert np.array_equal(d.astype(np.float16).astype(np.float32), d)             # HALF payload
    # a mis-decoded wavelet / Huffman stream is noise: a metric depth map is smooth almost everywhere
    assert np.abs(np.diff(d, axis=0)).mean() < 0.02 and np.abs(np.diff(d, axis=1)).mean() < 0.02
    m = S.read_png(os.path.join(SCENE, "mask.png")) > 127
    assert d[m].mean() < b[m].mean() - 0.3                                        # the object stands in front of its background
    assert np.abs(d[~m] - b[~m]).mean() < 0.3
    assert np.array_equal(d[::37, ::41], g["depth_slice"])


def test_load_scene_matches_reference_loader_steps(golden):
    g = golden("g11_scene.npz")
    sc = S.load_scene(SCENE, 512)
    assert sc["prompt"] == "banana fruits on the table"
    assert list(sc["transforms"].keys()) == ["edit_000", "edit_001", "edit_002"]
    assert sc["img"].shape == (1, 3, 512, 512) and 0.0 <= float(sc["img"].min()) and float(sc["img"].max()) <= 1.0
    assert sc["fg_mask"].shape == (1, 1, 512, 512) and set(np.unique(sc["fg_mask"].numpy())) <= {0.0, 1.0}
    assert np.array_equal(np.packbits(sc["fg_mask"].numpy() != 0), g["mask_bits"])
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(sc["depth"].numpy()) == str(g["depth_sha"]) and sha(sc["bg_depth"].numpy()) == str(g["bg_depth_sha"])
    kw = S.transform_args(sc["transforms"]["edit_001"])
    assert kw["rot_angle"] == 91.0 and torch.equal(kw["rot_axis"], torch.tensor([0.0, 1.0, 0.0]))
    assert torch.allclose(kw["translation"], torch.tensor([0.35, 0.0, 0.0]))
    # resize path: antialiased bilinear to another resolution, non-square crop
    x = torch.arange(6 * 10, dtype=torch.float32).reshape(1, 1, 6, 10)
    y = S.crop_and_resize(x, 3)
    assert y.shape == (1, 1, 3, 3)
    assert torch.allclose(y, torch.nn.functional.interpolate(x[..., 2:8], size=(3, 3), mode="bilinear", antialias=True))


def test_oracle_on_the_real_scene_vs_reference_golden(golden):
    """oracle == reference on real (estimated, noisy) depth: one of the scene's edits, full arrays."""
    from oracle import depth_ref as D
    g = golden("g11_scene.npz")
    sc = S.load_scene(SCENE, 512)
    t = sc["transforms"]["edit_001"]
    disp, corr, dbg = D.transform_depth_pc(sc["depth"], sc["bg_depth"], sc["fg_mask"], D.intrinsics_f32(),
                                           rot_angle=t["rotation_angle"], rot_axis=t["rotation_axis"],
                                           translation=t["translation"], return_debug=True)
    assert np.array_equal(corr.numpy(), g["edit_001_corr"].astype(np.int64))
    assert np.array_equal(np.packbits(dbg["cleaned"] != 0), g["edit_001_cleaned"])
    assert np.array_equal(dbg["zmap"][::37, ::41], g["edit_001_zmap_slice"])
    assert np.allclose(disp[0, 0].numpy()[::5, ::7], g["edit_001_disp_slice"], atol=1e-4, rtol=0)


def test_png_low_bit_depths_and_palette(tmp_path):
    """1/2/4-bit gray rows are packed MSB first and scaled to 0..255; palette images come back as RGB."""
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    rng = np.random.default_rng(7)
    for depth in (1, 2, 4):
        w, h = 13, 5
        a = rng.integers(0, 1 << depth, (h, w)).astype(np.uint8)
        per = 8 // depth
        stride = (w * depth + 7) // 8
        rows = b""
        for y in range(h):
            padded = np.zeros(stride * per, np.uint8)
            padded[:w] = a[y]
            packed = np.zeros(stride, np.uint8)
            for k in range(per):
                packed |= (padded[k::per] << ((per - 1 - k) * depth)).astype(np.uint8)
            rows += b"\x00" + packed.tobytes()
        p = tmp_path / f"g{depth}.png"
        p.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 0, 0, 0, 0)) +
                      chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b""))
        assert np.array_equal(S.read_png(str(p)), a * (255 // ((1 << depth) - 1)))
    # 8-bit palette
    pal = rng.integers(0, 256, (7, 3)).astype(np.uint8)
    idx = rng.integers(0, 7, (6, 9)).astype(np.uint8)
    rows = b"".join(b"\x00" + idx[y].tobytes() for y in range(6))
    p = tmp_path / "pal.png"
    p.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 9, 6, 8, 3, 0, 0, 0)) + chunk(b"PLTE", pal.tobytes()) +
                  chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b""))
    assert np.array_equal(S.read_png(str(p)), pal[idx])
    # interlaced files are refused, not mis-read
    p = tmp_path / "il.png"
    p.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 8, 0, 0, 0, 1)) +
                  chunk(b"IDAT", zlib.compress(b"\x00\x00\x00" * 2)) + chunk(b"IEND", b""))
    with pytest.raises(ValueError):
        S.read_png(str(p))


def test_exr_rejects_what_it_cannot_read(tmp_path):
    ch = {"Y": np.ones((4, 4), np.float16)}
    raw = bytearray(_exr_bytes(ch, 0))
    bad = bytes(raw).replace(b"compression\0compression\0\x01\x00\x00\x00\x00", b"compression\0compression\0\x01\x00\x00\x00\x05")   # PXR24
    (tmp_path / "pxr.exr").write_bytes(bad)
    with pytest.raises(ValueError):
        S.read_exr(str(tmp_path / "pxr.exr"))
    (tmp_path / "junk.exr").write_bytes(b"\0" * 64)
    with pytest.raises(ValueError):
        S.read_exr(str(tmp_path / "junk.exr"))


# ---- PIZ written by an encoder of the test's own (forward transforms; the reader only holds the inverses) -----------------
# OpenEXR's published PIZ block: u16 min/max non-zero bitmap byte, bitmap bytes, i32 length, Huffman block.  The forward
# wavelet, the forward LUT, the Huffman length assignment (heap), the table packer and the bit writer below share no code with
# diffusionhandles_amd/scene_io.py, which only has the decoding direction.

def _wenc14(a, b):
    a, b = a - 65536 if a >= 32768 else a, b - 65536 if b >= 32768 else b
    return ((a + b) >> 1) & 0xFFFF, (a - b) & 0xFFFF


def _wenc16(a, b):
    ao = (a + 0x8000) & 0xFFFF
    m, d = (ao + b) >> 1, ao - b
    if d < 0:
        m = (m + 0x8000) & 0xFFFF
    return m, d & 0xFFFF


def _wav2_encode(a, mx):
    """a: list of lists [ny][nx] of ints, in place; fine to coarse."""
    ny, nx = len(a), len(a[0])
    enc = _wenc14 if mx < (1 << 14) else _wenc16
    p, p2 = 1, 2
    while p2 <= min(nx, ny):
        y = 0
        while y <= ny - p2:
            x = 0
            while x <= nx - p2:
                i00, i01 = enc(a[y][x], a[y][x + p])
                i10, i11 = enc(a[y + p][x], a[y + p][x + p])
                a[y][x], a[y + p][x] = enc(i00, i10)
                a[y][x + p], a[y + p][x + p] = enc(i01, i11)
                x += p2
            if nx & p:
                a[y][x], a[y + p][x] = enc(a[y][x], a[y + p][x])
            y += p2
        if ny & p:
            x = 0
            while x <= nx - p2:
                a[y][x], a[y][x + p] = enc(a[y][x], a[y][x + p])
                x += p2
        p, p2 = p2, p2 << 1


class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):
        self.acc = (self.acc << nbits) | value
        self.n += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 255)
        self.acc &= (1 << self.n) - 1

    def done(self):
        total = 8 * len(self.out) + self.n
        if self.n:
            self.out.append((self.acc << (8 - self.n)) & 255)
        return bytes(self.out), total


def _huf_compress(words, use_runs=True):
    import heapq
    freq = {}
    for s in words:
        freq[s] = freq.get(s, 0) + 1
    im, iM = min(freq), max(freq) + 1              # iM: the run-length pseudo symbol, frequency 1
    freq[iM] = 1
    heap = [(f, s, (s,)) for s, f in freq.items()]
    heapq.heapify(heap)
    length = dict.fromkeys(freq, 0)
    while len(heap) > 1:
        fa, sa, ma = heapq.heappop(heap)
        fb, sb, mb = heapq.heappop(heap)
        for s in ma + mb:
            length[s] += 1
        heapq.heappush(heap, (fa + fb, min(sa, sb), ma + mb))
    assert max(length.values()) <= 58
    # canonical codes: longest lengths get the smallest codes; within a length, symbol order
    count = [0] * 60
    for l in length.values():
        count[l] += 1
    base, c = [0] * 60, 0
    for l in range(58, 0, -1):
        base[l], c = c, (c + count[l]) >> 1
    code = {}
    for s in sorted(length):
        code[s] = base[length[s]]
        base[length[s]] += 1
    tb = _Bits()                                   # packed table: 6-bit lengths with zero runs
    s = im
    while s <= iM:
        l = length.get(s, 0)
        if l == 0:
            run = 1
            while s + run <= iM and length.get(s + run, 0) == 0 and run < 255 + 6:
                run += 1
            if run >= 6:
                tb.put(63, 6)
                tb.put(run - 6, 8)
            elif run >= 2:
                tb.put(59 + run - 2, 6)
            else:
                tb.put(0, 6)
            s += run
        else:
            tb.put(l, 6)
            s += 1
    table, _ = tb.done()
    db = _Bits()
    i = 0
    while i < len(words):
        s, run = words[i], 1
        while use_runs and i + run < len(words) and words[i + run] == s and run < 256:
            run += 1
        db.put(code[s], length[s])
        if run >= 3:
            db.put(code[iM], length[iM])
            db.put(run - 1, 8)
        else:
            for _ in range(run - 1):
                db.put(code[s], length[s])
        i += run
    data, nbits = db.done()
    return struct.pack("<5I", im, iM, len(table), nbits, 0) + table + data


def _piz_block(rows, use_runs=True):
    """rows: per channel a uint16 array [ny, nx, words per pixel] -> PIZ block bytes."""
    used = np.zeros(65536, bool)
    for r in rows:
        used[r.reshape(-1)] = True
    used[0] = False                                # zero is never stored in the bitmap
    bitmap = np.packbits(used, bitorder="little")
    nz = np.nonzero(bitmap)[0]
    used[0] = True
    fwd = np.cumsum(used) - 1                      # value -> dense index (0 stays 0)
    mx = int(used.sum()) - 1
    words = []
    for r in rows:
        d = fwd[r]
        for j in range(r.shape[2]):
            plane = [[int(v) for v in line] for line in d[:, :, j]]
            _wav2_encode(plane, mx)
            d[:, :, j] = plane
        words += [int(v) for v in d.reshape(-1)]
    huf = _huf_compress(words, use_runs)
    if len(nz):
        head = struct.pack("<HH", nz[0], nz[-1]) + bitmap[nz[0]:nz[-1] + 1].tobytes()
    else:
        head = struct.pack("<HH", 8191, 0)
    return head + struct.pack("<i", len(huf)) + huf, mx


def _exr_piz_bytes(chans, use_runs=True):
    """Like _exr_bytes for compression 4: 32 scan lines per chunk."""
    names = sorted(chans)
    h, w = chans[names[0]].shape
    ref = _exr_bytes(chans, 0)
    n_none = h                                     # chunks of the uncompressed twin
    hdr_len = ref.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    hdr_end = len(ref) - sum(8 + w * sum(chans[n].dtype.itemsize for n in names) for _ in range(h)) - 8 * n_none
    hdr = ref[:hdr_len] + bytes([4]) + ref[hdr_len + 1:hdr_end]
    chunks, widest = [], 0
    for y in range(0, h, 32):
        ny = min(32, h - y)
        rows = [np.ascontiguousarray(chans[n][y:y + ny]).astype(chans[n].dtype.newbyteorder("<")).view("<u2")
                .reshape(ny, w, chans[n].dtype.itemsize // 2) for n in names]
        blk, mx = _piz_block(rows, use_runs)
        widest = max(widest, mx)
        raw = b"".join(chans[n][r].astype(chans[n].dtype.newbyteorder("<")).tobytes() for r in range(y, y + ny) for n in names)
        chunks.append((y, blk if len(blk) < len(raw) else raw))
    off0 = len(hdr) + 8 * len(chunks)
    offs, body = [], b""
    for y, data in chunks:
        offs.append(off0 + len(body))
        body += struct.pack("<ii", y, len(data)) + data
    return hdr + b"".join(struct.pack("<Q", o) for o in offs) + body, widest


@pytest.mark.parametrize("case", ["half14", "wide16", "odd", "flat"])
def test_exr_piz_round_trip_with_independent_encoder(tmp_path, case):
    rng = np.random.default_rng({"half14": 1, "wide16": 2, "odd": 3, "flat": 4}[case])
    if case == "half14":        # few distinct 16-bit values (< 2^14): the 14-bit wavelet
        yy, xx = np.mgrid[0:40, 0:48]
        ch = {"Y": (2.0 + 0.01 * xx + 0.02 * yy).astype(np.float16)}
    elif case == "wide16":      # float32 + uint32 planes, > 2^14 distinct words: the 16-bit (modulo) wavelet
        yy, xx = np.mgrid[0:70, 0:300]
        ch = {"Z": (1.0 + 0.001 * xx * yy + rng.random((70, 300)) * 1e-3).astype(np.float32),
              "id": rng.integers(0, 2 ** 32, (70, 300), dtype=np.uint32), "A": rng.random((70, 300)).astype(np.float16)}
    elif case == "odd":         # odd sizes at several levels, last chunk of 5 lines, a width below the chunk height
        ch = {"Y": np.round(rng.random((37, 21)) * 8).astype(np.float16), "Z": (rng.random((37, 21)) * 3).astype(np.float32)}
    else:                       # long runs: one value everywhere, then a step
        y = np.full((64, 64), 1.5, np.float16)
        y[40:, 10:] = 3.0
        ch = {"Y": y}
    data, widest = _exr_piz_bytes(ch)
    assert (widest >= (1 << 14)) == (case == "wide16")
    (tmp_path / "p.exr").write_bytes(data)
    got = S.read_exr(str(tmp_path / "p.exr"))
    for n, a in ch.items():
        assert np.array_equal(got[n], a.astype(got[n].dtype)), n
    if case in ("flat", "half14"):
        assert len(data) < len(_exr_bytes(ch, 0)) // 2         # the blocks really are the compressed form
        norun, _ = _exr_piz_bytes(ch, use_runs=False)
        (tmp_path / "q.exr").write_bytes(norun)
        assert np.array_equal(S.read_exr(str(tmp_path / "q.exr"))["Y"], ch["Y"].astype(np.float32))
        if case == "flat":
            assert len(data) < len(norun)                       # the run-length symbol was exercised


SCENE2 = os.path.join(os.path.dirname(__file__), "golden", "scene_dice")


def test_second_reference_scene_exr_and_oracle_vs_reference_golden(golden):
    """A second scene of the reference's test data (test/data/photogen/dice: other PIZ streams, another mask, a translation-only
    and an identity transform; tools/make_golden_scene.py scene_dice g15_scene_dice.npz): the EXR reader reproduces the depth
    the reference's functions were run on (hashes), and the oracle reproduces the reference's integer maps on it."""
    import hashlib
    from oracle import depth_ref as D
    g = golden("g15_scene_dice.npz")
    sc = S.load_scene(SCENE2, 512)
    assert sc["prompt"] == "a dice on the table" and list(sc["transforms"].keys()) == ["edit_000", "edit_001"]
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(sc["depth"].numpy()) == str(g["depth_sha"]) and sha(sc["bg_depth"].numpy()) == str(g["bg_depth_sha"])
    assert sha(sc["img"].numpy()) == str(g["img_sha"])
    assert np.array_equal(np.packbits(sc["fg_mask"].numpy() != 0), g["mask_bits"])
    assert np.array_equal(sc["depth"][0, 0].numpy()[::37, ::41], g["depth_slice"])
    d = sc["depth"][0, 0].numpy()
    assert np.abs(np.diff(d, axis=0)).mean() < 0.02 and np.abs(np.diff(d, axis=1)).mean() < 0.02      # a decoded depth map, not noise
    for name in ("edit_000", "edit_001"):
        t = sc["transforms"][name]
        disp, corr, dbg = D.transform_depth_pc(sc["depth"], sc["bg_depth"], sc["fg_mask"], D.intrinsics_f32(),
                                               rot_angle=t["rotation_angle"], rot_axis=t["rotation_axis"],
                                               translation=t["translation"], return_debug=True)
        assert np.array_equal(corr.numpy(), g[f"{name}_corr"].astype(np.int64)), name
        assert np.array_equal(np.packbits(dbg["cleaned"] != 0), g[f"{name}_cleaned"]), name
        assert np.array_equal(dbg["zmap"][::37, ::41], g[f"{name}_zmap_slice"]), name
        assert np.allclose(disp[0, 0].numpy()[::5, ::7], g[f"{name}_disp_slice"], atol=1e-4, rtol=0), name
