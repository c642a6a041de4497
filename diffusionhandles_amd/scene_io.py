"""Scene-directory I/O for the edit harness (SURVEY section 8c "what the build's own counterparts must reproduce").

The reference's harness (`test/test_diffusion_handles.py:208-263`, `test/utils.py:8-58`) reads a scene directory
`input.png, mask.png, depth.exr, bg_depth.exr, prompt.txt, transforms.json` through imageio/torchvision, neither of
which is installed here.  This module reads the same files with the standard library + NumPy only:

* `read_png`   8/16-bit gray, gray+alpha, RGB, RGBA and palette PNGs, non-interlaced (all five scanline filters)
* `read_exr`   single-part scanline OpenEXR, HALF/FLOAT/UINT channels, compression NONE / ZIPS / ZIP / PIZ
               (the reference's depth maps are one HALF channel `Y`, PIZ-compressed, `lineOrder` decreasing)
* `write_png`  8-bit gray / RGB
* `load_scene` the reference loader's steps: centre crop, antialiased bilinear resize, mask > 0.5, first prompt line,
               ordered transforms `{name: {translation, rotation_axis, rotation_angle}}`

Host-side plumbing only: nothing here is on the timed path.
"""
import json
import os
import struct
import zlib
from collections import OrderedDict

import numpy as np

# ---------------------------------------------------------------------------------------------- PNG

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _paeth_row(cur, prev, bpp):
    out = bytearray(cur)
    for i in range(len(out)):
        a = out[i - bpp] if i >= bpp else 0
        b = prev[i]
        c = prev[i - bpp] if i >= bpp else 0
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
        out[i] = (out[i] + pred) & 255
    return out


def _avg_row(cur, prev, bpp):
    out = bytearray(cur)
    for i in range(len(out)):
        a = out[i - bpp] if i >= bpp else 0
        out[i] = (out[i] + ((a + prev[i]) >> 1)) & 255
    return out


def _sub_row(cur, bpp):
    a = np.frombuffer(bytes(cur), dtype=np.uint8).reshape(-1, bpp).astype(np.uint32)
    return bytearray((np.cumsum(a, axis=0) & 255).astype(np.uint8).tobytes())


def read_png(path):
    """-> uint8 or uint16 array [H,W] (gray) or [H,W,C]."""
    with open(path, "rb") as f:
        b = f.read()
    if b[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG file")
    p, idat, hdr, plte = 8, [], None, None
    while p < len(b):
        n, tag = struct.unpack(">I4s", b[p:p + 8])
        data = b[p + 8:p + 8 + n]
        p += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"IDAT":
            idat.append(data)
        elif tag == b"PLTE":
            plte = np.frombuffer(data, dtype=np.uint8).reshape(-1, 3)
        elif tag == b"IEND":
            break
    if hdr is None:
        raise ValueError(f"{path}: no IHDR chunk")
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace:
        raise ValueError(f"{path}: interlaced PNGs are not supported")
    nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bits = nch * depth
    bpp = max(1, bits // 8)
    stride = (w * bits + 7) // 8
    raw = zlib.decompress(b"".join(idat))
    if len(raw) < h * (stride + 1):
        raise ValueError(f"{path}: truncated image data")
    rows, prev = [], bytearray(stride)
    for y in range(h):
        ft = raw[y * (stride + 1)]
        cur = raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)]
        if ft == 0:
            cur = bytearray(cur)
        elif ft == 1:
            cur = _sub_row(cur, bpp)
        elif ft == 2:
            cur = bytearray(((np.frombuffer(cur, dtype=np.uint8).astype(np.uint16) + np.frombuffer(bytes(prev), dtype=np.uint8)) & 255)
                            .astype(np.uint8).tobytes())
        elif ft == 3:
            cur = _avg_row(cur, prev, bpp)
        elif ft == 4:
            cur = _paeth_row(cur, prev, bpp)
        else:
            raise ValueError(f"{path}: bad filter type {ft}")
        rows.append(bytes(cur))
        prev = cur
    data = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(h, stride)
    if depth == 16:
        img = data.view(">u2").astype(np.uint16).reshape(h, w, nch)
    elif depth == 8:
        img = data.reshape(h, w, nch)
    else:                                           # 1/2/4-bit gray or palette: unpack MSB first
        per = 8 // depth
        shifts = np.arange(per - 1, -1, -1, dtype=np.uint8) * depth
        img = ((data[:, :, None] >> shifts[None, None, :]) & ((1 << depth) - 1)).reshape(h, stride * per)[:, :w, None]
        if ctype == 0:
            img = (img.astype(np.uint16) * (255 // ((1 << depth) - 1))).astype(np.uint8)
    if ctype == 3:
        if plte is None:
            raise ValueError(f"{path}: palette image without PLTE")
        img = plte[img[..., 0]]
    return img[..., 0] if img.shape[-1] == 1 else img


def write_png(path, img):
    """img: [H,W] or [H,W,3] float in [0,1] (or uint8).  `save_image` of test/utils.py:21-31 truncates (`* 255` ->
    uint8), so does this."""
    a = np.asarray(img)
    if a.dtype != np.uint8:
        a = (np.clip(a, 0, 1) * 255.0).astype(np.uint8)
    if a.ndim == 2:
        a = a[..., None]
    h, w, c = a.shape
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    hdr = struct.pack(">IIBBBBB", w, h, 8, {1: 0, 3: 2, 4: 6}[c], 0, 0, 0)
    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", hdr) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


# ---------------------------------------------------------------------------------------------- OpenEXR

_EXR_MAGIC = 20000630
_LINES = {0: 1, 1: 1, 2: 1, 3: 16, 4: 32}          # NONE, RLE, ZIPS, ZIP, PIZ
_PIX_BYTES = {0: 4, 1: 2, 2: 4}                    # UINT, HALF, FLOAT
_PIX_DTYPE = {0: "<u4", 1: "<f2", 2: "<f4"}


def _cstr(b, p):
    e = b.index(b"\0", p)
    return b[p:e].decode("latin-1"), e + 1


def _exr_header(b):
    magic, version = struct.unpack("<II", b[:8])
    if magic != _EXR_MAGIC:
        raise ValueError("not an OpenEXR file")
    if version & 0x200 or version & 0x1000 or version & 0x800:
        raise ValueError("tiled / multi-part / deep OpenEXR files are not supported")
    p, attrs = 8, {}
    while b[p] != 0:
        name, p = _cstr(b, p)
        typ, p = _cstr(b, p)
        (sz,) = struct.unpack("<I", b[p:p + 4])
        attrs[name] = (typ, b[p + 4:p + 4 + sz])
        p += 4 + sz
    p += 1
    chans, v, q = [], attrs["channels"][1], 0
    while v[q] != 0:
        cn, q = _cstr(v, q)
        pt, _lin, xs, ys = struct.unpack("<IB3xii", v[q:q + 16])
        q += 16
        if xs != 1 or ys != 1:
            raise ValueError("sub-sampled OpenEXR channels are not supported")
        chans.append((cn, pt))
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    return dict(channels=chans, compression=attrs["compression"][1][0], window=(x0, y0, x1, y1)), p


def _zip_undo(data, expected):
    t = np.frombuffer(zlib.decompress(data), dtype=np.uint8).astype(np.int64)
    if t.size != expected:
        raise ValueError("OpenEXR zip block has the wrong size")
    t = (np.cumsum(t - 128) + 128) & 255 if t.size else t         # t[i] = t[i-1] + t[i] - 128
    t = t.astype(np.uint8)
    half = (t.size + 1) // 2
    out = np.empty(t.size, dtype=np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


# ---- PIZ: 16-bit value bitmap + LUT, Huffman-coded wavelet coefficients (OpenEXR's published format) ------

def _huf_unpack_table(b, p, im, iM):
    """6-bit code lengths im..iM, MSB-first, with zero-run codes 59..62 (2..5 zeros) and 63 (+8 bits: 6..261 zeros)."""
    lens = np.zeros(65537, dtype=np.int64)
    c = lc = 0

    def bits(n):
        nonlocal c, lc, p
        while lc < n:
            c = ((c << 8) | b[p]) & 0xFFFFFFFFFFFF
            p += 1
            lc += 8
        lc -= n
        return (c >> lc) & ((1 << n) - 1)

    i = im
    while i <= iM:
        l = bits(6)
        if l == 63:
            i += bits(8) + 6
        elif l >= 59:
            i += l - 59 + 2
        else:
            lens[i] = l
            i += 1
    return lens, p


def _huf_canonical(lens):
    """symbol -> code: codes of a length are consecutive in symbol order; the base of length l is
    (base(l+1) + count(l+1)) >> 1, from base(58) = 0."""
    n = np.bincount(lens, minlength=59).astype(object)
    base, c = [0] * 59, 0
    for l in range(58, 0, -1):
        nc = (c + int(n[l])) >> 1
        base[l] = c
        c = nc
    table = {}
    for s in np.nonzero(lens)[0]:
        l = int(lens[s])
        table[(1 << l) | base[l]] = int(s)            # leading one marks the length
        base[l] += 1
    return table


def _huf_uncompress(b, n_raw):
    im, iM, _tl, nbits, _r = struct.unpack("<5I", b[:20])
    if im > 65536 or iM > 65536:
        raise ValueError("bad PIZ Huffman header")
    lens, p = _huf_unpack_table(b, 20, im, iM)
    if nbits > 8 * (len(b) - p):
        raise ValueError("PIZ Huffman data is truncated")
    table, rlc = _huf_canonical(lens), iM
    out = np.empty(n_raw, dtype=np.uint16)
    get = table.get
    no = 0
    key = 1
    pos, end = p * 8, p * 8 + nbits
    while pos < end:
        key = (key << 1) | ((b[pos >> 3] >> (7 - (pos & 7))) & 1)
        pos += 1
        s = get(key)
        if s is None:
            if key >> 59:
                raise ValueError("bad PIZ Huffman code")
            continue
        key = 1
        if s == rlc:
            if pos + 8 > end or no == 0:
                raise ValueError("bad PIZ run")
            cs = 0
            for _ in range(8):
                cs = (cs << 1) | ((b[pos >> 3] >> (7 - (pos & 7))) & 1)
                pos += 1
            if no + cs > n_raw:
                raise ValueError("PIZ run overflows the block")
            out[no:no + cs] = out[no - 1]
            no += cs
        else:
            if no >= n_raw:
                raise ValueError("PIZ data overflows the block")
            out[no] = s
            no += 1
    if no != n_raw:
        raise ValueError(f"PIZ block decoded {no} of {n_raw} values")
    return out


def _wdec14(l, h):
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    a = ls + (hs & 1) + (hs >> 1)
    return (a & 0xFFFF).astype(np.uint16), ((a - hs) & 0xFFFF).astype(np.uint16)


def _wdec16(l, h):
    m, d = l.astype(np.int32), h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xFFFF
    aa = (d + bb - 0x8000) & 0xFFFF
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(a, mx):
    """In-place inverse 2-D wavelet of a [ny, nx] uint16 array (coarse to fine; whole levels at once)."""
    ny, nx = a.shape
    dec = _wdec14 if mx < (1 << 14) else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2) if ny - p2 >= 0 else np.arange(0)
        xs = np.arange(0, nx - p2 + 1, p2) if nx - p2 >= 0 else np.arange(0)
        if len(ys) and len(xs):
            Y, X = np.ix_(ys, xs)
            px, p01, p10, p11 = a[Y, X], a[Y, X + p], a[Y + p, X], a[Y + p, X + p]
            i00, i10 = dec(px, p10)
            i01, i11 = dec(p01, p11)
            a[Y, X], a[Y, X + p] = dec(i00, i01)
            a[Y + p, X], a[Y + p, X + p] = dec(i10, i11)
        if nx & p and len(ys):                      # odd column left over at this level
            xo = len(xs) * p2
            i00, b10 = dec(a[ys, xo], a[ys + p, xo])
            a[ys, xo], a[ys + p, xo] = i00, b10
        if ny & p and len(xs):                      # odd row
            yo = len(ys) * p2
            i00, b01 = dec(a[yo, xs], a[yo, xs + p])
            a[yo, xs], a[yo, xs + p] = i00, b01
        p2 = p
        p >>= 1


def _piz_undo(data, chans, nx, ny):
    mn, mxnz = struct.unpack("<HH", data[:4])
    p = 4
    bitmap = np.zeros(8192, dtype=np.uint8)
    if mn <= mxnz:
        if mxnz >= 8192:
            raise ValueError("bad PIZ bitmap range")
        bitmap[mn:mxnz + 1] = np.frombuffer(data[p:p + mxnz - mn + 1], dtype=np.uint8)
        p += mxnz - mn + 1
    present = np.unpackbits(bitmap, bitorder="little").astype(bool)
    present[0] = True
    lut = np.zeros(65536, dtype=np.uint16)
    vals = np.nonzero(present)[0]
    lut[:len(vals)] = vals
    max_value = len(vals) - 1
    (length,) = struct.unpack("<i", data[p:p + 4])
    p += 4
    if length < 0 or p + length > len(data):
        raise ValueError("bad PIZ block length")
    sizes = [_PIX_BYTES[pt] // 2 for _, pt in chans]
    raw = _huf_uncompress(data[p:p + length], sum(nx * ny * s for s in sizes))
    planes, q = [], 0
    for s in sizes:
        blk = raw[q:q + nx * ny * s].reshape(ny, nx, s).copy()
        q += nx * ny * s
        for j in range(s):
            plane = np.ascontiguousarray(blk[:, :, j])
            _wav2_decode(plane, max_value)
            blk[:, :, j] = plane
        planes.append(lut[blk])
    # scan lines, channels in file order inside each line
    return b"".join(planes[c][y].astype("<u2").tobytes() for y in range(ny) for c in range(len(chans)))


def read_exr(path):
    """-> dict {channel name: float32 (or uint32) array [H,W]} of a single-part scanline file."""
    with open(path, "rb") as f:
        b = f.read()
    hdr, p = _exr_header(b)
    x0, y0, x1, y1 = hdr["window"]
    nx, ny = x1 - x0 + 1, y1 - y0 + 1
    comp, chans = hdr["compression"], hdr["channels"]
    if comp not in _LINES or comp == 1:
        raise ValueError(f"{path}: OpenEXR compression {comp} is not supported (NONE, ZIPS, ZIP, PIZ are)")
    lines = _LINES[comp]
    nchunks = (ny + lines - 1) // lines
    offsets = struct.unpack(f"<{nchunks}Q", b[p:p + 8 * nchunks])
    line_bytes = nx * sum(_PIX_BYTES[pt] for _, pt in chans)
    out = {cn: np.zeros((ny, nx), dtype=np.uint32 if pt == 0 else np.float32) for cn, pt in chans}
    for off in offsets:
        y, size = struct.unpack("<ii", b[off:off + 8])
        data = b[off + 8:off + 8 + size]
        rows = min(lines, y1 - y + 1)
        expected = rows * line_bytes
        if size < expected:
            if comp in (2, 3):
                data = _zip_undo(data, expected)
            elif comp == 4:
                data = _piz_undo(data, chans, nx, rows)
        elif size != expected:
            raise ValueError(f"{path}: bad chunk size")
        q = 0
        for r in range(rows):
            for cn, pt in chans:
                nb = nx * _PIX_BYTES[pt]
                out[cn][y - y0 + r] = np.frombuffer(data[q:q + nb], dtype=_PIX_DTYPE[pt])
                q += nb
    return out


def read_depth_exr(path):
    """`load_depth` (test/utils.py:33-42): the single channel of the file as float32 [H,W]."""
    ch = read_exr(path)
    for name in ("Y", "Z", "R", "depth"):
        if name in ch:
            return ch[name].astype(np.float32)
    return next(iter(ch.values())).astype(np.float32)


# ---------------------------------------------------------------------------------------------- scene directory

def crop_and_resize(img, size):
    """test/utils.py:54-58: centre crop to square, then bilinear resize with antialiasing (torchvision `resize`
    on a float tensor is `interpolate(mode='bilinear', align_corners=False, antialias=True)`).  img: torch [1,C,H,W]."""
    import torch.nn.functional as F
    h, w = img.shape[-2:]
    if h != w:
        s = min(h, w)
        top, left = int(round((h - s) / 2.0)), int(round((w - s) / 2.0))
        img = img[..., top:top + s, left:left + s]
    if img.shape[-1] != size:
        img = F.interpolate(img, size=(size, size), mode="bilinear", align_corners=False, antialias=True)
    return img


def load_image(path):
    """test/utils.py:8-19: float32 [C,H,W] in [0,1] (uint8 / 255)."""
    import torch
    a = read_png(path)
    if a.ndim == 2:
        a = a[..., None]
    scale = 255.0 if a.dtype == np.uint8 else 65535.0
    return torch.from_numpy(a.astype(np.float32) / scale).permute(2, 0, 1).contiguous()


def load_scene_geometry(scene_dir, img_res=512):
    """The geometric inputs of a scene only -- what `transform_depth` needs (test/test_diffusion_handles.py:213-215, 247-261):
    dict(transforms, fg_mask [1,1,R,R] in {0,1}, depth [1,1,R,R], bg_depth [1,1,R,R]).  Needs transforms.json, mask.png,
    depth.exr / .npy, bg_depth.exr / .npy; no image, no prompt."""
    import torch
    j = os.path.join
    with open(j(scene_dir, "transforms.json")) as f:
        transforms = json.load(f, object_pairs_hook=OrderedDict)
    mask = load_image(j(scene_dir, "mask.png"))[None]
    if mask.shape[1] > 1:
        mask = mask.mean(dim=1, keepdim=True)
    mask = (crop_and_resize(mask, img_res) > 0.5).to(torch.float32)

    def depth_of(stem):
        if os.path.exists(j(scene_dir, stem + ".exr")):
            d = read_depth_exr(j(scene_dir, stem + ".exr"))
        elif os.path.exists(j(scene_dir, stem + ".npy")):
            d = np.load(j(scene_dir, stem + ".npy")).astype(np.float32)
        else:
            raise FileNotFoundError(j(scene_dir, stem + ".exr"))
        return crop_and_resize(torch.from_numpy(np.ascontiguousarray(d))[None, None], img_res).to(torch.float32)

    return dict(transforms=transforms, fg_mask=mask, depth=depth_of("depth"), bg_depth=depth_of("bg_depth"))


def load_scene(scene_dir, img_res=512):
    """`load_diffhandles_inputs` (test/test_diffusion_handles.py:208-263) -> dict(transforms, prompt, img [1,3,R,R],
    fg_mask [1,1,R,R] in {0,1}, depth [1,1,R,R], bg_depth [1,1,R,R])."""
    j = os.path.join
    with open(j(scene_dir, "prompt.txt")) as f:
        lines = [ln for ln in f.read().splitlines() if len(ln) > 0]
    if not lines:
        raise ValueError(f"{scene_dir}: empty prompt")
    img = load_image(j(scene_dir, "input.png"))[None]
    img = crop_and_resize(img[:, :3], img_res)
    geo = load_scene_geometry(scene_dir, img_res)
    return dict(transforms=geo["transforms"], prompt=lines[0], img=img, fg_mask=geo["fg_mask"], depth=geo["depth"],
                bg_depth=geo["bg_depth"])


def transform_args(t):
    """One transforms.json entry -> keyword arguments of `transform_foreground` (test_diffusion_handles.py:137-150)."""
    import torch
    return dict(rot_angle=float(t.get("rotation_angle", 0.0)),
                rot_axis=torch.tensor(t.get("rotation_axis", [0.0, 1.0, 0.0]), dtype=torch.float32),
                translation=torch.tensor(t.get("translation", [0.0, 0.0, 0.0]), dtype=torch.float32))
