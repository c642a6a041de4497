"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of depth_transform_mode='mesh' (reference depth_transform.py:91-195, depth_to_mesh
:30-71, transform_points :438-458, renderer outputs 'world_position' + 'flat_vertex_color' of
pytorch3d_renderer.py:541-941).

GEOMETRY PINNED (g13, tools/make_golden_mesh.py): the vertices, the two counter-clockwise triangles per pixel quad and the
float32 Rodrigues motion of the masked vertices equal the reference's own depth_to_mesh / transform_points (pure torch).
RASTERISATION PARITY UNPINNED: the reference draws with pytorch3d (git HEAD, pyproject.toml:35), which is neither in
/root/reference nor installed, and the reference holds no test or golden vector for this mode.  This
file restates pytorch3d's published naive rasterisation rule (NDC +X left/+Y up, pixel-centre sampling,
blur_radius coverage, perspective-correct clipped barycentrics, |area| <= 1e-8 skipped, area < 0 culled,
nearest z, one face per pixel) in float32 NumPy with the SAME operation order as csrc/mesh.hip, so the
HIP path can be checked bit for bit against it; what anchors it to the reference's behaviour is the
comparison with the pinned point z-buffer path on smooth depth (tests/test_mesh_gpu.py).
"""
import numpy as np

F = np.float32
EPS = F(1e-8)


def _unproject(d, gx, invf):
    a = (d * invf).astype(F)
    X = (-(a * gx[None, :])).astype(F)
    Y = (-(a * gx[:, None])).astype(F)
    return X, Y, d.astype(F)


def rodrigues_f32(X, Y, Z, xf):
    """transform_points (depth_transform.py:438-458) in float32, fixed order; xf = 11 floats."""
    ax, ay, az, c, s, tx, ty, tz, cx, cy, cz = [F(v) for v in xf]
    q0, q1, q2 = X - cx, Y - cy, Z - cz
    dot = (q0 * ax + q1 * ay) + q2 * az
    k1 = F(1.0) - c
    c0, c1, c2 = ay * q2 - az * q1, az * q0 - ax * q2, ax * q1 - ay * q0
    r0 = (q0 * c + c0 * s) + (ax * dot) * k1
    r1 = (q1 * c + c1 * s) + (ay * dot) * k1
    r2 = (q2 * c + c2 * s) + (az * dot) * k1
    return ((r0 + cx) + tx).astype(F), ((r1 + cy) + ty).astype(F), ((r2 + cz) + tz).astype(F)


def _edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def _seg_dist2(px, py, ax, ay, bx, by):
    dx, dy = bx - ax, by - ay
    l2 = dx * dx + dy * dy
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ((px - ax) * dx + (py - ay) * dy) / l2
    t = np.where(l2 > EPS, np.clip(t, F(0), F(1)), F(0)).astype(F)
    qx, qy = ax + t * dx, ay + t * dy
    return (px - qx) * (px - qx) + (py - qy) * (py - qy)


def _hit(tri, px, py, blur):
    """tri: 9 broadcastable float32 arrays; returns (ok, b0, b1, b2, pz)."""
    x0, y0, z0, x1, y1, z1, x2, y2, z2 = tri
    area = _edge(x2, y2, x0, y0, x1, y1)
    ok = area > EPS                                   # area < 0: back face; |area| <= eps: degenerate
    with np.errstate(divide="ignore", invalid="ignore"):
        w0 = _edge(px, py, x1, y1, x2, y2) / area
        w1 = _edge(px, py, x2, y2, x0, y0) / area
        w2 = _edge(px, py, x0, y0, x1, y1) / area
        inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
        d = _seg_dist2(px, py, x0, y0, x1, y1)
        d = np.minimum(d, _seg_dist2(px, py, x1, y1, x2, y2))
        d = np.minimum(d, _seg_dist2(px, py, x2, y2, x0, y0))
        ok = ok & (inside | ~(d > blur))
        t0, t1, t2 = (w0 * z1) * z2, (z0 * w1) * z2, (z0 * z1) * w2
        den = (t0 + t1) + t2
        den = np.where(den > EPS, den, EPS).astype(F)
        p0, p1, p2 = [np.clip(t / den, F(0), F(1)).astype(F) for t in (t0, t1, t2)]
        sm = (p0 + p1) + p2
        sm = np.where(sm > EPS, sm, EPS).astype(F)
        b0, b1, b2 = (p0 / sm).astype(F), (p1 / sm).astype(F), (p2 / sm).astype(F)
        pz = ((b0 * z0 + b1 * z1) + b2 * z2).astype(F)
        ok = ok & (pz >= 0)
    return ok, b0, b1, b2, pz


def _centre(i, res):
    return (F(1.0) - (2 * np.asarray(i) + 1).astype(F) / F(res)).astype(F)


def mesh_reproject(depth, bg_depth, mask, gx, lin01, invf, f, xform, blur=1e-5, bounds=None, window=8):
    """depth, bg_depth [R,R] f32, mask [R,R] bool.  Returns dict(zmap, disparity, fg_flag, corr [N,4] int64)."""
    depth, bg_depth = depth.astype(F), bg_depth.astype(F)
    gx, lin01, invf, f, blur = gx.astype(F), lin01.astype(F), F(invf), F(f), F(blur)
    R = depth.shape[0]
    nq = (R - 1) * (R - 1)
    Xb, Yb, Zb = _unproject(bg_depth, gx, invf)
    vb = np.stack([(f * Xb) / Zb, (f * Yb) / Zb, Zb], axis=-1).astype(F).reshape(-1, 3)
    Xf, Yf, Zf = _unproject(depth, gx, invf)
    Xf, Yf, Zf = rodrigues_f32(Xf, Yf, Zf, xform)
    with np.errstate(divide="ignore", invalid="ignore"):
        vf = np.stack([(f * Xf) / Zf, (f * Yf) / Zf, Zf], axis=-1).astype(F).reshape(-1, 3)
    m = mask.reshape(-1)

    # faces: id -> vertex indices
    fid = np.arange(4 * nq)
    fg = fid >= 2 * nq
    loc = np.where(fg, fid - 2 * nq, fid)
    q, lower = loc >> 1, loc & 1
    y, x = q // (R - 1), q % (R - 1)
    v00 = y * R + x
    i0 = v00 + R
    i1 = np.where(lower == 1, v00 + R + 1, v00 + 1)
    i2 = np.where(lower == 1, v00 + 1, v00)
    exists = ~fg | (m[i0] & m[i1] & m[i2])
    V = np.where(fg[:, None, None], np.stack([vf[i0], vf[i1], vf[i2]], 1), np.stack([vb[i0], vb[i1], vb[i2]], 1))
    with np.errstate(invalid="ignore"):
        exists &= (V[:, 0, 2] > 0) & (V[:, 1, 2] > 0) & (V[:, 2, 2] > 0)
    tri = [V[:, k // 3, k % 3] for k in range(9)]

    pad = F(np.sqrt(blur) + F(1e-6))
    with np.errstate(invalid="ignore"):
        xmin, xmax = np.min(V[:, :, 0], 1) - pad, np.max(V[:, :, 0], 1) + pad
        ymin, ymax = np.min(V[:, :, 1], 1) - pad, np.max(V[:, :, 1], 1) + pad
        xmin, xmax, ymin, ymax = [np.nan_to_num(a, nan=0.0, posinf=4.0, neginf=-4.0) for a in (xmin, xmax, ymin, ymax)]
    c_lo = np.clip(np.floor(((1 - xmax) * R - 1) * 0.5).astype(np.int64) - 1, 0, R - 1)
    c_hi = np.clip(np.ceil(((1 - xmin) * R - 1) * 0.5).astype(np.int64) + 1, 0, R - 1)
    r_lo = np.clip(np.floor(((1 - ymax) * R - 1) * 0.5).astype(np.int64) - 1, 0, R - 1)
    r_hi = np.clip(np.ceil(((1 - ymin) * R - 1) * 0.5).astype(np.int64) + 1, 0, R - 1)

    zbuf = np.full(R * R, np.iinfo(np.uint64).max, dtype=np.uint64)

    def bid(face_idx, rows, cols):
        """face_idx [n], rows/cols [n, k] candidate pixels (may repeat / be out of the face's box)."""
        t = [a[face_idx][:, None] for a in tri]
        ok, _, _, _, pz = _hit(t, _centre(cols, R), _centre(rows, R), blur)
        ok &= (rows >= r_lo[face_idx][:, None]) & (rows <= r_hi[face_idx][:, None])
        ok &= (cols >= c_lo[face_idx][:, None]) & (cols <= c_hi[face_idx][:, None])
        key = (pz.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.broadcast_to(face_idx[:, None], pz.shape).astype(np.uint64)
        np.minimum.at(zbuf, (rows * R + cols)[ok], key[ok])

    small = exists & (c_hi - c_lo < window) & (r_hi - r_lo < window)
    idx = np.nonzero(small)[0]
    oy, ox = np.meshgrid(np.arange(window), np.arange(window), indexing="ij")
    for s0 in range(0, len(idx), 20000):
        fi = idx[s0:s0 + 20000]
        rows = np.minimum(r_lo[fi][:, None] + oy.reshape(1, -1), R - 1)
        cols = np.minimum(c_lo[fi][:, None] + ox.reshape(1, -1), R - 1)
        bid(fi, rows, cols)
    for fi in np.nonzero(exists & ~small)[0]:
        rr, cc = np.meshgrid(np.arange(r_lo[fi], r_hi[fi] + 1), np.arange(c_lo[fi], c_hi[fi] + 1), indexing="ij")
        bid(np.array([fi]), rr.reshape(1, -1), cc.reshape(1, -1))

    # resolve
    covered = zbuf != np.iinfo(np.uint64).max
    win = (zbuf & np.uint64(0xFFFFFFFF)).astype(np.int64)
    win = np.where(covered, win, 0)
    pix = np.arange(R * R)
    row, col = pix // R, pix % R
    t = [a[win] for a in tri]
    _, b0, b1, b2, pz = _hit(t, _centre(col, R), _centre(row, R), blur)
    zmap = np.where(covered, pz, F(0)).astype(F)
    wfg = covered & (win >= 2 * nq)
    wl = np.where(win >= 2 * nq, win - 2 * nq, win)
    wq, wlow = wl >> 1, wl & 1
    wy, wx = wq // (R - 1), wq % (R - 1)
    x0, y0 = wx, wy + 1
    x1, y1 = wx + 1, np.where(wlow == 1, wy + 1, wy)
    x2, y2 = np.where(wlow == 1, wx + 1, wx), wy
    u = ((b0 * lin01[x0] + b1 * lin01[x1]) + b2 * lin01[x2]).astype(F)
    v = ((b0 * lin01[y0] + b1 * lin01[y1]) + b2 * lin01[y2]).astype(F)
    sx = np.rint(u * F(R - 1)).astype(np.int64)
    sy = np.rint(v * F(R - 1)).astype(np.int64)
    sel = np.nonzero(wfg)[0]
    corr = np.stack([sx[sel], sy[sel], col[sel], row[sel]], axis=-1).astype(np.int64)
    with np.errstate(divide="ignore"):
        dsp = (F(1.0) / zmap).astype(F)
    lo, hi = (dsp.min(), dsp.max()) if bounds is None else (F(bounds[0]), F(bounds[1]))
    disparity = ((F(255.0) * (dsp - lo)) / (hi - lo)).astype(F)
    return dict(zmap=zmap.reshape(R, R), disparity=disparity.reshape(R, R), fg_flag=wfg.reshape(R, R), corr=corr)
