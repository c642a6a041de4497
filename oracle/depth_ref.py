"""ORACLE (test infrastructure, never on the product path).

CPU restatement (NumPy / torch-CPU) of the reference's depth re-projection path:
unproject -> SE(3) about the masked centroid -> pinhole projection -> z-buffer ->
mask clean-up -> correspondence filtering -> harmonic in-fill -> normalised disparity.

Pinned against the imported reference by tools/make_golden.py (fixtures under
tests/golden/).  Reference anchors (relative to /root/reference/diffhandles):
  normalize_depth            depth_transform.py:15-28
  depth_to_world_coords      depth_transform.py:589-641
  transform_point_cloud      depth_transform.py:461-533   (NumPy-2 promotion semantics)
  points_to_depth            depth_transform.py:643-747
  transform_depth_pc         depth_transform.py:198-363
  poisson_solve              depth_transform.py:535-587
  cv2 morphology             depth_transform.py:308-321   [ext: OpenCV absent -> parity unpinned]
"""
import numpy as np
import scipy.sparse
import scipy.sparse.linalg
import torch

FOV_DEG = 55.0


def intrinsics_f32():
    """K of guided_stable_diffuser.py:129-153 (55 deg fov, principal point 0)."""
    f = 1.0 / np.tan(0.5 * FOV_DEG * (np.pi / 180.0))
    return torch.tensor([[f, 0, 0], [0, f, 0], [0, 0, 1]], dtype=torch.float32)


def normalize_depth(depth, bounds=None):
    """255*(d-min)/(max-min) per sample; depth is a 4-D torch tensor."""
    if depth.dim() != 4:
        raise RuntimeError(f"Expected depth to have 4 dimensions, got {depth.dim()}")
    if bounds is None:
        flat = depth.reshape(depth.shape[0], -1)
        hi = flat.max(dim=-1).values[:, None, None, None]
        lo = flat.min(dim=-1).values[:, None, None, None]
    else:
        lo, hi = bounds
    return 255 * (depth - lo) / (hi - lo), (lo, hi)


def grid_axes(h, w):
    """The f32 pixel-centre coordinates the reference builds with torch.linspace."""
    m = max(h, w) - 1
    nw, nh = (w - 1) / m, (h - 1) / m
    xg = torch.linspace(-nw, nw, steps=w, dtype=torch.float32)
    yg = torch.linspace(-nh, nh, steps=h, dtype=torch.float32)
    return xg.numpy(), yg.numpy()


def unproject(depth_hw, K=None):
    """depth [H,W] f32 -> points [H,W,3] f32 in the flipped (pytorch3d) frame.

    Closed form of D * K^-1 @ [x,y,1], then diag(-1,-1,1): the off-diagonal zeros make
    every 3x3 product a single rounded multiply, so
      X = -fl32(fl32(D*invf) * xg[col]),  Y = -fl32(fl32(D*invf) * yg[row]),  Z = D.
    """
    if K is None:
        K = intrinsics_f32()
    d = np.asarray(depth_hw, dtype=np.float32)
    h, w = d.shape
    if h < 2 or w < 2:
        raise RuntimeError(f"Expected depth to have at least 2 pixels in each dimension, got {h} x {w}.")
    kinv = torch.linalg.inv(K).numpy()
    xg, yg = grid_axes(h, w)
    dfx = (d * kinv[0, 0]).astype(np.float32)
    dfy = (d * kinv[1, 1]).astype(np.float32)
    pts = np.empty((h, w, 3), dtype=np.float32)
    pts[..., 0] = -(dfx * xg[None, :])
    pts[..., 1] = -(dfy * yg[:, None])
    pts[..., 2] = d
    return pts


def masked_centroid_f32(points_hw3, mask_hw):
    """np.mean over the masked points: sequential row-major f32 accumulation / f32(N)."""
    sel = points_hw3[mask_hw.astype(bool)]
    return np.mean(sel, axis=0)


# `q . axis` of the Rodrigues formula: the reference calls np.dot (BLAS sgemv), whose summation order -- and use of fused
# multiply-adds -- belongs to the BLAS build and the host CPU, so for a general axis the reference's own result is machine-dependent
# in the last bit.  DOT_ORDER = "blas" follows the reference literally; "explicit" is one admissible order written out,
# (q0 a0 + q1 a1) + q2 a2 with every product and sum rounded to f32 -- the order the HIP kernel uses (csrc/geometry.hip k_points).
# tests/test_geometry_gpu.py holds the product to the explicit order bit for bit and to the BLAS order within a handful of pairs.
DOT_ORDER = "blas"


def rigid_transform(points_hw3, axis, angle_deg, translation, mask_hw):
    """Rodrigues rotation about the masked centroid + translation -> float64 [H,W,3].

    NumPy-2 (NEP 50) promotion: the f32 terms are multiplied by float64 cos/sin
    scalars, so term1/2/3 and the sum are float64; the cross and dot products and
    `points - centroid` stay float32.
    """
    p = np.asarray(points_hw3, dtype=np.float32)
    h, w, _ = p.shape
    ax = np.asarray(axis, dtype=np.float32)
    ax = ax / np.linalg.norm(ax)
    theta = np.radians(angle_deg)
    c, s = np.cos(theta), np.sin(theta)
    cen = masked_centroid_f32(p, mask_hw)
    q = (p - cen).reshape(-1, 3)                      # f32
    t1 = q * c                                        # f64
    cr = np.empty_like(q)
    cr[:, 0] = ax[1] * q[:, 2] - ax[2] * q[:, 1]
    cr[:, 1] = ax[2] * q[:, 0] - ax[0] * q[:, 2]
    cr[:, 2] = ax[0] * q[:, 1] - ax[1] * q[:, 0]
    t2 = cr * s                                       # f64
    if DOT_ORDER == "explicit":
        d = (q[:, 0] * ax[0] + q[:, 1] * ax[1]) + q[:, 2] * ax[2]        # f32 elementwise: one rounding per operation
    else:
        d = np.dot(q, ax)                             # f32 (BLAS order for general axes)
    t3 = ax * d[:, None] * (1 - c)                    # f32*f32 -> f32, then f64
    out = (t1 + t2 + t3).reshape(h, w, 3) + cen + np.array([translation[0], translation[1], translation[2]], dtype=np.float64)
    return out


def project_points(points_n3, K, out_hw):
    """float64 pinhole projection + clip + round-half-even -> integer pixel coords."""
    p = np.asarray(points_n3, dtype=np.float64)
    k = K.numpy().astype(np.float64) if isinstance(K, torch.Tensor) else np.asarray(K, dtype=np.float64)
    h, w = out_hw
    x, y, z = -p[:, 0], -p[:, 1], p[:, 2]
    u = (k[0, 0] * x) / z
    v = (k[1, 1] * y) / z
    m = max(h, w) - 1
    u = (u * 0.5 + 0.5) * m
    v = (v * 0.5 + 0.5) * m
    ui = np.around(np.clip(u, 0, w - 1)).astype(np.int64)
    vi = np.around(np.clip(v, 0, h - 1)).astype(np.int64)
    return ui, vi


def zbuffer(points_n3, flags, K, out_hw):
    """Winner per pixel = lexicographic min over (z, point index).

    Returns depth_map f32 [H,W] (inf where empty), fg_mask bool [H,W] (pixel won by a
    flagged point), u[vis], v[vis] (int64, point order), vis bool [N] -- the collapsed
    semantics of the reference's sequential loop when unflagged points precede flagged
    ones (its only call site).
    """
    p = np.asarray(points_n3, dtype=np.float64)
    h, w = out_hw
    ui, vi = project_points(p, K, out_hw)
    pix = vi * w + ui
    n = p.shape[0]
    order = np.lexsort((np.arange(n), p[:, 2], pix))
    first = np.ones(n, dtype=bool)
    first[1:] = pix[order][1:] != pix[order][:-1]
    winners = order[first]
    depth = np.full(h * w, np.inf)
    depth[pix[winners]] = p[winners, 2]
    flags = np.asarray(flags).astype(bool)
    fg = np.zeros(h * w, dtype=bool)
    fg[pix[winners]] = flags[winners]
    vis = np.zeros(n, dtype=bool)
    vis[winners] = flags[winners]
    return (depth.reshape(h, w).astype(np.float32), fg.reshape(h, w), ui[vis], vi[vis], vis)


def zbuffer_sequential(points_n3, flags, K, out_hw):
    """Literal sequential restatement (small inputs only) used to cross-check zbuffer()."""
    p = np.asarray(points_n3, dtype=np.float64)
    h, w = out_hw
    ui, vi = project_points(p, K, out_hw)
    depth = np.full((h, w), np.inf)
    owner = np.full((h, w), -1, dtype=np.int64)
    for i in range(p.shape[0]):
        if p[i, 2] < depth[vi[i], ui[i]]:
            depth[vi[i], ui[i]] = p[i, 2]
            owner[vi[i], ui[i]] = i
    flags = np.asarray(flags).astype(bool)
    vis = np.zeros(p.shape[0], dtype=bool)
    won = owner[owner >= 0]
    vis[won] = flags[won]
    fg = np.zeros((h, w), dtype=bool)
    fg[owner >= 0] = flags[owner[owner >= 0]]
    return depth.astype(np.float32), fg, ui[vis], vi[vis], vis


# ---------------------------------------------------------------------------------------
# OpenCV morphology restated [ext]; SURVEY Appendix C.  Parity with cv2 itself is unpinned.
# ---------------------------------------------------------------------------------------

def ellipse_kernel(kw, kh):
    """cv2.getStructuringElement(MORPH_ELLIPSE, (kw, kh)) -- classic row-span algorithm."""
    if kw == 1 and kh == 1:
        return np.ones((1, 1), dtype=np.uint8)
    r, c = kh // 2, kw // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    k = np.zeros((kh, kw), dtype=np.uint8)
    for i in range(kh):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            j1, j2 = max(c - dx, 0), min(c + dx + 1, kw)
            k[i, j1:j2] = 1
    return k


def _morph(img_u8, kernel, is_dilate):
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    h, w = img_u8.shape
    fill = 0 if is_dilate else 255
    pad = np.full((h + kh, w + kw), fill, dtype=np.uint8)
    pad[ay:ay + h, ax:ax + w] = img_u8
    out = np.full((h, w), fill, dtype=np.uint8)
    for i in range(kh):
        for j in range(kw):
            if kernel[i, j]:
                win = pad[i:i + h, j:j + w]
                out = np.maximum(out, win) if is_dilate else np.minimum(out, win)
    return out


def dilate(img_u8, kernel):
    return _morph(img_u8, kernel, True)


def erode(img_u8, kernel):
    return _morph(img_u8, kernel, False)


def morph_close(img_u8, kernel):
    return erode(dilate(img_u8, kernel), kernel)


def morph_open(img_u8, kernel):
    return dilate(erode(img_u8, kernel), kernel)


def clean_mask(raw_mask_bool, img_res):
    """CLOSE with ellipse(res//50) then OPEN with ellipse(res//250) on a {0,255} mask."""
    m = raw_mask_bool.astype(np.uint8) * 255
    kc = ellipse_kernel(img_res // 50, img_res // 50)
    ko = ellipse_kernel(max(img_res // 250, 1), max(img_res // 250, 1))   # cv2 rejects a 0x0 kernel (res < 250)
    return morph_open(morph_close(m, kc), ko)


def harmonic_fill(image, mask):
    """5-point Laplace in-fill: diag 4, -1 to masked neighbours, known neighbours on the
    right-hand side, zero Dirichlet beyond the image border (float64 direct solve)."""
    img = np.asarray(image)
    mk = np.asarray(mask).astype(bool)
    ys, xs = np.nonzero(mk)
    n = ys.size
    out = img.copy()
    if n == 0:
        return out
    h, w = img.shape
    idx = -np.ones((h, w), dtype=np.int64)
    idx[ys, xs] = np.arange(n)
    rows, cols, vals = [np.arange(n)], [np.arange(n)], [np.full(n, 4.0)]
    b = np.zeros(n)
    for dy, dx in ((-1, 0), (1, 0), (0, -1), (0, 1)):
        yy, xx = ys + dy, xs + dx
        inside = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        yc, xc = np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)
        unk = inside & mk[yc, xc]
        known = inside & ~mk[yc, xc]
        rows.append(np.nonzero(unk)[0]); cols.append(idx[yc[unk], xc[unk]]); vals.append(np.full(unk.sum(), -1.0))
        b[known] += img[yc[known], xc[known]]
    A = scipy.sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    sol = scipy.sparse.linalg.spsolve(A, b)
    out[ys, xs] = sol
    return out


def transform_depth_pc(depth, bg_depth, fg_mask, K=None, rot_angle=None, rot_axis=None,
                       translation=None, use_input_depth_normalization=False, return_debug=False):
    """Whole 'pc' mode edit: returns (disparity [1,1,H,W] f32 torch, correspondences [N,4] int64 torch)."""
    if K is None:
        K = intrinsics_f32()
    empty = torch.zeros((0, 4), dtype=torch.int64)
    bounds = None
    if use_input_depth_normalization:
        _, bounds = normalize_depth(1.0 / depth)
    if not bool(fg_mask.any()):
        return normalize_depth(1.0 / depth, bounds)[0], empty
    rot_angle = 0.0 if rot_angle is None else float(rot_angle)
    rot_axis = np.array([0, 1, 0], np.float32) if rot_axis is None else np.asarray(rot_axis, dtype=np.float32)
    translation = np.zeros(3, np.float32) if translation is None else np.asarray(translation, dtype=np.float32)
    if fg_mask.shape[-2] != fg_mask.shape[-1]:
        raise RuntimeError(f"Expected fg_mask to be square, got shape {fg_mask.shape[-2]} x {fg_mask.shape[-1]}.")
    res = fg_mask.shape[-1]
    mask = fg_mask[0, 0].numpy().astype(bool)
    bg_pts = unproject(bg_depth[0, 0].numpy(), K)
    pts = unproject(depth[0, 0].numpy(), K)
    moved = rigid_transform(pts, rot_axis, rot_angle, [float(t) for t in translation], mask)
    all_pts = np.vstack([bg_pts.reshape(-1, 3).astype(np.float64), moved.reshape(-1, 3)[mask.reshape(-1)]])
    flags = np.zeros(all_pts.shape[0], dtype=np.uint8)
    flags[res * res:] = 1
    zmap, raw_mask, tx, ty, vis = zbuffer(all_pts, flags, K, (res, res))
    disparity = normalize_depth(1.0 / torch.from_numpy(zmap)[None, None], bounds)[0][0, 0].numpy()
    fg_idx = np.nonzero(mask.reshape(-1))[0]
    src = fg_idx[vis[res * res:]]
    oy, ox = src // res, src % res
    cleaned = clean_mask(raw_mask, res)
    keep = cleaned[ty, tx] == 255
    corr = np.stack([ox[keep], oy[keep], tx[keep], ty[keep]], axis=-1).astype(np.int64).reshape(-1, 4)
    inpaint = (cleaned != 0) != raw_mask
    filled = harmonic_fill(disparity, inpaint.astype(np.uint8))
    out = torch.from_numpy(filled).to(torch.float32)[None, None]
    if return_debug:
        return out, torch.from_numpy(corr), dict(zmap=zmap, raw_mask=raw_mask, cleaned=cleaned, vis=vis,
                                                   tx=tx, ty=ty, inpaint=inpaint, disparity=disparity)
    return out, torch.from_numpy(corr)


def laplacian_blend(fg_depth, bg_depth, mask):
    """utils.solve_laplacian_depth (utils.py:49-102) restated: harmonic_fill with the background depth's
    5-point Laplacian (zero padding) on the right-hand side."""
    import scipy.ndimage
    fg = np.asarray(fg_depth)
    mk = np.asarray(mask).astype(bool)
    lap = scipy.ndimage.convolve(np.asarray(bg_depth), np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]]), mode="constant")
    ys, xs = np.nonzero(mk)
    n = ys.size
    out = fg.copy()
    if n == 0:
        return out
    h, w = fg.shape
    idx = -np.ones((h, w), dtype=np.int64)
    idx[ys, xs] = np.arange(n)
    rows, cols, vals = [np.arange(n)], [np.arange(n)], [np.full(n, 4.0)]
    b = np.zeros(n)
    for dy, dx in ((-1, 0), (1, 0), (0, -1), (0, 1)):
        yy, xx = ys + dy, xs + dx
        inside = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        yc, xc = np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)
        unk = inside & mk[yc, xc]
        known = inside & ~mk[yc, xc]
        rows.append(np.nonzero(unk)[0]); cols.append(idx[yc[unk], xc[unk]]); vals.append(np.full(unk.sum(), -1.0))
        b[known] += fg[yc[known], xc[known]]
    b -= lap[ys, xs]
    A = scipy.sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    out[ys, xs] = scipy.sparse.linalg.spsolve(A, b)
    return out


def set_foreground(depth, fg_mask, bg_depth):
    """DiffusionHandles.set_foreground (diffusion_handles.py:90-111) on [1,1,H,W] torch tensors."""
    import scipy.ndimage
    m = scipy.ndimage.binary_dilation(fg_mask[0, 0].numpy(), iterations=15)
    out = laplacian_blend(depth[0, 0].numpy(), bg_depth[0, 0].numpy(), m)
    return torch.from_numpy(out)[None, None]
