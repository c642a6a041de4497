"""Depth re-projection behind the reference's `transform_depth` API.

Mirrors /root/reference/diffhandles/depth_transform.py:73-89 (`transform_depth`),
:198-363 (`transform_depth_pc`), :15-28 (`normalize_depth`), :589-641
(`depth_to_world_coords`).  All arithmetic runs in the HIP library
(csrc/geometry.hip) through the C ABI; PyTorch only owns the device buffers.
`reproject_edits` is the batched form (K rigid transforms of one image in one launch
sequence) that the reference does not have.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def normalize_depth(depth, bounds=None, return_bounds=False):
    """255 * (d - min) / (max - min) per sample (depth_transform.py:15-28)."""
    if depth.dim() != 4:
        raise RuntimeError(f"Expected depth to have 4 dimensions, got {depth.dim()}")
    if bounds is None:
        flat = depth.reshape(depth.shape[0], -1)
        hi = flat.max(dim=-1).values[..., None, None, None]
        lo = flat.min(dim=-1).values[..., None, None, None]
    else:
        lo, hi = bounds
    out = 255 * (depth - lo) / (hi - lo)
    return (out, (lo, hi)) if return_bounds else out


_GRID_CACHE = {}


def _grids(h, w, device):
    key = (h, w, str(device))
    if key not in _GRID_CACHE:
        m = max(h, w) - 1
        gx = torch.linspace(-(w - 1) / m, (w - 1) / m, steps=w, dtype=torch.float32)   # host values, see DESIGN.md
        gy = torch.linspace(-(h - 1) / m, (h - 1) / m, steps=h, dtype=torch.float32)
        _GRID_CACHE[key] = (gx.to(device), gy.to(device))
    return _GRID_CACHE[key]


def _compute_device(t):
    if t.is_cuda:
        return t.device
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def _inv_focal(intrinsics):
    kinv = torch.linalg.inv(intrinsics.detach().to("cpu", torch.float32))
    return float(kinv[0, 0]), float(kinv[1, 1])


def depth_to_world_coords(depth, intrinsics, extrinsics_R=None, extrinsics_t=None):
    """[1,1,H,W] depth -> [H,W,3] float32 points (depth_transform.py:589-641)."""
    if depth.shape[0] != 1:
        raise ValueError("Only batch size 1 is supported")
    if extrinsics_R is not None or extrinsics_t is not None:
        raise NotImplementedError("only identity extrinsics are on the hot path")
    h, w = depth.shape[-2:]
    if h < 2 or w < 2:
        raise RuntimeError(f"Expected depth to have at least 2 pixels in each dimension, got {h} x {w}.")
    if h != w:
        raise RuntimeError("square depth maps only")
    dev = _compute_device(depth)
    d = depth.detach().to(dev, torch.float32).contiguous()
    gx, gy = _grids(h, w, dev)
    ifx, ify = _inv_focal(intrinsics)
    pts = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
    L = _lib.lib()
    _lib.check(L.dh_unproject(_lib.ptr(d), h, _lib.ptr(gx), _lib.ptr(gy), ifx, ify, _lib.ptr(pts), _lib.stream_ptr()),
               "dh_unproject")
    return pts.to(depth.device)


def _xform_rows(transforms):
    """(angle_deg, axis, translation) -> 8 doubles per edit, with NumPy's own cos/sin."""
    rows = np.zeros((len(transforms), 8), dtype=np.float64)
    for i, (angle, axis, trans) in enumerate(transforms):
        ax = np.asarray(axis.detach().cpu().numpy() if isinstance(axis, torch.Tensor) else axis, dtype=np.float32)
        ax = ax / np.linalg.norm(ax)
        tr = trans.detach().cpu() if isinstance(trans, torch.Tensor) else torch.as_tensor(trans, dtype=torch.float32)
        theta = np.radians(float(angle) if not isinstance(angle, torch.Tensor) else angle.item())
        rows[i, 0:3] = ax.astype(np.float64)
        rows[i, 3], rows[i, 4] = np.cos(theta), np.sin(theta)
        rows[i, 5:8] = [tr[0].item(), tr[1].item(), tr[2].item()]
    return rows


def reproject_edits(depth, bg_depth, fg_mask, intrinsics, transforms, use_input_depth_normalization=False,
                    return_debug=False, device_correspondences=False):
    """K edits of one image.  transforms: list of (rot_angle_deg, rot_axis[3], translation[3]).

    Returns a list of (disparity [1,1,H,W] f32 on depth.device, correspondences [N,4] int64 CPU),
    plus a dict of the intermediate device tensors when return_debug is set.  The reference hands the correspondences
    over as a CPU tensor (depth_transform.py:339-343) and so does `transform_depth`; device_correspondences=True leaves
    them where the kernels wrote them, for callers that feed them straight back to the device (the batched edit path:
    `process_correspondences` accepts either) -- K device-to-host copies and K uploads less per call.
    """
    if fg_mask.shape[-2] != fg_mask.shape[-1]:
        raise RuntimeError(f"Expected fg_mask to be square, got shape {fg_mask.shape[-2]} x {fg_mask.shape[-1]}.")
    if depth.dim() != 4 or depth.shape[0] != 1:
        raise ValueError("Only batch size 1 is supported")
    res = fg_mask.shape[-1]
    out_dev = depth.device
    dev = _compute_device(depth)
    L = _lib.lib()
    st = _lib.stream_ptr()
    d = depth.detach().to(dev, torch.float32).contiguous()
    bg = bg_depth.detach().to(dev, torch.float32).contiguous()
    mask_u8 = (fg_mask.detach().to(dev) != 0).to(torch.uint8).contiguous().view(-1)
    K = len(transforms)
    bounds = None
    if use_input_depth_normalization:
        disp_in = 1.0 / d
        bounds = torch.stack([disp_in.min(), disp_in.max()]).to(torch.float32).contiguous()
    n_fg = int(mask_u8.sum().item())
    if n_fg == 0:
        # empty foreground: the input disparity and no correspondences (depth_transform.py:203-216)
        lo_hi = None if bounds is None else (bounds[0], bounds[1])
        disp = normalize_depth(1.0 / d, bounds=lo_hi).to(out_dev)
        empty = torch.zeros((0, 4), dtype=torch.int64)
        res_list = [(disp, empty) for _ in range(K)]
        return (res_list, {}) if return_debug else res_list

    gx, gy = _grids(res, res, dev)
    ifx, ify = _inv_focal(intrinsics)
    kcpu = intrinsics.detach().to("cpu", torch.float32)
    fx, fy = float(kcpu[0, 0]), float(kcpu[1, 1])
    R2 = res * res
    fg_pix = torch.empty(R2, dtype=torch.int32, device=dev)
    n_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_small = torch.empty(4096, dtype=torch.uint8, device=dev)
    _lib.check(L.dh_fg_pixel_list(_lib.ptr(mask_u8), res, _lib.ptr(fg_pix), _lib.ptr(n_dev), _lib.ptr(ws_small),
                                  ws_small.numel(), st), "dh_fg_pixel_list")
    nbytes = ctypes.c_size_t()
    _lib.check(L.dh_reproject_workspace_bytes(res, n_fg, K, ctypes.byref(nbytes)), "dh_reproject_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    zmap = torch.empty((K, res, res), dtype=torch.float32, device=dev)
    raw = torch.empty((K, res, res), dtype=torch.uint8, device=dev)
    clean = torch.empty((K, res, res), dtype=torch.uint8, device=dev)
    disp = torch.empty((K, res, res), dtype=torch.float32, device=dev)
    vis = torch.empty((K, n_fg), dtype=torch.uint8, device=dev)
    txy = torch.empty((K, n_fg, 2), dtype=torch.int32, device=dev)
    corr = torch.empty((K, n_fg, 4), dtype=torch.int64, device=dev)
    counts = torch.zeros((K, 4), dtype=torch.int32, device=dev)
    rows = np.ascontiguousarray(_xform_rows(transforms))
    _lib.check(L.dh_reproject_edits(
        _lib.ptr(d), _lib.ptr(bg), _lib.ptr(fg_pix), n_fg, res, _lib.ptr(gx), _lib.ptr(gy), ifx, ify, fx, fy, K,
        rows.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _lib.ptr(bounds),
        _lib.ptr(zmap), _lib.ptr(raw), _lib.ptr(clean), _lib.ptr(disp), _lib.ptr(vis), _lib.ptr(txy),
        _lib.ptr(corr), _lib.ptr(counts), _lib.ptr(ws), nbytes.value, st), "dh_reproject_edits")
    counts_h = counts.cpu()
    _check_infill(counts_h[:, 3])
    out = []
    for e in range(K):
        n = int(counts_h[e, 0])
        out.append((disp[e][None, None].to(out_dev), corr[e, :n] if device_correspondences else corr[e, :n].cpu()))
    if return_debug:
        dbg = dict(zmap=zmap, raw_mask=raw, clean_mask=clean, vis=vis, target_xy=txy, counts=counts_h,
                   fg_pix=fg_pix[:n_fg], corr_dev=corr)
        return out, dbg
    return out


def _check_infill(iterations):
    """counts[..., 3] = CG iterations of the harmonic in-fill; -1 = the multi-workgroup solver's bounded grid barrier ran out
    (its workgroups were not co-resident: CU masks, another process holding the chip) -- an error, never a hang."""
    if bool((iterations < 0).any()):
        raise RuntimeError("harmonic in-fill: the multi-workgroup CG's grid barrier timed out (workgroups not co-resident); "
                           "the result is not valid")


def transform_depth_pc(depth, bg_depth, fg_mask, intrinsics, rot_angle=None, rot_axis=None, translation=None,
                       use_input_depth_normalization=False):
    if rot_angle is None:
        rot_angle = 0.0
    if rot_axis is None:
        rot_axis = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float32)
    if translation is None:
        translation = torch.tensor([0.0, 0.0, 0.0], dtype=torch.float32)
    (disp, corr), = reproject_edits(depth, bg_depth, fg_mask, intrinsics, [(rot_angle, rot_axis, translation)],
                                    use_input_depth_normalization)
    return disp, corr


MESH_BLUR_RADIUS = 1e-5      # depth_transform.py:152


def mesh_xform(depth, fg_mask, intrinsics, rot_angle, rot_axis, translation):
    """The 11 float32 numbers dh_mesh_reproject takes: unit axis, cos, sin, translation, centroid of the masked
    points (transform_points, depth_transform.py:438-458; the centroid is the sequential float32 mean the point
    path uses -- torch's own reduction order is not specified)."""
    dev = _compute_device(depth)
    res = depth.shape[-1]
    d = depth.detach().to(dev, torch.float32).contiguous()
    fg_pix = torch.nonzero((fg_mask.detach().to(dev)[0, 0] > 0.5).reshape(-1)).to(torch.int32).reshape(-1).contiguous()
    gx, gy = _grids(res, res, dev)
    ifx, ify = _inv_focal(intrinsics)
    cen = torch.empty(3, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().dh_masked_centroid(_lib.ptr(d), _lib.ptr(fg_pix), fg_pix.numel(), res, _lib.ptr(gx), _lib.ptr(gy),
                                             ifx, ify, _lib.ptr(cen), _lib.stream_ptr()), "dh_masked_centroid")
    ax = np.asarray(rot_axis.detach().cpu().numpy() if isinstance(rot_axis, torch.Tensor) else rot_axis, dtype=np.float32)
    ax = (ax / np.float32(np.linalg.norm(ax))).astype(np.float32)
    ang = np.float32(rot_angle.item() if isinstance(rot_angle, torch.Tensor) else rot_angle)
    theta = np.float32(ang * np.float32(np.pi / 180.0))
    tr = translation.detach().cpu().numpy() if isinstance(translation, torch.Tensor) else np.asarray(translation)
    c = cen.cpu().numpy()
    return np.array([ax[0], ax[1], ax[2], np.cos(theta), np.sin(theta), tr[0], tr[1], tr[2], c[0], c[1], c[2]],
                    dtype=np.float32)


def transform_depth_mesh(depth, bg_depth, fg_mask, intrinsics, rot_angle=None, rot_axis=None, translation=None,
                         use_input_depth_normalization=False, return_debug=False):
    """depth_transform.py:91-195 with our own HIP triangle rasteriser in place of pytorch3d (parity unpinned)."""
    if depth.dim() != 4 or depth.shape[0] != 1:
        raise ValueError("Only batch size 1 is supported")
    res = depth.shape[-1]
    if depth.shape[-2] != res:
        raise RuntimeError("square depth maps only")
    out_dev = depth.device
    dev = _compute_device(depth)
    d = depth.detach().to(dev, torch.float32).contiguous()
    bounds = None
    if use_input_depth_normalization:
        disp_in = 1.0 / d
        bounds = torch.stack([disp_in.min(), disp_in.max()]).to(torch.float32).contiguous()
    mask = (fg_mask.detach().to(dev)[0, 0] > 0.5)
    if not bool(mask.any()):
        lo_hi = None if bounds is None else (bounds[0], bounds[1])
        return normalize_depth(1.0 / d, bounds=lo_hi).to(out_dev), torch.zeros((0, 4), dtype=torch.int64)
    if rot_angle is None:
        rot_angle = 0.0
    if rot_axis is None:
        rot_axis = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float32)
    if translation is None:
        translation = torch.tensor([0.0, 0.0, 0.0], dtype=torch.float32)
    xf = mesh_xform(depth, fg_mask, intrinsics, rot_angle, rot_axis, translation)
    L = _lib.lib()
    bg = bg_depth.detach().to(dev, torch.float32).contiguous()
    m8 = mask.to(torch.uint8).contiguous()
    gx, _ = _grids(res, res, dev)
    lin01 = torch.linspace(0, 1, res, dtype=torch.float32).to(dev)
    ifx, _ = _inv_focal(intrinsics)
    f = float(intrinsics.detach().to("cpu", torch.float32)[0, 0])
    R2 = res * res
    zmap = torch.empty(R2, dtype=torch.float32, device=dev)
    disp = torch.empty(R2, dtype=torch.float32, device=dev)
    flag = torch.empty(R2, dtype=torch.uint8, device=dev)
    corr = torch.empty((R2, 4), dtype=torch.int64, device=dev)
    counts = torch.zeros(4, dtype=torch.int32, device=dev)
    nb = ctypes.c_size_t()
    _lib.check(L.dh_mesh_workspace_bytes(res, ctypes.byref(nb)))
    ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    xf_c = (ctypes.c_float * 11)(*[float(v) for v in xf])
    _lib.check(L.dh_mesh_reproject(_lib.ptr(d), _lib.ptr(bg), _lib.ptr(m8), res, _lib.ptr(gx), _lib.ptr(lin01), ifx, f,
                                   ctypes.cast(xf_c, ctypes.c_void_p), _lib.ptr(bounds), float(MESH_BLUR_RADIUS),
                                   _lib.ptr(zmap), _lib.ptr(disp), _lib.ptr(flag), _lib.ptr(corr), _lib.ptr(counts),
                                   _lib.ptr(ws), nb.value, _lib.stream_ptr()), "dh_mesh_reproject")
    n = int(counts[0].item())
    out = disp.view(1, 1, res, res).to(out_dev), corr[:n].cpu()
    if return_debug:
        return out + (dict(zmap=zmap.view(res, res), fg_flag=flag.view(res, res), xform=xf),)
    return out


def transform_depth(depth, bg_depth, fg_mask, intrinsics, rot_angle=None, rot_axis=None, translation=None,
                    use_input_depth_normalization=False, depth_transform_mode="pc"):
    """Same signature and return value as the reference's transform_depth (depth_transform.py:73-89)."""
    if depth_transform_mode == "pc":
        return transform_depth_pc(depth, bg_depth, fg_mask, intrinsics, rot_angle, rot_axis, translation,
                                  use_input_depth_normalization)
    if depth_transform_mode == "mesh":
        return transform_depth_mesh(depth, bg_depth, fg_mask, intrinsics, rot_angle, rot_axis, translation,
                                    use_input_depth_normalization)
    raise ValueError(f"Unknown depth transform mode '{depth_transform_mode}'.")


def laplacian_depth_blend(depth, bg_depth, fg_mask, dilate_iterations=15):
    """set_foreground's background-depth update (diffusion_handles.py:90-111): `depth` everywhere except
    inside the dilated foreground mask, where the background depth's Laplacian is integrated from the
    surrounding depth values.  [1,1,H,W] tensors in, [1,1,H,W] float32 out (on depth.device)."""
    res = depth.shape[-1]
    if depth.shape[-2] != res:
        raise RuntimeError("square depth maps only")
    out_dev = depth.device
    dev = _compute_device(depth)
    L = _lib.lib()
    d = depth.detach().to(dev, torch.float32).contiguous()
    bg = bg_depth.detach().to(dev, torch.float32).contiguous()
    m = (fg_mask.detach().to(dev) != 0).to(torch.uint8).contiguous()
    out = torch.empty((res, res), dtype=torch.float32, device=dev)
    counts = torch.zeros(4, dtype=torch.int32, device=dev)
    nbytes = ctypes.c_size_t()
    _lib.check(L.dh_laplacian_blend_workspace_bytes(res, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    _lib.check(L.dh_laplacian_blend(_lib.ptr(d), _lib.ptr(bg), _lib.ptr(m), res, int(dilate_iterations), _lib.ptr(out),
                                    _lib.ptr(counts), _lib.ptr(ws), nbytes.value, _lib.stream_ptr()), "dh_laplacian_blend")
    _check_infill(counts[3:4].cpu())
    return out[None, None].to(out_dev)
