"""Guidance energy behind the reference's losses.py API, computed by the HIP library.

`process_correspondences` (guided_stable_diffuser.py:490-584) and the two loss functions
(losses.py:4-40) keep their names and argument meaning.  Activations may be given the
reference way ([C,h,w], any float dtype) or channels-last ([h,w,C], the engine's native
layout, `channels_last=True`); the return value of the loss functions is a scalar tensor.
`energy_and_grad` is the fused form the denoising loop uses: it returns the loss and
d(loss)/d(activations) from one call, with no autograd graph.
"""
import ctypes

import numpy as np
import torch

from . import _lib

GRID = 64


class ProcessedCorrespondences(dict):
    """The reference's dict of int64 index arrays, plus the device-side lists the kernels use."""
    device_lists = None


def process_correspondences(correspondences, img_res, bg_erosion=0, grid=GRID, device=None):
    """[N,4] int64 (ox,oy,tx,ty) -> dict with original_x/y, transformed_x/y, background_x/y[_orig|_trans]."""
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    L = _lib.lib()
    corr = torch.as_tensor(correspondences).reshape(-1, 4).to(dev, torch.int64).contiguous()
    n = corr.shape[0]
    G2 = grid * grid
    pairs = torch.empty((max(n, 1), 2), dtype=torch.int32, device=dev)
    bg_lists = torch.empty((3, G2), dtype=torch.int32, device=dev)
    bg_masks = torch.empty((3, G2), dtype=torch.uint8, device=dev)
    counts = torch.zeros(4, dtype=torch.int32, device=dev)
    nbytes = ctypes.c_size_t()
    _lib.check(L.dh_cells_workspace_bytes(n, grid, ctypes.byref(nbytes)), "dh_cells_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    _lib.check(L.dh_cells_from_correspondences(_lib.ptr(corr) if n else _lib.c_p(0), n, int(img_res), grid,
                                               int(bg_erosion), _lib.ptr(pairs), _lib.ptr(bg_lists),
                                               _lib.ptr(bg_masks), _lib.ptr(counts), _lib.ptr(ws), nbytes.value,
                                               _lib.stream_ptr()), "dh_cells_from_correspondences")
    c = counts.cpu().tolist()
    pairs = pairs[:c[0]].contiguous()
    lists = [bg_lists[k, :c[1 + k]].contiguous() for k in range(3)]
    ph = pairs.cpu().numpy().astype(np.int64).reshape(-1, 2)
    lh = [l.cpu().numpy().astype(np.int64) for l in lists]
    out = ProcessedCorrespondences({
        "original_x": ph[:, 0] % grid, "original_y": ph[:, 0] // grid,
        "transformed_x": ph[:, 1] % grid, "transformed_y": ph[:, 1] // grid,
        "background_x": lh[0] % grid, "background_y": lh[0] // grid,
        "background_x_orig": lh[1] % grid, "background_y_orig": lh[1] // grid,
        "background_x_trans": lh[2] % grid, "background_y_trans": lh[2] // grid,
    })
    out.device_lists = dict(pairs=pairs, bg_both=lists[0], bg_orig=lists[1], bg_trans=lists[2],
                            bg_masks=bg_masks.view(3, grid, grid), grid=grid)
    return out


def _device_lists(pc, dev, grid):
    dl = getattr(pc, "device_lists", None)
    if dl is not None and dl["pairs"].device == dev:
        return dl
    g = grid
    mk = lambda y, x: torch.as_tensor(np.asarray(pc[y]) * g + np.asarray(pc[x]), dtype=torch.int32, device=dev)
    pairs = torch.stack([mk("original_y", "original_x"), mk("transformed_y", "transformed_x")], dim=-1).contiguous()
    return dict(pairs=pairs, bg_both=mk("background_y", "background_x"),
                bg_orig=mk("background_y_orig", "background_x_orig"),
                bg_trans=mk("background_y_trans", "background_x_trans"), grid=g)


_WS = {}


class EnergyPlan:
    """Per-edit constants of the guidance energy on one cell grid: device index lists, the
    target-cell -> source-cells CSR and the transformed-background flags (dh_energy_plan_build)."""

    def __init__(self, processed_correspondences, grid, device):
        L = _lib.lib()
        self.grid = int(grid)
        self.dl = _device_lists(processed_correspondences, device, self.grid)
        self.n_pairs = int(self.dl["pairs"].shape[0])
        nb = ctypes.c_size_t()
        _lib.check(L.dh_energy_plan_bytes(self.grid, self.n_pairs, ctypes.byref(nb)), "dh_energy_plan_bytes")
        self.nbytes = nb.value
        self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=device)
        _lib.check(L.dh_energy_plan_build(_lib.ptr(self.dl["pairs"]), self.n_pairs, _lib.ptr(self.dl["bg_trans"]),
                                          self.dl["bg_trans"].numel(), self.grid, _lib.ptr(self.buf), self.nbytes,
                                          _lib.stream_ptr()), "dh_energy_plan_build")
        self._ws = {}

    def workspace(self, C):
        if C not in self._ws:
            nb = ctypes.c_size_t()
            _lib.check(_lib.lib().dh_energy_planned_workspace_bytes(C, self.grid, ctypes.byref(nb)))
            self._ws[C] = (torch.empty(nb.value, dtype=torch.uint8, device=self.buf.device), nb.value)
        return self._ws[C]


def energy_and_grad_planned(act, act_orig, plan, fg_weight, bg_weight, grad_scale=1.0, want_loss=False, out=None):
    """Default-configuration evaluation through a prebuilt EnergyPlan: act / act_orig [grid,grid,C] channels-last,
    16-bit, contiguous.  Returns (loss[3] or None, grad like act); `out`: where the gradient is written (e.g. the engine's
    own cotangent buffer of that activation)."""
    _lib.require_gpu(act)
    h, w, C = act.shape
    if h != plan.grid or w != plan.grid or act.dtype not in (torch.float16, torch.bfloat16) or act_orig.dtype != act.dtype:
        raise ValueError("planned energy: maps must be 16-bit [grid, grid, C] of one dtype")
    a = act.detach().contiguous()
    o = act_orig.detach().contiguous()
    if out is not None and (out.shape != a.shape or out.dtype != a.dtype or not out.is_contiguous()):
        raise ValueError("planned energy: `out` must be a contiguous tensor like the activation")
    grad = torch.empty_like(a) if out is None else out
    loss = torch.zeros(3, dtype=torch.float32, device=a.device) if want_loss else None
    ws, wsb = plan.workspace(C)
    dl = plan.dl
    _lib.check(_lib.lib().dh_energy_fwd_bwd_planned(
        _lib.ptr(a), _lib.ptr(o), _lib.DTYPE_CODE[a.dtype], C, plan.grid, _lib.ptr(plan.buf), plan.nbytes, plan.n_pairs,
        _lib.ptr(dl["bg_orig"]), dl["bg_orig"].numel(), _lib.ptr(dl["bg_trans"]), dl["bg_trans"].numel(),
        float(fg_weight), float(bg_weight), float(grad_scale), _lib.ptr(loss), _lib.ptr(grad),
        _lib.DTYPE_CODE[a.dtype], _lib.ptr(ws), wsb, _lib.stream_ptr()), "dh_energy_fwd_bwd_planned")
    return loss, grad


def energy_and_grad(act, act_orig, processed_correspondences, fg_weight, bg_weight, fg_patch_size=1,
                    bg_patch_size=1, activations_size=(GRID, GRID), bg_loss_type="global_avg", grad_scale=1.0,
                    grad_dtype=None, channels_last=True, out=None):
    """One activation layer: returns (loss[3] = {total, fg, bg} f32 device tensor, grad like `act`).

    act / act_orig: [h,w,C] (channels_last) or [C,h,w] device tensors of the same dtype.
    """
    _lib.require_gpu(act)
    if bg_loss_type not in ("global_avg", "local_avg"):
        raise ValueError(f"Unknown background loss type: {bg_loss_type}")
    grid = int(activations_size[0])
    if not channels_last:
        act = act.permute(1, 2, 0)
        act_orig = act_orig.permute(1, 2, 0)
    a = act.detach().contiguous()
    o = act_orig.detach().to(a.dtype).contiguous()
    h, w, C = a.shape
    dev = a.device
    dl = _device_lists(processed_correspondences, dev, grid)
    gdt = a.dtype if grad_dtype is None else grad_dtype
    if out is not None and (tuple(out.shape) != (h, w, C) or out.dtype != gdt or not out.is_contiguous() or not channels_last):
        raise ValueError("energy: `out` must be a contiguous channels-last tensor like the activation")
    grad = torch.empty((h, w, C), dtype=gdt, device=dev) if out is None else out
    loss = torch.zeros(3, dtype=torch.float32, device=dev)
    L = _lib.lib()
    n_pairs = dl["pairs"].shape[0]
    nbytes = ctypes.c_size_t()
    _lib.check(L.dh_energy_workspace_bytes(C, grid, n_pairs, ctypes.byref(nbytes)), "dh_energy_workspace_bytes")
    key = (str(dev), nbytes.value)
    if key not in _WS:
        _WS.clear()
        _WS[key] = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    ws = _WS[key]
    _lib.check(L.dh_energy_fwd_bwd(
        _lib.ptr(a), _lib.ptr(o), _lib.DTYPE_CODE[a.dtype], C, h, w, grid,
        _lib.ptr(dl["pairs"]), n_pairs, _lib.ptr(dl["bg_both"]), dl["bg_both"].numel(),
        _lib.ptr(dl["bg_orig"]), dl["bg_orig"].numel(), _lib.ptr(dl["bg_trans"]), dl["bg_trans"].numel(),
        float(fg_weight), float(bg_weight), int(fg_patch_size), int(bg_patch_size),
        0 if bg_loss_type == "global_avg" else 1, float(grad_scale), _lib.ptr(loss), _lib.ptr(grad),
        _lib.DTYPE_CODE[gdt], _lib.ptr(ws), nbytes.value, _lib.stream_ptr()), "dh_energy_fwd_bwd")
    if not channels_last:
        grad = grad.permute(2, 0, 1)
    return loss, grad


def compute_foreground_loss(activations, activations_orig, processed_correspondences, patch_size, activations_size):
    """losses.py:4-17 -- [C,h,w] activations, returns the scalar foreground term."""
    loss, _ = energy_and_grad(activations, activations_orig, processed_correspondences, 1.0, 0.0, patch_size, 1,
                              activations_size, channels_last=False)
    return loss[1]


def compute_background_loss(activations, activations_orig, processed_correspondences, patch_size, activations_size,
                            loss_type="global_avg"):
    """losses.py:19-40 -- returns the scalar background term."""
    if loss_type not in ("global_avg", "local_avg"):
        raise ValueError(f"Unknown background loss type: {loss_type}")
    loss, _ = energy_and_grad(activations, activations_orig, processed_correspondences, 0.0, 1.0, 1, patch_size,
                              activations_size, bg_loss_type=loss_type, channels_last=False)
    return loss[2]
