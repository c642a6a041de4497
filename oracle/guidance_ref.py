"""ORACLE (test infrastructure, never on the product path).

CPU restatement of the guidance side of the edit loop:
  correspondences -> 64x64 cell index lists   guided_stable_diffuser.py:490-584
  guidance energy (fg / bg terms)              losses.py:4-84
  step/iteration weight schedule               guided_stable_diffuser.py:336-373, 622-665
Pinned against the imported reference by tools/make_golden.py.
"""
import numpy as np
import scipy.ndimage
import torch
import torch.nn.functional as F

GRID = 64


def cells_from_correspondences(corr, img_res, bg_erosion=0, grid=GRID):
    """corr [N,4] int64 (ox,oy,tx,ty) -> dict of int64 index arrays on the grid x grid cells.

    Duplicates are kept (one entry per pixel pair).  bg masks: cells not touched by any
    original / transformed coordinate, optionally eroded (cross element, border 0).
    """
    c = np.asarray(corr, dtype=np.int64).reshape(-1, 4)
    ox, oy, tx, ty = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    ok = (tx >= 0) & (tx < img_res) & (ty >= 0) & (ty < img_res)
    ox, oy, tx, ty = ox[ok], oy[ok], tx[ok], ty[ok]
    cell = img_res // grid
    ox, oy, tx, ty = ox // cell, oy // cell, tx // cell, ty // cell
    bg_o = np.ones((grid, grid), dtype=bool)
    bg_t = np.ones((grid, grid), dtype=bool)
    if ox.size:
        bg_o[oy, ox] = False
        bg_t[ty, tx] = False
    if bg_erosion > 0:
        bg_o = scipy.ndimage.binary_erosion(bg_o, iterations=bg_erosion)
        bg_t = scipy.ndimage.binary_erosion(bg_t, iterations=bg_erosion)
    by, bx = np.nonzero(bg_o & bg_t)
    byo, bxo = np.nonzero(bg_o)
    byt, bxt = np.nonzero(bg_t)
    return {
        "original_x": ox, "original_y": oy, "transformed_x": tx, "transformed_y": ty,
        "background_x": bx, "background_y": by,
        "background_x_orig": bxo, "background_y_orig": byo,
        "background_x_trans": bxt, "background_y_trans": byt,
    }


def _to64(x, size):
    return F.interpolate(x[None], size, mode="bilinear")[0]


def _local_avg_l1(f1, f2, x1, y1, x2, y2, patch):
    h, w = f1.shape[-2:]
    w1 = torch.zeros((h, w), dtype=f1.dtype, device=f1.device)
    w2 = torch.zeros((h, w), dtype=f2.dtype, device=f2.device)
    w1[y1, x1] = 1
    w2[y2, x2] = 1
    pool = lambda t: F.avg_pool2d(t, patch, stride=1, padding=patch // 2)
    a1 = pool((w1 * f1)[None]) / (pool(w1[None, None]) + 1e-10)
    a2 = pool((w2 * f2)[None]) / (pool(w2[None, None]) + 1e-10)
    diff = (a1[0][:, y1, x1] - a2[0][:, y2, x2]).abs()
    return diff.mean(dim=-1).mean()


def foreground_energy(act, act_orig, cells, patch, size):
    """mean_c mean_n | A_orig[c, oy, ox] - A_cur[c, ty, tx] | with masked local averages."""
    return _local_avg_l1(_to64(act_orig, size), _to64(act, size),
                         cells["original_x"], cells["original_y"],
                         cells["transformed_x"], cells["transformed_y"], patch)


def background_energy(act, act_orig, cells, patch, size, loss_type="global_avg"):
    fo, fc = _to64(act_orig, size), _to64(act, size)
    if loss_type == "global_avg":
        m1 = fo[..., cells["background_y_orig"], cells["background_x_orig"]].mean(dim=-1)
        m2 = fc[..., cells["background_y_trans"], cells["background_x_trans"]].mean(dim=-1)
        return (m1 - m2).abs().mean()
    if loss_type == "local_avg":
        return _local_avg_l1(fo, fc, cells["background_x"], cells["background_y"],
                             cells["background_x"], cells["background_y"], patch)
    raise ValueError(f"Unknown background loss type: {loss_type}")


# ---------------------------------------------------------------------------------------

LAYER_PATTERN = {0: ([0.0, 0.0, 7.5], [0.0, 0.0, 1.5]),
                 1: ([0.0, 5.0, 0.0], [0.0, 1.5, 0.0]),
                 2: ([0.0, 5.0, 7.5], [0.0, 1.5, 1.5])}
ITER_MULT = [(2.5, 1.25), (1.25, 2.5), (1.25, 1.25), (2.5, 2.5)]


def guidance_weights(t_idx, iteration, fg_weight, bg_weight, max_step, schedule="constant"):
    """(t_idx, iteration) -> (fg 3-list, bg 3-list); fg/bg_weight are the user values (x30 inside)."""
    wf, wb = fg_weight * 30, bg_weight * 30
    if schedule == "constant":
        ff, fb = np.linspace(wf, wf, max_step), np.linspace(wb, wb, max_step)
    elif schedule == "linear":
        ff, fb = np.linspace(wf, 0.0, max_step), np.linspace(wb, 0.0, max_step)
    elif schedule == "quadratic":
        ff, fb = np.linspace(np.sqrt(wf), 0.0, max_step) ** 2, np.linspace(np.sqrt(wb), 0.0, max_step) ** 2
    else:
        raise ValueError(f"Unknown guidance schedule type: {schedule}")
    if t_idx >= max_step:
        dfg, dbg = [0.0] * 3, [0.0] * 3
    else:
        pf, pb = LAYER_PATTERN[t_idx % 3]
        dfg = (np.array(pf) * ff[t_idx]).tolist()
        dbg = (np.array(pb) * fb[t_idx]).tolist()
    mf, mb = ITER_MULT[min(iteration, 3)]
    return [d * mf for d in dfg], [d * mb for d in dbg]
