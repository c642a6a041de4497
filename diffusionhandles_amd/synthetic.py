"""Synthetic inputs for tests and benchmarks (no weights / EXR files needed).

The analytic scene of SURVEY.md section 8d: a receding background plane with a sphere in
front of it; the foreground mask is the sphere's silhouette.
"""
import numpy as np
import torch

# the 6-edit transform set (angle deg about +y, translation) + two more for batch-8
TRANSFORMS = [
    (0.0, (0.0, 0.0, 0.0)),
    (15.0, (0.0, 0.0, 0.0)),
    (30.0, (0.0, 0.0, 0.0)),
    (60.0, (0.0, 0.0, 0.0)),
    (0.0, (0.3, 0.0, 0.2)),
    (0.0, (-1.0, 0.0, 0.0)),
    (-20.0, (0.5, 0.0, -0.3)),
    (45.0, (-0.4, 0.1, 0.4)),
]


def make_scene(res=512):
    """Returns depth, bg_depth, fg_mask as [1,1,res,res] float32 CPU tensors."""
    h = res
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(h, dtype=np.float64), indexing="ij")
    bg = 4.0 + 1.5 * (h - 1 - yy) / h
    cy, cx, rad = h / 2.0, 300.0 * h / 512.0, 110.0 * h / 512.0
    r2 = ((yy - cy) ** 2 + (xx - cx) ** 2) / (rad * rad)
    mask = r2 < 1.0
    sphere = 2.6 - 0.6 * np.sqrt(np.clip(1.0 - r2, 0.0, None))
    depth = np.where(mask, sphere, bg)
    t = lambda a: torch.from_numpy(a.astype(np.float32))[None, None].contiguous()
    return t(depth), t(bg), t(mask.astype(np.float32))


def make_image(res=512, seed=7):
    """A smooth synthetic RGB image in [0,1], [1,3,res,res]."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, 16, 16, generator=g)
    return torch.nn.functional.interpolate(low, size=(res, res), mode="bicubic", align_corners=False).clamp(0, 1)
