"""CPU: the GEOMETRY of depth_transform_mode='mesh' in oracle/mesh_ref.py against g13 -- what the reference's own
`depth_to_mesh` (depth_transform.py:30-71) and `transform_points` (:438-458) produced on the synthetic scene
(tools/make_golden_mesh.py).  pytorch3d is absent, so the rasterisation stays [ext] parity-unpinned; the vertices, the
two counter-clockwise triangles per pixel quad and the rigid motion of the masked vertices are pinned here."""
import numpy as np
import pytest
import torch

from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene
from oracle import depth_ref as D
from oracle import mesh_ref as M


def _fg_faces(mask_flat, R):
    q = np.arange((R - 1) * (R - 1))
    y, x = q // (R - 1), q % (R - 1)
    v00 = y * R + x
    f = np.stack([np.stack([v00 + R, v00 + 1, v00], -1), np.stack([v00 + R, v00 + R + 1, v00 + 1], -1)], 1).reshape(-1, 3)
    return f[mask_flat[f].all(axis=1)]


@pytest.mark.parametrize("R", [64, 128])
def test_mesh_geometry_matches_reference(golden, R):
    g = golden("g13_mesh.npz")
    depth, bg, mask = make_scene(R)
    m = (mask[0, 0] > 0.5).numpy()
    # faces: the oracle's rasteriser enumerates quads in this order with these vertex triples (mesh_ref.mesh_reproject)
    assert np.array_equal(_fg_faces(m.reshape(-1), R), g[f"r{R}_fg_faces"].astype(np.int64))
    assert int(g[f"r{R}_n_bg_faces"]) == 2 * (R - 1) * (R - 1)
    gx = torch.linspace(-1, 1, R).numpy().astype(np.float32)
    invf = np.float32(torch.linalg.inv(D.intrinsics_f32())[0, 0])
    idx = np.nonzero(m.reshape(-1))[0]
    for ti in (1, 3, 5):
        ang, tr = TRANSFORMS[ti]
        cen = g[f"r{R}_t{ti}_centroid"]
        th = np.float32(ang) * np.float32(np.pi / 180.0)
        xf = [0.0, 1.0, 0.0, np.cos(th, dtype=np.float32), np.sin(th, dtype=np.float32), *tr, *cen]
        X, Y, Z = M._unproject(depth[0, 0].numpy().astype(np.float32), gx, invf)
        # the centroid itself: a float32 mean of up to 10^4 masked vertices, whose value depends on the summation order
        # (torch's blocked sum here, NumPy's pairwise one, the product's sequential one): they agree to a few 1e-6, which
        # is why this mode is compared at a tolerance and not bit for bit with the reference
        pts = np.stack([X, Y, Z], -1).reshape(-1, 3)[idx]
        assert np.abs(pts.mean(axis=0, dtype=np.float32) - cen).max() < 1e-5
        Xm, Ym, Zm = M.rodrigues_f32(X, Y, Z, xf)
        moved = np.stack([Xm, Ym, Zm], -1).reshape(-1, 3)[idx]
        assert np.abs(moved - g[f"r{R}_t{ti}_fg_verts"]).max() < 2e-6, ti
