#!/usr/bin/env python3
"""g13: the GEOMETRY of depth_transform_mode='mesh' pinned to the reference.

pytorch3d is absent, so the reference's rasterisation cannot run; everything in front of it can: `depth_to_mesh`
(depth_transform.py:30-71: vertices, the two counter-clockwise triangles per pixel quad, the per-vertex (u, v, fg-flag)
colour) and `transform_points` (:438-458, torch float32 Rodrigues about the centroid of the masked vertices) are pure
torch + diffhandles.mesh.Mesh.  This script imports them from /root/reference (stubs as in tools/make_golden.py), runs them
on the synthetic scene, asserts oracle/mesh_ref.py's vertex / face construction equal on the spot (faces exactly, vertices to
float32 rounding of the centroid mean) and writes tests/golden/g13_mesh.npz.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as MG  # noqa: E402
from oracle import depth_ref as D  # noqa: E402
from oracle import mesh_ref as M  # noqa: E402
from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene  # noqa: E402


def oracle_geometry(depth, bg, mask, xf):
    """What oracle/mesh_ref.mesh_reproject builds in front of its rasteriser: world-space vertices and face list."""
    R = depth.shape[0]
    gx = torch.linspace(-1, 1, R).numpy().astype(np.float32)
    invf = np.float32(torch.linalg.inv(D.intrinsics_f32())[0, 0])
    Xb, Yb, Zb = M._unproject(bg.astype(np.float32), gx, invf)
    Xf, Yf, Zf = M._unproject(depth.astype(np.float32), gx, invf)
    Xf, Yf, Zf = M.rodrigues_f32(Xf, Yf, Zf, xf)
    return np.stack([Xb, Yb, Zb], -1).reshape(-1, 3), np.stack([Xf, Yf, Zf], -1).reshape(-1, 3)


def faces_of(mask_flat, R, fg):
    """Face list (vertex indices in the full R x R grid) in mesh_ref's order: per quad upper-left then lower-right."""
    nq = (R - 1) * (R - 1)
    q = np.arange(nq)
    y, x = q // (R - 1), q % (R - 1)
    v00 = y * R + x
    up = np.stack([v00 + R, v00 + 1, v00], -1)
    lo = np.stack([v00 + R, v00 + R + 1, v00 + 1], -1)
    f = np.stack([up, lo], 1).reshape(-1, 3)
    if fg:
        f = f[mask_flat[f].all(axis=1)]
    return f


def main():
    MG.install_stubs()
    import diffhandles.depth_transform as RD
    K = D.intrinsics_f32()
    store = {}
    for R in (64, 128):
        depth, bg, mask = make_scene(R)
        m = mask[0, 0] > 0.5
        bg_mesh = RD.depth_to_mesh(depth=bg, intrinsics=K)
        fg_mesh = RD.depth_to_mesh(depth=depth, intrinsics=K, mask=m)
        # faces: the reference compacts the masked vertices; map back to grid indices
        grid_idx = torch.nonzero(m.reshape(-1))[:, 0].numpy()
        ref_bg_faces = bg_mesh.faces.numpy()
        ref_fg_faces = grid_idx[fg_mesh.faces.numpy()]
        assert np.array_equal(ref_bg_faces, faces_of(None, R, False)), "bg faces differ"
        assert np.array_equal(ref_fg_faces, faces_of(m.reshape(-1).numpy(), R, True)), "fg faces differ"
        col = fg_mesh.vert_attributes["color"].values.detach().numpy() if hasattr(fg_mesh, "vert_attributes") else None
        lin = torch.linspace(0, 1, R).numpy()
        for ti in (1, 3, 5):
            ang, tr = TRANSFORMS[ti]
            axis = torch.tensor([0.0, 1.0, 0.0])
            moved = RD.transform_points(points=fg_mesh.verts.detach().clone(), rot_angle=torch.tensor(float(ang)), rot_axis=axis,
                                        translation=torch.tensor(tr, dtype=torch.float32)).numpy()
            cen = fg_mesh.verts.detach().mean(dim=0).numpy()
            th = np.float32(ang) * np.float32(np.pi / 180.0)
            xf = [0.0, 1.0, 0.0, np.cos(th, dtype=np.float32), np.sin(th, dtype=np.float32), *tr, *cen]
            vb, vf = oracle_geometry(depth[0, 0].numpy(), bg[0, 0].numpy(), m.numpy(), xf)
            assert np.array_equal(vb, bg_mesh.verts.detach().numpy()), "bg vertices differ"
            err = np.abs(vf[grid_idx] - moved).max()
            assert err < 2e-6, (R, ti, err)
            store[f"r{R}_t{ti}_fg_verts"] = moved.astype(np.float32)
            store[f"r{R}_t{ti}_centroid"] = cen.astype(np.float32)
            print(f"g13 res {R} transform {ti}: {len(ref_fg_faces)} fg faces, moved vertices max diff {err:.2e}")
        store[f"r{R}_fg_faces"] = ref_fg_faces.astype(np.int32)
        store[f"r{R}_n_bg_faces"] = np.int64(len(ref_bg_faces))
        if col is not None:
            # colour = (u, v, 1): u, v = linspace(0, 1) image coordinates of the vertex (meshgrid 'xy')
            yy, xx = np.divmod(grid_idx, R)
            assert np.allclose(col[:, 0], lin[xx]) and np.allclose(col[:, 1], lin[yy]) and np.all(col[:, 2] == 1.0)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g13_mesh.npz"), **store)
    print("wrote tests/golden/g13_mesh.npz")


if __name__ == "__main__":
    main()
