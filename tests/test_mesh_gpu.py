"""GPU tests of depth_transform_mode='mesh' (csrc/mesh.hip through the C ABI).

Parity is unpinned for this mode (pytorch3d is not available, see oracle/mesh_ref.py): the HIP rasteriser is
checked (1) bit for bit against the oracle's float32 NumPy restatement of the same rule, and (2) against the
PINNED point z-buffer path on a smooth scene, where both modes must name (almost) the same correspondences."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(res):
    from diffusionhandles_amd.synthetic import make_scene
    return make_scene(res)


def _intr():
    from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
    return GuidedStableDiffuser.get_depth_intrinsics()


CASES = [(0.0, (0.0, 1.0, 0.0), (0.0, 0.0, 0.0)), (25.0, (0.0, 1.0, 0.0), (0.3, 0.0, 0.2)),
         (-40.0, (0.0, 1.0, 0.0), (-0.6, 0.1, 0.4)), (17.0, (0.3, 1.0, -0.2), (0.1, -0.05, -0.3))]


@pytest.mark.parametrize("res", [32, 64])
def test_mesh_matches_oracle_bit_exact(res):
    from diffusionhandles_amd import depth_transform as DT
    from oracle import mesh_ref as M
    dev = torch.device("cuda:0")
    depth, bg_depth, mask = _scene(res)
    K = _intr()
    gx = torch.linspace(-1, 1, res, dtype=torch.float32).numpy()
    lin01 = torch.linspace(0, 1, res, dtype=torch.float32).numpy()
    invf = float(torch.linalg.inv(K)[0, 0])
    f = float(K[0, 0])
    for ang, axis, tr in CASES:
        disp, corr, dbg = DT.transform_depth_mesh(depth.to(dev), bg_depth.to(dev), mask.to(dev), K, ang,
                                                  torch.tensor(axis), torch.tensor(tr), return_debug=True)
        ref = M.mesh_reproject(depth[0, 0].numpy(), bg_depth[0, 0].numpy(), mask[0, 0].numpy() > 0.5, gx, lin01, invf, f,
                               dbg["xform"], blur=DT.MESH_BLUR_RADIUS)
        assert np.array_equal(dbg["fg_flag"].cpu().numpy().astype(bool), ref["fg_flag"])
        assert np.array_equal(corr.numpy(), ref["corr"])
        assert np.array_equal(dbg["zmap"].cpu().numpy(), ref["zmap"])
        assert np.allclose(disp[0, 0].cpu().numpy(), ref["disparity"], rtol=0, atol=1e-4)
        assert corr.dtype == torch.int64 and corr.device.type == "cpu" and disp.shape == (1, 1, res, res)


def test_mesh_agrees_with_point_zbuffer_on_smooth_depth():
    """The pinned 'pc' path and the mesh path re-project the same surface: the foreground silhouettes must
    overlap almost everywhere and shared target pixels must name source pixels within ~1 px."""
    from diffusionhandles_amd import depth_transform as DT
    dev = torch.device("cuda:0")
    res = 256
    depth, bg_depth, mask = _scene(res)
    K = _intr()
    for ang, axis, tr in CASES[:3]:
        args = (depth.to(dev), bg_depth.to(dev), mask.to(dev), K, ang, torch.tensor(axis), torch.tensor(tr))
        dm, cm = DT.transform_depth(*args, depth_transform_mode="mesh")
        dp, cp = DT.transform_depth(*args, depth_transform_mode="pc")
        tm = {(int(c[2]), int(c[3])): (int(c[0]), int(c[1])) for c in cm}
        tp = {(int(c[2]), int(c[3])): (int(c[0]), int(c[1])) for c in cp}
        both = set(tm) & set(tp)
        # the point path leaves holes where the moved points spread apart; the mesh fills them: pc targets are a subset
        assert len(both) > 0.97 * len(tp) and len(tm) < 1.6 * len(tp), (len(tm), len(tp), len(both))
        err = np.array([max(abs(tm[k][0] - tp[k][0]), abs(tm[k][1] - tp[k][1])) for k in both])
        assert (err <= 2).mean() > 0.97, (err <= 2).mean()
        # disparity maps agree away from the silhouette
        diff = (dm - dp).abs()[0, 0]
        assert float(diff.median()) < 1.0


def test_mesh_empty_mask_and_dispatch():
    from diffusionhandles_amd import depth_transform as DT
    dev = torch.device("cuda:0")
    depth, bg_depth, mask = _scene(64)
    K = _intr()
    disp, corr = DT.transform_depth(depth.to(dev), bg_depth.to(dev), torch.zeros_like(mask).to(dev), K, 10.0,
                                    depth_transform_mode="mesh")
    assert corr.shape == (0, 4) and corr.dtype == torch.int64
    assert torch.allclose(disp.cpu(), DT.normalize_depth(1.0 / depth))
    with pytest.raises(ValueError):
        DT.transform_depth(depth.to(dev), bg_depth.to(dev), mask.to(dev), K, depth_transform_mode="voxels")


def test_mesh_real_scene_matches_oracle_bit_exact():
    """Estimated (noisy, discontinuous) depth of a scene of the reference's test data, down-sampled to 128, with the
    scene's own transforms (identity, 91 degrees + shift, pure translation)."""
    import os
    from diffusionhandles_amd import depth_transform as DT
    from diffusionhandles_amd import scene_io as S
    from oracle import mesh_ref as M
    res = 128
    dev = torch.device("cuda:0")
    sc = S.load_scene(os.path.join(os.path.dirname(__file__), "golden", "scene_banana_fruits"), res)
    depth, bg_depth, mask = sc["depth"], sc["bg_depth"], sc["fg_mask"]
    K = _intr()
    gx = torch.linspace(-1, 1, res, dtype=torch.float32).numpy()
    lin01 = torch.linspace(0, 1, res, dtype=torch.float32).numpy()
    invf, f = float(torch.linalg.inv(K)[0, 0]), float(K[0, 0])
    for name, t in sc["transforms"].items():
        kw = S.transform_args(t)
        disp, corr, dbg = DT.transform_depth_mesh(depth.to(dev), bg_depth.to(dev), mask.to(dev), K, kw["rot_angle"],
                                                  kw["rot_axis"], kw["translation"], return_debug=True)
        ref = M.mesh_reproject(depth[0, 0].numpy(), bg_depth[0, 0].numpy(), mask[0, 0].numpy() > 0.5, gx, lin01, invf, f,
                               dbg["xform"], blur=DT.MESH_BLUR_RADIUS)
        assert np.array_equal(dbg["fg_flag"].cpu().numpy().astype(bool), ref["fg_flag"]), name
        assert np.array_equal(corr.numpy(), ref["corr"]), name
        assert np.array_equal(dbg["zmap"].cpu().numpy(), ref["zmap"]), name
        assert np.allclose(disp[0, 0].cpu().numpy(), ref["disparity"], rtol=0, atol=1e-4), name
        assert corr.shape[0] > 100, name
