"""CPU: the oracle (oracle/*.py) against the golden vectors captured from the reference
(tools/make_golden.py).  Integer outputs bit-exact; float outputs bit-exact where the
generator asserted equality, else at the stated tolerance."""
import hashlib

import numpy as np
import torch

from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene
from oracle import depth_ref as D
from oracle import guidance_ref as G
from oracle import loop_ref as L


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_unproject_matches_reference(golden):
    g = golden("g1_unproject.npz")
    depth, bg, _ = make_scene(512)
    pts = D.unproject(depth[0, 0].numpy())
    assert sha(pts) == str(g["points_sha"])
    assert sha(D.unproject(bg[0, 0].numpy())) == str(g["bg_points_sha"])
    assert np.array_equal(pts[::37, ::41], g["points_slice"])
    rd = torch.rand(1, 1, 512, 512, generator=torch.Generator().manual_seed(3)) * 5 + 0.5
    assert sha(D.unproject(rd[0, 0].numpy())) == str(g["rand_points_sha"])


def test_rigid_transform_matches_reference(golden):
    g = golden("g2_rigid.npz")
    depth, _, mask = make_scene(512)
    pts = D.unproject(depth[0, 0].numpy())
    m = mask[0, 0].numpy().astype(bool)
    for ti, (ang, tr) in enumerate(TRANSFORMS[:6]):
        out = D.rigid_transform(pts, [0, 1, 0], ang, tr, m)
        assert out.dtype == np.float64
        assert sha(out) == str(g[f"t{ti}_sha"]), f"transform {ti}"


def test_zbuffer_and_edit_match_reference(golden):
    g = golden("g3_zbuffer.npz")
    depth, bg, mask = make_scene(512)
    for ti in (0, 2, 5):
        ang, tr = TRANSFORMS[ti]
        disp, corr, dbg = D.transform_depth_pc(depth, bg, mask, rot_angle=ang, rot_axis=[0, 1, 0], translation=tr,
                                               return_debug=True)
        assert sha(dbg["zmap"]) == str(g[f"t{ti}_zmap_sha"])
        assert np.array_equal(np.packbits(dbg["raw_mask"]), g[f"t{ti}_mask"])
        assert np.array_equal(dbg["tx"], g[f"t{ti}_u"]) and np.array_equal(dbg["ty"], g[f"t{ti}_v"])
        assert np.array_equal(np.packbits(dbg["vis"][512 * 512:]), g[f"t{ti}_vis"])
        assert np.array_equal(corr.numpy(), g[f"t{ti}_corr"].astype(np.int64))
        assert np.array_equal(np.packbits(dbg["cleaned"] != 0), g[f"t{ti}_cleaned"])
        assert np.allclose(disp[0, 0].numpy()[::5, ::7], g[f"t{ti}_disp_slice"], atol=1e-4, rtol=0)


def test_zbuffer_tie_semantics(golden):
    g = golden("g3_zbuffer.npz")
    K = D.intrinsics_f32()
    for fn in (D.zbuffer, D.zbuffer_sequential):
        z, m, u, v, vis = fn(g["ties_pts"], g["ties_flags"], K, (32, 32))
        assert np.array_equal(z, g["ties_zmap"]) and np.array_equal(m, g["ties_mask"])
        assert np.array_equal(u, g["ties_u"]) and np.array_equal(v, g["ties_v"]) and np.array_equal(vis, g["ties_vis"])


def test_empty_mask_edge_case():
    depth, bg, mask = make_scene(64)
    disp, corr = D.transform_depth_pc(depth, bg, torch.zeros_like(mask))
    assert corr.shape == (0, 4) and corr.dtype == torch.int64
    assert torch.equal(disp, D.normalize_depth(1.0 / depth)[0])


def test_cells_match_reference(golden):
    g3, g4 = golden("g3_zbuffer.npz"), golden("g4_cells.npz")
    corr = g3["t2_corr"].astype(np.int64)
    for er in (0, 5, 10):
        c = G.cells_from_correspondences(corr, 512, er)
        for k, v in c.items():
            assert np.array_equal(v, g4[f"e{er}_{k}"].astype(np.int64)), (er, k)
    # off-image targets are dropped, empty input gives full background
    bad = np.array([[1, 1, -1, 5], [2, 2, 600, 5], [8, 8, 16, 24]], dtype=np.int64)
    c = G.cells_from_correspondences(bad, 512, 0)
    assert c["original_x"].tolist() == [1] and c["transformed_y"].tolist() == [3]
    assert G.cells_from_correspondences(np.zeros((0, 4), np.int64), 512, 0)["background_x"].size == 4096


def test_energy_matches_reference(golden):
    g3, g5 = golden("g3_zbuffer.npz"), golden("g5_energy.npz")
    corr = g3["t2_corr"].astype(np.int64)
    cells = {"e0": G.cells_from_correspondences(corr, 512, 0), "e5": G.cells_from_correspondences(corr, 512, 5)}
    for li in range(3):
        cur, org = torch.from_numpy(g5[f"l{li}_cur"]), torch.from_numpy(g5[f"l{li}_org"])
        for patch in (1, 3):
            for name in ("e0", "e5"):
                for kind in ("fg", "bg_global_avg", "bg_local_avg"):
                    a = cur.clone().requires_grad_(True)
                    if kind == "fg":
                        l = G.foreground_energy(a, org, cells[name], patch, (64, 64))
                    else:
                        l = G.background_energy(a, org, cells[name], patch, (64, 64), kind[3:])
                    gr, = torch.autograd.grad(l, a)
                    key = f"l{li}_p{patch}_{name}_{kind}"
                    assert np.float32(l.item()) == g5[key + "_loss"], key
                    assert np.array_equal(gr.numpy(), g5[key + "_grad"]), key


def test_schedule_matches_reference(golden):
    g6 = golden("g6_schedule.npz")
    for sched in ("constant", "linear", "quadratic"):
        tab = g6[sched]
        for t in range(50):
            for it in range(4):
                f, b = G.guidance_weights(t, it, 1.5, 1.25, 38, sched)
                assert f == tab[t, it, 0].tolist() and b == tab[t, it, 1].tolist()
    # act0 never carries weight; guidance stops at max_step
    assert all(G.guidance_weights(t, 0, 1.5, 1.25, 38)[0][0] == 0.0 for t in range(50))
    assert G.guidance_weights(38, 0, 1.5, 1.25, 38) == ([0.0] * 3, [0.0] * 3)


def test_misc_matches_reference(golden):
    g9 = golden("g9_misc.npz")
    depth, _, _ = make_scene(512)
    disp = D.normalize_depth(1.0 / depth)[0]
    assert np.array_equal(disp[0, 0].numpy()[::5, ::7], g9["disp_slice"])
    assert np.array_equal(L.init_depth(disp, (64, 64)).numpy(), g9["depth64"])
    out = D.harmonic_fill(g9["poisson_img"].copy(), g9["poisson_mask"])
    assert np.allclose(out, g9["poisson_out"], atol=1e-9)


def test_morphology_kernels():
    k10 = D.ellipse_kernel(10, 10)
    spans = [(int(r.argmax()), int(r.sum())) for r in k10]
    # per-row dx = 0,3,4,5,5,5,5,5,4,3 around column 5
    assert [s[1] for s in spans] == [1, 7, 9, 10, 10, 10, 10, 10, 9, 7]
    assert D.ellipse_kernel(2, 2).tolist() == [[0, 1], [1, 1]]
    img = np.zeros((20, 20), np.uint8); img[8:12, 8:12] = 255; img[9, 9] = 0
    closed = D.morph_close(img, D.ellipse_kernel(3, 3))
    assert closed[9, 9] == 255 and closed.sum() >= img.sum()


def test_ddim_scheduler_properties():
    s = L.DDIM()
    assert s.timesteps.tolist() == list(range(980, -1, -20))
    x = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    e = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(1))
    # invert_step then step with the same eps is the identity (DDIM is deterministic)
    y = s.invert_step(e, 500, x)
    assert torch.allclose(s.step(e, 500, y), x, atol=1e-5)


def test_laplacian_blend_matches_reference(golden):
    g9 = golden("g9_misc.npz")
    out = D.laplacian_blend(g9["poisson_img"].copy(), g9["laplacian_bg"], g9["poisson_mask"].astype(bool))
    assert np.allclose(out, g9["laplacian_out"], atol=1e-9)


def test_mesh_oracle_identity_transform_is_a_near_identity_map():
    """oracle/mesh_ref.py (parity unpinned, see its header): with no motion every foreground pixel must map to
    itself up to the half-pixel difference between the unprojection grid (edge-aligned linspace) and the
    rasteriser's pixel centres, the foreground flag must reproduce the mask interior, and depth must be kept."""
    import torch
    from oracle import mesh_ref as M
    from diffusionhandles_amd.synthetic import make_scene
    res = 48
    depth, bg_depth, mask = make_scene(res)
    f = 1.0 / np.tan(np.radians(27.5))
    gx = torch.linspace(-1, 1, res, dtype=torch.float32).numpy()
    lin01 = torch.linspace(0, 1, res, dtype=torch.float32).numpy()
    m = mask[0, 0].numpy() > 0.5
    xf = np.array([0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 2.5], dtype=np.float32)
    out = M.mesh_reproject(depth[0, 0].numpy(), bg_depth[0, 0].numpy(), m, gx, lin01, np.float32(1.0 / f), np.float32(f), xf)
    c = out["corr"]
    assert len(c) > 0.8 * m.sum()
    assert np.abs(c[:, 0] - c[:, 2]).max() <= 1 and np.abs(c[:, 1] - c[:, 3]).max() <= 1
    inner = m & np.roll(m, 1, 0) & np.roll(m, -1, 0) & np.roll(m, 1, 1) & np.roll(m, -1, 1)
    assert out["fg_flag"][inner].all()
    far = ~(m | np.roll(m, 2, 0) | np.roll(m, -2, 0) | np.roll(m, 2, 1) | np.roll(m, -2, 1))
    assert not out["fg_flag"][far].any()
    assert np.abs(out["zmap"][inner] - depth[0, 0].numpy()[inner]).max() < 0.05
