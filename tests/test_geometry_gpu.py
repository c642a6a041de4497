"""GPU parity (through the C ABI): depth re-projection, cell lists and guidance energy
against the oracle and the committed golden vectors.  Integer outputs bit-exact."""
import numpy as np
import pytest
import torch

from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    return torch.device("cuda:0")


def _edits(res, idx):
    from oracle import depth_ref as D
    from diffusionhandles_amd import depth_transform as DT
    depth, bg, mask = make_scene(res)
    K = D.intrinsics_f32()
    dev = _dev()
    tf = [(TRANSFORMS[i][0], torch.tensor([0.0, 1.0, 0.0]), torch.tensor(TRANSFORMS[i][1])) for i in idx]
    out, dbg = DT.reproject_edits(depth.to(dev), bg.to(dev), mask.to(dev), K, tf, return_debug=True)
    return depth, bg, mask, K, out, dbg


def test_unproject_bit_exact(golden):
    from oracle import depth_ref as D
    from diffusionhandles_amd import depth_transform as DT
    depth, bg, _ = make_scene(512)
    pts = DT.depth_to_world_coords(depth.to(_dev()), D.intrinsics_f32()).cpu().numpy()
    assert np.array_equal(pts, D.unproject(depth[0, 0].numpy()))
    assert np.array_equal(pts[::37, ::41], golden("g1_unproject.npz")["points_slice"])


def test_batched_edits_bit_exact_vs_golden_and_oracle(golden):
    from oracle import depth_ref as D
    g = golden("g3_zbuffer.npz")
    idx = [0, 1, 2, 3, 4, 5]
    depth, bg, mask, K, out, dbg = _edits(512, idx)
    for e, ti in enumerate(idx):
        disp, corr = out[e]
        assert corr.dtype == torch.int64 and corr.device.type == "cpu"
        assert np.array_equal(corr.numpy(), g[f"t{ti}_corr"].astype(np.int64)), f"corr t{ti}"
        assert np.array_equal(np.packbits(dbg["raw_mask"][e].cpu().numpy() != 0), g[f"t{ti}_mask"]), f"mask t{ti}"
        assert np.array_equal(np.packbits(dbg["clean_mask"][e].cpu().numpy() != 0), g[f"t{ti}_cleaned"])
        assert np.array_equal(np.packbits(dbg["vis"][e].cpu().numpy() != 0), g[f"t{ti}_vis"]), f"vis t{ti}"
        assert np.array_equal(dbg["zmap"][e].cpu().numpy()[::37, ::41], g[f"t{ti}_zmap_slice"]), f"zmap t{ti}"
        vis = dbg["vis"][e].cpu().numpy() != 0
        txy = dbg["target_xy"][e].cpu().numpy()
        assert np.array_equal(txy[vis, 0], g[f"t{ti}_u"]) and np.array_equal(txy[vis, 1], g[f"t{ti}_v"])
        d = disp[0, 0].cpu().numpy()
        assert np.allclose(d[::5, ::7], g[f"t{ti}_disp_slice"], atol=2e-3, rtol=0), f"disp t{ti}"
        assert abs(float(d.astype(np.float64).sum()) - float(g[f"t{ti}_disp_sum"])) < 1.0
    # full-array check of one edit against the oracle run here
    disp_o, corr_o, dbg_o = D.transform_depth_pc(depth, bg, mask, K, rot_angle=TRANSFORMS[3][0], rot_axis=[0, 1, 0],
                                                 translation=TRANSFORMS[3][1], return_debug=True)
    assert np.array_equal(dbg["zmap"][3].cpu().numpy(), dbg_o["zmap"])
    assert np.array_equal(out[3][1].numpy(), corr_o.numpy())
    assert np.abs(out[3][0].cpu().numpy() - disp_o.numpy()).max() < 2e-3


def test_general_axis_and_small_res_vs_oracle():
    """res 256 scene, non axis-aligned rotation: <= a handful of differing map entries allowed
    (np.dot goes through BLAS on the oracle side)."""
    from oracle import depth_ref as D
    from diffusionhandles_amd import depth_transform as DT
    depth, bg, mask = make_scene(256)
    K = D.intrinsics_f32()
    dev = _dev()
    axis = torch.tensor([0.3, 0.9, -0.2])
    (disp, corr), = DT.reproject_edits(depth.to(dev), bg.to(dev), mask.to(dev), K, [(25.0, axis, torch.tensor([0.1, -0.05, 0.2]))])
    disp_o, corr_o = D.transform_depth_pc(depth, bg, mask, K, rot_angle=25.0, rot_axis=axis.numpy(), translation=[0.1, -0.05, 0.2])
    a = set(map(tuple, corr.numpy().tolist())); b = set(map(tuple, corr_o.numpy().tolist()))
    assert len(a ^ b) <= 4
    # ... and EXACT against the oracle with the dot product's order written out (the only machine-dependent operation of the path)
    D.DOT_ORDER = "explicit"
    try:
        disp_x, corr_x = D.transform_depth_pc(depth, bg, mask, K, rot_angle=25.0, rot_axis=axis.numpy(), translation=[0.1, -0.05, 0.2])
    finally:
        D.DOT_ORDER = "blas"
    assert np.array_equal(corr.numpy(), corr_x.numpy()), len(a ^ set(map(tuple, corr_x.numpy().tolist())))
    assert np.abs(disp.cpu().numpy() - disp_x.numpy()).max() < 2e-3
    # axis-aligned at the same size must be exact
    (disp, corr), = DT.reproject_edits(depth.to(dev), bg.to(dev), mask.to(dev), K, [(40.0, torch.tensor([0.0, 1.0, 0.0]), torch.tensor([0.2, 0.0, 0.1]))])
    disp_o, corr_o = D.transform_depth_pc(depth, bg, mask, K, rot_angle=40.0, rot_axis=[0, 1, 0], translation=[0.2, 0.0, 0.1])
    assert np.array_equal(corr.numpy(), corr_o.numpy())
    assert np.abs(disp.cpu().numpy() - disp_o.numpy()).max() < 2e-3


def test_edge_cases_empty_mask_and_input_normalisation():
    from oracle import depth_ref as D
    from diffusionhandles_amd import depth_transform as DT
    depth, bg, mask = make_scene(256)
    K = D.intrinsics_f32()
    dev = _dev()
    disp, corr = DT.transform_depth(depth.to(dev), bg.to(dev), torch.zeros_like(mask).to(dev), K)
    assert corr.shape == (0, 4) and corr.dtype == torch.int64
    assert torch.equal(disp.cpu(), D.normalize_depth(1.0 / depth)[0])
    disp, corr = DT.transform_depth(depth.to(dev), bg.to(dev), mask.to(dev), K, rot_angle=10.0, use_input_depth_normalization=True)
    disp_o, corr_o = D.transform_depth_pc(depth, bg, mask, K, rot_angle=10.0, use_input_depth_normalization=True)
    assert np.array_equal(corr.numpy(), corr_o.numpy())
    assert np.abs(disp.cpu().numpy() - disp_o.numpy()).max() < 2e-3
    with pytest.raises(ValueError):
        DT.transform_depth(depth.to(dev), bg.to(dev), mask.to(dev), K, depth_transform_mode="nope")


@pytest.mark.parametrize("res", [512, 768])
def test_full_size_properties_identity_round_trip_and_monotone_translation(res):
    """Size-independent properties at the BASELINE resolutions (the oracle's Python z-buffer takes ~1 s per edit
    there, so full sizes are checked through invariants): (1) the identity transform maps every foreground pixel to
    itself, leaves no hole, and returns the normalised input disparity; (2) a pure translation towards the camera
    keeps every correspondence's source inside the mask, targets unique, and never shrinks the silhouette;
    (3) the K-batched call equals K single calls bit for bit."""
    from diffusionhandles_amd import depth_transform as DT
    from oracle import depth_ref as D
    depth, bg, mask = make_scene(res)
    K = D.intrinsics_f32()
    dev = _dev()
    d, b, m = depth.to(dev), bg.to(dev), mask.to(dev)
    axis = torch.tensor([0.0, 1.0, 0.0])
    zero = torch.tensor([0.0, 0.0, 0.0])
    (disp, corr), = DT.reproject_edits(d, b, m, K, [(0.0, axis, zero)])
    c = corr.numpy()
    # (the reference's OPEN(2x2 ellipse) clean-up of the re-projected mask trims ~1 % of silhouette pixels even here)
    assert int(0.98 * mask.sum()) <= len(c) <= int(mask.sum())
    assert np.array_equal(c[:, 0], c[:, 2]) and np.array_equal(c[:, 1], c[:, 3])
    key = c[:, 1] * res + c[:, 0]
    assert (np.diff(key) > 0).all() and (mask[0, 0].numpy()[c[:, 1], c[:, 0]] > 0.5).all()   # row-major source order
    diff = (disp.cpu() - D.normalize_depth(1.0 / depth)[0]).abs().flatten()
    assert float((diff <= 2e-3).float().mean()) > 0.995       # all but the trimmed / in-filled silhouette pixels
    tfs = [(0.0, axis, torch.tensor([0.0, 0.0, -0.4])), (20.0, axis, torch.tensor([0.2, 0.0, 0.0])),
           (-35.0, axis, torch.tensor([-0.3, 0.05, 0.3]))]
    batched = DT.reproject_edits(d, b, m, K, tfs)
    mk = mask[0, 0].numpy() > 0.5
    for i, tf in enumerate(tfs):
        (ds, cs), = DT.reproject_edits(d, b, m, K, [tf])
        assert torch.equal(cs, batched[i][1]) and torch.equal(ds, batched[i][0])
        c = cs.numpy()
        assert mk[c[:, 1], c[:, 0]].all()
        assert len({(int(x), int(y)) for x, y in c[:, 2:4]}) == len(c)                  # one source per target pixel
        assert (c[:, 2:4] >= 0).all() and (c[:, 2:4] < res).all()
        assert torch.isfinite(ds).all() and float(ds.min()) >= -1e-3 and float(ds.max()) <= 255.001
    assert len(batched[0][1]) >= int(0.95 * mask.sum())        # moving closer never shrinks the visible silhouette much


def test_zbuffer_determinism():
    a = _edits(512, [3])[3 + 1]
    b = _edits(512, [3])[3 + 1]
    assert torch.equal(a[0][1], b[0][1]) and torch.equal(a[0][0], b[0][0])


def test_cells_bit_exact(golden):
    from diffusionhandles_amd import losses as LS
    from oracle import guidance_ref as G
    g3, g4 = golden("g3_zbuffer.npz"), golden("g4_cells.npz")
    corr = torch.from_numpy(g3["t2_corr"].astype(np.int64))
    for er in (0, 5, 10):
        pc = LS.process_correspondences(corr, 512, er)
        for k in pc:
            assert np.array_equal(np.asarray(pc[k]), g4[f"e{er}_{k}"].astype(np.int64)), (er, k)
    bad = torch.tensor([[1, 1, -1, 5], [2, 2, 600, 5], [8, 8, 16, 24]], dtype=torch.int64)
    pc = LS.process_correspondences(bad, 512, 0)
    ref = G.cells_from_correspondences(bad.numpy(), 512, 0)
    for k in ref:
        assert np.array_equal(np.asarray(pc[k]), ref[k]), k
    pc = LS.process_correspondences(torch.zeros((0, 4), dtype=torch.int64), 512, 3)
    ref = G.cells_from_correspondences(np.zeros((0, 4), np.int64), 512, 3)
    for k in ref:
        assert np.array_equal(np.asarray(pc[k]), ref[k]), k


def test_energy_vs_golden(golden):
    """fp32 inputs: loss rel 1e-5, gradient abs 1e-6 (f32 summation order differs)."""
    from diffusionhandles_amd import losses as LS
    g3, g5 = golden("g3_zbuffer.npz"), golden("g5_energy.npz")
    corr = torch.from_numpy(g3["t2_corr"].astype(np.int64))
    dev = _dev()
    cells = {"e0": LS.process_correspondences(corr, 512, 0), "e5": LS.process_correspondences(corr, 512, 5)}
    for li in range(3):
        cur = torch.from_numpy(g5[f"l{li}_cur"]).to(dev)
        org = torch.from_numpy(g5[f"l{li}_org"]).to(dev)
        for patch in (1, 3):
            for name in ("e0", "e5"):
                for kind in ("fg", "bg_global_avg", "bg_local_avg"):
                    key = f"l{li}_p{patch}_{name}_{kind}"
                    if kind == "fg":
                        loss, grad = LS.energy_and_grad(cur, org, cells[name], 1.0, 0.0, patch, 1, (64, 64), channels_last=False)
                        val = loss[1].item()
                    else:
                        loss, grad = LS.energy_and_grad(cur, org, cells[name], 0.0, 1.0, 1, patch, (64, 64),
                                                        bg_loss_type=kind[3:], channels_last=False)
                        val = loss[2].item()
                    ref_l, ref_g = float(g5[key + "_loss"]), g5[key + "_grad"]
                    assert abs(val - ref_l) <= 2e-5 * max(1.0, abs(ref_l)), (key, val, ref_l)
                    err = np.abs(grad.cpu().numpy() - ref_g).max()
                    assert err <= 2e-6 + 1e-4 * np.abs(ref_g).max(), (key, err, np.abs(ref_g).max())


def test_energy_fp16_and_scaling():
    from diffusionhandles_amd import losses as LS
    from oracle import guidance_ref as G
    dev = _dev()
    gen = torch.Generator().manual_seed(4)
    corr = torch.stack([torch.randint(0, 512, (5000,), generator=gen) for _ in range(4)], dim=-1)
    pc = LS.process_correspondences(corr, 512, 0)
    cur = torch.randn(64, 64, 320, generator=gen).half()
    org = torch.randn(64, 64, 320, generator=gen).half()
    loss, grad = LS.energy_and_grad(cur.to(dev), org.to(dev), pc, 3.0, 2.0, grad_scale=256.0)
    a = cur.float().permute(2, 0, 1).clone().requires_grad_(True)
    cells = G.cells_from_correspondences(corr.numpy(), 512, 0)
    ref = 3.0 * G.foreground_energy(a, org.float().permute(2, 0, 1), cells, 1, (64, 64)) + \
        2.0 * G.background_energy(a, org.float().permute(2, 0, 1), cells, 1, (64, 64))
    gr, = torch.autograd.grad(ref, a)
    assert abs(loss[0].item() - ref.item()) < 1e-4 * abs(ref.item())
    got = grad.float().cpu().permute(2, 0, 1) / 256.0
    assert (got - gr).abs().max() <= 2e-3 * gr.abs().max() + 1e-7
    # determinism of the gradient (integer sign sums, single writer per element)
    loss2, grad2 = LS.energy_and_grad(cur.to(dev), org.to(dev), pc, 3.0, 2.0, grad_scale=256.0)
    assert torch.equal(grad, grad2)


def test_energy_planned_path_is_bit_identical_to_general_path():
    """dh_energy_fwd_bwd_planned (CSR built once per edit, 16-bit reads, 3 kernels) vs dh_energy_fwd_bwd:
    identical gradient bits, loss within f32 rounding; also empty-pair and empty-background edge cases."""
    from diffusionhandles_amd import losses as LS
    dev = _dev()
    gen = torch.Generator().manual_seed(9)
    for dtype in (torch.float16, torch.bfloat16):
        for C in (320, 640, 1280):
            for n in (7000, 0):
                corr = torch.stack([torch.randint(40, 470, (n,), generator=gen) for _ in range(4)], dim=-1)
                pc = LS.process_correspondences(corr, 512, 0)
                cur = torch.randn(64, 64, C, generator=gen).to(dtype).to(dev)
                org = torch.randn(64, 64, C, generator=gen).to(dtype).to(dev)
                plan = LS.EnergyPlan(pc, 64, dev)
                for fw, bw in ((3.0, 2.0), (0.0, 1.5), (2.5, 0.0)):
                    if n == 0 and fw != 0.0:
                        continue
                    l0, g0 = LS.energy_and_grad(cur, org, pc, fw, bw, grad_scale=256.0)
                    l1, g1 = LS.energy_and_grad_planned(cur, org, plan, fw, bw, grad_scale=256.0, want_loss=True)
                    assert torch.equal(g0, g1), (dtype, C, n, fw, bw)
                    assert torch.allclose(l0, l1, rtol=1e-6, atol=1e-7), (l0, l1)
                    _, g2 = LS.energy_and_grad_planned(cur, org, plan, fw, bw, grad_scale=256.0)
                    assert torch.equal(g1, g2)


def test_set_foreground_laplacian_blend_vs_oracle():
    """set_foreground: f64 CG on the GPU vs the oracle's sparse direct solve; depth values O(1..6) -> 1e-4."""
    from oracle import depth_ref as D
    from diffusionhandles_amd import depth_transform as DT
    depth, bg, mask = make_scene(256)
    bg = bg + 0.05 * torch.sin(torch.arange(256, dtype=torch.float32) / 9.0)[None, None, None, :]
    out = DT.laplacian_depth_blend(depth.to(_dev()), bg.to(_dev()), mask.to(_dev()))
    ref = D.set_foreground(depth, mask, bg)
    assert out.shape == (1, 1, 256, 256) and out.dtype == torch.float32
    assert float((out.cpu() - ref).abs().max()) < 1e-4
    # outside the dilated mask the depth is untouched
    import scipy.ndimage
    m = scipy.ndimage.binary_dilation(mask[0, 0].numpy(), iterations=15)
    assert torch.equal(out.cpu()[0, 0][~torch.from_numpy(m)], depth[0, 0][~torch.from_numpy(m)])


@pytest.mark.parametrize("side,path", [(60, "on-chip (<= 8192 unknowns)"), (150, "16 workgroups (<= 65536)"),
                                       (290, "single workgroup in global memory (> 65536)")])
def test_laplacian_blend_every_cg_kernel_vs_oracle(side, path):
    """The three CG kernels behind the harmonic in-fill / set_foreground (the on-chip cg_fill_lds, which runs inside k_cg_fill_multi; the sixteen-workgroup pipelined form with its bounded
    grid barrier, k_cg_fill: the fallback nothing else in the suite reaches, advisor round 3) on square holes sized for each,
    res 512, against the oracle's sparse direct solve of the same system (utils.solve_laplacian_depth, pinned by g9)."""
    from oracle import depth_ref as D
    from diffusionhandles_amd import depth_transform as DT
    res = 512
    depth, bg, _ = make_scene(res)
    bg = bg + 0.05 * torch.sin(torch.arange(res, dtype=torch.float32) / 9.0)[None, None, None, :]
    mask = torch.zeros(1, 1, res, res, dtype=torch.bool)
    a = (res - side) // 2
    mask[0, 0, a:a + side, a:a + side] = True
    import scipy.ndimage
    n_unknown = int(scipy.ndimage.binary_dilation(mask[0, 0].numpy(), iterations=15).sum())
    lo, hi = {60: (0, 8192), 150: (8193, 65536), 290: (65537, res * res)}[side]
    assert lo <= n_unknown <= hi, (n_unknown, path)
    out = DT.laplacian_depth_blend(depth.to(_dev()), bg.to(_dev()), mask.to(_dev()))
    ref = D.set_foreground(depth, mask, bg)
    err = float((out.cpu() - ref).abs().max())
    print(f"laplacian blend, {n_unknown} unknowns, {path}: max abs err {err:.2e}")
    assert err < 2e-4, (path, err)


def test_real_scene_edits_bit_exact_vs_reference_golden(golden):
    """A scene of the reference's own test data (estimated depth, PIZ OpenEXR; tests/golden/scene_banana_fruits) with
    its three transforms (identity, 91 degrees + shift, pure translation): integer maps bit-exact against what the
    reference produced (tools/make_golden_scene.py)."""
    import os
    from oracle import depth_ref as D
    from diffusionhandles_amd import depth_transform as DT
    from diffusionhandles_amd import scene_io as S
    g = golden("g11_scene.npz")
    sc = S.load_scene(os.path.join(os.path.dirname(__file__), "golden", "scene_banana_fruits"), 512)
    dev = _dev()
    names = list(sc["transforms"].keys())
    tf = []
    for n in names:
        kw = S.transform_args(sc["transforms"][n])
        tf.append((kw["rot_angle"], kw["rot_axis"], kw["translation"]))
    out, dbg = DT.reproject_edits(sc["depth"].to(dev), sc["bg_depth"].to(dev), sc["fg_mask"].to(dev), D.intrinsics_f32(), tf,
                                  return_debug=True)
    for e, n in enumerate(names):
        disp, corr = out[e]
        assert np.array_equal(corr.numpy(), g[f"{n}_corr"].astype(np.int64)), n
        assert np.array_equal(np.packbits(dbg["raw_mask"][e].cpu().numpy() != 0), g[f"{n}_mask"]), n
        assert np.array_equal(np.packbits(dbg["clean_mask"][e].cpu().numpy() != 0), g[f"{n}_cleaned"]), n
        assert np.array_equal(np.packbits(dbg["vis"][e].cpu().numpy() != 0), g[f"{n}_vis"]), n
        assert np.array_equal(dbg["zmap"][e].cpu().numpy()[::37, ::41], g[f"{n}_zmap_slice"]), n
        d = disp[0, 0].cpu().numpy()
        assert np.allclose(d[::5, ::7], g[f"{n}_disp_slice"], atol=2e-3, rtol=0), n
        assert abs(float(d.astype(np.float64).sum()) - float(g[f"{n}_disp_sum"])) < 1.0 + 1e-6 * abs(float(g[f"{n}_disp_sum"])), n
    # and through the public entry point, one edit at a time
    kw = S.transform_args(sc["transforms"]["edit_002"])
    disp1, corr1 = DT.transform_depth(sc["depth"].to(dev), sc["bg_depth"].to(dev), sc["fg_mask"].to(dev), D.intrinsics_f32(),
                                      rot_angle=kw["rot_angle"], rot_axis=kw["rot_axis"], translation=kw["translation"])
    assert np.array_equal(corr1.numpy(), g["edit_002_corr"].astype(np.int64))


def test_second_real_scene_edits_bit_exact_vs_reference_golden(golden):
    """The second scene of the reference's test data (tests/golden/scene_dice, g15): both transforms (a translation and the
    identity) in one batched call, integer maps bit-exact against what the reference produced."""
    import os
    from oracle import depth_ref as D
    from diffusionhandles_amd import depth_transform as DT
    from diffusionhandles_amd import scene_io as S
    g = golden("g15_scene_dice.npz")
    sc = S.load_scene(os.path.join(os.path.dirname(__file__), "golden", "scene_dice"), 512)
    dev = _dev()
    names = list(sc["transforms"].keys())
    tf = []
    for n in names:
        kw = S.transform_args(sc["transforms"][n])
        tf.append((kw["rot_angle"], kw["rot_axis"], kw["translation"]))
    out, dbg = DT.reproject_edits(sc["depth"].to(dev), sc["bg_depth"].to(dev), sc["fg_mask"].to(dev), D.intrinsics_f32(), tf,
                                  return_debug=True)
    for e, n in enumerate(names):
        disp, corr = out[e]
        assert np.array_equal(corr.numpy(), g[f"{n}_corr"].astype(np.int64)), n
        assert np.array_equal(np.packbits(dbg["raw_mask"][e].cpu().numpy() != 0), g[f"{n}_mask"]), n
        assert np.array_equal(np.packbits(dbg["clean_mask"][e].cpu().numpy() != 0), g[f"{n}_cleaned"]), n
        assert np.array_equal(np.packbits(dbg["vis"][e].cpu().numpy() != 0), g[f"{n}_vis"]), n
        assert np.array_equal(dbg["zmap"][e].cpu().numpy()[::37, ::41], g[f"{n}_zmap_slice"]), n
        d = disp[0, 0].cpu().numpy()
        assert np.allclose(d[::5, ::7], g[f"{n}_disp_slice"], atol=2e-3, rtol=0), n


def test_set_foreground_on_the_real_scene_vs_oracle():
    """The reference's scene at full size: two independently estimated depth maps (image / in-painted background) blended
    around the object; GPU CG vs the oracle's sparse direct solve."""
    import os
    from oracle import depth_ref as D
    from diffusionhandles_amd import depth_transform as DT
    from diffusionhandles_amd import scene_io as S
    sc = S.load_scene(os.path.join(os.path.dirname(__file__), "golden", "scene_banana_fruits"), 512)
    out = DT.laplacian_depth_blend(sc["depth"].to(_dev()), sc["bg_depth"].to(_dev()), sc["fg_mask"].to(_dev()))
    ref = D.set_foreground(sc["depth"], sc["fg_mask"], sc["bg_depth"])
    assert out.shape == (1, 1, 512, 512)
    assert float((out.cpu() - ref).abs().max()) < 1e-4
    assert float((out.cpu() - sc["bg_depth"]).abs().max()) > 1e-3          # the blend did change the background depth
