#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (imported from
/root/reference through sys.modules stubs) on seeded synthetic
inputs, and cross-check the oracle (oracle/*.py) against it on the spot.

Only runs in the build container (needs /root/reference).  The fixtures are data (inputs
are re-creatable from seeds; expected outputs are stored); no reference source is copied.

cv2 is absent here: the reference's transform_depth_pc is run with a `cv2` stand-in whose
getStructuringElement / morphologyEx / dilate are the oracle's restatement of OpenCV
(SURVEY Appendix C).  Everything upstream of the morphology is pinned by the reference
alone; the cleaned mask and what follows it are pinned only up to that restatement
("parity unpinned" at the cv2 boundary).
"""
import hashlib
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import depth_ref as D  # noqa: E402
from oracle import guidance_ref as G  # noqa: E402
from oracle import loop_ref as L  # noqa: E402
from oracle import unet_torch as U  # noqa: E402
from diffusionhandles_amd.synthetic import make_scene, TRANSFORMS  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    pkg = mod("diffhandles")
    pkg.__path__ = [os.path.join(REF, "diffhandles")]
    cv2 = mod("cv2", MORPH_ELLIPSE=2, MORPH_CLOSE=3, MORPH_OPEN=2)
    cv2.getStructuringElement = lambda shape, ksize: D.ellipse_kernel(ksize[0], ksize[1])

    def morphologyEx(img, op, kernel):
        return D.morph_close(img, kernel) if op == cv2.MORPH_CLOSE else D.morph_open(img, kernel)
    cv2.MORPH_CLOSE, cv2.MORPH_OPEN = 3, 2
    cv2.morphologyEx = morphologyEx
    cv2.dilate = lambda img, kernel: D.dilate(img, kernel)
    dummy = type("Dummy", (), {})
    mod("pytorch3d")
    mod("pytorch3d.renderer", FoVPerspectiveCameras=dummy, MeshRasterizer=dummy, RasterizationSettings=dummy,
        TexturesUV=dummy, look_at_view_transform=dummy)
    mod("pytorch3d.renderer.mesh")
    mod("pytorch3d.renderer.mesh.shader", ShaderBase=torch.nn.Module)
    mod("pytorch3d.renderer.blending", BlendParams=dummy, sigmoid_alpha_blend=dummy, _get_background_color=dummy)
    mod("pytorch3d.structures", Meshes=dummy)
    mod("pytorch3d.structures.meshes", join_meshes_as_scene=dummy)
    mod("diffusers", AutoencoderKL=dummy, UNet2DConditionModel=dummy, DDIMScheduler=dummy)

    class VIP:
        def __init__(self, vae_scale_factor=None):
            pass

        def postprocess(self, x, output_type="pt"):
            return (x / 2 + 0.5).clamp(0, 1)
    mod("diffusers.image_processor", VaeImageProcessor=VIP)
    mod("diffusers.configuration_utils", FrozenDict=dict)
    mod("diffusers.utils", deprecate=lambda *a, **k: None)
    mod("diffusers.utils.torch_utils",
        randn_tensor=lambda shape, generator=None, device=None, dtype=None: torch.randn(shape, generator=generator, dtype=dtype))
    # the vendored U-Net files are imported for real (diffusers leaf primitives: tools/diffusers_standins.py)
    import diffusers_standins
    diffusers_standins.import_reference_unet(REF)


class RefScheduler(L.DDIM):
    """Oracle DDIM restatement with the attribute surface the reference loops touch."""

    def __init__(self):
        super().__init__()
        self.config = SimpleNamespace(num_train_timesteps=1000)
        self.order = 1

    def set_timesteps(self, n, device=None):
        super().set_timesteps(n)

    def scale_model_input(self, x, t):
        return x

    def step(self, eps, t, x, eta=0.0, generator=None, return_dict=False):
        return (L.DDIM.step(self, eps, t, x),)


class FakeVAE:
    """Deterministic linear stand-in (8x avg-pool encode / nearest decode)."""
    config = SimpleNamespace(scaling_factor=L.VAE_SCALE, block_out_channels=(1, 1, 1, 1))

    def encode(self, x):
        z = torch.nn.functional.avg_pool2d(x, 8)
        z = torch.cat([z, z.mean(dim=1, keepdim=True)], dim=1)
        return {"latent_dist": SimpleNamespace(mean=z)}

    def decode(self, z, return_dict=True):
        img = torch.nn.functional.interpolate(z[:, :3], scale_factor=8.0, mode="nearest")
        return (img,) if return_dict is False else {"sample": img}


def text_embedding(prompt, dim, seed_base=1000):
    g = torch.Generator().manual_seed(seed_base + sum(prompt.encode()))
    return torch.randn(1, 77, dim, generator=g)


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    import diffhandles.depth_transform as RD
    import diffhandles.losses as RL
    import diffhandles.utils as RU
    import diffhandles.guided_stable_diffuser as RG
    import diffhandles.stable_null_inverter as RN

    K = RG.GuidedStableDiffuser.get_depth_intrinsics()
    assert torch.equal(K, D.intrinsics_f32())
    depth, bg_depth, mask = make_scene(512)
    res = 512

    # ---- G1 unproject ----------------------------------------------------------------
    ref_pts = RD.depth_to_world_coords(depth, K).numpy()
    ora_pts = D.unproject(depth[0, 0].numpy(), K)
    assert np.array_equal(ref_pts, ora_pts), "G1 unproject mismatch"
    ref_bg = RD.depth_to_world_coords(bg_depth, K).numpy()
    assert np.array_equal(ref_bg, D.unproject(bg_depth[0, 0].numpy(), K))
    g = dict(points_sha=sha(ref_pts), bg_points_sha=sha(ref_bg), points_slice=ref_pts[::37, ::41].copy())
    # a random-depth case too
    rd = (torch.rand(1, 1, 512, 512, generator=torch.Generator().manual_seed(3)) * 5 + 0.5)
    rp = RD.depth_to_world_coords(rd, K).numpy()
    assert np.array_equal(rp, D.unproject(rd[0, 0].numpy(), K))
    g["rand_points_sha"] = sha(rp)
    np.savez_compressed(os.path.join(OUT, "g1_unproject.npz"), **g)
    print("G1 ok")

    # ---- G2 / G3 / full pc edit ------------------------------------------------------
    m = mask[0, 0].numpy().astype(bool)
    g2, g3 = {}, {}
    for ti, (ang, tr) in enumerate(TRANSFORMS[:6]):
        axis = np.array([0, 1, 0], np.float32)
        ref_rot, ref_ids = RD.transform_point_cloud(ref_pts, axis, ang, tr[0], tr[1], tr[2], m)
        ora_rot = D.rigid_transform(ora_pts, axis, ang, tr, m)
        assert ref_rot.dtype == np.float64 and np.array_equal(ref_rot, ora_rot), f"G2 mismatch t{ti}"
        g2[f"t{ti}_sha"] = sha(ref_rot)
        g2[f"t{ti}_slice"] = ref_rot[::37, ::41].copy()
        allp = np.vstack([ref_bg.reshape(-1, 3), ref_rot.reshape(-1, 3)[m.reshape(-1)]])
        flags = np.zeros(allp.shape[0], np.uint8)
        flags[res * res:] = 1
        zr, mr, ur, vr, visr = RD.points_to_depth(torch.from_numpy(allp), K, (res, res), point_mask=torch.from_numpy(flags))
        zo, mo, uo, vo, viso = D.zbuffer(allp, flags, K, (res, res))
        assert np.array_equal(zr[0, 0].numpy(), zo) and np.array_equal(mr, mo), f"G3 maps mismatch t{ti}"
        assert np.array_equal(ur, uo) and np.array_equal(vr, vo) and np.array_equal(visr, viso), f"G3 idx mismatch t{ti}"
        g3[f"t{ti}_zmap_sha"] = sha(zo)
        g3[f"t{ti}_zmap_slice"] = zo[::37, ::41].copy()
        g3[f"t{ti}_mask"] = np.packbits(mo)
        g3[f"t{ti}_u"] = uo.astype(np.int16)
        g3[f"t{ti}_v"] = vo.astype(np.int16)
        g3[f"t{ti}_vis"] = np.packbits(viso[res * res:])
        # whole edit through the reference (cv2 = restated morphology)
        disp_r, corr_r = RD.transform_depth_pc(depth, bg_depth, mask, K, rot_angle=ang,
                                               rot_axis=torch.tensor([0.0, 1.0, 0.0]), translation=torch.tensor(tr))
        disp_o, corr_o, dbg = D.transform_depth_pc(depth, bg_depth, mask, K, rot_angle=ang,
                                                   rot_axis=[0.0, 1.0, 0.0], translation=tr, return_debug=True)
        assert torch.equal(corr_r, corr_o), f"corr mismatch t{ti}"
        assert torch.allclose(disp_r, disp_o, atol=1e-4, rtol=0), f"disp mismatch t{ti} {(disp_r-disp_o).abs().max()}"
        g3[f"t{ti}_corr"] = corr_r.numpy().astype(np.int16)
        g3[f"t{ti}_cleaned"] = np.packbits(dbg["cleaned"] != 0)
        g3[f"t{ti}_disp_slice"] = disp_r[0, 0].numpy()[::5, ::7].copy()
        g3[f"t{ti}_disp_sum"] = np.float64(disp_r.double().sum().item())
        print(f"  edit t{ti}: N_vis={int(viso.sum())} N_corr={corr_r.shape[0]} inpaint={int(dbg['inpaint'].sum())}")
    # small sequential cross-check of the collapsed z-buffer semantics incl. ties
    rng = np.random.default_rng(5)
    pts_s = np.concatenate([rng.uniform(-1, 1, (3000, 2)), rng.integers(2, 5, (3000, 1)).astype(np.float64)], axis=1)
    fl_s = np.zeros(3000, np.uint8); fl_s[2000:] = 1
    a = RD.points_to_depth(torch.from_numpy(pts_s), K, (32, 32), point_mask=torch.from_numpy(fl_s))
    b = D.zbuffer(pts_s, fl_s, K, (32, 32)); c = D.zbuffer_sequential(pts_s, fl_s, K, (32, 32))
    for x, y, z in zip((a[0][0, 0].numpy(),) + tuple(a[1:]), b, c):
        assert np.array_equal(x, y) and np.array_equal(y, z), "tie semantics"
    g3["ties_pts"] = pts_s; g3["ties_flags"] = fl_s
    g3["ties_zmap"] = b[0]; g3["ties_mask"] = b[1]; g3["ties_u"] = b[2]; g3["ties_v"] = b[3]; g3["ties_vis"] = b[4]
    np.savez_compressed(os.path.join(OUT, "g2_rigid.npz"), **g2)
    np.savez_compressed(os.path.join(OUT, "g3_zbuffer.npz"), **g3)
    print("G2/G3 ok")

    # ---- G4 process_correspondences ----------------------------------------------------
    g4 = {}
    corr = torch.from_numpy(g3["t2_corr"].astype(np.int64))
    for er in (0, 5, 10):
        r = RG.GuidedStableDiffuser.process_correspondences(None, corr, img_res=512, bg_erosion=er)
        o = G.cells_from_correspondences(corr.numpy(), 512, er)
        for k in r:
            assert np.array_equal(np.asarray(r[k]), o[k]), f"G4 {k} er={er}"
            g4[f"e{er}_{k}"] = np.asarray(r[k]).astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "g4_cells.npz"), **g4)
    print("G4 ok")

    # ---- G5 losses ---------------------------------------------------------------------
    g5 = {}
    gen = torch.Generator().manual_seed(11)
    shapes = [(16, 32, 32), (12, 64, 64), (8, 64, 64)]
    cells = G.cells_from_correspondences(corr.numpy(), 512, 0)
    cells5 = G.cells_from_correspondences(corr.numpy(), 512, 5)
    for li, shp in enumerate(shapes):
        cur = torch.randn(shp, generator=gen)
        org = torch.randn(shp, generator=gen)
        g5[f"l{li}_cur"] = cur.numpy(); g5[f"l{li}_org"] = org.numpy()
        for patch in (1, 3):
            for name, cl in (("e0", cells), ("e5", cells5)):
                for kind in ("fg", "bg_global_avg", "bg_local_avg"):
                    a = cur.clone().requires_grad_(True)
                    b = cur.clone().requires_grad_(True)
                    if kind == "fg":
                        lr = RL.compute_foreground_loss(a, org, cl, patch, (64, 64))
                        lo = G.foreground_energy(b, org, cl, patch, (64, 64))
                    else:
                        lt = kind[3:]
                        lr = RL.compute_background_loss(a, org, cl, patch, (64, 64), loss_type=lt)
                        lo = G.background_energy(b, org, cl, patch, (64, 64), lt)
                    gr, = torch.autograd.grad(lr, a)
                    go, = torch.autograd.grad(lo, b)
                    assert torch.equal(lr, lo) and torch.equal(gr, go), f"G5 {li} {patch} {name} {kind}"
                    key = f"l{li}_p{patch}_{name}_{kind}"
                    g5[key + "_loss"] = np.float32(lr.item())
                    g5[key + "_grad"] = gr.numpy()
    np.savez_compressed(os.path.join(OUT, "g5_energy.npz"), **g5)
    print("G5 ok")

    # ---- G6 schedule -------------------------------------------------------------------
    g6 = {}
    for sched in ("constant", "linear", "quadratic"):
        fgw, bgw = 1.5 * 30, 1.25 * 30
        ms = 38
        if sched == "constant":
            ff, fb = np.linspace(fgw, fgw, ms), np.linspace(bgw, bgw, ms)
        elif sched == "linear":
            ff, fb = np.linspace(fgw, 0.0, ms), np.linspace(bgw, 0.0, ms)
        else:
            ff, fb = np.linspace(np.sqrt(fgw), 0.0, ms) ** 2, np.linspace(np.sqrt(bgw), 0.0, ms) ** 2
        den = []
        for t in range(ms):
            pf, pb = G.LAYER_PATTERN[t % 3]
            den.append((t, (np.array(pf) * ff[t]).tolist(), (np.array(pb) * fb[t]).tolist()))
        den.append((ms, [0.0] * 3, [0.0] * 3))
        opt = [(0, [2.5] * 3, [1.25] * 3), (1, [1.25] * 3, [2.5] * 3), (2, [1.25] * 3, [1.25] * 3), (3, [2.5] * 3, [2.5] * 3)]
        ref_s = RG.StepGuidanceWeightSchedule(denoising_steps=den, optimization_steps=opt)
        tab = np.zeros((50, 4, 2, 3))
        for t in range(50):
            for it in range(4):
                rf, rb = ref_s(t, it)
                of, ob = G.guidance_weights(t, it, 1.5, 1.25, ms, sched)
                assert rf == of and rb == ob, f"G6 {sched} {t} {it}"
                tab[t, it, 0], tab[t, it, 1] = rf, rb
        g6[sched] = tab
    np.savez_compressed(os.path.join(OUT, "g6_schedule.npz"), **g6)
    print("G6 ok")

    # ---- G9 init_depth / normalize_depth; G10 poisson / laplacian ----------------------
    g9 = {}
    disp = RD.normalize_depth(1.0 / depth)
    assert torch.equal(disp, D.normalize_depth(1.0 / depth)[0])
    stub = object.__new__(RG.GuidedStableDiffuser)
    stub.unet = SimpleNamespace(sample_size=64, config=SimpleNamespace(out_channels=4))
    d64 = RG.GuidedStableDiffuser.init_depth(stub, disp)
    assert torch.equal(d64, L.init_depth(disp, (64, 64)))
    g9["disp_slice"] = disp[0, 0].numpy()[::5, ::7].copy(); g9["depth64"] = d64.numpy()
    rng = np.random.default_rng(9)
    img = rng.uniform(0, 255, (48, 48)); hole = np.zeros((48, 48), np.uint8)
    hole[10:20, 5:30] = 1; hole[0:3, 40:48] = 1; hole[30, 30] = 1
    pr = RD.poisson_solve(img.copy(), hole)
    po = D.harmonic_fill(img.copy(), hole)
    assert np.allclose(pr, po, atol=1e-9)
    g9["poisson_img"] = img; g9["poisson_mask"] = hole; g9["poisson_out"] = pr
    bgd = rng.uniform(1, 5, (48, 48))
    g9["laplacian_bg"] = bgd
    g9["laplacian_out"] = RU.solve_laplacian_depth(img.copy(), bgd, hole.astype(bool))
    np.savez_compressed(os.path.join(OUT, "g9_misc.npz"), **g9)
    print("G9/G10 ok")

    # ---- G7 / G8 loops at the TINY config.  (1) The reference's loops and the oracle's loops both drive
    # oracle/unet_torch.py: trajectories must be EQUAL, which pins the loop logic.  (2) G7b below re-runs the
    # reference's loops on the REFERENCE'S OWN U-Net class (model/unet_2d_condition.py on the diffusers leaf
    # stand-ins, tools/diffusers_standins.py; 2e-6 away from the oracle U-Net, g12) and bounds the difference ----
    torch.manual_seed(0)
    unet = U.init_synthetic_(U.UNetTorch(U.TINY), seed=0).eval()
    import diffusers_standins as DS
    ref_unet = DS.import_reference_unet()(**DS.sd2_depth_kwargs(U.TINY)).eval()
    ref_unet.load_state_dict(unet.state_dict(), strict=True)
    for p in list(unet.parameters()) + list(ref_unet.parameters()):
        p.requires_grad_(True)   # the reference keeps weights requiring grad
    cdim = U.TINY["cross_attention_dim"]
    prompt = "a sphere on a plane"
    conf = SimpleNamespace(bg_weight=1.25, fg_weight=1.5, fg_patch_size=1, bg_patch_size=1, use_depth=True,
                           save_denoising_steps=False, bg_loss_type="global_avg", num_timesteps=50,
                           num_optsteps=3, guidance_max_step=38, guidance_schedule_type="constant",
                           bg_erosion=0, seed=2773)
    gd = object.__new__(RG.GuidedStableDiffuser)
    gd.conf = conf
    gd.scheduler = RefScheduler()
    gd.unet = unet
    gd.device = torch.device("cpu")
    tok = lambda texts, **kw: SimpleNamespace(input_ids=SimpleNamespace(to=lambda dev, _t=texts: _t))
    tok_obj = type("Tok", (), {"model_max_length": 77, "__call__": lambda self, texts, **kw: tok(texts)})()
    gd.tokenizer = tok_obj
    gd.text_encoder = lambda ids: (text_embedding(ids[0], cdim),)
    gd.vae = FakeVAE()
    cond = text_embedding(prompt, cdim)
    unc0 = text_embedding("", cdim)
    g7 = {}
    sys.stdout.flush()
    import time
    t0 = time.time()
    # G8 inversion (reference) ------------------------------------------------------
    from diffusionhandles_amd.synthetic import make_image
    img = make_image(512)
    inv = RN.StableNullInverter(gd)
    (_, recon), init_noise, unc = inv.invert(img, disp, prompt, num_inner_steps=5)
    print(f"  ref inversion {time.time()-t0:.1f}s")
    lat0 = gd.vae.encode(img * 2 - 1)["latent_dist"].mean * L.VAE_SCALE
    o_lat, o_unc = L.null_text_inversion(unet, L.DDIM(), lat0, disp, unc0, cond, num_inner_steps=5)
    d1 = (o_lat[-1] - init_noise).abs().max().item(); d2 = (o_unc - unc).abs().max().item()
    print(f"  oracle-vs-ref inversion: noise {d1:.3e} uncond {d2:.3e}")
    assert d1 < 1e-4 and d2 < 1e-4
    g7["inv_init_noise"] = init_noise.numpy(); g7["inv_uncond_first"] = unc[:3].numpy(); g7["inv_uncond_last"] = unc[-2:].numpy()
    g7["inv_uncond_sum"] = unc.double().sum(dim=(1, 2, 3)).numpy()
    # G7 initial + guided (reference) ---------------------------------------------
    t0 = time.time()
    with torch.no_grad():
        acts, latent_img, unc_r, noise_r = gd.initial_inference(init_latents=init_noise, depth=disp, uncond_embeddings=unc, prompt=prompt)
    o_acts, o_latent, _, _ = L.initial_inference(unet, L.DDIM(), init_noise, disp, unc, cond)
    d3 = max((a - b).abs().max().item() for a, b in zip(acts, o_acts)); d4 = (latent_img - o_latent).abs().max().item()
    print(f"  initial_inference {time.time()-t0:.1f}s  oracle-vs-ref acts {d3:.3e} latent {d4:.3e}")
    assert d3 < 1e-3 and d4 < 1e-3
    g7["init_latent"] = latent_img.numpy()
    g7["init_acts_t0"] = np.concatenate([a[0].reshape(-1)[::97].numpy() for a in acts])
    g7["init_acts_t49"] = np.concatenate([a[49].reshape(-1)[::97].numpy() for a in acts])
    corr2 = torch.from_numpy(g3["t2_corr"].astype(np.int64))
    disp_e, _ = D.transform_depth_pc(depth, bg_depth, mask, K, rot_angle=TRANSFORMS[2][0], rot_axis=[0, 1, 0], translation=TRANSFORMS[2][1])
    t0 = time.time()
    edited = gd.guided_inference(latents=init_noise, depth=disp_e, uncond_embeddings=unc, prompt=prompt,
                                 activations_orig=acts, correspondences=corr2)
    torch.set_grad_enabled(True)
    print(f"  ref guided_inference {time.time()-t0:.1f}s")
    rec = {}
    o_final = L.guided_inference(unet, L.DDIM(), init_noise, disp_e, unc, cond, acts, corr2.numpy(), conf, record=rec)
    o_img = (gd.vae.decode(o_final / L.VAE_SCALE, return_dict=False)[0] / 2 + 0.5).clamp(0, 1)
    d5 = (o_img - edited).abs().max().item()
    print(f"  oracle-vs-ref guided image diff {d5:.3e}")
    assert d5 < 5e-3
    g7["guided_final_latent"] = o_final.numpy()
    g7["guided_image_slice"] = edited[0, :, ::16, ::16].numpy()
    g7["guided_steps"] = torch.stack(rec["step"])[::7].numpy()
    g7["guided_opt_first"] = torch.stack(rec["opt"][:6]).numpy()
    # ---- G7b: the same loops on the reference's own U-Net class ---------------------------------------
    gd.unet = ref_unet
    t0 = time.time()
    with torch.no_grad():
        acts_b, latent_b, _, _ = gd.initial_inference(init_latents=init_noise, depth=disp, uncond_embeddings=unc, prompt=prompt)
    d6 = max(((a - b).norm() / b.norm()).item() for a, b in zip(acts_b, acts)); d7 = ((latent_b - latent_img).norm() / latent_img.norm()).item()
    print(f"  G7b initial_inference on the reference U-Net class {time.time()-t0:.1f}s: acts rel {d6:.3e} latent rel {d7:.3e}")
    assert d6 < 1e-4 and d7 < 1e-4
    t0 = time.time()
    edited_b, steps_b = gd.guided_inference(latents=init_noise, depth=disp_e, uncond_embeddings=unc, prompt=prompt,
                                            activations_orig=acts, correspondences=corr2, save_denoising_steps=True)
    torch.set_grad_enabled(True)
    # save_denoising_steps: one list per timestep in 'opt' holding [image after the optimisation loop, image after the
    # DDIM step]; 'post-opt' stays empty (reference :385-386, 446-448, 476-478)
    assert len(steps_b["opt"]) == 50 and all(len(x) == 2 for x in steps_b["opt"]) and len(steps_b["post-opt"]) == 0
    dec = lambda z: (gd.vae.decode(z / L.VAE_SCALE, return_dict=False)[0] / 2 + 0.5).clamp(0, 1)
    # the energy is an L1 loss: its gradient is a sign, so two U-Nets 2e-6 apart follow the same trajectory only until
    # the first sign flips; the first steps are compared tightly, the end of the trajectory by its mean
    d8 = max((steps_b["opt"][t][0] - dec(rec["opt"][3 * t + 2])).abs().max().item() for t in range(2))
    d8s = max((steps_b["opt"][t][1] - dec(rec["step"][t])).abs().max().item() for t in range(2))
    d8f = (edited_b - edited).abs().mean().item()
    print(f"  G7b guided_inference on the reference U-Net class {time.time()-t0:.1f}s: first two steps image max diff "
          f"{d8:.3e} (after opt) {d8s:.3e} (after step); final image mean abs diff {d8f:.3e}")
    # measured 4.3e-3 / 6.4e-3 (a handful of sign flips move single latent elements by 0.1 x weight) and 7.2e-3
    assert d8 < 2e-2 and d8s < 2e-2 and d8f < 5e-2
    # inversion: the DDIM loop in full, the null-text optimisation for the first 3 timesteps (the Adam update is
    # +-lr per element at step 1 whatever the gradient's size, so two fp32 U-Nets 2e-6 apart give unconds that
    # differ by up to 2 lr on near-zero-gradient elements: measured and stored, bounded only by 2.5 lr)
    inv_b = RN.StableNullInverter(gd)
    ctx = torch.cat([unc0, cond])
    d64 = gd.init_depth(disp)
    with torch.no_grad():
        _, lat_b = inv_b.ddim_inversion(img, ctx, d64)
    d9 = (lat_b[-1] - init_noise).abs().max().item()
    inv_b.num_ddim_steps = 3
    unc_b = inv_b.null_optimization(lat_b, ctx, d64, 5, 1e-5)
    torch.set_grad_enabled(True)
    d10 = (unc_b - unc[:3]).abs().max().item(); d11 = (unc_b - unc[:3]).abs().mean().item()
    print(f"  G7b inversion on the reference U-Net class: noise max diff {d9:.3e}; uncond (3 timesteps) max {d10:.3e} mean {d11:.3e}")
    assert d9 < 1e-4 and d10 < 2.5e-2 and d11 < 2e-3
    g7["refclass_diffs"] = np.array([d6, d7, d8, d8s, d8f, d9, d10, d11])
    np.savez_compressed(os.path.join(OUT, "g7_loops.npz"), **g7)
    print("G7/G8 ok")


if __name__ == "__main__":
    main()
