"""GPU parity of the product loops (guided_inference / initial_inference / null-text
inversion on the native engine, fp16) against the oracle loops (torch fp32 + autograd) with the
same TINY U-Net weights, text embeddings and inputs.  The oracle loops are pinned bit-exact to
the reference's own loops on CPU (tests/test_loops_golden.py); here the tolerance is the fp16
engine tolerance, tight on the first steps and loose after 50 chaotic steps."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from diffusionhandles_amd.synthetic import TRANSFORMS, make_image, make_scene

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.fixture(scope="module")
def rig():
    from diffusionhandles_amd import conf as C
    from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
    from diffusionhandles_amd.stable_null_inverter import StableNullInverter
    from diffusionhandles_amd.unet import HipUNet
    from oracle import depth_ref as D
    from oracle import unet_torch as U
    ref = U.init_synthetic_(U.UNetTorch(U.TINY), seed=0).to(dev()).eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.half().float())
    hip = HipUNet(dict(U.TINY, text_len=77), dtype=torch.float16, max_batch=2)
    hip.load_state_dict(ref.state_dict())
    conf = C.load_default().guided_diffuser
    gd = GuidedStableDiffuser(conf, unet=hip, unet_config=dict(U.TINY, text_len=77)).to(dev())
    inv = StableNullInverter(gd)
    depth, bg, mask = make_scene(512)
    disp = D.normalize_depth(1.0 / depth)[0]
    prompt = "a sphere on a plane"
    cond = gd._encode([prompt])
    unc0 = gd._encode([""])
    return SimpleNamespace(ref=ref, hip=hip, gd=gd, inv=inv, depth=depth, bg=bg, mask=mask, disp=disp, prompt=prompt,
                           cond=cond, unc0=unc0, conf=conf)


def test_initial_inference_matches_oracle(rig):
    from oracle import loop_ref as L
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(1, 4, 64, 64, generator=g)
    unc = rig.unc0[None].expand(50, -1, -1, -1).contiguous()
    acts, lat, _, _ = rig.gd.initial_inference(noise.to(dev()), rig.disp.to(dev()), unc, rig.prompt)
    o_acts, o_lat, _, _ = L.initial_inference(rig.ref, L.DDIM(), noise.to(dev()), rig.disp.to(dev()), unc, rig.cond)
    assert acts[0].shape == (50, 128, 32, 32) and acts[2].shape == (50, 64, 64, 64)
    for k in range(3):
        assert rel(acts[k][0], o_acts[k][0]) < 1e-2, k
    e = rel(lat, o_lat)
    print("initial_inference final latent rel err", e)
    assert e < 5e-2
    rig.acts = [a.float() for a in o_acts]
    rig.noise = noise


def test_guided_inference_matches_oracle(rig):
    from oracle import depth_ref as D
    from oracle import loop_ref as L
    from diffusionhandles_amd.depth_transform import transform_depth
    if not hasattr(rig, "acts"):
        test_initial_inference_matches_oracle(rig)
    ang, tr = TRANSFORMS[2]
    K = rig.gd.get_depth_intrinsics()
    disp_e, corr = transform_depth(rig.depth.to(dev()), rig.bg.to(dev()), rig.mask.to(dev()), K, rot_angle=ang,
                                   rot_axis=torch.tensor([0.0, 1.0, 0.0]), translation=torch.tensor(tr))
    unc = rig.unc0[None].expand(50, -1, -1, -1).contiguous()
    rec_p, rec_o = {}, {}
    img = rig.gd.guided_inference(rig.noise.to(dev()), disp_e, unc, rig.prompt, rig.acts, corr, record=rec_p)
    o_final = L.guided_inference(rig.ref, L.DDIM(), rig.noise.to(dev()), disp_e.to(dev()), unc, rig.cond,
                                 [a.to(dev()) for a in rig.acts], corr.numpy(), rig.conf, record=rec_o)
    assert img.shape == (1, 3, 512, 512) and float(img.min()) >= 0 and float(img.max()) <= 1
    # the first latent update (pure guidance gradient): compare the update itself
    x0 = rig.noise.to(dev())
    up_p, up_o = rec_p["opt"][0] - x0, rec_o["opt"][0] - x0
    e0 = rel(up_p, up_o)
    print("first guidance update rel err", e0, "update norm", up_o.norm().item())
    assert e0 < 5e-2
    for i in range(3):
        assert rel(rec_p["step"][i], rec_o["step"][i]) < 2e-2, i
    e_fin = rel(rig.gd.last_latents, o_final)
    print("guided final latent rel err", e_fin)
    assert e_fin < 0.25
    assert len(rec_p["opt"]) == 38 * 3 and len(rec_p["step"]) == 50


def test_null_inversion_matches_oracle(rig):
    from oracle import loop_ref as L
    img = make_image(512).to(dev())
    (_, recon), init_noise, unc = rig.inv.invert(img, rig.disp.to(dev()), rig.prompt, num_inner_steps=5, max_timesteps=2)
    lat0 = rig.gd.vae.encode(img * 2 - 1)["latent_dist"].mean * 0.18215
    o_lat, o_unc = L.null_text_inversion(rig.ref, L.DDIM(), lat0, rig.disp.to(dev()), rig.unc0, rig.cond,
                                         num_inner_steps=5, null_steps=2)
    assert init_noise.shape == (1, 4, 64, 64) and unc.shape == (2, 1, 77, 64)
    e = rel(init_noise, o_lat[-1])
    print("ddim inversion rel err", e)
    assert e < 2e-2
    d_p, d_o = unc[0] - rig.unc0, o_unc[0] - rig.unc0
    e2 = rel(d_p, d_o)
    print("null-text update rel err", e2, "norm", d_o.norm().item())
    assert e2 < 0.2


def test_batched_edits_match_single_edits(rig):
    """BASELINE config 3 shape (K transforms of one image in one U-Net batch), K=2 with the TINY engine:
    the batched trajectory equals the two single-edit trajectories up to the engine's batch-dependent
    tile selection (fp16)."""
    from diffusionhandles_amd.depth_transform import reproject_edits
    from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
    from diffusionhandles_amd.unet import HipUNet
    from oracle import unet_torch as U
    if not hasattr(rig, "acts"):
        test_initial_inference_matches_oracle(rig)
    hip4 = HipUNet(dict(U.TINY, text_len=77), dtype=torch.float16, max_batch=4)
    hip4.load_state_dict(rig.ref.state_dict())
    gd4 = GuidedStableDiffuser(rig.conf, unet=hip4, unet_config=dict(U.TINY, text_len=77)).to(dev())
    K = rig.gd.get_depth_intrinsics()
    tfs = [(TRANSFORMS[i][0], torch.tensor([0.0, 1.0, 0.0]), torch.tensor(TRANSFORMS[i][1])) for i in (2, 4)]
    edits = reproject_edits(rig.depth.to(dev()), rig.bg.to(dev()), rig.mask.to(dev()), K, tfs)
    unc = rig.unc0[None].expand(50, -1, -1, -1).contiguous()
    imgs = gd4.guided_inference_batch(rig.noise.to(dev()), [d for d, _ in edits], unc, rig.prompt, rig.acts, [c for _, c in edits])
    assert imgs.shape == (2, 3, 512, 512)
    batched = gd4.last_latents.clone()
    for e, (d, c) in enumerate(edits):
        rig.gd.guided_inference(rig.noise.to(dev()), d, unc, rig.prompt, rig.acts, c)
        err = rel(batched[e:e + 1], rig.gd.last_latents)
        print("batched vs single edit", e, err)
        assert err < 5e-2


def test_guided_inference_is_bit_deterministic(rig):
    """No float atomics anywhere on the path (split-K slabs in slice order, integer sign sums, fixed-tree GroupNorm /
    attention merges): two runs of the whole 50-step guided loop give identical latents, bit for bit."""
    from diffusionhandles_amd.depth_transform import transform_depth
    if not hasattr(rig, "acts"):
        test_initial_inference_matches_oracle(rig)
    ang, tr = TRANSFORMS[3]
    K = rig.gd.get_depth_intrinsics()
    disp_e, corr = transform_depth(rig.depth.to(dev()), rig.bg.to(dev()), rig.mask.to(dev()), K, rot_angle=ang,
                                   rot_axis=torch.tensor([0.0, 1.0, 0.0]), translation=torch.tensor(tr))
    unc = rig.unc0[None].expand(50, -1, -1, -1).contiguous()
    outs = []
    for _ in range(2):
        rig.gd.guided_inference(rig.noise.to(dev()), disp_e, unc, rig.prompt, rig.acts, corr)
        outs.append(rig.gd.last_latents.clone())
    assert torch.equal(outs[0], outs[1])


def test_scene_harness_end_to_end_on_the_reference_scene(tmp_path):
    """tools/run_edit.py on the scene directory of the reference's test data (PNG + PIZ OpenEXR inputs), full-size
    U-Net with seeded random weights: the counterpart of the reference's test_diffusion_handles.py writes its
    outputs, and the disparity it writes is the re-projection the golden vectors pin."""
    import json
    import os
    import subprocess
    import sys
    from diffusionhandles_amd import scene_io as S
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "edit")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_edit.py"), "--scene",
                        os.path.join(root, "tests", "golden", "scene_banana_fruits"), "--out", out, "--skip-inversion",
                        "--no-identity-cache", "--max-edits", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert [e["name"] for e in rep["edits"]] == ["edit_000", "edit_001"] and rep["resolution"] == 512
    for f in ("recon.png", "edit_000.png", "edit_000_disparity.png", "edit_001.png", "edit_001_disparity.png", "report.json"):
        assert os.path.exists(os.path.join(out, f)), f
    img = S.read_png(os.path.join(out, "edit_001.png"))
    disp = S.read_png(os.path.join(out, "edit_001_disparity.png"))
    assert img.shape == (512, 512, 3) and disp.shape == (512, 512) and disp.max() == 255 and disp.min() < 64
    assert not os.path.exists(os.path.join(out, "identity.npz"))
