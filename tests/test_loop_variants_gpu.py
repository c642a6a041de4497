"""GPU: the product's guided loop over the reference's option space (/root/reference/test/config/*.yaml, carried here as the
configuration dictionaries stored in tests/golden/g14_loop_variants.npz) against the oracle loop, TINY U-Net, fp16 engine.

Every variant runs INSIDE GuidedStableDiffuser.guided_step: the general (non-planned) energy path for 'local_avg', the eroded
background masks, the falling weight schedules over 50 guided steps, and use_depth = false (a 4-channel engine, no depth
concatenation).  The product's steps are teacher-forced (started from the oracle's latent), and the oracle itself is tied to
the reference-generated fixture on the way."""
import json
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene

pytestmark = pytest.mark.gpu

VARIANTS = ["bg_erosion_10_local_avg", "local_avg_bg_loss", "linear_schedule", "quadratic_schedule", "no_depth", "bg_erosion",
            "quadratic_schedule_local_avg", "bg_erosion_15_local_avg", "bg_erosion_local_avg", "full_debug"]
NT = 4


def text_embedding(prompt, dim, seed_base=1000):
    """The seeded prompt embeddings the fixture generator used (tools/make_golden.py::text_embedding)."""
    g = torch.Generator().manual_seed(seed_base + sum(prompt.encode()))
    return torch.randn(1, 77, dim, generator=g)


class FixedTokenizer:
    model_max_length = 77

    def __call__(self, texts, **kw):
        return SimpleNamespace(input_ids=SimpleNamespace(to=lambda dev, _t=texts: _t))


class FixedTextEncoder:
    def __init__(self, dim):
        self.dim, self.dev = dim, "cpu"

    def to(self, dev):
        self.dev = dev
        return self

    def __call__(self, texts):
        return (torch.cat([text_embedding(t, self.dim) for t in texts]).to(self.dev),)


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.fixture(scope="module")
def scene():
    from diffusionhandles_amd.depth_transform import transform_depth
    from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
    from oracle import depth_ref as D
    dev = torch.device("cuda:0")
    depth, bg, mask = make_scene(512)
    disp = D.normalize_depth(1.0 / depth)[0]
    ang, tr = TRANSFORMS[2]
    disp_e, corr = transform_depth(depth.to(dev), bg.to(dev), mask.to(dev), GuidedStableDiffuser.get_depth_intrinsics(),
                                   rot_angle=ang, rot_axis=torch.tensor([0.0, 1.0, 0.0]), translation=torch.tensor(tr))
    return SimpleNamespace(dev=dev, disp=disp.to(dev), disp_e=disp_e, corr=corr)


@pytest.mark.parametrize("name", VARIANTS)
def test_guided_steps_under_variant_config(name, scene, golden):
    from diffusionhandles_amd import conf as C
    from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
    from diffusionhandles_amd.unet import HipUNet
    from oracle import loop_ref as L
    from oracle import unet_torch as U
    dev = scene.dev
    g14 = golden("g14_loop_variants.npz")
    raw = json.loads(str(g14[name + ".conf"]))
    conf = C.load_default().guided_diffuser
    for k, v in raw.items():
        setattr(conf, k, v)
    cfg = dict(U.TINY) if conf.use_depth else dict(U.TINY, in_channels=4)
    ref = U.init_synthetic_(U.UNetTorch(cfg), seed=0).to(dev).eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.half().float())
    hip = HipUNet(dict(cfg, text_len=77), dtype=torch.float16, max_batch=2)
    hip.load_state_dict(ref.state_dict())
    gd = GuidedStableDiffuser(conf, unet=hip, unet_config=dict(cfg, text_len=77), tokenizer=FixedTokenizer(),
                              text_encoder=FixedTextEncoder(cfg["cross_attention_dim"])).to(dev)
    prompt = "a sphere on a plane"
    cond = gd._encode([prompt])
    unc = gd._encode([""])[None].expand(50, -1, -1, -1).contiguous()
    noise = torch.from_numpy(g14["noise"]).to(dev)

    class Short(L.DDIM):
        def set_timesteps(self, n):
            super().set_timesteps(n)
            self.timesteps = self.timesteps[:NT]
    o_acts, _, _, _ = L.initial_inference(ref, Short(), noise, scene.disp, unc, cond, use_depth=conf.use_depth)
    rec_o = {}
    L.guided_inference(ref, Short(), noise, scene.disp_e.to(dev), unc, cond, o_acts, scene.corr.numpy(), conf, record=rec_o)
    iters = conf.num_optsteps
    # the product's initial inference under this configuration (4 channels without depth): first timestep's activations
    with torch.no_grad():
        conf_steps = conf.num_timesteps
        acts_p, _, _, _ = gd.initial_inference(noise, scene.disp, unc, prompt)
    assert acts_p[0].shape[0] == conf_steps
    for k in range(3):
        assert rel(acts_p[k][0], o_acts[k][0]) < 1e-2, (name, k)
    worst = dict(step=0.0, upd=0.0, upd3=0.0)
    with torch.no_grad(), gd.on_stream():
        gd.scheduler.set_timesteps(conf.num_timesteps)
        ts = gd.scheduler.timesteps
        st = gd.prepare_guidance(scene.disp_e, prompt, [a.float() for a in o_acts], scene.corr)
        assert (st.plan is None) == (conf.bg_loss_type != "global_avg")       # 'local_avg' takes the general energy path
        assert (st.depth_nhwc is None) == (not conf.use_depth)
        for i in range(NT):
            x_in = noise if i == 0 else rec_o["step"][i - 1]
            rec = {}
            x_out = gd.guided_step(st, x_in.permute(0, 2, 3, 1).contiguous(), i, ts[i], unc[i], record=rec)
            e = rel(x_out.permute(0, 3, 1, 2), rec_o["step"][i])
            # the first optimisation iteration's update is the guidance gradient itself; after three iterations of an L1
            # energy (sign-valued gradient) flipped signs accumulate, so the whole update is held loosely
            eu = rel(rec["opt"][0] - x_in, rec_o["opt"][iters * i] - x_in)
            eu3 = rel(rec["opt"][iters - 1] - x_in, rec_o["opt"][iters * i + iters - 1] - x_in)
            worst["step"], worst["upd"], worst["upd3"] = max(worst["step"], e), max(worst["upd"], eu), max(worst["upd3"], eu3)
            # use_depth false, first step: the U-Net input of the edit IS the input the original activations were recorded with
            # (no edited depth channel), so current == original activations exactly in the fp32 oracle and every foreground
            # pair that maps a cell onto itself sits on the kink of the L1 energy: sign(0) = 0 there in fp32, +-1 from 16-bit
            # rounding in the engine (measured: energy gradient 36 % off on those cells, engine backward 1.8e-3 for the same
            # cotangent).  That step is held by its outcome only; from step 1 on the latents have moved and the check is full.
            degenerate = (not conf.use_depth) and i == 0
            assert e < 2e-2 and (degenerate or (eu < 6e-2 and eu3 < 0.2)), (name, i, e, eu, eu3)
    print(name, "worst", worst)
    # the oracle run here (fp32 on the GPU, weights rounded to fp16) follows the reference-generated trajectory of the same
    # configuration (fp32 CPU, unrounded weights) at the first update, before the L1 signs can flip
    up_here = rec_o["opt"][0].cpu() - noise.cpu()
    up_ref = torch.from_numpy(g14[name + ".opt_t0"][0]) - noise.cpu()
    # (not for use_depth false: its first update sits on the L1 kink, see above -- there even two fp32 evaluations of the same
    # network disagree on the signs of the self-mapped cells, measured 0.26)
    assert (not conf.use_depth) or rel(up_here, up_ref) < 0.1, (name, rel(up_here, up_ref))


def test_full_debug_saves_the_denoising_steps_like_the_reference(scene, golden):
    """test/config/full_debug.yaml: save_denoising_steps.  The reference returns (image, {'opt': [[image after the optimisation
    loop, image after the DDIM step] per timestep], 'post-opt': []}) with the images on the CPU
    (guided_stable_diffuser.py:329-334, 385-386, 444-447, 477-479, 485-486), also for the unguided tail where the first image
    is the unchanged latent."""
    from diffusionhandles_amd import conf as C
    from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
    from diffusionhandles_amd.unet import HipUNet
    from oracle import unet_torch as U
    dev = scene.dev
    g14 = golden("g14_loop_variants.npz")
    raw = json.loads(str(g14["full_debug.conf"]))
    assert raw["save_denoising_steps"] is True
    conf = C.load_default().guided_diffuser
    for k, v in raw.items():
        setattr(conf, k, v)
    cfg = dict(U.TINY)
    ref = U.init_synthetic_(U.UNetTorch(cfg), seed=0).to(dev).eval()
    hip = HipUNet(dict(cfg, text_len=77), dtype=torch.float16, max_batch=2)
    hip.load_state_dict(ref.state_dict())
    gd = GuidedStableDiffuser(conf, unet=hip, unet_config=dict(cfg, text_len=77), tokenizer=FixedTokenizer(),
                              text_encoder=FixedTextEncoder(cfg["cross_attention_dim"])).to(dev)
    prompt = "a sphere on a plane"
    unc = gd._encode([""])[None].expand(50, -1, -1, -1).contiguous()
    noise = torch.from_numpy(g14["noise"]).to(dev)
    with torch.no_grad():
        acts, _, _, _ = gd.initial_inference(noise, scene.disp, unc, prompt)
    rec = {}
    image, steps = gd.guided_inference(noise, scene.disp_e, unc, prompt, acts, scene.corr,
                                       save_denoising_steps=conf.save_denoising_steps, record=rec)
    assert sorted(steps) == ["opt", "post-opt"] and steps["post-opt"] == []
    nt, it, gmax = conf.num_timesteps, conf.num_optsteps, conf.guidance_max_step
    assert len(steps["opt"]) == nt and all(len(s) == 2 for s in steps["opt"])
    assert len(rec["opt"]) == it * gmax and len(rec["step"]) == nt
    assert all(im.device.type == "cpu" and im.shape == image.shape for s in steps["opt"] for im in s)
    with torch.no_grad():
        for t in (0, 1, gmax - 1, gmax, nt - 1):
            before = rec["opt"][it * t + it - 1] if t < gmax else rec["step"][t - 1]       # unguided tail: the latent is untouched
            assert torch.equal(steps["opt"][t][0], gd.decode_latent_image(before).cpu()), t
            assert torch.equal(steps["opt"][t][1], gd.decode_latent_image(rec["step"][t]).cpu()), t
    assert torch.equal(steps["opt"][-1][1], image.cpu())
    # and the plain call returns the image alone
    plain = gd.guided_inference(noise, scene.disp_e, unc, prompt, acts, scene.corr)
    assert torch.is_tensor(plain) and torch.equal(plain, image)
