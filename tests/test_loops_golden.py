"""CPU: the oracle's loops (oracle/loop_ref.py) against the trajectories captured from the
reference's own initial_inference / guided_inference / StableNullInverter.invert run with
the TINY stand-in U-Net (tools/make_golden.py).  Bounded: a few steps of each loop."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from diffusionhandles_amd.synthetic import TRANSFORMS, make_image, make_scene
from oracle import depth_ref as D
from oracle import loop_ref as L
from oracle import unet_torch as U


def text_embedding(prompt, dim, seed_base=1000):
    g = torch.Generator().manual_seed(seed_base + sum(prompt.encode()))
    return torch.randn(1, 77, dim, generator=g)


@pytest.fixture(scope="module")
def setup():
    torch.manual_seed(0)
    unet = U.init_synthetic_(U.UNetTorch(U.TINY), seed=0).eval()
    depth, bg, mask = make_scene(512)
    disp = D.normalize_depth(1.0 / depth)[0]
    cond = text_embedding("a sphere on a plane", U.TINY["cross_attention_dim"])
    unc0 = text_embedding("", U.TINY["cross_attention_dim"])
    return unet, depth, bg, mask, disp, cond, unc0


def test_guided_first_steps_match_reference(setup, golden):
    unet, depth, bg, mask, disp, cond, unc0 = setup
    g7, g3 = golden("g7_loops.npz"), golden("g3_zbuffer.npz")
    noise = torch.from_numpy(g7["inv_init_noise"])
    conf = SimpleNamespace(bg_weight=1.25, fg_weight=1.5, fg_patch_size=1, bg_patch_size=1, bg_loss_type="global_avg",
                           num_timesteps=50, num_optsteps=3, guidance_max_step=38,
                           guidance_schedule_type="constant", bg_erosion=0, seed=2773)
    # two guided steps of the loop: needs acts_orig for t_idx 0,1 and the uncond list
    unc = torch.from_numpy(g7["inv_uncond_first"])          # [3,1,77,C]
    sched = L.DDIM()
    depth64 = L.init_depth(disp, (64, 64))
    acts = ([], [], [])
    x = noise
    with torch.no_grad():
        for i in range(2):
            t = sched.timesteps[i]
            out = unet(torch.cat([x, depth64], dim=1), t, encoder_hidden_states=cond, return_dict=False)
            for k in range(3):
                acts[k].append(out[4 + k][0])
            x = sched.step(L._eps_cfg(unet, x, depth64, t, unc[i], cond), t, x)
    ref0 = np.concatenate([a[0].reshape(-1)[::97].numpy() for a in acts])
    assert np.allclose(ref0, g7["init_acts_t0"], atol=1e-5)
    disp_e, corr = D.transform_depth_pc(depth, bg, mask, rot_angle=TRANSFORMS[2][0], rot_axis=[0, 1, 0],
                                        translation=TRANSFORMS[2][1])
    assert np.array_equal(corr.numpy(), g3["t2_corr"].astype(np.int64))

    class TwoSteps(L.DDIM):
        def set_timesteps(self, n):
            super().set_timesteps(n)
            self.timesteps = self.timesteps[:2]
    rec = {}
    L.guided_inference(unet, TwoSteps(), noise, disp_e, unc, cond, [torch.stack(a) for a in acts], corr.numpy(), conf,
                       record=rec)
    got = torch.stack(rec["opt"][:6]).numpy()
    assert np.allclose(got, g7["guided_opt_first"], atol=2e-5), np.abs(got - g7["guided_opt_first"]).max()
    assert np.allclose(rec["step"][0].numpy(), g7["guided_steps"][0], atol=2e-5)


def test_inversion_first_steps_match_reference(setup, golden):
    unet, depth, bg, mask, disp, cond, unc0 = setup
    g7 = golden("g7_loops.npz")
    img = make_image(512)
    z = torch.nn.functional.avg_pool2d(img * 2 - 1, 8)
    lat0 = torch.cat([z, z.mean(dim=1, keepdim=True)], dim=1) * L.VAE_SCALE
    # full DDIM inversion (50 no-grad forwards) then 2 null-text timesteps
    lat, unc = L.null_text_inversion(unet, L.DDIM(), lat0, disp, unc0, cond, num_inner_steps=5, null_steps=2)
    assert np.allclose(lat[-1].numpy(), g7["inv_init_noise"], atol=1e-5)
    assert np.allclose(unc.numpy(), g7["inv_uncond_first"][:2], atol=1e-5)


VARIANTS = ["bg_erosion_10_local_avg", "local_avg_bg_loss", "linear_schedule", "quadratic_schedule", "no_depth", "bg_erosion_15_local_avg"]


@pytest.mark.parametrize("name", VARIANTS)
def test_guided_variants_match_reference(name, golden):
    """The oracle loop under the reference's own variant configurations (test/config/<name>.yaml: eroded background masks,
    'local_avg' background loss, linear / quadratic schedules over 50 guided steps, use_depth false with a 4-channel U-Net)
    against the trajectories the REFERENCE's guided_inference produced with them (tools/make_golden_variants.py, g14).
    Two timesteps: t_idx 1 is where the falling schedules first differ from the constant one."""
    import json
    g14 = golden("g14_loop_variants.npz")
    conf = SimpleNamespace(**json.loads(str(g14[name + ".conf"])))
    cfg = dict(U.TINY) if conf.use_depth else dict(U.TINY, in_channels=4)
    torch.manual_seed(0)
    unet = U.init_synthetic_(U.UNetTorch(cfg), seed=0).eval()
    depth, bg, mask = make_scene(512)
    disp = D.normalize_depth(1.0 / depth)[0]
    cond = text_embedding("a sphere on a plane", U.TINY["cross_attention_dim"])
    unc = text_embedding("", U.TINY["cross_attention_dim"])[None].expand(50, -1, -1, -1).contiguous()
    noise = torch.from_numpy(g14["noise"])
    disp_e, corr = D.transform_depth_pc(depth, bg, mask, rot_angle=TRANSFORMS[2][0], rot_axis=[0, 1, 0],
                                        translation=TRANSFORMS[2][1])

    class TwoSteps(L.DDIM):
        def set_timesteps(self, n):
            super().set_timesteps(n)
            self.timesteps = self.timesteps[:2]
    acts, _, _, _ = L.initial_inference(unet, TwoSteps(), noise, disp, unc, cond, use_depth=conf.use_depth)
    rec = {}
    L.guided_inference(unet, TwoSteps(), noise, disp_e, unc, cond, acts, corr.numpy(), conf, record=rec)
    got = torch.stack(rec["opt"]).numpy()
    assert np.allclose(got[:3], g14[name + ".opt_t0"], atol=2e-5), np.abs(got[:3] - g14[name + ".opt_t0"]).max()
    assert np.allclose(got[:, 0, :, ::4, ::4], g14[name + ".opt_all"][:6], atol=5e-5)
    assert np.allclose(torch.stack(rec["step"]).numpy(), g14[name + ".step"][:2], atol=5e-5)
    # the reference's loop and the oracle's agreed to this when the fixture was written (acts, latent, after-opt, after-step)
    assert float(g14[name + ".ref_minus_oracle"].max()) < 2e-4
