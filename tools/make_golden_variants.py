#!/usr/bin/env python3
"""g14: the guided loop over the reference's OPTION SPACE (test/config/*.yaml), pinned to the reference's own loop.

tools/make_golden.py pins the loops (g7) with the default configuration only.  This generator reads the reference's
variant configuration files where they lie (/root/reference/test/config/<name>.yaml: eroded background masks, the
'local_avg' background loss, the linear / quadratic weight schedules with guidance_max_step 50, use_depth false), runs the
REFERENCE's `initial_inference` and `guided_inference(save_denoising_steps=True)` with each of them on the TINY stand-in U-Net
(4 input channels for use_depth false) for the first NT timesteps, asserts the oracle loop (oracle/loop_ref.py) equal on the
spot, and stores the trajectories in tests/golden/g14_loop_variants.npz.

The stand-in VAE decodes a latent to an un-clamped affine image of all four channels, so the images the reference returns
are the latents themselves (lat = (img - 0.5) * 20): every recorded point is compared in full.

Only runs in the build container (needs /root/reference).  Fixtures are data; no reference source is copied.
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as MG  # noqa: E402
from oracle import depth_ref as D  # noqa: E402
from oracle import loop_ref as L  # noqa: E402
from oracle import unet_torch as U  # noqa: E402
from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene  # noqa: E402

VARIANTS = ["bg_erosion_10_local_avg", "local_avg_bg_loss", "linear_schedule", "quadratic_schedule", "no_depth",
            "bg_erosion", "quadratic_schedule_local_avg", "bg_erosion_15_local_avg", "bg_erosion_local_avg", "full_debug"]
# (the eleventh file, mesh_depth_transform.yaml, has the default guided_diffuser block: it differs in depth_transform_mode only,
#  which tools/make_golden_mesh.py pins)
NT = 4            # timesteps of each loop that are run and stored (t_idx 0..3: every layer phase, schedule fall-off visible)
IMG_GAIN = 0.05   # image = latent * IMG_GAIN / 2 + 0.5 stays inside (0, 1) for |latent| < 20


class AffineVAE:
    """decode(z) = z * scaling * IMG_GAIN on all four channels (no up-sampling): an invertible stand-in."""
    config = SimpleNamespace(scaling_factor=L.VAE_SCALE, block_out_channels=(1, 1, 1, 1))

    def decode(self, z, return_dict=True):
        img = z * (L.VAE_SCALE * IMG_GAIN)
        return (img,) if return_dict is False else {"sample": img}


def img_to_latent(img):
    return (img - 0.5) * (2.0 / IMG_GAIN)


class ShortScheduler(MG.RefScheduler):
    def set_timesteps(self, n, device=None):
        super().set_timesteps(n)
        self.timesteps = self.timesteps[:NT]


class ShortDDIM(L.DDIM):
    def set_timesteps(self, n):
        super().set_timesteps(n)
        self.timesteps = self.timesteps[:NT]


def main():
    MG.install_stubs()
    import diffhandles.guided_stable_diffuser as RG
    depth, bg_depth, mask = make_scene(512)
    disp = D.normalize_depth(1.0 / depth)[0]
    disp_e, corr = D.transform_depth_pc(depth, bg_depth, mask, rot_angle=TRANSFORMS[2][0], rot_axis=[0, 1, 0],
                                        translation=TRANSFORMS[2][1])
    cdim = U.TINY["cross_attention_dim"]
    prompt = "a sphere on a plane"
    cond, unc0 = MG.text_embedding(prompt, cdim), MG.text_embedding("", cdim)
    unc = unc0[None].expand(50, -1, -1, -1).contiguous()
    noise = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(5))
    out = {}
    for name in VARIANTS:
        with open(os.path.join(MG.REF, "test", "config", name + ".yaml")) as fh:
            raw = yaml.safe_load(fh)
        conf = SimpleNamespace(**raw["guided_diffuser"])
        conf.save_denoising_steps = False
        cfg = dict(U.TINY) if conf.use_depth else dict(U.TINY, in_channels=4)
        torch.manual_seed(0)
        unet = U.init_synthetic_(U.UNetTorch(cfg), seed=0).eval()
        for p in unet.parameters():
            p.requires_grad_(True)      # the reference keeps weights requiring grad
        gd = object.__new__(RG.GuidedStableDiffuser)
        gd.conf = conf
        gd.scheduler = ShortScheduler()
        gd.unet = unet
        gd.device = torch.device("cpu")
        gd.tokenizer = type("Tok", (), {"model_max_length": 77, "__call__": lambda self, texts, **kw: SimpleNamespace(
            input_ids=SimpleNamespace(to=lambda dev, _t=texts: _t))})()
        gd.text_encoder = lambda ids: (MG.text_embedding(ids[0], cdim),)
        gd.vae = AffineVAE()
        with torch.no_grad():
            acts, latent_img, _, _ = gd.initial_inference(init_latents=noise, depth=disp, uncond_embeddings=unc, prompt=prompt)
        o_acts, o_latent, _, _ = L.initial_inference(unet, ShortDDIM(), noise, disp, unc, cond, use_depth=conf.use_depth)
        d_a = max((a - b).abs().max().item() for a, b in zip(acts, o_acts))
        d_l = (latent_img - o_latent).abs().max().item()
        assert acts[0].shape[0] == NT and d_a < 1e-4 and d_l < 1e-4, (name, d_a, d_l)
        _, steps = gd.guided_inference(latents=noise, depth=disp_e, uncond_embeddings=unc, prompt=prompt,
                                       activations_orig=acts, correspondences=corr, save_denoising_steps=True)
        torch.set_grad_enabled(True)
        assert len(steps["opt"]) == NT and all(len(s) == 2 for s in steps["opt"])
        ref_opt = torch.stack([img_to_latent(s[0]) for s in steps["opt"]])       # after the optimisation loop of timestep t
        ref_step = torch.stack([img_to_latent(s[1]) for s in steps["opt"]])      # after its DDIM step
        rec = {}
        L.guided_inference(unet, ShortDDIM(), noise, disp_e, unc, cond, acts, corr.numpy(), conf, record=rec)
        iters = conf.num_optsteps
        o_opt = torch.stack([rec["opt"][iters * t + iters - 1] for t in range(NT)])
        o_step = torch.stack(rec["step"])
        d_o, d_s = (o_opt - ref_opt).abs().max().item(), (o_step - ref_step).abs().max().item()
        moved = (ref_opt[0] - noise).abs().max().item()
        print(f"{name:32s} acts {d_a:.2e} latent {d_l:.2e} | guided: after-opt {d_o:.2e} after-step {d_s:.2e} "
              f"(first update max {moved:.3f})", flush=True)
        # f32 images carry the latents with ~1e-5 absolute resolution (IMG_GAIN): the comparison is as tight as that allows
        assert d_o < 2e-4 and d_s < 2e-4 and moved > 1e-3, name
        out[name + ".conf"] = np.array(json.dumps(raw["guided_diffuser"]))
        out[name + ".opt_all"] = torch.stack(rec["opt"])[:, 0, :, ::4, ::4].numpy()      # every iteration, strided
        out[name + ".opt_t0"] = torch.stack(rec["opt"][:iters]).numpy()                   # first timestep in full
        out[name + ".step"] = o_step.numpy()
        out[name + ".ref_minus_oracle"] = np.array([d_a, d_l, d_o, d_s])
    out["noise"] = noise.detach().numpy()
    np.savez_compressed(os.path.join(MG.OUT, "g14_loop_variants.npz"), **out)
    print("g14 ok")


if __name__ == "__main__":
    main()
