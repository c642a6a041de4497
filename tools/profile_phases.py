#!/usr/bin/env python3
"""Where the time of the per-image phase and of a batched edit goes (diagnostic, run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import DiffusionHandles
from diffusionhandles_amd import conf as C
from diffusionhandles_amd.depth_transform import normalize_depth, reproject_edits
from diffusionhandles_amd.synthetic import TRANSFORMS, make_image, make_scene

dev = torch.device("cuda:0")
K = 8
conf = C.load_default()
dh = DiffusionHandles(conf, dtype=torch.float16, max_batch=2 * K, vae="sd").to(dev)
gd, inv = dh.diffuser, dh.inverter
depth, bg, mask = (t.to(dev) for t in make_scene(512))
img = make_image(512).to(dev)
prompt = "a sphere on a plane"
disp = normalize_depth(1.0 / depth)


def T(label, fn, n=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{label:60s} {dt*1e3:9.2f} ms", flush=True)
    return r


with torch.no_grad(), gd.on_stream():
    d64 = gd.init_depth(disp).permute(0, 2, 3, 1).contiguous()
    ctx = gd.init_prompt(prompt)
    unc, cond = ctx.chunk(2)
    unc = unc.clone().contiguous(); cond = cond.contiguous()
    lat = T("vae encode", lambda: inv.image2latent(img))
    x = lat.permute(0, 2, 3, 1).contiguous()
    t = gd.scheduler.timesteps[3]
    T("forward B=1 no save (eps)", lambda: inv.get_noise_pred_single(x, t, cond, d64), 5)
    T("forward B=1 save (eps)", lambda: inv.get_noise_pred_single(x, t, unc, d64, save=True), 5)
    d_eps = torch.randn(1, 64, 64, 4, device=dev) * 1e-3
    T("backward eps -> text", lambda: gd.unet.backward(None, d_eps, want_sample_grad=False, want_text_grad=True), 5)
    T("backward eps -> sample", lambda: gd.unet.backward(None, d_eps, want_sample_grad=True, want_text_grad=False), 5)
    T("cfg eps B=2", lambda: gd._cfg_eps(x, d64, t, unc, cond), 5)
    target = x + 0.01 * torch.randn_like(x)
    T("null_step (5 inner)", lambda: inv.null_step(x, unc.clone(), cond, d64, 3, target, 5, 0.0), 2)
T("invert_input_image", lambda: dh.invert_input_image(img, depth, prompt))
T("invert_input_image (2nd)", lambda: dh.invert_input_image(img, depth, prompt))
T = T
uncond = gd._encode([""])[None].expand(50, -1, -1, -1).contiguous()
noise = torch.randn(1, 4, 64, 64, device=dev)
acts, _, _, init_noise = T("initial_inference", lambda: gd.initial_inference(noise, disp, uncond, prompt))
tfs = [(TRANSFORMS[i][0], torch.tensor([0.0, 1.0, 0.0]), torch.tensor(TRANSFORMS[i][1])) for i in range(K)]
with torch.no_grad():
    edits = T("reproject_edits K=8", lambda: reproject_edits(depth, bg, mask, gd.get_depth_intrinsics(), tfs))
    sts = T("prepare_guidance x8", lambda: [gd.prepare_guidance(d, prompt, acts, c) for d, c in edits])
    x8 = init_noise.permute(0, 2, 3, 1).contiguous().expand(K, -1, -1, -1).contiguous()
    gd.scheduler.set_timesteps(50)
    ts = gd.scheduler.timesteps
    with gd.on_stream():
        for i in (0, 1, 2, 40):
            gd.guided_step_batch(sts, x8, i, ts[i], uncond[i])
        for i in (0, 1, 2, 40):
            T(f"guided_step_batch t_idx={i}", lambda: gd.guided_step_batch(sts, x8, i, ts[i], uncond[i]), 2)
    T("decode 8 latents", lambda: gd.decode_latent_image(torch.randn(8, 4, 64, 64, device=dev)))
    T("decode 8 latents (2nd)", lambda: gd.decode_latent_image(torch.randn(8, 4, 64, 64, device=dev)))
    T("decode 1 latent", lambda: gd.decode_latent_image(torch.randn(1, 4, 64, 64, device=dev)))
    T("transform_foreground_batch", lambda: dh.transform_foreground_batch(depth, prompt, mask, bg, uncond, init_noise, acts, tfs))
