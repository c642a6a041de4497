#!/usr/bin/env python3
"""Same-process A/B of the guided step with and without the engine's in-place I/O (dh_unet_io_ptr / dh_pack_sample /
dh_latent_update_strided): identical kernels, the copies and torch.cat around the passes present or not.  Interleaved repeats."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionhandles_amd import DiffusionHandles
from diffusionhandles_amd import conf as C
from diffusionhandles_amd.depth_transform import normalize_depth, transform_depth
from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene

dev = torch.device("cuda:0")
conf = C.load_default()
dh = DiffusionHandles(conf).to(dev)
gd = dh.diffuser
depth, bg, mask = (t.to(dev) for t in make_scene(512))
T, gmax = 50, 38
unc = gd._encode([""])[None].expand(T, -1, -1, -1).contiguous()
torch.manual_seed(2773)
noise = torch.randn(1, 4, 64, 64).to(dev)
acts, _, _, init_noise = gd.initial_inference(noise, normalize_depth(1.0 / depth), unc, "a sphere on a plane")
ang, tr = TRANSFORMS[2]
disp_e, corr = transform_depth(depth, bg, mask, gd.get_depth_intrinsics(), rot_angle=ang, rot_axis=torch.tensor([0.0, 1.0, 0.0]),
                               translation=torch.tensor(tr))
st = gd.prepare_guidance(disp_e, "a sphere on a plane", acts, corr)
gd.scheduler.set_timesteps(T)
ts = gd.scheduler.timesteps
x0 = init_noise.permute(0, 2, 3, 1).contiguous()


def run(n):
    x = x0
    with torch.no_grad(), gd.on_stream():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            x = gd.guided_step(st, x0 if i % gmax == 0 else x, i % gmax, ts[i % gmax], unc[i % gmax])
        torch.cuda.synchronize()
    return n / (time.perf_counter() - t0), x


finals = {}
for rep in range(3):
    for ip in (False, True):
        gd._inplace_io = ip
        run(6)
        sps, x = run(38)
        finals[ip] = x
        print(f"in-place I/O {ip}: {sps:.2f} steps/s", flush=True)
print("bit-identical trajectories:", bool(torch.equal(finals[False], finals[True])))
