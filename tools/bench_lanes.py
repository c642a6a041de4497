#!/usr/bin/env python3
"""Concurrent edit lanes in one process (GuidedStableDiffuser.fork: two engine arenas + streams on one copy of the weights):
whole-edit throughput for (edits, batch, streams) combinations and guided steps/s of B = 1 edits on 1 / 2 / 3 lanes.
Run on the GPU box: python3 tools/bench_lanes.py > gpurun_out/lanes.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionhandles_amd import conf as C
from diffusionhandles_amd.depth_transform import normalize_depth, transform_depth
from diffusionhandles_amd.diffusion_handles import DiffusionHandles
from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene
from diffusionhandles_amd.unet import SD2_DEPTH

dev = torch.device("cuda:0")
conf = C.load_default()
MAXB = 32 if os.environ.get("DH_LANES_CASES") == "batch16" else 16
dh = DiffusionHandles(conf, dtype=torch.float16, unet_config=dict(SD2_DEPTH), max_batch=MAXB, vae="sd-native",
                      text_encoder="sd2-native").to(dev)
gd = dh.diffuser
depth, bg_depth, mask = (t.to(dev) for t in make_scene(512))
prompt = "a sphere on a plane"
T = conf.guided_diffuser.num_timesteps
gmax = conf.guided_diffuser.guidance_max_step
uncond = gd._encode([""])[None].expand(T, -1, -1, -1).contiguous()
torch.manual_seed(conf.guided_diffuser.seed)
noise = torch.randn(1, 4, 64, 64).to(dev)
acts, _, _, init_noise = gd.initial_inference(noise, normalize_depth(1.0 / depth), uncond, prompt)
Y = torch.tensor([0.0, 1.0, 0.0])


def tfs(n):
    return [(TRANSFORMS[i % 8][0], Y, torch.tensor(TRANSFORMS[i % 8][1])) for i in range(n)]


print("weights GB", gd.unet.weight_bytes() / 1e9, f"workspace GB (max_batch {MAXB})", gd.unet.workspace_bytes() / 1e9, flush=True)
# whole edits: (edits, batch, streams)
ref = {}
CASES = ((8, 8, 1), (8, 4, 1), (8, 4, 2), (16, 8, 1), (16, 8, 2), (8, 2, 2), (8, 1, 2), (8, 1, 1), (12, 4, 3))
if os.environ.get("DH_LANES_CASES") == "queues":    # short list for A/Bs of runtime settings (GPU_MAX_HW_QUEUES)
    CASES = ((16, 8, 1), (16, 8, 2), (24, 8, 3))
elif os.environ.get("DH_LANES_CASES") == "wide":      # more lanes of full batches
    CASES = ((24, 8, 1), (24, 8, 3), (32, 8, 1), (32, 8, 4), (16, 8, 2))
if os.environ.get("DH_LANES_CASES") == "batch16":   # sixteen edits as ONE batch of 16 (CFG pass at 32) against 2 x 8 on one stream / two lanes
    CASES = ((16, 8, 1), (16, 16, 1), (16, 8, 2), (32, 16, 1), (32, 16, 2))
for n, batch, streams in CASES:
    with torch.no_grad():
        dh.transform_foreground_batch(depth, prompt, mask, bg_depth, uncond, init_noise, acts, tfs(min(n, 2 * batch)), streams=streams, batch=batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        imgs, _ = dh.transform_foreground_batch(depth, prompt, mask, bg_depth, uncond, init_noise, acts, tfs(n), streams=streams, batch=batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    key = (n, batch)
    same = ""
    if streams == 1:
        ref[key] = imgs.clone()
    elif key in ref:
        same = f"  bit-identical to one stream: {bool(torch.equal(ref[key], imgs))}"
    lanes = gd.lanes(streams) if streams > 1 else [gd]
    ws = sum(l.unet.workspace_bytes() for l in lanes) / 1e9
    print(f"edits {n:2d} batch {batch} streams {streams}: {dt:6.3f} s  {n / dt:5.3f} edits/s   arenas {ws:.1f} GB{same}", flush=True)

# guided steps/s of single edits on 1..3 lanes
disp_e, corr = transform_depth(depth, bg_depth, mask, gd.get_depth_intrinsics(), rot_angle=TRANSFORMS[2][0], rot_axis=Y,
                               translation=torch.tensor(TRANSFORMS[2][1]))
for nl in ((1, 2, 3, 4) if os.environ.get("DH_LANES_CASES") == "wide" else (1, 2, 3)):
    lanes = gd.lanes(nl)
    x0 = init_noise.to(dev, torch.float32).permute(0, 2, 3, 1).contiguous()
    with torch.no_grad():
        sts = []
        for ln in lanes:
            ln.scheduler.set_timesteps(T)
            sts.append(ln.prepare_guidance(disp_e, prompt, acts, corr))
        ts = lanes[0].scheduler.timesteps
        xs = [x0 for _ in lanes]

        def rounds(n, i0):
            for i in range(n):
                for li, ln in enumerate(lanes):
                    with torch.cuda.stream(ln._stream):
                        xs[li] = ln.guided_step(sts[li], xs[li] if (i0 + i) % gmax else x0, (i0 + i) % gmax, ts[(i0 + i) % gmax], uncond[(i0 + i) % gmax])
        torch.cuda.synchronize()          # the guidance states were prepared on the current stream
        rounds(3, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rounds(20, 3)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"B=1 guided steps on {nl} lane(s): {20 * nl / dt:6.2f} steps/s total ({dt / 20 * 1e3:.2f} ms per round)", flush=True)
