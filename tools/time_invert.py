#!/usr/bin/env python3
"""Time the per-image phases: null-text inversion (a1 + a2) and initial inference (a14)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import DiffusionHandles
from diffusionhandles_amd import conf as C
from diffusionhandles_amd.synthetic import make_image, make_scene
dev = torch.device("cuda:0")
dh = DiffusionHandles(C.load_default(), dtype=torch.float16).to(dev)
depth, bg, mask = (t.to(dev) for t in make_scene(512))
img = make_image(512).to(dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    nt, noise = dh.invert_input_image(img, depth, "a sphere on a plane")
    torch.cuda.synchronize(); t1 = time.time()
    dh.generate_input_image(depth, "a sphere on a plane", nt, noise)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"rep {rep}: invert {t1-t0:.2f} s, initial inference {t2-t1:.2f} s", flush=True)
st = getattr(dh.inverter, "last_stats", None)
print("inverter stats:", st)
