#!/usr/bin/env python3
"""Batched K=8 reprojection of the bench scene (run under rocprofv3 --stats for per-kernel times)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd.depth_transform import reproject_edits
from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene
dev = torch.device("cuda:0")
res = int(os.environ.get("DH_RES", "512"))
depth, bg_depth, mask = (t.to(dev) for t in make_scene(res))
K = 8
tfs = [(TRANSFORMS[i % 8][0], torch.tensor([0.0, 1.0, 0.0]), torch.tensor(TRANSFORMS[i % 8][1])) for i in range(K)]
intr = GuidedStableDiffuser.get_depth_intrinsics(dev)
for _ in range(2):
    reproject_edits(depth, bg_depth, mask, intr, tfs)
torch.cuda.synchronize()
t0 = time.time()
n = 5
for _ in range(n):
    out = reproject_edits(depth, bg_depth, mask, intr, tfs)
torch.cuda.synchronize()
out, dbg = reproject_edits(depth, bg_depth, mask, intr, tfs, return_debug=True)
cn = dbg["counts"]
print(f"K={K} res={res}: {(time.time()-t0)/n*1e3:.2f} ms per call; correspondences {[int(c.shape[0]) for _, c in out]}; "
      f"in-fill pixels {[int(v) for v in cn[:, 2]]}; CG iterations {[int(v) for v in cn[:, 3]]}")
