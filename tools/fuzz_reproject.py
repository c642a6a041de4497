#!/usr/bin/env python3
"""Randomised parity sweep of the re-projection (HIP vs the CPU oracle) on the synthetic scene and on the real scene of
the reference's test data: y-axis rotations must agree bit for bit; general axes may differ in a few map entries
(np.dot goes through BLAS on the oracle side).  One-off tool, run on the GPU box:
  python tools/fuzz_reproject.py [n_transforms] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusionhandles_amd import depth_transform as DT  # noqa: E402
from diffusionhandles_amd import scene_io as S  # noqa: E402
from diffusionhandles_amd.synthetic import make_scene  # noqa: E402
from oracle import depth_ref as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
K = D.intrinsics_f32()
sc = S.load_scene(os.path.join(ROOT, "tests", "golden", "scene_banana_fruits"), 512)
scenes = {"synthetic": make_scene(512), "banana": (sc["depth"], sc["bg_depth"], sc["fg_mask"])}
bad = 0
for name, (depth, bg, mask) in scenes.items():
    tfs = []
    for i in range(n):
        axis = [0.0, 1.0, 0.0] if i % 3 else list(rng.normal(size=3))
        tfs.append((float(rng.uniform(-95, 95)), torch.tensor(axis, dtype=torch.float32),
                    torch.tensor([rng.uniform(-1, 1), rng.uniform(-0.3, 0.3), rng.uniform(-1, 1)], dtype=torch.float32)))
    t0 = time.time()
    out = DT.reproject_edits(depth.to(dev), bg.to(dev), mask.to(dev), K, tfs)
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    for i, (ang, axis, tr) in enumerate(tfs):
        disp_o, corr_o = D.transform_depth_pc(depth, bg, mask, K, rot_angle=ang, rot_axis=axis.tolist(), translation=tr.tolist())
        disp, corr = out[i]
        y_axis = i % 3 != 0
        if corr.shape == corr_o.shape and torch.equal(corr, corr_o):
            nd = 0
        else:
            a = {tuple(r) for r in corr.tolist()}
            b = {tuple(r) for r in corr_o.tolist()}
            nd = len(a ^ b)
        dd = float((disp.cpu() - disp_o).abs().max())
        ok = (nd == 0) if y_axis else (nd <= 40)
        bad += 0 if ok and dd < 0.5 else 1
        print(f"{name:9s} t{i:02d} angle {ang:7.2f} axis {'y' if y_axis else 'general'} N={corr_o.shape[0]:6d} differing pairs {nd:4d} max|ddisp| {dd:.2e} {'ok' if ok else 'MISMATCH'}")
    print(f"{name}: {n} edits on the GPU in {t_gpu*1e3:.1f} ms")
print("FAILED" if bad else "all ok")
sys.exit(1 if bad else 0)
