#!/usr/bin/env python3
"""Guidance energy + gradient (HIP general and planned paths) vs the oracle's autograd on the correspondences of the real
scene's three edits and on random correspondence sets (duplicates, single pair, erosion 0/5/10, both background loss
types, patch 1/3).  One-off tool for the GPU box: python tools/fuzz_energy.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusionhandles_amd import losses as LS  # noqa: E402
from oracle import guidance_ref as G  # noqa: E402

dev = torch.device("cuda:0")
g11 = np.load(os.path.join(ROOT, "tests", "golden", "g11_scene.npz"))
gen = torch.Generator().manual_seed(11)
sets = {n: torch.from_numpy(g11[f"{n}_corr"].astype(np.int64)) for n in ("edit_000", "edit_001", "edit_002")}
sets["random_dups"] = torch.stack([torch.randint(100, 140, (3000,), generator=gen) for _ in range(4)], dim=-1)
sets["single"] = torch.tensor([[17, 300, 400, 90]])
bad = 0
for name, corr in sets.items():
    for erosion in (0, 5, 10):
        pc = LS.process_correspondences(corr, 512, erosion)
        cells = G.cells_from_correspondences(corr.numpy(), 512, erosion)
        for C, patch, bgt in ((320, 1, "global_avg"), (64, 3, "local_avg"), (640, 1, "local_avg")):
            cur = torch.randn(C, 64, 64, generator=gen)
            org = torch.randn(C, 64, 64, generator=gen)
            loss, grad = LS.energy_and_grad(cur.to(dev), org.to(dev), pc, 2.0, 1.5, patch, patch, (64, 64), bg_loss_type=bgt,
                                            channels_last=False)
            a = cur.clone().requires_grad_(True)
            ref = 2.0 * G.foreground_energy(a, org, cells, patch, (64, 64)) + \
                1.5 * G.background_energy(a, org, cells, patch, (64, 64), loss_type=bgt)
            gr, = torch.autograd.grad(ref, a)
            e_l = abs(loss[0].item() - ref.item()) / max(1e-6, abs(ref.item()))
            e_g = float((grad.cpu() - gr).abs().max()) / max(1e-12, float(gr.abs().max()))
            ok = e_l < 2e-5 and e_g < 2e-4
            bad += 0 if ok else 1
            print(f"{name:12s} erosion {erosion:2d} C={C:4d} patch {patch} {bgt:10s} N={corr.shape[0]:6d} loss rel {e_l:.1e} grad rel {e_g:.1e} {'ok' if ok else 'MISMATCH'}")
print("FAILED" if bad else "all ok")
sys.exit(1 if bad else 0)
