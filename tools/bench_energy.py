#!/usr/bin/env python3
"""Planned guidance-energy evaluation on the bench scene's correspondences (run under rocprofv3 --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import losses as LS
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
n = 24906
oy = torch.randint(200, 400, (n,), generator=gen); ox = torch.randint(180, 420, (n,), generator=gen)
corr = torch.stack([ox, oy, ox + 40, oy + 3], dim=-1)
pc = LS.process_correspondences(corr, 512, 0)
plan = LS.EnergyPlan(pc, 64, dev)
for C in (320, 640):
    cur = torch.randn(64, 64, C, generator=gen).half().to(dev); org = torch.randn(64, 64, C, generator=gen).half().to(dev)
    for _ in range(50):
        LS.energy_and_grad_planned(cur, org, plan, 3.0, 2.0, grad_scale=256.0)
torch.cuda.synchronize()
