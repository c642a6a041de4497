#!/usr/bin/env python3
"""Planned guidance-energy evaluation on the bench scene's real correspondences (run under rocprofv3 --kernel-trace).

Prints one JSON line with what each kernel of an evaluation touches (bytes), so that tools/hbm_report.py can turn the
kernel trace into achieved GB/s per kernel:  DH_RES=512|768, C = 320 and 640 (act2 and act1 of the SD-2-depth U-Net)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import losses as LS
from diffusionhandles_amd.depth_transform import transform_depth
from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene
dev = torch.device("cuda:0")
res = int(os.environ.get("DH_RES", "512"))
grid = res // 8
depth, bg, mask = (t.to(dev) for t in make_scene(res))
ang, tr = TRANSFORMS[2]
_, corr = transform_depth(depth, bg, mask, GuidedStableDiffuser.get_depth_intrinsics(), rot_angle=ang,
                          rot_axis=torch.tensor([0.0, 1.0, 0.0]), translation=torch.tensor(tr))
pc = LS.process_correspondences(corr, res, 0, grid=grid, device=dev)
plan = LS.EnergyPlan(pc, grid, dev)
n1, n2 = int(plan.dl["bg_orig"].numel()), int(plan.dl["bg_trans"].numel())
cells = torch.stack([torch.as_tensor(pc["original_y"]) * grid + torch.as_tensor(pc["original_x"]),
                     torch.as_tensor(pc["transformed_y"]) * grid + torch.as_tensor(pc["transformed_x"])], dim=1)
uniq = int(torch.unique(cells, dim=0).shape[0])
gen = torch.Generator().manual_seed(1)
info = dict(res=res, grid=grid, pairs=int(corr.shape[0]), unique_cell_pairs=uniq, n_bg_orig=n1, n_bg_trans=n2, layers=[])
for C in (320, 640):
    cur = torch.randn(grid, grid, C, generator=gen).half().to(dev)
    org = torch.randn(grid, grid, C, generator=gen).half().to(dev)
    for _ in range(50):
        LS.energy_and_grad_planned(cur, org, plan, 3.0, 2.0, grad_scale=256.0)
    G2 = grid * grid
    info["layers"].append(dict(C=C, algorithmic_bytes=3 * G2 * C * 2, kernels=dict(
        k_colsum_q=(n1 + n2) * C * 2 + (n1 + n2) * 4 * (C // 64) + 8 * C * 4,
        k_energy_grad=2 * G2 * C * 2 + uniq * C * 2 + uniq * 8 + G2 * 9)))
torch.cuda.synchronize()
print(json.dumps(info))
