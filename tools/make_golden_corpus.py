#!/usr/bin/env python3
"""Golden vectors of the WHOLE shipped test set (round 6): the 20 PhotoGen scenes x their transforms = the 90 edits that
/root/reference/test/test_diffusion_handles.py:216-225 runs (test/data/photogen/photogen.json).

Per edit the REFERENCE's own `transform_depth_pc` (imported from /root/reference as in make_golden.py, with the cv2 stand-in that
is the oracle's restatement of the morphology) is run on the scene's estimated depth maps and mask, the oracle is
asserted equal on the spot (integer maps bit-equal, disparity <= 1e-4), and what is stored is SMALL: SHA-256 of the
correspondences, of the raw z-buffer mask, of the cleaned mask and of the visibility flags, the number of correspondences, the
number of unknown (in-filled) pixels, a strided slice and the sum of the disparity.  The inputs are data fixtures: copies of the
scenes' depth.exr / bg_depth.exr / mask.png / transforms.json under tests/golden/photogen/ (no reference source travels).
Writes tests/golden/g16_corpus.npz + g16_corpus.json (the human-readable table).  Only runs in the build container."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402
from make_golden import D, OUT, sha  # noqa: E402

from diffusionhandles_amd import scene_io as S  # noqa: E402


def main():
    MG.install_stubs()
    import diffhandles.depth_transform as RD
    import diffhandles.guided_stable_diffuser as RG
    K = RG.GuidedStableDiffuser.get_depth_intrinsics()
    root = os.path.join(OUT, "photogen")
    with open(os.path.join(root, "photogen.json")) as f:
        test_set = json.load(f)
    g, table = {}, []
    t00 = time.time()
    for scene, edits in test_set.items():
        sc = S.load_scene_geometry(os.path.join(root, scene), 512)
        depth, bg_depth, mask = sc["depth"], sc["bg_depth"], sc["fg_mask"]
        g[f"{scene}/depth_sha"] = sha(depth.numpy())
        g[f"{scene}/bg_depth_sha"] = sha(bg_depth.numpy())
        g[f"{scene}/mask_sha"] = sha(np.packbits(mask.numpy() != 0))
        for name in edits:
            t = sc["transforms"][name]
            ang, tr, axis = float(t["rotation_angle"]), [float(v) for v in t["translation"]], [float(v) for v in t["rotation_axis"]]
            t0 = time.time()
            disp_r, corr_r = RD.transform_depth_pc(depth, bg_depth, mask, K, rot_angle=ang, rot_axis=torch.tensor(axis),
                                                   translation=torch.tensor(tr))
            disp_o, corr_o, dbg = D.transform_depth_pc(depth, bg_depth, mask, K, rot_angle=ang, rot_axis=axis, translation=tr,
                                                       return_debug=True)
            assert torch.equal(corr_r, corr_o), (scene, name)
            assert torch.allclose(disp_r, disp_o, atol=1e-4, rtol=0), (scene, name, float((disp_r - disp_o).abs().max()))
            key = f"{scene}/{name}"
            corr = np.ascontiguousarray(corr_r.numpy().astype(np.int64))
            g[key + "/corr_sha"] = sha(corr)
            g[key + "/raw_mask_sha"] = sha(np.packbits(dbg["raw_mask"] != 0))
            g[key + "/cleaned_sha"] = sha(np.packbits(dbg["cleaned"] != 0))
            g[key + "/vis_sha"] = sha(np.packbits(dbg["vis"][512 * 512:] != 0))      # the foreground points (the background grid comes first)
            g[key + "/n_corr"] = np.int64(corr.shape[0])
            g[key + "/n_inpaint"] = np.int64(int(dbg["inpaint"].sum()))
            g[key + "/disp_slice"] = disp_r[0, 0].numpy()[::17, ::19].copy()
            g[key + "/disp_sum"] = np.float64(disp_r.double().sum().item())
            table.append(dict(scene=scene, edit=name, angle=ang, translation=tr, mask_px=int(mask.sum().item()),
                              n_corr=int(corr.shape[0]), n_inpaint=int(dbg["inpaint"].sum())))
            print(f"  {key}: angle {ang} t {tr}: N_corr={corr.shape[0]} inpaint={int(dbg['inpaint'].sum())} ({time.time() - t0:.1f}s)",
                  flush=True)
    np.savez_compressed(os.path.join(OUT, "g16_corpus.npz"), **g)
    with open(os.path.join(OUT, "g16_corpus.json"), "w") as f:
        json.dump(table, f, indent=0)
    print(f"g16_corpus.npz ok: {len(table)} edits in {time.time() - t00:.0f}s")


if __name__ == "__main__":
    main()
