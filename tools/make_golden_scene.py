#!/usr/bin/env python3
"""Golden vectors on a REAL scene of the reference's test data (tests/golden/scene_banana_fruits, a copy of the data
files test/data/photogen/banana_fruits/{input.png,mask.png,depth.exr,bg_depth.exr,prompt.txt,transforms.json}).

The scene is read with diffusionhandles_amd.scene_io (imageio / OpenEXR are not installed), the REFERENCE's own
depth_transform functions (imported from /root/reference as in make_golden.py) are run on it for the scene's three
transforms, the oracle is cross-checked against them on the spot, and the expected integer maps go to
tests/golden/g11_scene.npz.  Only runs in the build container.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402
from make_golden import D, OUT, sha  # noqa: E402

from diffusionhandles_amd import scene_io as S  # noqa: E402


def main(scene="scene_banana_fruits", out_name="g11_scene.npz"):
    MG.install_stubs()
    import diffhandles.depth_transform as RD
    import diffhandles.guided_stable_diffuser as RG
    K = RG.GuidedStableDiffuser.get_depth_intrinsics()
    sc = S.load_scene(os.path.join(OUT, scene), 512)
    depth, bg_depth, mask = sc["depth"], sc["bg_depth"], sc["fg_mask"]
    res = 512
    g = dict(depth_sha=sha(depth.numpy()), bg_depth_sha=sha(bg_depth.numpy()), mask_bits=np.packbits(mask.numpy() != 0),
             img_sha=sha(sc["img"].numpy()), depth_slice=depth[0, 0].numpy()[::37, ::41].copy())
    ref_pts = RD.depth_to_world_coords(depth, K).numpy()
    assert np.array_equal(ref_pts, D.unproject(depth[0, 0].numpy(), K))
    ref_bg = RD.depth_to_world_coords(bg_depth, K).numpy()
    m = mask[0, 0].numpy().astype(bool)
    for name, t in sc["transforms"].items():
        ang, tr, axis = float(t["rotation_angle"]), [float(v) for v in t["translation"]], t["rotation_axis"]
        ref_rot, _ = RD.transform_point_cloud(ref_pts, np.array(axis, np.float32), ang, tr[0], tr[1], tr[2], m)
        assert np.array_equal(ref_rot, D.rigid_transform(ref_pts, np.array(axis, np.float32), ang, tr, m)), name
        allp = np.vstack([ref_bg.reshape(-1, 3), ref_rot.reshape(-1, 3)[m.reshape(-1)]])
        flags = np.zeros(allp.shape[0], np.uint8)
        flags[res * res:] = 1
        zr, mr, ur, vr, visr = RD.points_to_depth(torch.from_numpy(allp), K, (res, res), point_mask=torch.from_numpy(flags))
        zo, mo, uo, vo, viso = D.zbuffer(allp, flags, K, (res, res))
        assert np.array_equal(zr[0, 0].numpy(), zo) and np.array_equal(mr, mo), name
        assert np.array_equal(ur, uo) and np.array_equal(vr, vo) and np.array_equal(visr, viso), name
        disp_r, corr_r = RD.transform_depth_pc(depth, bg_depth, mask, K, rot_angle=ang, rot_axis=torch.tensor(axis),
                                               translation=torch.tensor(tr))
        disp_o, corr_o, dbg = D.transform_depth_pc(depth, bg_depth, mask, K, rot_angle=ang, rot_axis=axis, translation=tr,
                                                   return_debug=True)
        assert torch.equal(corr_r, corr_o), name
        assert torch.allclose(disp_r, disp_o, atol=1e-4, rtol=0), name
        g[f"{name}_zmap_sha"] = sha(zo)
        g[f"{name}_zmap_slice"] = zo[::37, ::41].copy()
        g[f"{name}_mask"] = np.packbits(mo)
        g[f"{name}_vis"] = np.packbits(viso[res * res:])
        g[f"{name}_corr"] = corr_r.numpy().astype(np.int16)
        g[f"{name}_cleaned"] = np.packbits(dbg["cleaned"] != 0)
        g[f"{name}_disp_slice"] = disp_r[0, 0].numpy()[::5, ::7].copy()
        g[f"{name}_disp_sum"] = np.float64(disp_r.double().sum().item())
        print(f"  {name}: angle {ang} t {tr}: N_vis={int(viso.sum())} N_corr={corr_r.shape[0]} inpaint={int(dbg['inpaint'].sum())}")
    np.savez_compressed(os.path.join(OUT, out_name), **g)
    print(out_name, "ok")


if __name__ == "__main__":
    # g11: test/data/photogen/banana_fruits; g15 (round 5): test/data/photogen/dice -- a second pair of estimated depth maps, so
    # that the EXR reader and the integer maps are pinned on two scenes (copies of the reference's DATA files under tests/golden/)
    if len(sys.argv) > 1:
        main(sys.argv[1], sys.argv[2])
    else:
        main()
        main("scene_dice", "g15_scene_dice.npz")
