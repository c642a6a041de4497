#!/usr/bin/env python3
"""End-to-end edit harness: the build's counterpart of the reference's test/test_diffusion_handles.py
(invert -> reconstruct -> set_foreground -> transform_foreground per transform), on the synthetic scene or on a
scene directory, writing PNGs and the .npz identity cache with the reference's keys
(null_text_emb, init_noise, activations1..3, latent_image; test_diffusion_handles.py:106-113).

  python tools/run_edit.py --out /tmp/edit                       # synthetic sphere-on-plane scene
  python tools/run_edit.py --scene DIR --out /tmp/edit           # DIR laid out like the reference's test/data/<set>/<scene>:
                                                                 #   input.png, mask.png, depth.exr, bg_depth.exr (or .npy),
                                                                 #   prompt.txt, transforms.json {name: {translation,
                                                                 #   rotation_axis, rotation_angle}}
  python tools/run_edit.py --scene tests/golden/scene_banana_fruits --out /tmp/edit   # a scene of the reference's test data
PNG / OpenEXR are read by diffusionhandles_amd.scene_io (no imaging library offline).
Real weights: DIFFHANDLES_UNET_SAFETENSORS / DIFFHANDLES_VAE_SAFETENSORS / DIFFHANDLES_TEXT_ENCODER_DIR /
DIFFHANDLES_TOKENIZER_DIR (otherwise seeded random U-Net weights and the synthetic side modules).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=None)
    ap.add_argument("--out", default="edit_out")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--mode", default="pc", choices=["pc", "mesh"])
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--skip-inversion", action="store_true", help="generate the image from noise instead of inverting an input")
    ap.add_argument("--no-identity-cache", action="store_true",
                    help="neither read nor write the input-image identity cache (1 GB at 512x512)")
    ap.add_argument("--identity-cache", default=None,
                    help="path of the .npz identity cache (default <out>/identity.npz).  Like the reference's "
                         "--cache_input_image_identity (test_diffusion_handles.py:85-113): loaded when it exists, written otherwise")
    ap.add_argument("--skip-existing", action="store_true",
                    help="skip edits whose <name>.png exists, and the whole scene when all do (test_diffusion_handles.py:133-135, 216-225)")
    ap.add_argument("--max-edits", type=int, default=0, help="run only the first N transforms")
    args = ap.parse_args()
    from diffusionhandles_amd import DiffusionHandles
    from diffusionhandles_amd import conf as C
    from diffusionhandles_amd.scene_io import load_scene, transform_args, write_png
    from diffusionhandles_amd.synthetic import TRANSFORMS, make_image, make_scene
    from diffusionhandles_amd.unet import SD2_DEPTH
    os.makedirs(args.out, exist_ok=True)
    dev = torch.device("cuda:0")
    conf = C.load_default()
    conf.depth_transform_mode = args.mode
    if args.scene:
        sc = load_scene(args.scene, args.res)
        img, depth, bg_depth, mask, prompt, res = sc["img"], sc["depth"], sc["bg_depth"], sc["fg_mask"], sc["prompt"], args.res
        transforms = [dict(name=n, **transform_args(t)) for n, t in sc["transforms"].items()]
    else:
        res = args.res
        depth, bg_depth, mask = make_scene(res)
        img = make_image(res)
        prompt = "a sphere on a plane"
        transforms = [dict(name=f"edit{i}", rot_angle=float(TRANSFORMS[i][0]), rot_axis=torch.tensor([0.0, 1.0, 0.0]),
                           translation=torch.tensor(TRANSFORMS[i][1], dtype=torch.float32)) for i in (2, 4)]
    if args.max_edits > 0:
        transforms = transforms[:args.max_edits]
    for i, tf in enumerate(transforms):
        tf.setdefault("name", f"edit{i}")
    exists = {tf["name"]: os.path.exists(os.path.join(args.out, tf["name"] + ".png")) for tf in transforms}
    if args.skip_existing and transforms and all(exists.values()):
        report = dict(resolution=res, mode=args.mode, skipped_scene=True, edits=[dict(name=n, skipped=True) for n in exists])
        print(json.dumps(report))
        return
    dh = DiffusionHandles(conf, dtype=torch.float16 if args.dtype == "fp16" else torch.bfloat16,
                          unet_config=dict(SD2_DEPTH, sample_size=res // 8)).to(dev)
    depth, bg_depth, mask, img = depth.to(dev), bg_depth.to(dev), mask.to(dev), img.to(dev)
    t0 = time.time()
    cache = None if args.no_identity_cache else (args.identity_cache or os.path.join(args.out, "identity.npz"))
    identity_from_cache = cache is not None and os.path.exists(cache)
    if identity_from_cache:
        # the input-image identity as the reference caches it (and as its web services pass it around,
        # webapp/webapps/diffhandles_webapp.py:82-94): float32 arrays under the reference's keys
        with np.load(cache) as z:
            null_text = torch.from_numpy(z["null_text_emb"]).to(dev)
            noise = torch.from_numpy(z["init_noise"]).to(dev)
            acts = [torch.from_numpy(z[f"activations{i + 1}"]).to(dev) for i in range(3)]
            latent = torch.from_numpy(z["latent_image"]).to(dev)
    else:
        null_text, noise = (None, None)
        if not args.skip_inversion:
            null_text, noise = dh.invert_input_image(img, depth, prompt)
        null_text, noise, acts, latent = dh.generate_input_image(depth, prompt, null_text, noise)
        if cache is not None:
            os.makedirs(os.path.dirname(os.path.abspath(cache)), exist_ok=True)
            np.savez(cache, null_text_emb=null_text.float().cpu().numpy(),
                     init_noise=noise.float().cpu().numpy(), activations1=acts[0].float().cpu().numpy(),
                     activations2=acts[1].float().cpu().numpy(), activations3=acts[2].float().cpu().numpy(),
                     latent_image=latent.float().cpu().numpy())
    bg_depth = dh.set_foreground(depth, mask, bg_depth)
    torch.cuda.synchronize()
    t_identity = time.time() - t0
    recon = dh.diffuser.decode_latent_image(latent)
    write_png(os.path.join(args.out, "recon.png"), recon[0].permute(1, 2, 0).float().cpu().numpy())
    report = dict(resolution=res, mode=args.mode, identity_s=round(t_identity, 2), identity_from_cache=bool(identity_from_cache),
                  edits=[])
    for tf in transforms:
        if args.skip_existing and exists[tf["name"]]:
            report["edits"].append(dict(name=tf["name"], skipped=True))
            continue
        t0 = time.time()
        out = dh.transform_foreground(depth, prompt, mask, bg_depth, null_text, noise, acts, rot_angle=tf["rot_angle"],
                                      rot_axis=tf["rot_axis"], translation=tf["translation"])
        torch.cuda.synchronize()
        dt = time.time() - t0
        edited, disparity = out[0], out[1]
        name = tf["name"]
        write_png(os.path.join(args.out, f"{name}.png"), edited[0].permute(1, 2, 0).float().cpu().numpy())
        write_png(os.path.join(args.out, f"{name}_disparity.png"), (disparity[0, 0] / disparity.max()).float().cpu().numpy())
        report["edits"].append(dict(name=name, seconds=round(dt, 3)))
    json.dump(report, open(os.path.join(args.out, "report.json"), "w"), indent=1)
    # the results page of the reference's harness (test/generate_results_webpage.py: one row per edit with input, mask,
    # depth, background depth, reconstruction, edit, edited disparity), written without a template engine
    norm = lambda d: ((d - d.min()) / (d.max() - d.min() + 1e-12)).float().cpu().numpy()
    write_png(os.path.join(args.out, "input.png"), img[0].permute(1, 2, 0).float().cpu().numpy())
    write_png(os.path.join(args.out, "mask.png"), mask[0, 0].float().cpu().numpy())
    write_png(os.path.join(args.out, "depth.png"), norm(1.0 / depth[0, 0]))
    write_png(os.path.join(args.out, "bg_depth.png"), norm(1.0 / bg_depth[0, 0]))
    cols = ["input", "mask", "depth", "bg_depth", "recon"]
    rows = []
    for e in report["edits"]:
        e.setdefault("seconds", "skipped")
        cells = "".join(f'<td><img src="{c}.png" width="192"></td>' for c in cols)
        cells += f'<td><img src="{e["name"]}.png" width="192"></td><td><img src="{e["name"]}_disparity.png" width="192"></td>'
        rows.append(f'<tr><th>{e["name"]}<br>{e["seconds"]} s</th>{cells}</tr>')
    head = "".join(f"<th>{c}</th>" for c in ["edit"] + cols + ["edited image", "edited disparity"])
    with open(os.path.join(args.out, "summary.html"), "w") as f:
        f.write(f"<!doctype html><html><head><meta charset='utf-8'><title>{prompt}</title></head><body><h3>{prompt} "
                f"({res}x{res}, {args.mode})</h3><table border='1' cellspacing='0' cellpadding='4'><tr>{head}</tr>{''.join(rows)}</table></body></html>")
    print(json.dumps(report))


if __name__ == "__main__":
    main()
