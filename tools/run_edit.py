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


def load_config(path):
    """--config PATH.yaml (test_diffusion_handles.py:47, --config_path): a configuration file of the reference's layout
    (test/config/*.yaml: the 13 guided_diffuser keys + depth_transform_mode).  Keys the file does not name keep the
    defaults of config/default.yaml; unknown keys are an error (a typo must not silently run the default edit)."""
    from diffusionhandles_amd import conf as C
    conf = C.load_default()
    if path is None:
        return conf
    over = C.load(path) or {}
    for k, v in over.items():
        if k == "guided_diffuser":
            for kk, vv in (v or {}).items():
                if kk not in conf.guided_diffuser:
                    raise ValueError(f"{path}: unknown key guided_diffuser.{kk}")
                conf.guided_diffuser[kk] = vv
        elif k in conf:
            conf[k] = v
        else:
            raise ValueError(f"{path}: unknown key {k}")
    return conf


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=None)
    ap.add_argument("--out", default="edit_out")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--mode", default=None, choices=["pc", "mesh"], help="depth_transform_mode (default: the configuration's)")
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
    # the reference harness's outer loop (test_diffusion_handles.py:302-323, 42-75)
    ap.add_argument("--test-set", default=None,
                    help="JSON {scene name: [transform names]} (the reference's data/photogen/photogen.json): every scene is read "
                         "from <input-dir>/<scene> and written to <out>/<scene>")
    ap.add_argument("--input-dir", default=None, help="directory of the scene directories of --test-set (default: the JSON's directory)")
    ap.add_argument("--config", default=None, help="configuration YAML (the reference's --config_path, test/config/*.yaml)")
    ap.add_argument("--max-scenes", type=int, default=0, help="--test-set: only the first N scenes")
    return ap.parse_args()


def main():
    args = parse()
    conf = load_config(args.config)
    if args.mode is not None:
        conf.depth_transform_mode = args.mode
    args.mode = conf.depth_transform_mode
    state = {"dh": None}            # the engine is built on first use: a run that skips every scene never touches the GPU

    def handles(res):
        from diffusionhandles_amd import DiffusionHandles
        from diffusionhandles_amd.unet import SD2_DEPTH
        if state["dh"] is None:
            ucfg = dict(SD2_DEPTH, sample_size=res // 8)
            if not conf.guided_diffuser.use_depth:
                ucfg["in_channels"] = 4          # use_depth: false (test/config/no_depth.yaml): no depth channel beside the latent
            state["dh"] = DiffusionHandles(conf, dtype=torch.float16 if args.dtype == "fp16" else torch.bfloat16,
                                           unet_config=ucfg).to(torch.device("cuda:0"))
        return state["dh"]

    if args.test_set is None:
        print(json.dumps(run_scene(args, conf, handles, args.scene, args.out, None)))
        return
    # ---- the test set: one output directory per scene, the configuration saved beside them, one summary page ----------------
    import yaml
    with open(args.test_set) as f:
        from collections import OrderedDict
        dataset = json.load(f, object_pairs_hook=OrderedDict)
    input_dir = args.input_dir or os.path.dirname(os.path.abspath(args.test_set))
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, "config.yaml"), "w") as f:            # (test_diffusion_handles.py:52-55)
        yaml.safe_dump(json.loads(json.dumps(conf)), f, sort_keys=False)
    names = list(dataset.items())
    if args.max_scenes > 0:
        names = names[:args.max_scenes]
    reports = []
    for idx, (scene, transform_names) in enumerate(names):
        sys.stderr.write(f"[{idx + 1}/{len(names)}] {scene}: {len(transform_names)} transforms\n")
        sub = argparse.Namespace(**vars(args))
        # the identity cache of a scene lives in its own output directory unless the caller disabled it
        sub.identity_cache = None
        rep = run_scene(sub, conf, handles, os.path.join(input_dir, scene), os.path.join(args.out, scene), list(transform_names))
        rep["scene"] = scene
        reports.append(rep)
    set_name = os.path.splitext(os.path.basename(args.test_set))[0]
    rows = "".join(f'<tr><td><a href="{r["scene"]}/summary.html">{r["scene"]}</a></td><td>{len(r["edits"])}</td>'
                   f'<td>{"skipped" if r.get("skipped_scene") else r.get("identity_s", "")}</td></tr>' for r in reports)
    with open(os.path.join(args.out, f"{set_name}_summary.html"), "w") as f:
        f.write(f"<!doctype html><html><head><meta charset='utf-8'><title>{set_name}</title></head><body><h3>{set_name}: "
                f"{len(reports)} scenes</h3><table border='1' cellspacing='0' cellpadding='4'><tr><th>scene</th><th>edits</th>"
                f"<th>identity s</th></tr>{rows}</table></body></html>")
    total = dict(test_set=set_name, scenes=reports, config=args.config, depth_transform_mode=conf.depth_transform_mode,
                 edits_run=sum(1 for r in reports for e in r["edits"] if not e.get("skipped")),
                 edits_skipped=sum(1 for r in reports for e in r["edits"] if e.get("skipped")))
    json.dump(total, open(os.path.join(args.out, "report.json"), "w"), indent=1)
    print(json.dumps(total))


def run_scene(args, conf, handles, scene, out, transform_names):
    """One scene (the body of the reference's loop, test_diffusion_handles.py:66-175): identity (inversion + initial inference,
    or the cache), set_foreground, one transform_foreground per transform.  transform_names: the subset / order the test set
    lists for this scene (names the scene's transforms.json does not have are skipped with a warning, :126-128)."""
    from diffusionhandles_amd.scene_io import load_scene, transform_args, write_png
    from diffusionhandles_amd.synthetic import TRANSFORMS, make_image, make_scene
    os.makedirs(out, exist_ok=True)
    dev = torch.device("cuda:0")
    if scene:
        sc = load_scene(scene, args.res)
        img, depth, bg_depth, mask, prompt, res = sc["img"], sc["depth"], sc["bg_depth"], sc["fg_mask"], sc["prompt"], args.res
        transforms = [dict(name=n, **transform_args(t)) for n, t in sc["transforms"].items()]
    else:
        res = args.res
        depth, bg_depth, mask = make_scene(res)
        img = make_image(res)
        prompt = "a sphere on a plane"
        transforms = [dict(name=f"edit{i}", rot_angle=float(TRANSFORMS[i][0]), rot_axis=torch.tensor([0.0, 1.0, 0.0]),
                           translation=torch.tensor(TRANSFORMS[i][1], dtype=torch.float32)) for i in (2, 4)]
    if transform_names is not None:
        have = {tf["name"]: tf for tf in transforms}
        for n in transform_names:
            if n not in have:
                sys.stderr.write(f"WARNING: transform {n} not found for scene {scene}; skipping\n")
        transforms = [have[n] for n in transform_names if n in have]
    if args.max_edits > 0:
        transforms = transforms[:args.max_edits]
    for i, tf in enumerate(transforms):
        tf.setdefault("name", f"edit{i}")
    exists = {tf["name"]: os.path.exists(os.path.join(out, tf["name"] + ".png")) for tf in transforms}
    if args.skip_existing and transforms and all(exists.values()):
        return dict(resolution=res, mode=args.mode, skipped_scene=True, edits=[dict(name=n, skipped=True) for n in exists])
    dh = handles(res)
    depth, bg_depth, mask, img = depth.to(dev), bg_depth.to(dev), mask.to(dev), img.to(dev)
    t0 = time.time()
    cache = None if args.no_identity_cache else (args.identity_cache or os.path.join(out, "identity.npz"))
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
    write_png(os.path.join(out, "recon.png"), recon[0].permute(1, 2, 0).float().cpu().numpy())
    report = dict(resolution=res, mode=args.mode, identity_s=round(t_identity, 2), identity_from_cache=bool(identity_from_cache),
                  edits=[])
    for tf in transforms:
        if args.skip_existing and exists[tf["name"]]:
            report["edits"].append(dict(name=tf["name"], skipped=True))
            continue
        t0 = time.time()
        res_ = dh.transform_foreground(depth, prompt, mask, bg_depth, null_text, noise, acts, rot_angle=tf["rot_angle"],
                                      rot_axis=tf["rot_axis"], translation=tf["translation"])
        torch.cuda.synchronize()
        dt = time.time() - t0
        edited, disparity = res_[0], res_[1]
        name = tf["name"]
        write_png(os.path.join(out, f"{name}.png"), edited[0].permute(1, 2, 0).float().cpu().numpy())
        write_png(os.path.join(out, f"{name}_disparity.png"), (disparity[0, 0] / disparity.max()).float().cpu().numpy())
        report["edits"].append(dict(name=name, seconds=round(dt, 3)))
    json.dump(report, open(os.path.join(out, "report.json"), "w"), indent=1)
    # the results page of the reference's harness (test/generate_results_webpage.py: one row per edit with input, mask,
    # depth, background depth, reconstruction, edit, edited disparity), written without a template engine
    norm = lambda d: ((d - d.min()) / (d.max() - d.min() + 1e-12)).float().cpu().numpy()
    write_png(os.path.join(out, "input.png"), img[0].permute(1, 2, 0).float().cpu().numpy())
    write_png(os.path.join(out, "mask.png"), mask[0, 0].float().cpu().numpy())
    write_png(os.path.join(out, "depth.png"), norm(1.0 / depth[0, 0]))
    write_png(os.path.join(out, "bg_depth.png"), norm(1.0 / bg_depth[0, 0]))
    cols = ["input", "mask", "depth", "bg_depth", "recon"]
    rows = []
    for e in report["edits"]:
        e.setdefault("seconds", "skipped")
        cells = "".join(f'<td><img src="{c}.png" width="192"></td>' for c in cols)
        cells += f'<td><img src="{e["name"]}.png" width="192"></td><td><img src="{e["name"]}_disparity.png" width="192"></td>'
        rows.append(f'<tr><th>{e["name"]}<br>{e["seconds"]} s</th>{cells}</tr>')
    head = "".join(f"<th>{c}</th>" for c in ["edit"] + cols + ["edited image", "edited disparity"])
    with open(os.path.join(out, "summary.html"), "w") as f:
        f.write(f"<!doctype html><html><head><meta charset='utf-8'><title>{prompt}</title></head><body><h3>{prompt} "
                f"({res}x{res}, {args.mode})</h3><table border='1' cellspacing='0' cellpadding='4'><tr>{head}</tr>{''.join(rows)}</table></body></html>")
    return report


if __name__ == "__main__":
    main()
