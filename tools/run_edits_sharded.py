#!/usr/bin/env python3
"""BASELINE config 4: N independent edits of one image sharded over the GPUs of a node, 8 per GPU as one batched pass,
no collective on the per-edit path.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
      tools/run_edits_sharded.py --edits 64 --out /tmp/edits
  python tools/run_edits_sharded.py --edits 8 --out /tmp/edits          # one GPU
  python tools/run_edits_sharded.py --gpus 8 --edits 64 --out /tmp/edits # starts its own 8 ranks (bench.launch_ranks)

One process per GPU.  Rank 0 computes the per-image identity once (initial inference; --invert adds the null-text
inversion of the input image) and hands it to the other ranks with one broadcast per tensor (parallel.broadcast_identity;
--recompute-identity makes every rank compute it instead: no inter-GPU traffic at all).  Every rank then runs its
round-robin share of the edits in batches of --batch (parallel.run_edits -> transform_foreground_batch); results are
gathered once at the end (control plane) and rank 0 writes <out>/edit_XXX.png, edit_XXX_disparity.png and report.json with
the whole-job edits/s (barrier-bracketed, MAX over ranks).  The launcher starts before anything touches the GPU.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_transforms(n):
    """n rigid transforms in the range of the reference's shipped transforms.json files (rotation about +y, -35..60 degrees;
    translations within [-1, 1] x [-0.3, 0.3] x [-0.5, 0.5])."""
    g = torch.Generator().manual_seed(64)
    out = []
    for i in range(n):
        ang = float(-35.0 + 95.0 * torch.rand(1, generator=g))
        tr = (torch.rand(3, generator=g) * 2 - 1) * torch.tensor([1.0, 0.3, 0.5])
        if i % 3 == 0:
            tr = torch.zeros(3)
        if i % 3 == 1:
            ang = 0.0
        out.append(dict(name=f"edit_{i:03d}", rot_angle=ang, rot_axis=torch.tensor([0.0, 1.0, 0.0]), translation=tr))
    return out


def edit_plan(n_edits, world, batch, streams):
    """Which edits every rank runs, in which batches, on which lane: exactly what parallel.run_edits does with
    parallel.shard_edits (round-robin over ranks, then chunks of `batch` in order, chunk i on lane i % streams).  Pure
    arithmetic: the --dry-run-launch output and its CPU test (config 4: 64 edits on 8 GPUs = 8 x one batch of 8)."""
    from diffusionhandles_amd import parallel
    plan = []
    for r in range(world):
        mine = parallel.shard_edits(list(range(n_edits)), r, world)
        b = max(1, batch)
        chunks = [mine[i:i + b] for i in range(0, len(mine), b)]
        plan.append({"rank": r, "edits": mine, "batches": [{"lane": i % max(1, streams), "edits": c} for i, c in enumerate(chunks)]})
    return plan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edits", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--out", default="edits_out")
    ap.add_argument("--scene", default=None, help="scene directory (reference layout); default: the synthetic scene")
    ap.add_argument("--invert", action="store_true", help="null-text inversion of the input image for the identity")
    ap.add_argument("--recompute-identity", action="store_true", help="every rank computes the identity (no broadcast)")
    ap.add_argument("--no-images", action="store_true", help="do not write PNGs (timing runs)")
    ap.add_argument("--streams", type=int, default=1,
                    help="concurrent edit lanes per GPU process (engine arenas + streams on one copy of the weights): a rank's "
                         "batches of --batch edits run on this many lanes; images are bit-identical to --streams 1")
    ap.add_argument("--gpus", type=int, default=1, help="without a launcher in front (RANK unset): start this many ranks")
    ap.add_argument("--launch-timeout", type=float, default=7200.0, help="--gpus N launcher: kill still-running ranks after this many seconds")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="print the launch plan (per-rank environment) and the edit plan (which edits, in which batches, on which "
                         "lane, on every rank) as one JSON line and exit")
    args = ap.parse_args()
    if "RANK" not in os.environ and (args.gpus > 1 or args.dry_run_launch):
        # one process per GPU, started before anything here touches the GPU (the same launcher as bench.py)
        import importlib.util
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("dh_bench", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        argv = [a for a in sys.argv[1:] if a != "--dry-run-launch"]
        raise SystemExit(bench.launch_ranks(args, argv, script=__file__, plan={"plan": edit_plan(args.edits, args.gpus, args.batch, args.streams)}))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU over RCCL; DH_BENCH_BACKEND=gloo (+ ranks folded onto the visible devices) exists only so that the
    # multi-rank control flow can be exercised on a single-GPU box (tests/test_multirank_gpu.py), as in bench.py
    backend = os.environ.get("DH_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    from diffusionhandles_amd import DiffusionHandles, parallel
    from diffusionhandles_amd import conf as C
    from diffusionhandles_amd.scene_io import load_scene, transform_args, write_png
    from diffusionhandles_amd.synthetic import make_image, make_scene
    from diffusionhandles_amd.unet import SD2_DEPTH
    conf = C.load_default()
    dh = DiffusionHandles(conf, dtype=torch.float16, unet_config=dict(SD2_DEPTH, sample_size=args.res // 8),
                          max_batch=max(2, 2 * args.batch), vae="sd-native").to(dev)
    if args.scene:
        sc = load_scene(args.scene, args.res)
        img, depth, bg_depth, mask, prompt = sc["img"], sc["depth"], sc["bg_depth"], sc["fg_mask"], sc["prompt"]
        edits = [dict(name=n, **transform_args(t)) for n, t in sc["transforms"].items()]
        edits = (edits * (args.edits // len(edits) + 1))[:args.edits]
    else:
        depth, bg_depth, mask = make_scene(args.res)
        img, prompt = make_image(args.res), "a sphere on a plane"
        edits = make_transforms(args.edits)
    img, depth, bg_depth, mask = (t.to(dev) for t in (img, depth, bg_depth, mask))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():       # untimed warm-up of the decoder
        lat = args.res // 8
        dh.diffuser.decode_latent_image(torch.zeros(min(args.batch, -(-len(edits) // world)), 4, lat, lat, device=dev))
    # ---- per-image identity: once per image ------------------------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    identity = None
    if rank == 0 or args.recompute_identity:
        null_text, noise = dh.invert_input_image(img, depth, prompt) if args.invert else (None, None)
        null_text, noise, acts, _ = dh.generate_input_image(depth, prompt, null_text, noise)
        identity = (null_text.contiguous(), noise, acts)
    if not args.recompute_identity:
        identity = parallel.broadcast_identity(identity, src=0, device=dev)
    barrier()
    t_identity = time.perf_counter() - t0
    # ---- the edits: this rank's share, `batch` at a time ------------------------------------------------------------
    t0 = time.perf_counter()
    local = parallel.run_edits(dh, identity, edits, depth, mask, bg_depth, prompt, batch=args.batch, streams=args.streams)
    torch.cuda.synchronize()
    t_own = time.perf_counter() - t0                  # this rank's own share, before it waits for the others
    barrier()
    t_edits = time.perf_counter() - t0
    per_rank = [t_own]
    if world > 1:
        wire = dev if backend == "nccl" else "cpu"
        tt = torch.tensor([t_edits], dtype=torch.float64, device=wire)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_edits = float(tt.item())
        mine = torch.tensor([t_own], dtype=torch.float64, device=wire)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        per_rank = [float(g.item()) for g in got]
    results = parallel.gather_results([(gi, None if args.no_images else im, None if args.no_images else dp) for gi, im, dp in local])
    if rank == 0:
        os.makedirs(args.out, exist_ok=True)
        assert sorted(gi for gi, _, _ in results) == list(range(len(edits)))
        if not args.no_images:
            for gi, im, dp in results:
                write_png(os.path.join(args.out, f"{edits[gi]['name']}.png"), (im.permute(1, 2, 0).clamp(0, 1) * 255).round().byte().numpy())
                write_png(os.path.join(args.out, f"{edits[gi]['name']}_disparity.png"), dp[0, 0].clamp(0, 255).round().byte().numpy())
        rep = {"edits": len(edits), "n_gpus": world, "edits_per_gpu": -(-len(edits) // world), "batch": args.batch,
               "concurrent_streams": args.streams,
               "resolution": args.res, "identity_s": round(t_identity, 3), "identity": ("inversion + " if args.invert else "") +
               "initial inference on " + ("every rank" if args.recompute_identity else "rank 0, broadcast"),
               "edits_s": round(t_edits, 3), "edits_per_s": round(len(edits) / t_edits, 4),
               "per_rank": {"edits": [len(parallel.shard_edits(list(range(len(edits))), r, world)) for r in range(world)],
                            "edits_s": [round(v, 3) for v in per_rank],
                            "what": "each rank's share and its own wall time for it (rank order); edits_per_s divides by the MAX incl. the barrier"},
               "edits_per_s_with_identity": round(len(edits) / (t_edits + t_identity), 4)}
        with open(os.path.join(args.out, "report.json"), "w") as f:
            json.dump(rep, f, indent=1)
        print(json.dumps(rep))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
