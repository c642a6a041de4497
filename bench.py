#!/usr/bin/env python3
"""Headline benchmark: guided-denoise steps/sec at 512x512 (SD2-depth), whole job over N GPUs.

One "step" = one iteration of the guided-inference loop in its guided phase
(reference guided_stable_diffuser.py:377-479): 3 x {U-Net forward B=1 with activation capture,
guidance energy + gradient, backward-to-latent, latent update} + the CFG U-Net forward (B=2)
+ the DDIM step.  Workload = BASELINE.json configs[1] (single 512x512 edit, SD2-depth fp16):
synthetic scene, seeded random weights of the exact architecture, the per-image identity
(original activations, null-text list, initial noise) resident in HBM before the timed region.
N > 1: one process per GPU, each an independent edit, no collective on the data path
(weak scaling); torch.distributed is used only for the timing barrier and the MAX reduction.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STEP_TFLOP = 6.99          # algorithmic TFLOP of one guided-denoise step at 512^2
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra steps run with the HIP-event GEMM bracket")
    ap.add_argument("--res", type=int, default=512, help="image resolution (512 = headline; 768 = BASELINE config 5)")
    ap.add_argument("--no-time-edit", dest="time_edit", action="store_false", help="skip the whole-edit timing")
    ap.add_argument("--batch-edits", type=int, default=0, help="also time K edits of one image in one U-Net batch (config 3)")
    return ap.parse_args()


def cpu_baseline():
    """The oracle (torch fp32 CPU restatement of the same U-Net, kind 'port') timed on this host's cores on a bounded
    sample of the step (about 10-20 s of CPU work): one cold forward, one warm forward (B=1) and one warm
    forward + backward-to-input (B=1, gradient of the two guided activations), at the full SD-2-depth size.  A guided
    step is 3 x (forward + backward, B=1) + one CFG forward at B=2, timed as two B=1 forwards; the energy and the
    elementwise updates are left out (they favour the CPU figure)."""
    from oracle import unet_torch as U
    threads = min(os.cpu_count() or 1, 32)      # more threads than this oversubscribes torch's CPU kernels
    torch.set_num_threads(threads)
    unet = U.UNetTorch(U.SD2_DEPTH).eval()      # default torch init: values do not matter for timing
    for p_ in unet.parameters():
        p_.requires_grad_(False)
    g = torch.Generator().manual_seed(1)
    cond = torch.randn(1, 77, 1024, generator=g)
    x = torch.randn(1, 5, 64, 64, generator=g)
    times = []
    with torch.no_grad():
        for _ in range(2):
            t0 = time.time()
            unet(x, torch.tensor(940), encoder_hidden_states=cond, return_dict=False)
            times.append(time.time() - t0)
    t_fwd = times[-1]
    t_fb, note = None, ""
    try:
        xg = x.clone().requires_grad_(True)
        t0 = time.time()
        out = unet(xg, torch.tensor(940), encoder_hidden_states=cond, return_dict=False)
        (out[5].float().sum() + out[6].float().sum()).backward()
        t_fb = time.time() - t0
    except Exception as exc:              # noqa: BLE001 - e.g. host memory: fall back to the FLOP-scaled forward
        note = f"; forward+backward not timed ({type(exc).__name__}), step scaled from the forward by 6.99/0.804 TFLOP"
    step_s = 3.0 * t_fb + 2.0 * t_fwd if t_fb is not None else t_fwd * STEP_TFLOP / 0.804
    fb = f", forward+backward-to-input B=1 = {t_fb:.2f}s" if t_fb is not None else ""
    return {"value": 1.0 / step_s, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": f"oracle torch-CPU fp32 full SD2-depth U-Net: forward B=1 = {t_fwd:.2f}s warm ({times[0]:.2f}s cold){fb}; "
                      f"step = 3 x (fwd+bwd) + CFG forward at B=2 (2 x fwd) = {step_s:.1f}s{note}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # one rank per GPU; DH_BENCH_BACKEND=gloo (+ ranks folded onto the visible devices) exists only so that the
    # multi-rank control flow can be exercised on a single-GPU box
    backend = os.environ.get("DH_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from diffusionhandles_amd import DiffusionHandles, _lib
    from diffusionhandles_amd import conf as C
    from diffusionhandles_amd.depth_transform import normalize_depth, transform_depth
    from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene

    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    conf = C.load_default()
    from diffusionhandles_amd.unet import SD2_DEPTH
    lat = args.res // 8
    ucfg = dict(SD2_DEPTH, sample_size=lat)
    dh = DiffusionHandles(conf, dtype=dtype, unet_config=ucfg, max_batch=max(2, 2 * args.batch_edits)).to(dev)
    gd = dh.diffuser
    depth, bg_depth, mask = (t.to(dev) for t in make_scene(args.res))
    prompt = "a sphere on a plane"
    disparity = normalize_depth(1.0 / depth)
    T = conf.guided_diffuser.num_timesteps
    uncond = gd._encode([""])[None].expand(T, -1, -1, -1).contiguous()
    torch.manual_seed(conf.guided_diffuser.seed)
    noise = torch.randn(1, 4, lat, lat).to(dev)
    # per-image identity, resident in HBM before the timed region
    acts, _, _, init_noise = gd.initial_inference(noise, disparity, uncond, prompt)
    ang, tr = TRANSFORMS[2 + rank % 4]
    disp_e, corr = transform_depth(depth, bg_depth, mask, gd.get_depth_intrinsics(), rot_angle=ang,
                                   rot_axis=torch.tensor([0.0, 1.0, 0.0]), translation=torch.tensor(tr))
    st = gd.prepare_guidance(disp_e, prompt, acts, corr)
    gd.scheduler.set_timesteps(T)
    timesteps = gd.scheduler.timesteps
    x0 = init_noise.to(dev, torch.float32).permute(0, 2, 3, 1).contiguous()
    gmax = conf.guided_diffuser.guidance_max_step

    state = {"x": x0, "i": 0}

    def one_step():
        t_idx = state["i"] % gmax
        if t_idx == 0:
            state["x"] = x0
        state["x"] = gd.guided_step(st, state["x"], t_idx, timesteps[t_idx], uncond[t_idx])
        state["i"] += 1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad(), gd.on_stream():
        for _ in range(args.warmup):
            one_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        barrier()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(state["x"]).all(), "latents diverged"

    # roofline of the dominant kernel (k_gemm: MFMA implicit GEMM), HIP events on its stream
    import ctypes
    L = _lib.lib()
    roof = None
    if rank == 0:
        with torch.no_grad(), gd.on_stream():
            _lib.check(L.dh_gemm_profile_begin())
            for _ in range(max(1, args.profile_steps)):
                # the bracketed pass runs eagerly (event records cannot be timed inside a captured graph); a device-side
                # spin lets the host enqueue the whole step first, so the brackets see back-to-back launches like the
                # timed (graph-replayed) region instead of host launch gaps
                torch.cuda._sleep(int(0.06 * 2.4e9))
                one_step()
            ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
            _lib.check(L.dh_gemm_profile_end(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
        ach = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        # HBM traffic per k_gemm launch: PMC counters cannot be read from inside the process, so this is the
        # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE measurement (separate passes, gfx950 x2 fetch
        # correction) of the same U-Net fwd+bwd launch mix; null when the workload differs from the measured one.
        traffic = None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_gemm_traffic.json")
        if os.path.exists(pmc) and args.res == 512 and args.dtype == "fp16":
            with open(pmc) as fh:
                traffic = round(json.load(fh)["traffic_bytes_per_launch"])
        roof = {"bound": "mfma", "kernel": "k_gemm (MFMA implicit GEMM: conv3x3 + linear, fwd + input-gradient)",
                "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic, "traffic_unit": "bytes/launch (offline PMC)", "launches_per_step": int(n.value // max(1, args.profile_steps)),
                "avg_launch_us": round(ms.value * 1e3 / max(1, n.value), 2),
                "flops_per_launch": round(fl.value / max(1, n.value) / 1e9, 3),
                "step_tflop_algorithmic": STEP_TFLOP,
                "step_frac_of_mfma_peak": round(world * args.steps / elapsed * STEP_TFLOP / MFMA_PEAK_TFLOPS / world, 4)}

    batch_info = None
    if rank == 0 and args.batch_edits > 1:
        from diffusionhandles_amd.depth_transform import reproject_edits
        K = args.batch_edits
        tfs = [(TRANSFORMS[i % 8][0], torch.tensor([0.0, 1.0, 0.0]), torch.tensor(TRANSFORMS[i % 8][1])) for i in range(K)]
        edits = reproject_edits(depth, bg_depth, mask, gd.get_depth_intrinsics(), tfs)
        sts = [gd.prepare_guidance(d, prompt, acts, c) for d, c in edits]
        xb = x0.expand(K, -1, -1, -1).contiguous()
        with torch.no_grad(), gd.on_stream():
            for i in range(2):
                gd.guided_step_batch(sts, xb, i, timesteps[i], uncond[i])
            torch.cuda.synchronize()
            tb = time.perf_counter()
            nb = max(3, args.steps // 4)
            for i in range(nb):
                xb2 = gd.guided_step_batch(sts, xb, i % gmax, timesteps[i % gmax], uncond[i % gmax])
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb
        batch_info = {"edits_in_batch": K, "ms_per_batched_step": round(tb / nb * 1e3, 2),
                      "edit_steps_per_s": round(K * nb / tb, 2),
                      "frac_of_mfma_peak": round(K * nb / tb * STEP_TFLOP / MFMA_PEAK_TFLOPS, 4)}

    # ---- secondary measurements (rank 0, after the timed region).  They never gate the headline line: a failure
    # here is reported in place of the numbers.
    def secondary():
        # HBM-bound pieces of the path: guidance energy fwd+bwd and the batched K=8 reprojection,
        # timed with events on the stream they are launched on; bytes are the algorithmic figures of BASELINE.md section 3
        hbm = None
        if rank == 0:
            from diffusionhandles_amd.depth_transform import reproject_edits
            HBM_PEAK = 8000.0

            def timed(fn, n=20, graph=False):
                for _ in range(3):
                    fn()
                if graph:       # replay through a captured graph: device time of the launches, no host gaps between them
                    try:
                        torch.cuda.synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
                            fn()
                        fn = g.replay
                        fn()
                    except Exception:
                        pass
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                e1.synchronize()
                return e0.elapsed_time(e1) / n * 1e-3

            with torch.no_grad(), gd.on_stream():
                hbm = []
                cur = [o[1] for o in st.orig]                    # another timestep's activations stand in for "current"
                for layers, tag in (((2,), "t%3==0: act2"), ((1,), "t%3==1: act1"), ((1, 2), "t%3==2: act1+act2")):
                    fgw, bgw = st.schedule(2 if len(layers) == 2 else (0 if layers == (2,) else 1), 0)
                    def run():
                        for k in layers:
                            gd._energy_grad(st, k, cur[k], 0, fgw[k], bgw[k])
                    sec = timed(run, graph=True)
                    nbytes = sum(3 * cur[k].numel() * cur[k].element_size() for k in layers)
                    hbm.append({"kernel": f"guidance energy fwd+bwd ({tag})", "bound": "hbm", "bytes": nbytes,
                                "us": round(sec * 1e6, 2), "achieved": round(nbytes / sec / 1e9, 1), "peak": HBM_PEAK,
                                "unit": "GB/s", "frac": round(nbytes / sec / 1e9 / HBM_PEAK, 4)})
                K = 8
                tfs8 = [(TRANSFORMS[i % 8][0], torch.tensor([0.0, 1.0, 0.0]), torch.tensor(TRANSFORMS[i % 8][1])) for i in range(K)]
                sec = timed(lambda: reproject_edits(depth, bg_depth, mask, gd.get_depth_intrinsics(), tfs8), n=5)
                per_edit = 9e6 * (args.res / 512.0) ** 2
                hbm.append({"kernel": "batched K=8 unproject -> SE(3) -> z-buffer -> index maps (whole reproject_edits call, "
                                      "host glue included)", "bound": "hbm", "bytes": int(K * per_edit),
                            "us": round(sec * 1e6, 1), "achieved": round(K * per_edit / sec / 1e9, 2), "peak": HBM_PEAK,
                            "unit": "GB/s", "frac": round(K * per_edit / sec / 1e9 / HBM_PEAK, 5)})

        # one whole edit (the other half of BASELINE.json's metric): transform_foreground = re-projection + 38 guided +
        # 12 unguided steps + decode, with the per-image identity (inversion, original activations) already cached
        edit_info = None
        if rank == 0 and args.time_edit:
            rot = dict(rot_angle=ang, rot_axis=torch.tensor([0.0, 1.0, 0.0]), translation=torch.tensor(tr))
            with torch.no_grad():
                dh.transform_foreground(depth, prompt, mask, bg_depth, uncond, init_noise, acts, **rot)
                torch.cuda.synchronize()
                te = time.perf_counter()
                dh.transform_foreground(depth, prompt, mask, bg_depth, uncond, init_noise, acts, **rot)
                torch.cuda.synchronize()
                te = time.perf_counter() - te
            edit_info = {"edits_per_s": round(1.0 / te, 4), "s_per_edit": round(te, 3),
                         "what": "transform_foreground: z-buffer + index maps + 38 guided + 12 unguided steps + decode "
                                 "(synthetic VAE stand-in), identity cached"}

        return hbm, edit_info

    hbm, edit_info = None, None
    if rank == 0:
        try:
            hbm, edit_info = secondary()
        except Exception as exc:          # noqa: BLE001 - report, do not lose the headline measurement
            hbm = {"error": f"{type(exc).__name__}: {exc}"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.res == 512:
        try:
            cpu = cpu_baseline()
        except Exception as exc:          # noqa: BLE001
            cpu = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        value = world * args.steps / elapsed
        out = {
            "metric": f"guided-denoise steps/sec at {args.res}x{args.res} (SD2-depth)",
            "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": f"single {args.res}x{args.res} edit per GPU, SD2-depth (865.7M params, seeded random weights), guided "
                                   "phase: 3 x (fwd + energy + bwd-to-latent) + CFG fwd (B=2) + DDIM step",
                       "resolution": args.res, "edits_per_gpu": 1, "correspondences": int(corr.shape[0]),
                       "parallelism": "independent edits, one process per GPU, no collectives"},
            "roofline": roof, "hbm_kernels": hbm, "cpu_baseline": cpu, "batched_edits": batch_info, "whole_edit": edit_info,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
