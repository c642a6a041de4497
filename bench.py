#!/usr/bin/env python3
"""Headline benchmark: guided-denoise steps/sec at 512x512 (SD2-depth), whole job over N GPUs.

One "step" = one iteration of the guided-inference loop in its guided phase
(reference guided_stable_diffuser.py:377-479): 3 x {U-Net forward B=1 with activation capture,
guidance energy + gradient, backward-to-latent, latent update} + the CFG U-Net forward (B=2)
+ the DDIM step.  Workload = BASELINE.json configs[1] (single 512x512 edit, SD2-depth fp16):
synthetic scene, seeded random weights of the exact architecture, the per-image identity
(original activations, null-text list, initial noise) resident in HBM before the timed region.
N > 1: one process per GPU, each an independent edit, no collective on the data path
(weak scaling); torch.distributed is used only for the timing barriers and the MAX reductions.
`python bench.py --gpus N` with no launcher in front starts the N ranks itself (launch_ranks: the parent never touches
the GPU; --launch-timeout bounds the job); under `python -m torch.distributed.run ... bench.py --gpus N` the ranks are used as
given.  Profiling: put a profiler in front of a `--gpus 1` run (or of one rank) only -- `rocprofv3 ... -- python3 bench.py --gpus N`
would make the profiler-initialised parent spawn the ranks.

Next to the headline the same run reports (rank 0 unless stated; none of it inside the timed region):
  roofline        dominant kernel (k_gemm_dma) by HIP events on its stream, + committed PMC traffic
  hbm_kernels     guidance energy and batched re-projection, achieved GB/s
  phases          BASELINE config 2 "with and without the per-image phase": null-text inversion (50 timesteps x <= 5
                  inner Adam steps), initial inference, one whole edit (re-projection + 38 guided + 12 unguided steps +
                  AutoencoderKL decode), edits/s with the identity cached and including it
  batched_edits   BASELINE config 3: K = 8 edits of one image per U-Net batch
  edits           BASELINE config 4's unit on every rank: 8 edits per GPU as one batch, whole-job edits/s (MAX over ranks)
  res768_bf16     BASELINE config 5: 768x768, bf16 U-Net, f32 guidance energy / cotangent / latent update
  cpu_baseline    the oracle on the host cores: U-Net step, and per piece z-buffer / cells / energy

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STEP_TFLOP = {512: 6.99, 768: 20.0}   # algorithmic TFLOP of one guided-denoise step
MFMA_PEAK_TFLOPS = 2500.0              # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
HBM_PEAK = 8000.0


GEMM_SOURCES = ("gemm.hip", "gemm_pp.hip", "gemm_k.h", "unet_kernels.h", "unet_engine.cpp")


def gemm_sources_sha(root=ROOT):
    """SHA-256 over the sources that decide which GEMM launches a pass makes and what they move (tile policy, split-K slabs,
    staging): a committed PMC traffic figure is only reported next to a live measurement when it was collected on THIS code
    (tools/pmc_summarise.py stamps the same hash into profiles/rNN_pmc_gemm_traffic.json)."""
    import hashlib
    h = hashlib.sha256()
    for name in GEMM_SOURCES:
        with open(os.path.join(root, "diffusionhandles_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()


def committed_traffic(root=ROOT, res=512, dtype="fp16"):
    """(bytes per launch, file name, note) of the newest profiles/rNN_pmc_gemm_traffic.json -- or (None, name, why) when it was
    collected on other GEMM sources than the ones in the tree (stale), or for another workload."""
    import glob
    files = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_pmc_gemm_traffic.json")), reverse=True)
    if not files or res != 512 or dtype != "fp16":
        return None, None, "no PMC collection for this workload"
    name = os.path.basename(files[0])
    with open(files[0]) as fh:
        rec = json.load(fh)
    if rec.get("sources_sha256") != gemm_sources_sha(root):
        return None, name, f"{name} was collected on other GEMM sources (sources_sha256 differs): stale, not reported"
    return round(rec["traffic_bytes_per_launch"]), name, "offline PMC (separate FETCH_SIZE / WRITE_SIZE passes) on these sources"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--res", type=int, default=512, help="image resolution of the headline (512 = BASELINE config 2)")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra steps run with the HIP-event GEMM bracket")
    ap.add_argument("--batch-edits", type=int, default=8, help="K edits of one image per U-Net batch (config 3 / 4); 0 = skip")
    ap.add_argument("--streams", type=int, default=2,
                    help="concurrent edit lanes per GPU for the edits.concurrent record (engine arenas + streams on one copy of "
                         "the weights; never the single-edit headline); <= 1 = skip.  Two since round 5: a third lane's stream can "
                         "land on the hardware queue of another lane (profiles/r05_lanes_wide.txt: 3 lanes 2.09, 2 lanes 2.27 edits/s; "
                         "round 4's GPU_MAX_HW_QUEUES A/B shows the same split), and with the round-5 GEMMs one stream already fills more of the chip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-phases", dest="phases", action="store_false", help="skip inversion / initial inference / whole-edit timing")
    ap.add_argument("--no-res768", dest="res768", action="store_false", help="skip the 768x768 bf16 record (config 5)")
    ap.add_argument("--no-time-edit", dest="phases", action="store_false", help=argparse.SUPPRESS)
    ap.add_argument("--launch-timeout", type=float, default=3600.0,
                    help="--gpus N launcher: seconds after which still-running ranks are killed and the job returns 124 (0 = no limit)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="print the per-rank environment / command the N-rank launcher would start (one JSON line) and exit")
    return ap.parse_args()


def rank_environments(n, port, base=None):
    """The environment of each of the n ranks the launcher starts: one process per GPU, rendezvous on 127.0.0.1."""
    envs = []
    for r in range(n):
        env = dict(os.environ if base is None else base)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        envs.append(env)
    return envs


def launch_ranks(args, argv, script=None, plan=None):
    """`python bench.py --gpus N` without a launcher in front (RANK unset): start N copies of this script, one rank per GPU
    (the reference's own multi-GPU shape is one process per device too, webapp/start_webapps_in_tmux.sh:21-43).  The parent
    never touches the GPU - no torch.cuda call, no library load - and nothing is exec'd: children are ordinary
    subprocesses, rank 0's stdout (the one JSON line) passes through, the worst child return code is returned."""
    import socket
    import subprocess
    cmd = [sys.executable, os.path.abspath(script or __file__)] + argv
    launch_timeout = getattr(args, "launch_timeout", 0) or 0
    # the rendezvous port is found by bind / close, so another process can take it before rank 0 listens: a job whose rank 0
    # (the rendezvous host) fails inside the first seconds is started again on a fresh port (twice at most)
    import threading
    worst = 0
    for attempt in range(3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        envs = rank_environments(args.gpus, port)
        if args.dry_run_launch:
            keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
            print(json.dumps(dict({"launch": [{"cmd": cmd, "env": {k: e[k] for k in keys}} for e in envs],
                                   "launch_timeout_s": launch_timeout}, **(plan or {}))))
            return 0
        started = time.time()
        # rank 0's stderr is passed through AND its tail kept: a retry needs evidence that the port was taken
        procs = [subprocess.Popen(cmd, env=e, stdout=None if r == 0 else subprocess.DEVNULL,
                                  stderr=subprocess.PIPE if r == 0 else None) for r, e in enumerate(envs)]
        rank0 = procs[0]
        tail = bytearray()

        def tee(pipe=rank0.stderr):
            for chunk in iter(lambda: pipe.read1(65536), b""):
                sys.stderr.buffer.write(chunk)
                sys.stderr.buffer.flush()
                tail.extend(chunk)
                del tail[:-65536]
        th = threading.Thread(target=tee, daemon=True)
        th.start()
        rcs = []
        deadline = started + launch_timeout if launch_timeout > 0 else None      # overall limit: a rank that hangs ends the job
        first_failure = None
        timed_out = False
        while procs:
            for p in list(procs):
                rc = p.poll()
                if rc is not None:
                    procs.remove(p)
                    rcs.append(rc)
                    if rc != 0 and first_failure is None:   # a rank died: the others wait at a barrier; give them a minute, then end them
                        first_failure = time.time()
                        if p is rank0 and port_clash(bytes(tail)):
                            deadline = first_failure         # the rendezvous never existed: nothing to wait for
                        else:
                            deadline = min(deadline, first_failure + 60) if deadline else first_failure + 60
            if deadline is not None and time.time() > deadline and procs:
                for p in procs:
                    p.kill()
                if first_failure is None and not timed_out:
                    timed_out = True
                    rcs.append(124)                          # timed out (the value `timeout` returns), once
            time.sleep(0.2)
        th.join(timeout=5)
        worst = max((abs(rc) for rc in rcs), default=0)
        if worst != 0 and rank0.returncode not in (0, None) and port_clash(bytes(tail)) and attempt < 2:
            continue                                         # rank 0 could not listen on the port found by bind / close: once more on another
        return worst
    return worst


def port_clash(stderr_tail):
    """Evidence in rank 0's stderr that the rendezvous port was taken between the parent's bind / close and rank 0's listen --
    the only failure the launcher starts the job again for (a bad argument, an import error or an OOM is final)."""
    t = stderr_tail.lower()
    return b"eaddrinuse" in t or b"address already in use" in t


class Sections:
    """Secondary measurements never cost the headline line, but they do not fail silently either: every exception is kept
    (section name + message), the JSON line carries them as a top-level "errors" list and the process exits non-zero AFTER
    printing the line when the list is not empty (finish)."""

    def __init__(self):
        self.errors = []

    def note(self, name, exc):
        self.errors.append({"section": name, "error": f"{type(exc).__name__}: {exc}"})
        return {"error": f"{type(exc).__name__}: {exc}"}

    def run(self, name, fn):
        """fn() or, when it raises, {"error": ...} with the failure recorded."""
        if os.environ.get("DH_BENCH_INJECT_FAIL") == name:      # test hook (tests/test_bench_errors.py, the GPU suite)
            return self.note(name, RuntimeError("injected failure (DH_BENCH_INJECT_FAIL)"))
        try:
            return fn()
        except Exception as exc:          # noqa: BLE001 - never lose the headline line to a secondary measurement
            return self.note(name, exc)

    def finish(self, out, fd):
        """Write the one JSON line (with "errors") to fd; returns the process exit code: 0, or 3 when a section failed."""
        out["errors"] = list(self.errors)
        os.write(fd, (json.dumps(out) + "\n").encode())
        return 3 if self.errors else 0


def cpu_baseline():
    """The oracle (kind 'port') timed on this host's cores on a bounded sample (about 20 s of CPU work): the torch fp32
    restatement of the full SD-2-depth U-Net (one cold + one warm forward, one forward + backward-to-input, B=1) scaled
    to a guided step (3 x (forward + backward) + the CFG forward at B=2 as two B=1 forwards), and per piece of the edit:
    z-buffered re-projection, correspondences -> cells, guidance energy + gradient (NumPy / torch-CPU oracles)."""
    from oracle import depth_ref as D
    from oracle import guidance_ref as G
    from oracle import unet_torch as U
    from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene
    host = os.cpu_count() or 1
    # 32 threads: measured on the 256-core GPU host, torch's CPU kernels at 256 threads take 136 s for the forward that
    # takes 2.3 s at 32 (oversubscribed intra-op pool); DH_CPU_BASELINE_THREADS overrides
    threads = int(os.environ.get("DH_CPU_BASELINE_THREADS", min(host, 32)))
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
        acts = [out[5][0].detach(), out[6][0].detach()]
    except Exception as exc:              # noqa: BLE001 - e.g. host memory: fall back to the FLOP-scaled forward
        note = f"; forward+backward not timed ({type(exc).__name__}), step scaled from the forward by 6.99/0.804 TFLOP"
        acts = [torch.randn(640, 64, 64), torch.randn(320, 64, 64)]
    del unet
    step_s = 3.0 * t_fb + 2.0 * t_fwd if t_fb is not None else t_fwd * STEP_TFLOP[512] / 0.804
    fb = f", forward+backward-to-input B=1 = {t_fb:.2f}s" if t_fb is not None else ""
    pieces = {}
    try:
        depth, bg, mask = make_scene(512)
        ang, tr = TRANSFORMS[2]
        t0 = time.time()
        _, corr = D.transform_depth_pc(depth, bg, mask, rot_angle=ang, rot_axis=[0, 1, 0], translation=tr)
        pieces["reproject_s"] = round(time.time() - t0, 3)
        t0 = time.time()
        cells = G.cells_from_correspondences(corr.numpy(), 512, 0)
        pieces["cells_s"] = round(time.time() - t0, 3)
        t0 = time.time()
        for a in acts:             # one energy evaluation of the t%3==2 phase: act1 and act2, foreground + background, with gradient
            cur = (a + 0.01 * torch.randn(a.shape)).requires_grad_(True)
            e = G.foreground_energy(cur, a, cells, 1, (64, 64)) + G.background_energy(cur, a, cells, 1, (64, 64), "global_avg")
            torch.autograd.grad(e, cur)
        pieces["energy_fwd_bwd_s"] = round(time.time() - t0, 3)
    except Exception as exc:              # noqa: BLE001
        pieces["error"] = f"{type(exc).__name__}: {exc}"
    return {"value": 1.0 / step_s, "unit": "steps/s", "cores": threads, "host_cores": host, "kind": "port",
            "sample": f"oracle torch-CPU fp32 full SD2-depth U-Net: forward B=1 = {t_fwd:.2f}s warm ({times[0]:.2f}s cold){fb}; "
                      f"step = 3 x (fwd+bwd) + CFG forward at B=2 (2 x fwd) = {step_s:.1f}s{note}",
            "pieces": dict(pieces, what="oracle on one 512x512 edit: z-buffered re-projection (NumPy), correspondences -> cells, "
                                        "energy + gradient of act1 and act2 (torch-CPU autograd)")}


def main():
    args = parse()
    if "RANK" not in os.environ and (args.gpus > 1 or args.dry_run_launch):
        argv = [a for a in sys.argv[1:] if a != "--dry-run-launch"]
        raise SystemExit(launch_ranks(args, argv))
    # rank 0's stdout carries ONLY the one JSON line, in every backend: whatever the libraries print to fd 1 while the job runs
    # (gloo's "[Gloo] Rank 0 is connected ..." banner, MIOpen notes) goes to stderr; the line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    global torch
    import torch
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
    from diffusionhandles_amd.depth_transform import normalize_depth, reproject_edits, transform_depth
    from diffusionhandles_amd.synthetic import TRANSFORMS, make_image, make_scene
    from diffusionhandles_amd.unet import SD2_DEPTH

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        tt = torch.tensor([seconds], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def per_rank(seconds):
        """every rank's own figure, in rank order (a slow rank stays visible next to the MAX the headline uses)"""
        if dist is None:
            return [seconds]
        mine = torch.tensor([seconds], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        return [float(g.item()) for g in got]

    K = max(0, args.batch_edits)
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    conf = C.load_default()
    lat = args.res // 8
    # the SD AutoencoderKL and the SD-2 CLIP text tower with random weights, both on the engine's kernels ("sd-native" /
    # "sd2-native"): the decode on an edit's critical path, the encode and the prompt embedding of the per-image phase are real ones
    dh = DiffusionHandles(conf, dtype=dtype, unet_config=dict(SD2_DEPTH, sample_size=lat), max_batch=max(2, 2 * K),
                          vae="sd-native", text_encoder="sd2-native").to(dev)
    gd = dh.diffuser
    depth, bg_depth, mask = (t.to(dev) for t in make_scene(args.res))
    prompt = "a sphere on a plane"
    disparity = normalize_depth(1.0 / depth)
    T = conf.guided_diffuser.num_timesteps
    gmax = conf.guided_diffuser.guidance_max_step
    Y = torch.tensor([0.0, 1.0, 0.0])

    # ---- the per-image phase (BASELINE config 2: "50 guided steps + null inversion"), timed on rank 0 --------------------
    phases = None
    uncond = gd._encode([""])[None].expand(T, -1, -1, -1).contiguous()
    torch.manual_seed(conf.guided_diffuser.seed)
    noise = torch.randn(1, 4, lat, lat).to(dev)
    # untimed warm-up of the decoder and of the engine's graphs (the PyTorch-ROCm encoder's first call lets MIOpen pick /
    # compile its convolution kernels: seconds that belong to no phase; it runs inside the inversion warm-up below)
    with torch.no_grad():
        gd.decode_latent_image(torch.zeros(1, 4, lat, lat, device=dev))
        if K > 1:
            gd.decode_latent_image(torch.zeros(K, 4, lat, lat, device=dev))
    if rank == 0 and args.phases:
        img = make_image(args.res).to(dev)
        dh.inverter.invert(img, disparity, prompt, num_inner_steps=5, max_timesteps=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        null_text, inv_noise = dh.invert_input_image(img, depth, prompt)
        torch.cuda.synchronize()
        phases = {"inversion_s": round(time.perf_counter() - t0, 3),
                  "inversion_inner_steps": int(sum(dh.inverter.inner_steps_taken)),
                  "inversion_what": "StableNullInverter.invert: VAE encode, 50 DDIM-inversion forwards, 50 timesteps x (cond forward + "
                                    "<= 5 x (forward + backward-to-text + Adam) + unconditional forward B=1 for the CFG step, "
                                    "which reuses the cond forward)"}
        del null_text, inv_noise
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acts, _, _, init_noise = gd.initial_inference(noise, disparity, uncond, prompt)
    torch.cuda.synchronize()
    if phases is not None:
        phases["initial_inference_s"] = round(time.perf_counter() - t0, 3)

    # ---- headline: guided-denoise steps of one edit per GPU ----------------------------------------------------------------
    ang, tr = TRANSFORMS[2 + rank % 4]
    disp_e, corr = transform_depth(depth, bg_depth, mask, gd.get_depth_intrinsics(), rot_angle=ang, rot_axis=Y,
                                   translation=torch.tensor(tr))
    st = gd.prepare_guidance(disp_e, prompt, acts, corr)
    gd.scheduler.set_timesteps(T)
    timesteps = gd.scheduler.timesteps
    x0 = init_noise.to(dev, torch.float32).permute(0, 2, 3, 1).contiguous()
    state = {"x": x0, "i": 0}

    def one_step():
        t_idx = state["i"] % gmax
        if t_idx == 0:
            state["x"] = x0
        state["x"] = gd.guided_step(st, state["x"], t_idx, timesteps[t_idx], uncond[t_idx])
        state["i"] += 1

    with torch.no_grad(), gd.on_stream():
        for _ in range(args.warmup):
            one_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        barrier()
        elapsed = time.perf_counter() - t0
    elapsed_by_rank = per_rank(elapsed)
    elapsed = max_over_ranks(elapsed)
    assert torch.isfinite(state["x"]).all(), "latents diverged"

    # ---- roofline of the dominant kernel (k_gemm_dma: MFMA implicit GEMM), HIP events on its stream ------------------------
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
            alg = ctypes.c_double()
            _lib.check(L.dh_gemm_profile_bytes(ctypes.byref(alg)))
        ach = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        # HBM traffic per k_gemm launch: PMC counters cannot be read from inside the process, so this is the
        # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE measurement (separate passes, gfx950 x2 fetch
        # correction) of the same U-Net fwd+bwd launch mix -- reported only when it was collected on the GEMM sources in the
        # tree (committed_traffic: a hash of csrc/gemm*.hip, gemm_k.h, unet_kernels.h, unet_engine.cpp), null otherwise.
        traffic, traffic_file, traffic_note = committed_traffic(ROOT, args.res, args.dtype)
        step_tf = STEP_TFLOP.get(args.res)
        nl = max(1, n.value)
        sec = ms.value * 1e-3
        alg_per = alg.value / nl
        gbps = alg.value / sec / 1e9 if sec > 0 else 0.0
        intensity = fl.value / alg.value if alg.value > 0 else 0.0          # algorithmic FLOP per algorithmic byte of the launch mix
        ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK * 1e9)                   # 312.5 FLOP/B: below it the HBM side of the roofline bounds the mix
        hbm_side = intensity < ridge
        roof = {"bound": "hbm" if hbm_side else "mfma",
                "kernel": "k_gemm_dma (+ k_gemm_pp for the GEGLU forward launches): MFMA implicit GEMM, conv3x3 + linear, fwd + input-gradient",
                "achieved": round(gbps, 1) if hbm_side else round(ach, 2), "peak": HBM_PEAK if hbm_side else MFMA_PEAK_TFLOPS,
                "unit": "GB/s" if hbm_side else "TFLOP/s",
                "frac": round(gbps / HBM_PEAK, 4) if hbm_side else round(ach / MFMA_PEAK_TFLOPS, 4),
                "bound_note": f"algorithmic intensity of the launch mix {intensity:.0f} FLOP/B against the ridge {ridge:.1f} FLOP/B (2.5 PFLOP/s / 8 TB/s): "
                              + ("the HBM side bounds it; " if hbm_side else "the MFMA side bounds it; ") + "both fractions are given (frac_mfma, frac_hbm)",
                "intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                "achieved_tflops": round(ach, 2), "frac_mfma": round(ach / MFMA_PEAK_TFLOPS, 4),
                "achieved_gbps_algorithmic": round(gbps, 1), "frac_hbm": round(gbps / HBM_PEAK, 4),
                "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_file, "traffic_note": traffic_note,
                "launches_per_step": int(n.value // max(1, args.profile_steps)),
                "algorithmic_bytes_per_launch": round(alg_per),
                "traffic_over_algorithmic": round(traffic / alg_per, 2) if traffic and alg.value > 0 else None,
                "frac_hbm_traffic": round(traffic * nl / sec / 1e9 / HBM_PEAK, 4) if traffic and sec > 0 else None,
                "algorithmic_bytes_note": "every operand once in 16-bit storage (A source, W, output, residual; include/diffhandles_hip.h "
                                          "dh_gemm_profile_bytes), live over the profiled guided steps; traffic = PMC of the U-Net fwd+bwd launch mix",
                "avg_launch_us": round(ms.value * 1e3 / nl, 2),
                "flops_per_launch": round(fl.value / nl / 1e9, 3),
                "step_tflop_algorithmic": step_tf,
                "step_frac_of_mfma_peak": round(args.steps / elapsed * step_tf / MFMA_PEAK_TFLOPS, 4) if step_tf else None,
                "step_frac_note": "algorithmic TFLOP of the reference's step (the truncated optimisation forwards execute less)"}

    # ---- batched edits: K transforms of one image in one U-Net batch (config 3), and config 4's unit on every rank -----------
    batch_info, edits_info = None, None
    def batched_section():
        nonlocal batch_info, edits_info
        tfs = [(TRANSFORMS[(i + rank) % 8][0], Y, torch.tensor(TRANSFORMS[(i + rank) % 8][1])) for i in range(K)]
        with torch.no_grad():
            # batched guided steps (every rank runs them: they also capture the batch-K graphs the whole-edit timing replays)
            edits = reproject_edits(depth, bg_depth, mask, gd.get_depth_intrinsics(), tfs, device_correspondences=True)
            sts = [gd.prepare_guidance(d, prompt, acts, c) for d, c in edits]
            xb = x0.expand(K, -1, -1, -1).contiguous()
            with gd.on_stream():
                for i in range(3):
                    gd.guided_step_batch(sts, xb, i, timesteps[i], uncond[i])
                gd.guided_step_batch(sts, xb, gmax, timesteps[gmax], uncond[gmax])       # an unguided step: CFG pass only
                torch.cuda.synchronize()
                tb = time.perf_counter()
                nb = max(3, args.steps // 4)
                for i in range(nb):
                    gd.guided_step_batch(sts, xb, i % gmax, timesteps[i % gmax], uncond[i % gmax])
                torch.cuda.synchronize()
                tb = time.perf_counter() - tb
            if rank == 0:
                batch_info = {"edits_in_batch": K, "ms_per_batched_step": round(tb / nb * 1e3, 2),
                              "edit_steps_per_s": round(K * nb / tb, 2),
                              "frac_of_mfma_peak": round(K * nb / tb * STEP_TFLOP[512] / MFMA_PEAK_TFLOPS, 4) if args.res == 512 else None}
                # roofline of the GEMM launches of a batched step (k_gemm_pp: the eight-wave ping-pong loop on the grids that fill the
                # chip; k_gemm_dma on the rest), HIP start / stop events per launch like the headline's record
                with gd.on_stream():
                    _lib.check(L.dh_gemm_profile_begin())
                    for i in range(3):
                        torch.cuda._sleep(int(0.1 * 2.4e9))
                        gd.guided_step_batch(sts, xb, i, timesteps[i], uncond[i])
                    bms, bn_, bfl, balg = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
                    _lib.check(L.dh_gemm_profile_end(ctypes.byref(bms), ctypes.byref(bn_), ctypes.byref(bfl)))
                    _lib.check(L.dh_gemm_profile_bytes(ctypes.byref(balg)))
                bach = bfl.value / (bms.value * 1e-3) / 1e12 if bms.value > 0 else 0.0
                batch_info["roofline"] = {
                    "bound": "mfma", "kernel": "k_gemm_pp + k_gemm_dma launches of three batched guided steps (t_idx 0, 1, 2)",
                    "achieved": round(bach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(bach / MFMA_PEAK_TFLOPS, 4),
                    "launches": int(bn_.value), "avg_launch_us": round(bms.value * 1e3 / max(1, bn_.value), 2),
                    "flops_per_launch": round(bfl.value / max(1, bn_.value) / 1e9, 3),
                    "algorithmic_bytes_per_launch": round(balg.value / max(1, bn_.value)),
                    "algorithmic_TBps": round(balg.value / (bms.value * 1e-3) / 1e12, 3) if bms.value > 0 else None}
            del sts, edits
            # K whole edits per GPU as one batch (re-projection of K transforms, 50 batched steps, K decodes), every rank
            barrier()
            te = time.perf_counter()
            imgs, _ = dh.transform_foreground_batch(depth, prompt, mask, bg_depth, uncond, init_noise, acts, tfs)
            barrier()
            te_mine = time.perf_counter() - te
            te_by_rank = per_rank(te_mine)
            te = max_over_ranks(te_mine)
            assert torch.isfinite(imgs).all()
            edits_info = {"edits_per_gpu": K, "edits_per_s": round(world * K / te, 4), "s_per_batch": round(te, 3),
                          "per_rank": {"s_per_batch": [round(v, 3) for v in te_by_rank],
                                       "edits_per_s": [round(K / v, 4) for v in te_by_rank],
                                       "what": "each rank's own wall time / rate for its batch (rank order); the whole-job figure divides by the MAX"},
                          "concurrent_streams": 1,
                          "what": f"{K} edits of one image per GPU as one batch: re-projection of {K} transforms, 38 guided + 12 "
                                  "unguided batched steps, AutoencoderKL decode (native decoder, random weights); identity cached; MAX over ranks"}
            del imgs
            # the same unit on concurrent lanes of ONE process: `streams` engine arenas + HIP streams on one copy of the weights,
            # streams x K edits per GPU, every lane running batches of K (bit-identical to the one-stream batches,
            # tests/test_loops_gpu.py::test_lanes_are_bit_identical_to_one_stream).  Its own record, never the headline.
            S = max(1, args.streams)
            if S > 1:
                tfs2 = [(TRANSFORMS[(i + rank) % 8][0], Y, torch.tensor(TRANSFORMS[(i + rank) % 8][1])) for i in range(S * K)]
                dh.transform_foreground_batch(depth, prompt, mask, bg_depth, uncond, init_noise, acts, tfs2, streams=S, batch=K)   # captures the lanes' graphs
                barrier()
                tc = time.perf_counter()
                imgs2, _ = dh.transform_foreground_batch(depth, prompt, mask, bg_depth, uncond, init_noise, acts, tfs2, streams=S, batch=K)
                barrier()
                tc = max_over_ranks(time.perf_counter() - tc)
                assert torch.isfinite(imgs2).all()
                lanes = gd.lanes(S)
                edits_info["concurrent"] = {
                    "concurrent_streams": S, "edits_per_gpu": S * K, "batch_per_stream": K,
                    "edits_per_s": round(world * S * K / tc, 4), "s_total": round(tc, 3),
                    "hbm_weights_bytes": int(gd.unet.weight_bytes()), "hbm_weights_copies": 1,
                    "hbm_arena_bytes_per_lane": int(lanes[-1].unet.workspace_bytes()),
                    "what": f"{S} lanes in one process per GPU (GuidedStableDiffuser.fork: private activation / gradient arenas, hipGraphs "
                            f"and stream per lane; U-Net weights resident once), {S * K} edits per GPU as {S} concurrent batches of {K}; "
                            "images bit-identical to the one-stream batches; MAX over ranks"}
                del imgs2

    sections = Sections()
    if K > 1:
        if world > 1:
            batched_section()          # every rank meets the same barriers: an exception must end the job, not leave ranks waiting
        else:
            failed = sections.run("batched_section", batched_section)
            if failed is not None:
                edits_info = failed

    # ---- secondary measurements (rank 0).  They never gate the headline line: a failure is reported in place of the numbers.
    def hbm_records(gd_, st_, depth_, bg_, mask_, res):
        def timed(fn, n=20, graph=False):
            # (a graph-replayed record is warmed with 60 replays and timed over 100: round 4's first record -- 20 replays of a 17 us
            #  evaluation right behind seconds of batched edits -- read 33 us where rocprofv3 says 17: 0.4 ms of tiny launches do
            #  not bring the clocks back up)
            for _ in range(3):
                fn()
            if graph:       # replay through a captured graph: device time of the launches, no host gaps between them
                eager = fn
                try:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
                        fn()
                    fn = g.replay
                    fn()
                    graphed.append(True)
                except Exception as exc:          # noqa: BLE001 - the record says so ("graph": false) and the failure is listed
                    sections.note("hbm_records.graph_capture", exc)
                    fn = eager
                    graphed.append(False)
            if graph and graphed and graphed[-1]:
                n = max(n, 100)
                for _ in range(60):
                    fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / n * 1e-3

        out = []
        graphed = []
        with torch.no_grad(), gd_.on_stream():
            cur = [o[1] for o in st_.orig]                    # another timestep's activations stand in for "current"
            for layers, tag in (((2,), "t%3==0: act2"), ((1,), "t%3==1: act1"), ((1, 2), "t%3==2: act1+act2")):
                fgw, bgw = st_.schedule(2 if len(layers) == 2 else (0 if layers == (2,) else 1), 0)

                def run():
                    for k in layers:
                        gd_._energy_grad(st_, k, cur[k], 0, fgw[k], bgw[k])
                sec = timed(run, graph=True)
                nbytes = sum(3 * cur[k].numel() * cur[k].element_size() for k in layers)
                out.append({"kernel": f"guidance energy fwd+bwd ({tag}) at {res}x{res}", "bound": "hbm", "bytes": nbytes,
                            "us": round(sec * 1e6, 2), "achieved": round(nbytes / sec / 1e9, 1), "peak": HBM_PEAK,
                            "unit": "GB/s", "frac": round(nbytes / sec / 1e9 / HBM_PEAK, 4), "graph": bool(graphed and graphed[-1])})
            tfs8 = [(TRANSFORMS[i % 8][0], Y, torch.tensor(TRANSFORMS[i % 8][1])) for i in range(8)]
            sec = timed(lambda: reproject_edits(depth_, bg_, mask_, gd_.get_depth_intrinsics(), tfs8, device_correspondences=True), n=5)
            per_edit = 9e6 * (res / 512.0) ** 2
            out.append({"kernel": f"batched K=8 unproject -> SE(3) -> z-buffer -> index maps at {res}x{res} (whole reproject_edits "
                                  "call, host glue included)", "bound": "hbm", "bytes": int(8 * per_edit),
                        "us": round(sec * 1e6, 1), "achieved": round(8 * per_edit / sec / 1e9, 2), "peak": HBM_PEAK,
                        "unit": "GB/s", "frac": round(8 * per_edit / sec / 1e9 / HBM_PEAK, 5)})
        return out

    hbm = None
    if rank == 0:
        hbm = sections.run("hbm_records", lambda: hbm_records(gd, st, depth, bg_depth, mask, args.res))

    # one whole edit: transform_foreground = re-projection + 38 guided + 12 unguided steps + AutoencoderKL decode
    def whole_edit():
        rot = dict(rot_angle=ang, rot_axis=Y, translation=torch.tensor(tr))
        with torch.no_grad():
            dh.transform_foreground(depth, prompt, mask, bg_depth, uncond, init_noise, acts, **rot)
            torch.cuda.synchronize()
            te = time.perf_counter()
            dh.transform_foreground(depth, prompt, mask, bg_depth, uncond, init_noise, acts, **rot)
            torch.cuda.synchronize()
            te = time.perf_counter() - te
            lat_img = torch.randn(1, 4, lat, lat, device=dev)
            gd.decode_latent_image(lat_img)
            torch.cuda.synchronize()
            td = time.perf_counter()
            gd.decode_latent_image(lat_img)
            torch.cuda.synchronize()
            td = time.perf_counter() - td
        per_image = phases["inversion_s"] + phases["initial_inference_s"]
        phases.update({"edit_s": round(te, 3), "vae_decode_s": round(td, 4),
                       "edits_per_s_identity_cached": round(1.0 / te, 4),
                       "edits_per_s_with_inversion_and_initial_inference": round(1.0 / (te + per_image), 4),
                       "edit_what": "transform_foreground: z-buffer + index maps + 38 guided + 12 unguided steps + AutoencoderKL "
                                    "decode (the SD VAE decoder on the engine's kernels, csrc/vae_engine.cpp, random weights)"})

    if rank == 0 and phases is not None:
        failed = sections.run("phases.whole_edit", whole_edit)
        if failed is not None:
            phases["error"] = failed["error"]

    # ---- BASELINE config 5: 768x768, bf16 U-Net + f32 guidance (energy, cotangent seed, latent gradient and update in f32) ----
    res768 = None
    if rank == 0 and args.res768 and args.res == 512:
        del dh, gd, st, acts
        torch.cuda.empty_cache()
        res768 = sections.run("res768_bf16", lambda: bench_768(conf, dev, prompt, TRANSFORMS, Y, max(5, args.steps // 2)))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.res == 512:
        cpu = sections.run("cpu_baseline", cpu_baseline)
        if cpu is not None and "error" in cpu.get("pieces", {}):
            sections.errors.append({"section": "cpu_baseline.pieces", "error": cpu["pieces"]["error"]})

    if rank == 0:
        value = world * args.steps / elapsed
        out = {
            "metric": f"guided-denoise steps/sec at {args.res}x{args.res} (SD2-depth)",
            "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "per_rank": {"ms_per_step": [round(v / args.steps * 1e3, 3) for v in elapsed_by_rank],
                         "steps_per_s": [round(args.steps / v, 3) for v in elapsed_by_rank], "backend": backend if world > 1 else None,
                         "what": "each rank's own timed region (rank order); value = world x steps / MAX over ranks"},
            "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": f"single {args.res}x{args.res} edit per GPU, SD2-depth (865.9M params, seeded random weights), guided "
                                   "phase: 3 x (fwd + energy + bwd-to-latent) + CFG fwd (B=2) + DDIM step",
                       "resolution": args.res, "edits_per_gpu": 1, "correspondences": int(corr.shape[0]),
                       "parallelism": "independent edits, one process per GPU, no collectives"},
            "roofline": roof, "hbm_kernels": hbm, "cpu_baseline": cpu, "phases": phases, "batched_edits": batch_info,
            "edits": edits_info, "res768_bf16": res768,
        }
        sys.stdout.flush()
        rc = sections.finish(out, json_fd)
    else:
        rc = 0
    if dist is not None:
        dist.destroy_process_group()
    if rc:
        raise SystemExit(rc)       # after the line: a broken secondary record (config 3 / 5, HBM kernels, CPU baseline) is not a pass


def bench_768(conf, dev, prompt, TRANSFORMS, Y, steps):
    """Guided-denoise steps/s of one 768x768 edit with the bf16 engine (96x96 latents), identity from initial_inference."""
    from diffusionhandles_amd import DiffusionHandles
    from diffusionhandles_amd.depth_transform import normalize_depth, transform_depth
    from diffusionhandles_amd.synthetic import make_scene
    from diffusionhandles_amd.unet import SD2_DEPTH
    res, lat = 768, 96
    dh = DiffusionHandles(conf, dtype=torch.bfloat16, unet_config=dict(SD2_DEPTH, sample_size=lat), max_batch=2).to(dev)
    gd = dh.diffuser
    depth, bg_depth, mask = (t.to(dev) for t in make_scene(res))
    T, gmax = conf.guided_diffuser.num_timesteps, conf.guided_diffuser.guidance_max_step
    uncond = gd._encode([""])[None].expand(T, -1, -1, -1).contiguous()
    torch.manual_seed(conf.guided_diffuser.seed)
    noise = torch.randn(1, 4, lat, lat).to(dev)
    acts, _, _, init_noise = gd.initial_inference(noise, normalize_depth(1.0 / depth), uncond, prompt)
    ang, tr = TRANSFORMS[2]
    disp_e, corr = transform_depth(depth, bg_depth, mask, gd.get_depth_intrinsics(), rot_angle=ang, rot_axis=Y,
                                   translation=torch.tensor(tr))
    st = gd.prepare_guidance(disp_e, prompt, acts, corr)
    gd.scheduler.set_timesteps(T)
    ts = gd.scheduler.timesteps
    x0 = init_noise.to(dev, torch.float32).permute(0, 2, 3, 1).contiguous()
    x = x0
    with torch.no_grad(), gd.on_stream():
        for i in range(3):
            x = gd.guided_step(st, x, i, ts[i], uncond[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            t_idx = i % gmax
            x = gd.guided_step(st, x0 if t_idx == 0 else x, t_idx, ts[t_idx], uncond[t_idx])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        assert torch.isfinite(x).all()
        # the HBM-bound warp / energy at high resolution: achieved GB/s of one energy evaluation (act1 + act2, bf16)
        cur = [o[1] for o in st.orig]
        fgw, bgw = st.schedule(2, 0)
        for _ in range(3):
            for k in (1, 2):
                gd._energy_grad(st, k, cur[k], 0, fgw[k], bgw[k])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            for k in (1, 2):
                gd._energy_grad(st, k, cur[k], 0, fgw[k], bgw[k])
        e1.record()
        e1.synchronize()
        sec = e0.elapsed_time(e1) / 20 * 1e-3
    nbytes = sum(3 * cur[k].numel() * cur[k].element_size() for k in (1, 2))
    return {"metric": "guided-denoise steps/sec at 768x768 (SD2-depth)", "value": round(steps / el, 3), "unit": "steps/s",
            "ms_per_step": round(el / steps * 1e3, 2), "steps": steps, "dtype": "bf16",
            "guidance": "energy, cotangent seed, latent gradient and latent update in f32; bf16 storage between the U-Net kernels, f32 accumulation",
            "correspondences": int(corr.shape[0]),
            "frac_of_mfma_peak": round(steps / el * STEP_TFLOP[768] / MFMA_PEAK_TFLOPS, 4),
            "energy_hbm": {"bytes": nbytes, "us": round(sec * 1e6, 1), "achieved": round(nbytes / sec / 1e9, 1), "unit": "GB/s",
                           "frac": round(nbytes / sec / 1e9 / HBM_PEAK, 4), "what": "energy + gradient of act1 and act2 at 96x96 cells (eager launches)"}}


if __name__ == "__main__":
    main()
