#!/usr/bin/env python3
"""k_gemm_pp (ping-pong main loop) against k_gemm_dma on the batch-8 / 96x96-latent GEMM shapes, same process, interleaved
rounds, L2-warm (same operands every launch) and COLD (a ring of operand sets larger than the 256 MB Infinity Cache, the
in-situ condition: weights come from HBM, activations from wherever the previous kernel left them).  Run on the GPU box:
    python3 tools/bench_gemm_pp.py [shapes]      shapes: b8 (default) | l96 | all"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DH_DBG_PRETILED", "1")
import torch
from diffusionhandles_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
part = torch.empty(64 << 20, dtype=torch.float32, device=dev)
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)      # > Infinity Cache: a write pass over it evicts the operands

B8 = [(32768, 320, 2880, (8, 64, 320)), (32768, 320, 5760, (8, 64, 640)), (32768, 640, 2880, (8, 64, 320)), (8192, 640, 5760, (8, 32, 640)),
      (8192, 640, 11520, (8, 32, 1280)), (8192, 1280, 5760, (8, 32, 640)), (2048, 1280, 11520, (8, 16, 1280)),
      (32768, 320, 320, None), (32768, 960, 320, None), (32768, 320, 1280, None), (8192, 640, 640, None), (8192, 640, 2560, None),
      (2048, 1280, 1280, None), (2048, 1280, 5120, None)]
L96 = [(9216, 320, 2880, (1, 96, 320)), (2304, 640, 5760, (1, 48, 640)), (9216, 320, 320, None), (9216, 320, 1280, None)]
B16 = [(65536, 320, 2880, (16, 64, 320)), (16384, 640, 5760, (16, 32, 640)), (4096, 1280, 11520, (16, 16, 1280)), (65536, 320, 320, None)]


def run(M, N, K, conv, iters=20, rounds=3):
    dt = torch.float16
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    if conv:
        Bn, H, Cin = conv
        A = (torch.rand(Bn * H * H, Cin, generator=g, device=dev) * 2 - 1).to(dt); lda = Cin
        geo = (H, H, Cin, H, H, 1, 0); mode = 1
    else:
        A = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).to(dt); lda = K; geo = (0, 0, 0, 0, 0, 1, 0); mode = 0
    W = (torch.rand(N, K, generator=g, device=dev) * 2 - 1).to(dt)
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, generator=g, device=dev).to(dt)
    C = torch.empty(M, N, dtype=dt, device=dev)

    def call():
        L.dh_dbg_gemm(0, P(A), lda, P(W), M, N, K, mode, *geo, P(bias), P(None), 0, 1, P(R), N, P(C), N, 0, P(part), part.numel(), _lib.stream_ptr())
    res = {}
    # arms: k_gemm_dma (family 1), then k_gemm_pp main-loop variants (family 2; DH_PP_VARS=0,2,4,6 needs tools/bin/libdh_pp_variants.so)
    arms = [(1, 0)] + [(0 if os.environ.get("DH_PP_FAMILY0") else 2, v) for v in VARS]
    for fam, var in arms:
        L.dh_dbg_gemm_family(fam); L.dh_dbg_gemm_pp_variant(var, None)
        for _ in range(3): call()
    torch.cuda.synchronize()
    for rnd in range(rounds):
        for fam, var in arms:
            L.dh_dbg_gemm_family(fam); L.dh_dbg_gemm_pp_variant(var, None)
            fam = (fam, var)
            call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): call()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(("warm", fam), []).append(e0.elapsed_time(e1) * 1e3 / iters)
            # cold: flush the caches between launches, time each launch by its own events
            tot = 0.0
            for _ in range(6):
                flush.fill_(rnd)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); call(); e1.record(); torch.cuda.synchronize()
                tot += e0.elapsed_time(e1) * 1e3
            res.setdefault(("cold", fam), []).append(tot / 6)
    L.dh_dbg_gemm_family(0); L.dh_dbg_gemm_pp_variant(-1, None)
    fl = 2.0 * M * N * K
    med = lambda v: sorted(v)[len(v) // 2]
    out = f"M={M:6d} N={N:5d} K={K:6d} {'conv' if conv else 'dense':5s}:"
    for fam, var in arms:
        w, c = med(res[("warm", (fam, var))]), med(res[("cold", (fam, var))])
        out += f" | {'dma' if fam == 1 else 'pp' + str(var)} warm {w:6.1f} us {fl/w/1e6:5.0f} TF cold {c:6.1f} us"
    print(out, flush=True)


# the B = 1 / B = 2 launches the split-K branch of gemm_pp_plan takes (input gradients of the 64 x 64-level convolutions whose inputs
# have 640 / 960 channels, the 32 x 32 level at B = 2 ...): policy (DH_PP_FAMILY0=1: family 0 instead of "k_gemm_pp wherever it can run")
B1 = [(4096, 640, 2880, (1, 64, 320)), (4096, 960, 2880, (1, 64, 320)), (4096, 640, 5760, (1, 64, 640)), (2048, 1280, 5760, (2, 32, 640)),
      (2048, 640, 5760, (2, 32, 640)), (2048, 1280, 11520, (2, 32, 1280)), (1024, 1280, 11520, (4, 16, 1280)), (4096, 1280, 5760, (4, 32, 640))]
VARS = [int(v) for v in os.environ.get("DH_PP_VARS", "-1").split(",")]
which = sys.argv[1] if len(sys.argv) > 1 else "b8"
KEY = [B8[0], B8[3], B8[6], B8[7], B8[9], B8[11]]       # one per class: the shapes the review names
shapes = {"b8": B8, "l96": L96, "b16": B16, "all": B8 + B16 + L96, "key": KEY, "b1": B1}[which]
for s in shapes:
    run(*s)
