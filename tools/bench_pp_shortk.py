#!/usr/bin/env python3
"""Short-K dense GEMMs and the GEGLU-epilogue GEMMs of the batch-8 pass: k_gemm_dma | k_gemm_pp (persistent workgroups) |
k_gemm_pp with one workgroup per work item, same process, interleaved rounds, warm and cold.
    python3 tools/bench_pp_shortk.py          DH_PP_ABL=1: timing ablations of the measurement build (no epilogue / one K tile)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DH_DBG_PRETILED", "1")
import torch
from diffusionhandles_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
dt = torch.float16
ARMS = [("dma", 1, 0), ("pp", 2, 0), ("pp-np", 2, 4)]        # -np: one workgroup per work item (not persistent)
# DH_PP_ABL=1: the pp arms again without their epilogue (0x100) and with one K tile only (0x200) -- needs tools/bin/libdh_pp_variants.so
if os.environ.get("DH_PP_ABL"):
    ARMS = [("pp", 2, 0), ("pp-noepi", 2, 0x100), ("pp-1tile", 2, 0x200), ("np", 2, 4), ("np-noepi", 2, 0x104), ("np-1tile", 2, 0x204)]


def set_arm(fam, two):
    L.dh_dbg_gemm_family(fam); L.dh_dbg_gemm_pp_ablate(two & 0x300); L.dh_dbg_gemm_pp_persist(0 if two & 4 else 1)


def measure(name, call, flops, iters=20, rounds=3):
    res = {}
    for _, fam, two in ARMS:
        set_arm(fam, two)
        for _ in range(3): call()
    torch.cuda.synchronize()
    for rnd in range(rounds):
        for arm, fam, two in ARMS:
            set_arm(fam, two)
            call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): call()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(("warm", arm), []).append(e0.elapsed_time(e1) * 1e3 / iters)
            tot = 0.0
            for _ in range(6):
                flush.fill_(rnd)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); call(); e1.record(); torch.cuda.synchronize()
                tot += e0.elapsed_time(e1) * 1e3
            res.setdefault(("cold", arm), []).append(tot / 6)
    set_arm(0, 0)
    med = lambda v: sorted(v)[len(v) // 2]
    out = f"{name:44s}:"
    for arm, _, _ in ARMS:
        w, c = med(res[("warm", arm)]), med(res[("cold", arm)])
        out += f" | {arm:8s} warm {w:6.1f} us {flops / w / 1e6:5.0f} TF cold {c:6.1f} us"
    print(out, flush=True)


DENSE = [] if os.environ.get("DH_GLU_SMALL") else [(32768, 320, 320, True), (32768, 320, 320, False), (32768, 960, 320, False), (32768, 320, 1280, True),
                         (32768, 1280, 320, False), (8192, 640, 640, True), (8192, 1920, 640, False), (8192, 640, 2560, True),
                         (65536, 320, 320, True), (16384, 640, 640, True)]
for (M, N, K, withR) in DENSE:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).to(dt)
    W = (torch.rand(N, K, generator=g, device=dev) * 2 - 1).to(dt)
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, generator=g, device=dev).to(dt) if withR else None
    C = torch.empty(M, N, dtype=dt, device=dev)
    call = lambda: L.dh_dbg_gemm(0, P(A), K, P(W), M, N, K, 0, 0, 0, 0, 0, 0, 1, 0, P(bias), P(None), 0, 1, P(R), N, P(C), N, 0, P(None), 0,
                                 _lib.stream_ptr())
    measure(f"dense M={M} N={N} K={K} {'+R' if withR else '  '}", call, 2.0 * M * N * K)

GLU = [(32768, 1280, 320), (8192, 2560, 640), (2048, 5120, 1280)]
if os.environ.get("DH_GLU_SMALL"):      # the B = 1 / 2 / 4 passes and the B = 16 CFG pass
    GLU = [(4096, 1280, 320), (8192, 1280, 320), (16384, 1280, 320), (65536, 1280, 320), (1024, 2560, 640), (2048, 2560, 640), (4096, 2560, 640),
           (16384, 2560, 640), (512, 5120, 1280), (1024, 5120, 1280), (4096, 5120, 1280), (512, 5120, 1280)]
for (M, Fd, K) in GLU:
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, generator=g, device=dev).to(dt)
    W = (torch.randn(2 * Fd, K, generator=g, device=dev) / K ** 0.5).to(dt)
    bias = torch.randn(2 * Fd, device=dev)
    pre = torch.empty(M, 2 * Fd, dtype=dt, device=dev)
    y = torch.empty(M, Fd, dtype=dt, device=dev)
    K2 = Fd // 4
    A2 = torch.randn(M, K2, generator=g, device=dev).to(dt)
    Wb = (torch.randn(Fd, K2, generator=g, device=dev) / K2 ** 0.5).to(dt)
    dx = torch.empty(M, 2 * Fd, dtype=dt, device=dev)
    S = _lib.stream_ptr()
    measure(f"geglu fwd save   M={M} F={Fd} K={K}", lambda: L.dh_dbg_gemm_glu(0, 0, P(A), K, P(W), M, 2 * Fd, K, P(bias), P(pre), P(y), P(None), P(None), S),
            2.0 * M * 2 * Fd * K)
    measure(f"geglu fwd nosave M={M} F={Fd} K={K}", lambda: L.dh_dbg_gemm_glu(0, 0, P(A), K, P(W), M, 2 * Fd, K, P(bias), P(None), P(y), P(None), P(None), S),
            2.0 * M * 2 * Fd * K)
    measure(f"geglu bwd        M={M} F={Fd} K={K2}", lambda: L.dh_dbg_gemm_glu(0, 1, P(A2), K2, P(Wb), M, Fd, K2, P(None), P(None), P(None), P(pre), P(dx), S),
            2.0 * M * Fd * K2)
