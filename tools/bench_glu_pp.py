#!/usr/bin/env python3
"""GEGLU-epilogue GEMMs of the batch-8 pass: k_gemm_dma (family 1) vs the policy / k_gemm_pp (family 0), warm and cold."""
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
for (M, Fd, K) in [(32768, 1280, 320), (8192, 2560, 640), (2048, 5120, 1280)]:
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, generator=g, device=dev).to(dt)
    W = (torch.randn(2 * Fd, K, generator=g, device=dev) / K ** 0.5).to(dt)
    bias = torch.randn(2 * Fd, device=dev)
    pre = torch.empty(M, 2 * Fd, dtype=dt, device=dev)
    y = torch.empty(M, Fd, dtype=dt, device=dev)
    A2 = torch.randn(M, 4 * 0 + K * 4 if False else Fd // 4 * 4 // 4 * 1, generator=g, device=dev).to(dt) if False else None
    # backward: dy [M][Fd] = A2 [M][K2] Wb[Fd][K2]^T with K2 = the block width (ff.net.2: C = Fd / 4)
    K2 = Fd // 4
    A2 = torch.randn(M, K2, generator=g, device=dev).to(dt)
    Wb = (torch.randn(Fd, K2, generator=g, device=dev) / K2 ** 0.5).to(dt)
    dx = torch.empty(M, 2 * Fd, dtype=dt, device=dev)
    calls = {
        "fwd save": lambda: L.dh_dbg_gemm_glu(0, 0, P(A), K, P(W), M, 2 * Fd, K, P(bias), P(pre), P(y), P(None), P(None), _lib.stream_ptr()),
        "fwd nosave": lambda: L.dh_dbg_gemm_glu(0, 0, P(A), K, P(W), M, 2 * Fd, K, P(bias), P(None), P(y), P(None), P(None), _lib.stream_ptr()),
        "bwd": lambda: L.dh_dbg_gemm_glu(0, 1, P(A2), K2, P(Wb), M, Fd, K2, P(None), P(None), P(None), P(pre), P(dx), _lib.stream_ptr()),
    }
    for name, call in calls.items():
        out = f"M={M} F={Fd} K={K if name != 'bwd' else K2} {name:10s}:"
        for fam in (1, 0):
            L.dh_dbg_gemm_family(fam)
            for _ in range(3): call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call()
            e1.record(); torch.cuda.synchronize()
            warm = e0.elapsed_time(e1) * 100
            tot = 0.0
            for i in range(5):
                flush.fill_(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); call(); e1.record(); torch.cuda.synchronize()
                tot += e0.elapsed_time(e1) * 1e3
            out += f" | {'dma' if fam == 1 else 'pp '} warm {warm:6.1f} us cold {tot / 5:6.1f} us"
        L.dh_dbg_gemm_family(0)
        print(out, flush=True)
