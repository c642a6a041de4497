#!/usr/bin/env python3
"""Diagnostics for k_gemm_pp: which part of the epilogue / which tile goes wrong (error maps per 16x16 block)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
torch.set_printoptions(linewidth=250, edgeitems=1000, threshold=100000)

def gemm(A, W, M, N, K, fam, bias=None, R=None):
    C = torch.full((M, N), -7.0, dtype=torch.float16, device=dev)
    L.dh_dbg_gemm_family(fam)
    rc = L.dh_dbg_gemm(0, P(A), K, P(W), M, N, K, 0, 0, 0, 0, 0, 0, 1, 0, P(bias), P(None), 0, 1, P(R), N, P(C), N, 0, P(None), 0, _lib.stream_ptr())
    torch.cuda.synchronize()
    L.dh_dbg_gemm_family(0)
    assert rc == 0
    return C

shapes = [(512, 320, 320), (300, 320, 128), (100, 128, 192), (128, 640, 320), (1024, 960, 384)]
for (M, N, K) in shapes:
    print(f"==== M={M} N={N} K={K}")
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, generator=g, device=dev).half()
    W = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    bias = torch.randn(N, generator=g, device=dev)
    R = torch.randn(M, N, generator=g, device=dev).half()
    base = A.float() @ W.float().t()
    for tag, b, r in (("plain", None, None), ("bias", bias, None), ("R", None, R), ("bias+R", bias, R)):
        ref = base + (b if b is not None else 0) + (r.float() if r is not None else 0)
        C = gemm(A, W, M, N, K, 2, b, r).float()
        err = (C - ref).abs() > 4e-3 + 4e-3 * ref.abs()
        print(f"{tag}: bad frac {err.float().mean().item():.4f}")
        if err.any():
            Mp, Np = (M + 15) // 16 * 16, N
            e = torch.zeros(Mp, Np, device=dev); e[:M] = err.float()
            mp = (e.view(Mp // 16, 16, Np // 16, 16).mean(dim=(1, 3)) * 100).round().int()
            print(mp[:8])
            i = int(err.float().sum(dim=1).argmax())
            print("worst row", i, "C-ref first 40 cols:", [round(v, 2) for v in (C - ref)[i, :40].tolist()])
            if b is not None:
                print("   bias first 40:", [round(v, 2) for v in b[:40].tolist()])
            break
