#!/usr/bin/env python3
"""Tile-level timeline of a PERSISTENT k_gemm_pp workgroup (stamping variant of tools/bin/libdh_pp_variants.so): for the first
tiles workgroup 0 walks -- cycles in: setup (tile coordinates, lane offsets), prologue issue + first wait, K loop, epilogue
(issue only: the stores drain behind it), and what the next tile's first wait costs.
    DIFFHANDLES_LIB=tools/bin/libdh_pp_variants.so python3 tools/pp_tile_timeline.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DH_DBG_PRETILED", "1")
import numpy as np
import torch
from diffusionhandles_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def run(M, N, K, var=20, withR=False, abl=0):
    dt = torch.float16
    g = torch.Generator(device=dev).manual_seed(1)
    A = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).to(dt)
    W = (torch.rand(N, K, generator=g, device=dev) * 2 - 1).to(dt)
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, generator=g, device=dev).to(dt) if withR else None
    C = torch.empty(M, N, dtype=dt, device=dev)
    ts = torch.zeros(1024, dtype=torch.int64, device=dev)
    L.dh_dbg_gemm_family(2); L.dh_dbg_gemm_pp_ablate(abl)
    for it in range(3):
        ts.zero_()
        L.dh_dbg_gemm_pp_variant(var | 1, P(ts))
        L.dh_dbg_gemm(0, P(A), K, P(W), M, N, K, 0, 0, 0, 0, 0, 0, 1, 0, P(bias), P(None), 0, 1, P(R), N, P(C), N, 0, P(None), 0, _lib.stream_ptr())
        torch.cuda.synchronize()
    L.dh_dbg_gemm_family(0); L.dh_dbg_gemm_pp_variant(-1, None); L.dh_dbg_gemm_pp_ablate(0)
    t = ts.cpu().numpy()
    nt = 1 if abl & 0x200 else K // 64
    seg = 1 if var & 4 else 2
    TM = 4                                     # 256-row tiles (the shapes below): stamps after the column vectors and after each row block
    per = 5 + 6 * nt * seg + 1 + TM
    print(f"== dense M={M} N={N} K={K}{' +R' if withR else ''} variant {var} ablate {abl:#x}: {per} stamps per tile")
    for grp in (0, 1):
        s = t[grp * 512:(grp + 1) * 512]
        n = int((s != 0).sum())
        ntile = n // per
        print(f"  group {grp}: {ntile} tiles stamped")
        prev_end = None
        for i in range(ntile):
            q = s[i * per:(i + 1) * per].astype(np.int64)
            setup, first = q[1] - q[0], q[2] - q[1]
            le = per - 2 - (1 + TM)                # index of the loop-end stamp
            loop = q[le] - q[2]
            epi = q[per - 1] - q[le]
            epi_parts = " ".join(str(int(q[le + 1 + k] - q[le + k])) for k in range(1 + TM))
            gap = (q[0] - prev_end) if prev_end is not None else 0
            prev_end = q[per - 1]
            print(f"    tile {i}: setup {setup:6d}  prologue issue + first wait {first:6d}  K loop {loop:6d} ({loop // max(1, nt * seg)} per segment)  epilogue issue {epi:6d} (vectors, rows: {epi_parts})"
                  f"  gap before {gap:5d}  | tile total {q[per - 1] - q[0]:6d} cyc")


SHAPES = [(32768, 1280, 320, False), (32768, 960, 320, False), (65536, 320, 320, True)]
if os.environ.get("DH_TL_SMALL"):
    # few workgroups: is the epilogue's store issue bounded per CU or by the chip's write path?  (8, 32, 64, 128, 256 tiles of 256 x 160)
    SHAPES = [(2048, 160, 320, False), (8192, 160, 320, False), (2048, 1280, 320, False), (4096, 1280, 320, False), (8192, 1280, 320, False)]
for (M, N, K, withR) in SHAPES:
    run(M, N, K, 20, withR)
run(32768, 1280, 320, 20, False, 0x100)
