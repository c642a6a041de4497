#!/usr/bin/env python3
"""Micro-benchmark of the flash-attention kernels on the U-Net's shapes (run on the GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
def run(B, H, Nq, Nk, iters=20, bwd=True):
    C = H * 64; dt = torch.float16
    q = torch.randn(B, Nq, C, device=dev).to(dt); k = torch.randn(B, Nk, C, device=dev).to(dt); v = torch.randn(B, Nk, C, device=dev).to(dt)
    do = torch.randn(B, Nq, C, device=dev).to(dt)
    o = torch.empty_like(q); lse = torch.empty(B, H, Nq, dtype=torch.float32, device=dev); delta = torch.empty_like(lse)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    def call(mode):
        if mode == 0: L.dh_dbg_attention(0, P(q), C, P(k), P(v), C, P(o), C, P(lse), P(None), P(delta), P(None), P(None), P(None), B, H, Nq, Nk, _lib.stream_ptr())
        else: L.dh_dbg_attention(0, P(q), C, P(k), P(v), C, P(o), C, P(lse), P(do), P(delta), P(dq), P(dk), P(dv), B, H, Nq, Nk, _lib.stream_ptr())
    res = []
    for mode in (0, 1):
        for _ in range(3): call(mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): call(mode)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / iters)
    f = 4.0 * B * H * Nq * Nk * 64
    print(f"B={B} H={H:2d} Nq={Nq:5d} Nk={Nk:5d}: fwd {res[0]:7.1f} us ({f/res[0]/1e6:6.1f} TF/s)   fwd+delta+dq+dkv {res[1]:7.1f} us ({3.5*f/(res[1])/1e6:6.1f} TF/s)")
CFGS = [(1, 5, 4096, 4096), (2, 5, 4096, 4096), (1, 10, 1024, 1024), (1, 20, 256, 256), (1, 5, 4096, 77), (1, 10, 1024, 77), (1, 5, 9216, 9216)]
if os.environ.get("DH_ATTN_CFGS") == "b8":      # the batched-edits mode
    CFGS = [(8, 5, 4096, 4096), (8, 10, 1024, 1024), (8, 20, 256, 256), (8, 5, 4096, 77), (8, 10, 1024, 77)]
if os.environ.get("DH_ATTN_CFGS") == "n1024":
    CFGS = [(1, 10, 1024, 1024)]
if os.environ.get("DH_ATTN_CFGS") == "n4096":   # one shape (counter passes)
    CFGS = [(1, 5, 4096, 4096)]
for cfg in CFGS:
    run(*cfg)
