#!/usr/bin/env python3
"""Attention backward of one layer: dQ then dK/dV on one stream (the engine's order) against delta, then dQ and dK/dV SIDE BY SIDE on
two streams (fork / join events) -- what the two kernels gain from each other's idle CUs.  B = 1, 2, 8; self-attention N = 4096 / 1024 /
256 and the 96 x 96 level N = 9216.  python3 tools/bench_attn_pair.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
dt = torch.float16
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for (B, H, N) in [(1, 5, 4096), (1, 10, 1024), (1, 20, 256), (2, 5, 4096), (2, 10, 1024), (8, 5, 4096), (8, 10, 1024), (1, 5, 9216)]:
    C = 64 * H
    qkv = torch.randn(B * N, 3 * C, device=dev).to(dt)
    o = torch.empty(B * N, C, dtype=dt, device=dev)
    d_o = torch.randn(B * N, C, device=dev).to(dt)
    lse = torch.empty(B * H * N, dtype=torch.float32, device=dev)
    delta = torch.empty(B * H * N, dtype=torch.float32, device=dev)
    dqkv = torch.zeros(B * N, 3 * C, dtype=dt, device=dev)
    q, k, v = qkv, qkv[:, C:], qkv[:, 2 * C:]
    dq, dk, dv = dqkv, dqkv[:, C:], dqkv[:, 2 * C:]
    _lib.check(L.dh_dbg_attention(0, P(q), 3 * C, P(k), P(v), 3 * C, P(o), C, P(lse), P(None), P(None), P(None), P(None), P(None), B, H, N, N,
                                  ctypes.c_void_p(s1.cuda_stream)))
    torch.cuda.synchronize()
    res = {}
    outs = {}
    for mode in ("sequential", "side by side"):
        st2 = ctypes.c_void_p(s2.cuda_stream) if mode != "sequential" else ctypes.c_void_p(0)
        call = lambda: _lib.check(L.dh_dbg_attention_bwd_pair(0, P(q), 3 * C, P(k), P(v), 3 * C, P(o), C, P(lse), P(d_o), P(delta), P(dq), P(dk), P(dv),
                                                               B, H, N, N, ctypes.c_void_p(s1.cuda_stream), st2))
        for _ in range(3): call()
        torch.cuda.synchronize()
        outs[mode] = dqkv.clone()
        ts = []
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s1):
                e0.record()
                for _ in range(20): call()
                e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        res[mode] = sorted(ts)[1]
    same = bool(torch.equal(outs["sequential"], outs["side by side"]))
    print(f"B={B} H={H:2d} N={N:5d}: dQ then dK/dV {res['sequential']:7.1f} us | delta, then dQ || dK/dV {res['side by side']:7.1f} us   ({res['side by side'] / res['sequential'] - 1:+.1%})  same bits: {same}", flush=True)
