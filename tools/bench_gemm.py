#!/usr/bin/env python3
"""Micro-benchmark of the MFMA implicit-GEMM kernel on the U-Net's shapes (run on the GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
part = torch.empty(64 << 20, dtype=torch.float32, device=dev)

def run(M, N, K, conv=None, iters=30):
    dt = torch.float16
    if conv:
        B, H, Cin = conv
        A = torch.randn(B * H * H, Cin, device=dev).to(dt); lda = Cin
        geo = (H, H, Cin, H, H, 1, 0); mode = 1
    else:
        A = torch.randn(M, K, device=dev).to(dt); lda = K; geo = (0, 0, 0, 0, 0, 1, 0); mode = 0
    W = torch.randn(N, K, device=dev).to(dt)
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, dtype=dt, device=dev)
    def call():
        L.dh_dbg_gemm(0, P(A), lda, P(W), M, N, K, mode, *geo, P(bias), P(None), 0, 1, P(None), N, P(C), N, 0, P(part), part.numel(), _lib.stream_ptr())
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"M={M:5d} N={N:5d} K={K:6d} {'conv' if conv else 'dense':5s}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s")

shapes = [
    (4096, 320, 2880, (1, 64, 320)), (4096, 320, 5760, (1, 64, 640)), (4096, 320, 320, None), (4096, 960, 320, None),
    (4096, 2560, 320, None), (4096, 320, 1280, None), (1024, 640, 5760, (1, 32, 640)), (1024, 640, 640, None),
    (1024, 5120, 640, None), (256, 1280, 11520, (1, 16, 1280)), (256, 1280, 1280, None), (64, 1280, 11520, (1, 8, 1280)),
    (64, 1280, 23040, (1, 8, 2560)), (8192, 320, 2880, (2, 64, 320)), (4096, 640, 5760, (1, 64, 640)), (154, 24960, 1024, None),
]
import os
if os.environ.get('DH_SHAPES') == 'abl':
    shapes = [(4096, 640, 5760, (1, 64, 640)), (256, 1280, 11520, (1, 16, 1280)), (8192, 1280, 11520, (8, 32, 1280))]
if os.environ.get('DH_SHAPES') == 'b1':
    shapes = [(4096, 320, 2880, (1, 64, 320)), (4096, 640, 5760, (1, 64, 640)), (4096, 320, 1280, None), (1024, 640, 5760, (1, 32, 640)), (4096, 1280, 320, None)]
if os.environ.get('DH_SHAPES') == 'b8':     # the batched-edits mode (8 images per pass)
    shapes = [(32768, 320, 2880, (8, 64, 320)), (32768, 320, 5760, (8, 64, 640)), (8192, 640, 5760, (8, 32, 640)),
              (8192, 640, 11520, (8, 32, 1280)), (2048, 1280, 11520, (8, 16, 1280)), (512, 1280, 11520, (8, 8, 1280)),
              (32768, 320, 320, None), (32768, 960, 320, None), (32768, 2560, 320, None), (32768, 320, 1280, None),
              (8192, 5120, 640, None), (8192, 640, 2560, None), (2048, 1280, 1280, None), (2048, 10240, 1280, None)]
if os.environ.get('DH_SHAPES') == 'lin':    # the long-K linears that split K over workgroups + a plain reduce launch
    shapes = [(256, 1280, 5120, None), (256, 1280, 10240, None), (1024, 640, 2560, None), (1024, 640, 5120, None), (64, 1280, 5120, None),
              (64, 1280, 10240, None), (256, 10240, 1280, None), (1024, 5120, 640, None), (4096, 320, 2560, None)]
for s in shapes:
    run(*s)
