#!/usr/bin/env python3
"""Experiment: does pulling the NEXT GEMM's weights into the Infinity Cache from a side stream speed up a chain of
weight-cold, latency-bound GEMMs?  24 GEMMs with distinct weights (more than the 256 MB cache in total) are captured in
one graph (a) alone, (b) with a reduction over the next GEMM's weights on a second stream, started when the current GEMM starts."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import _lib

os.environ["DH_DBG_PRETILED"] = "1"
dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
part = torch.empty(64 << 20, dtype=torch.float32, device=dev)


def chain(M, N, K, n, prefetch, ahead=1, group=1):
    dt = torch.float16
    A = [torch.randn(M, K if i == 0 else N, device=dev).to(dt) for i in range(2)]
    Wall = torch.randn(n, N, K, device=dev).to(dt)
    Ws = [Wall[i] for i in range(n)]
    C = [torch.empty(M, N, dtype=dt, device=dev) for _ in range(2)]
    sink = torch.zeros(n, device=dev)
    main, side = torch.cuda.Stream(), torch.cuda.Stream()

    def body():
        evs = []
        for i in range(n):
            if prefetch and i % group == 0 and i + ahead * group < n:
                ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
                with torch.cuda.stream(side):
                    lo = i + ahead * group
                    sink[lo] = Wall[lo:lo + group].view(torch.int16).amax()
            L.dh_dbg_gemm(0, P(A[0]), K, P(Ws[i]), M, N, K, 0, 0, 0, 0, 0, 0, 1, 0, P(None), P(None), 0, 1, P(None), N, P(C[i & 1]), N, 0,
                          P(part), part.numel(), ctypes.c_void_p(main.cuda_stream))
        if prefetch:
            main.wait_stream(side)

    with torch.cuda.stream(main):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            body()
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(5):
            g.replay()
        e1.record(main)
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5 / n
    print(f"M={M} N={N} K={K} x{n} (W {N*K*2/1e6:.1f} MB each) prefetch={prefetch} ahead={ahead} group={group}: {us:7.2f} us per GEMM", flush=True)


for shape in [(256, 1280, 11520), (64, 1280, 11520), (1024, 640, 5760), (256, 1280, 1280)]:
    n = 24 if shape[1] * shape[2] * 2 > 8e6 else 96
    chain(*shape, n, False)
    chain(*shape, n, True, 1, 1)
    chain(*shape, n, True, 1, 4)
    chain(*shape, n, True, 1, 8)
