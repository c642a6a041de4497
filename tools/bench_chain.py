#!/usr/bin/env python3
"""What one launch of OUR kernels costs inside a hipGraph, in isolation: N dependent launches of one kernel captured in
a graph and replayed (device time per launch from events), next to torch's trivial kernel in the same process.
Diagnostic for the launch-bound B=1 step (run on the GPU box)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionhandles_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.lib()


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def time_graph(name, fn, n=200, reps=5):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        e1.synchronize()
    print(f"{name:60s} {e0.elapsed_time(e1) * 1e3 / (n * reps):7.2f} us per launch")


def main():
    os.environ["DH_DBG_PRETILED"] = "1"      # the GEMM hook then issues one launch (weights untiled: timing only)
    g = torch.Generator(device=dev).manual_seed(0)
    dt = torch.float16
    one = torch.zeros(64, device=dev)
    time_graph("torch add_ on 64 floats", lambda: one.add_(1.0))
    big = torch.zeros(4096 * 320, device=dev, dtype=dt)
    time_graph("torch add_ on 4096x320 halves", lambda: big.add_(1.0))
    for rows, C in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
        x = torch.randn(rows, C, generator=g, device=dev).to(dt)
        y = torch.empty_like(x)
        gamma = torch.ones(C, device=dev)
        beta = torch.zeros(C, device=dev)
        stats = torch.empty(rows * 2, dtype=torch.float32, device=dev)
        st = lambda: _lib.stream_ptr()
        time_graph(f"layernorm fwd {rows}x{C}", lambda: lib.dh_dbg_layernorm(0, P(x), P(gamma), P(beta), P(y), P(stats), None, None, None, rows, C, 1e-5, st()))
        xx = torch.randn(rows, 8 * C, generator=g, device=dev).to(dt)
        yy = torch.empty(rows, 4 * C, dtype=dt, device=dev)
        time_graph(f"geglu fwd {rows}x{4 * C}", lambda: lib.dh_dbg_geglu(0, P(xx), P(yy), None, None, rows, 4 * C, st()))
        # groupnorm forward (partial + apply = 2 launches per call)
        HW = rows
        sc = torch.empty(1 << 16, dtype=torch.float32, device=dev)
        gst = torch.empty(64, dtype=torch.float32, device=dev)
        time_graph(f"groupnorm fwd {rows}x{C} (2 launches per call)",
                   lambda: lib.dh_dbg_groupnorm(0, P(x), P(gamma), P(beta), P(y), P(gst), None, None, P(sc), 1, HW, C, 32, 1e-5, 1, 0, st()))
        # dense GEMM rows x C x C (pre-tiled weights so that the hook issues one launch)
        W = (torch.randn(C, C, generator=g, device=dev) / C ** 0.5).to(dt)
        Cc = torch.empty(rows, C, dtype=dt, device=dev)
        bias = torch.zeros(C, device=dev)

        def gemm():
            lib.dh_dbg_gemm(0, P(x), C, P(W), rows, C, C, 0, 0, 0, 0, 0, 0, 1, 0, P(bias), None, 0, 1, None, C, P(Cc), C, 0,
                            None, 0, st())
        # first call tiles (env read once): do one untimed call before the env var matters
        time_graph(f"gemm {rows}x{C}x{C} (no split-K)", gemm)


if __name__ == "__main__":
    main()
