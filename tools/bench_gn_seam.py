#!/usr/bin/env python3
"""GroupNorm as one launch (seam between the slice workgroups) against statistics + apply launches: the outputs of the two
forms bit for bit, and the device time per GroupNorm inside a replayed hipGraph of N dependent ones (run on the GPU box)."""
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


def time_graph(fn, n=200, reps=5):
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
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


def main():
    gen = torch.Generator(device=dev).manual_seed(0)
    dt = torch.float16
    bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
    for B, HW, C in ((1, 4096, 320), (2, 4096, 320), (1, 4096, 640), (1, 4096, 960), (1, 1024, 640), (2, 1024, 640),
                     (1, 1024, 1920), (1, 256, 1280), (1, 256, 2560), (1, 64, 1280), (2, 64, 2560), (8, 4096, 320)):
        x = (torch.randn(B * HW, C, generator=gen, device=dev) * 1.7 + 0.3).to(dt)
        dy = torch.randn(B * HW, C, generator=gen, device=dev).to(dt) if bwd else None
        gamma = torch.randn(C, generator=gen, device=dev) * 0.2 + 1.0
        beta = torch.randn(C, generator=gen, device=dev) * 0.1
        outs = []
        times = []
        for seam in (0, 1):
            lib.dh_dbg_gn_seam(seam)
            y = torch.zeros_like(x)
            dx = torch.zeros_like(x) if bwd else None
            stats = torch.zeros(B * 32 * 2, dtype=torch.float32, device=dev)
            scr = torch.zeros(B * 64 * 4096, dtype=torch.float32, device=dev)
            st = lambda: _lib.stream_ptr()
            fn = lambda: lib.dh_dbg_groupnorm(0, P(x), P(gamma), P(beta), P(y), P(stats), P(dy), P(dx), P(scr), B, HW, C, 32,
                                              1e-5, 1, 0, st())
            fn()
            torch.cuda.synchronize()
            outs.append((y.clone(), stats.clone(), dx.clone() if bwd else None))
            times.append(time_graph(fn))
        failed = lib.dh_dbg_gn_seam(0)
        same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        if bwd:
            same = same and torch.equal(outs[0][2], outs[1][2])
        print(f"B={B} HW={HW:5d} C={C:5d}  two launches {times[0]:6.2f} us   one launch {times[1]:6.2f} us   "
              f"bit-identical {same}   seam failed {failed}", flush=True)


if __name__ == "__main__":
    main()
