#!/usr/bin/env python3
"""Per-node cost of a hipGraph of N dependent tiny kernels as a function of N (a runtime that submitted long graphs in
segments with a stall per segment would show it); under rocprofv3 --kernel-trace, tools/gap_analyze.py lists the gaps."""
import sys
import torch
dev = torch.device("cuda:0")
x = torch.zeros(256, device=dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        x.add_(1.0)
    torch.cuda.synchronize()
    for n in [int(a) for a in sys.argv[1:]] or [100, 400, 1000, 4000, 16000]:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                x.add_(1.0)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
        print(f"graph of {n:6d} dependent kernels: {e0.elapsed_time(e1) * 1e3 / 5 / n:.3f} us per node ({e0.elapsed_time(e1) / 5:.3f} ms per replay)")
