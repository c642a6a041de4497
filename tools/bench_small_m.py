#!/usr/bin/env python3
"""Weight streaming rate of the small-M GEMM classes of the B = 1 step (16 x 16 and 8 x 8 latents: M = 256, 64): every distinct
(M, N, K, kind) the engine dispatches at those levels (DH_GEMM_LOG of the tuning build), timed through the same dispatch
(dh_dbg_gemm: GEMM + its split-K reduce) with the weights WARM (back-to-back launches) and COLD (a 320-MB fill between launches:
weights from HBM, the in-situ condition for a layer's first use in a pass).  Bytes = the weight matrix (N x K x 2) -- the A
operand and the output of these launches are 0.03 - 3 MB.
    DIFFHANDLES_LIB=tools/bin/libdiffhandles_hip_tuning.so DH_GEMM_LOG=1 python3 tools/time_unet.py 1 2> gemmlog_b1.txt      (the shapes)
    python3 tools/bench_small_m.py gemmlog_b1.txt                                                                        (product library)"""
import collections, ctypes, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DH_DBG_PRETILED", "1")
import torch
from diffusionhandles_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
part = torch.empty(64 << 20, dtype=torch.float32, device=dev)
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
dt = torch.float16

shapes = collections.Counter()
for line in open(sys.argv[1]):
    m = re.search(r"GEMMLOG M=(\d+) N=(\d+) K=(\d+) mode=(\d)", line)
    if m and int(m.group(1)) <= 256 and int(m.group(4)) in (0, 1) and int(m.group(2)) >= 64 and int(m.group(1)) >= 64:
        shapes[tuple(int(g) for g in m.groups())] += 1
rows = []
for (M, N, K, mode), cnt in sorted(shapes.items()):
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    if mode == 1:
        H = int(round(M ** 0.5)); Cin = K // 9
        A = torch.randn(M, Cin, generator=g, device=dev).to(dt); lda = Cin; geo = (H, H, Cin, H, H, 1, 0)
    else:
        A = torch.randn(M, K, generator=g, device=dev).to(dt); lda = K; geo = (0, 0, 0, 0, 0, 1, 0)
    W = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).to(dt)
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, dtype=dt, device=dev)
    call = lambda: L.dh_dbg_gemm(0, P(A), lda, P(W), M, N, K, mode, *geo, P(bias), P(None), 0, 1, P(None), N, P(C), N, 0, P(part), part.numel(),
                                 _lib.stream_ptr())
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    warm = e0.elapsed_time(e1) * 1e3 / 20
    tot = 0.0
    for i in range(8):
        flush.fill_(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    cold = tot / 8
    wb = N * K * 2
    rows.append((M, N, K, mode, cnt, wb, warm, cold))
print("# small-M GEMM classes of the B = 1 forward + backward pass; GEMM + its split-K reduce per launch; 'logged' = dispatches in the log")
print("# (the engine dispatches a launch once per eager pass or graph capture: relative weights of the classes, not launches per step)")
print(f"{'M':>5} {'N':>6} {'K':>6} kind    logged   W MB | warm us   TB/s | cold us   TB/s | TFLOP/s cold")
tw = tc = 0.0
for (M, N, K, mode, per, wb, warm, cold) in rows:
    print(f"{M:5d} {N:6d} {K:6d} {'conv ' if mode else 'dense'} {per:8d} {wb / 1e6:6.1f} | {warm:7.1f} {wb / warm / 1e6:6.2f} | {cold:7.1f} {wb / cold / 1e6:6.2f} | {2.0 * M * N * K / cold / 1e6:6.0f}")
    tw += per * warm; tc += per * cold
print(f"# logged-count-weighted mean: {tw / sum(r[4] for r in rows):.1f} us warm, {tc / sum(r[4] for r in rows):.1f} us cold per launch")
