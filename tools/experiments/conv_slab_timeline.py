#!/usr/bin/env python3
"""In-kernel timeline of k_conv_slab (measurement build: conv_slab.hip compiled with -DDH_CS_TIMELINE, DIFFHANDLES_LIB=tools/bin/libdh_cs_tl.so):
workgroup (0,0,0), wave 0 (a multiplying wave: per step "work done" and "barrier passed") and wave 4 (a loader: per step "pieces issued",
"counted wait passed", "barrier passed").  Ticks are shader clocks."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import _lib

os.environ["DH_DBG_PRETILED"] = "1"
dev = torch.device("cuda:0")
L = _lib.lib()
L.dh_dbg_conv_slab_timeline.argtypes = [ctypes.c_void_p]
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
part = torch.empty(64 << 20, dtype=torch.float32, device=dev)
ts = torch.zeros(512, dtype=torch.int64, device=dev)


def run(B, H, Cin, N, cold=False):
    dt = torch.float16
    M, K = B * H * H, 9 * Cin
    A = torch.randn(M, Cin, device=dev).to(dt)
    W = torch.randn(N, K, device=dev).to(dt)
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, dtype=dt, device=dev)
    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev) if cold else None

    def call():
        L.dh_dbg_gemm(0, P(A), Cin, P(W), M, N, K, 1, H, H, Cin, H, H, 1, 0, P(bias), P(None), 0, 1, P(None), N, P(C), N, 0, P(part), part.numel(), _lib.stream_ptr())
    for _ in range(3):
        call()
    if cold:
        junk.add_(1.0)
    ts.zero_()
    L.dh_dbg_conv_slab_timeline(P(ts))
    call()
    L.dh_dbg_conv_slab_timeline(None)
    torch.cuda.synchronize()
    t = ts.cpu().tolist()
    c, l = t[:256], t[256:]
    nsteps = 0
    while 3 + 2 * nsteps < 256 and c[3 + 2 * nsteps]:
        nsteps += 1
    t0 = min(x for x in (c[0], l[0]) if x)
    print(f"B={B} H={H} Cin={Cin} N={N} {'COLD' if cold else 'warm'}: {nsteps} steps; consumer prologue barrier at {c[1] - t0}, loader prologue issue {l[0] - t0} wait-> {l[1] - t0}")
    cw, cb, li, lw_, lb = [], [], [], [], []
    prev_c, prev_l = c[1], l[1]
    for S in range(nsteps):
        cw.append(c[2 + 2 * S] - prev_c); cb.append(c[3 + 2 * S] - c[2 + 2 * S]); prev_c = c[3 + 2 * S]
        li.append(l[2 + 3 * S] - prev_l); lw_.append(l[3 + 3 * S] - l[2 + 3 * S]); lb.append(l[4 + 3 * S] - l[3 + 3 * S]); prev_l = l[4 + 3 * S]
    f = lambda v: " ".join(f"{x:5d}" for x in v[:30])
    print("   consumer work   :", f(cw))
    print("   consumer barrier:", f(cb))
    print("   loader issue    :", f(li))
    print("   loader wait     :", f(lw_))
    print("   loader barrier  :", f(lb))
    print(f"   per step: consumer work {sum(cw) / nsteps:.0f} + barrier {sum(cb) / nsteps:.0f}; loader issue {sum(li) / nsteps:.0f} + wait {sum(lw_) / nsteps:.0f} + barrier {sum(lb) / nsteps:.0f}; "
          f"loop {c[1 + 2 * nsteps] - c[1]} ticks")


for cold in (False, True):
    run(1, 64, 320, 320, cold)
    run(1, 32, 640, 640, cold)
