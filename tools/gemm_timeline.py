#!/usr/bin/env python3
"""In-kernel timeline of the first workgroup of a GEMM launch (tuning build: DIFFHANDLES_LIB=tools/bin/libdiffhandles_hip_tuning.so):
s_memtime stamps at kernel start / prologue DMA issued / in front of the K loop / first tile landed / K loop done / merge and
LayerNorm transform done / stores issued / stores drained.  Diagnostic for the fixed cost of the small GEMMs."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import _lib

os.environ["DH_DBG_PRETILED"] = "1"
dev = torch.device("cuda:0")
L = _lib.lib()
L.dh_dbg_gemm_timeline.argtypes = [ctypes.c_void_p]
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
part = torch.empty(64 << 20, dtype=torch.float32, device=dev)
ts = torch.zeros(8 + 3 * 40, dtype=torch.int64, device=dev)
NAMES = ["start->prologue DMA issued", "->in front of K loop", "->first tile landed", "->K loop done", "->merge/LN done",
         "->stores issued", "->stores drained"]


def run(M, N, K, conv=None, cold=False):
    dt = torch.float16
    if conv:
        B, H, Cin = conv
        A = torch.randn(B * H * H, Cin, device=dev).to(dt); lda = Cin; geo = (H, H, Cin, H, H, 1, 0); mode = 1
    else:
        A = torch.randn(M, K, device=dev).to(dt); lda = K; geo = (0, 0, 0, 0, 0, 1, 0); mode = 0
    W = torch.randn(N, K, device=dev).to(dt)
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev).to(dt)
    C = torch.empty(M, N, dtype=dt, device=dev)
    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev) if cold else None       # 1 GiB: evicts L2 / MALL

    def call():
        L.dh_dbg_gemm(0, P(A), lda, P(W), M, N, K, mode, *geo, P(bias), P(None), 0, 1, P(R), N, P(C), N, 0, P(part), part.numel(), _lib.stream_ptr())
    for _ in range(3):
        call()
    acc = torch.zeros(7, dtype=torch.float64)
    wait = torch.zeros(40, dtype=torch.float64)
    work = torch.zeros(40, dtype=torch.float64)
    n = 10
    for _ in range(n):
        if cold:
            junk.add_(1.0)
        ts.zero_()
        L.dh_dbg_gemm_timeline(P(ts))
        call()
        L.dh_dbg_gemm_timeline(None)
        torch.cuda.synchronize()
        t = ts.cpu().double()
        acc += t[1:8] - t[:7]
        tt = t[8:].view(40, 3)
        ok = tt[:, 2] > 0
        wait += torch.where(ok, tt[:, 1] - tt[:, 0], torch.zeros(40, dtype=torch.float64))
        work += torch.where(ok, tt[:, 2] - tt[:, 1], torch.zeros(40, dtype=torch.float64))
        ntl = int(ok.sum())
    acc /= n
    wait /= n
    work /= n
    tot = acc.sum().item()
    print(f"M={M} N={N} K={K} {'conv' if conv else 'dense'} {'COLD' if cold else 'warm'}: total {tot:.0f} ticks")
    for nm, v in zip(NAMES, acc.tolist()):
        print(f"    {nm:32s} {v:8.0f} ticks {100 * v / tot:5.1f} %")
    # per K tile of workgroup (0,0,0), wave 0 (its own K range when the tile splits K over wave groups): ticks (10 ns) spent waiting
    # for the tile to land (counted vmcnt + barrier) and spent reading fragments / multiplying / issuing the next tile's DMA
    if ntl:
        w, k = wait[:ntl], work[:ntl]
        print(f"    K loop, {ntl} tiles of wave 0: waiting for the tile {w.sum().item():7.0f} ticks ({100 * w.sum().item() / (w.sum().item() + k.sum().item()):4.1f} %), "
              f"fragments + MFMA + next DMA issue {k.sum().item():7.0f} ticks; per tile wait {w.mean().item():5.1f} work {k.mean().item():5.1f}")
        print("      wait per tile:", " ".join(f"{v:.0f}" for v in w.tolist()))
        print("      work per tile:", " ".join(f"{v:.0f}" for v in k.tolist()))


# calibrate ticks: a kernel of known duration would be needed; the shares are what matters here
for cold in (False, True):
    run(4096, 320, 320, cold=cold)
    run(1024, 640, 640, cold=cold)
    run(256, 1280, 1280, cold=cold)
    run(4096, 320, 2880, (1, 64, 320), cold=cold)
    run(256, 1280, 11520, (1, 16, 1280), cold=cold)
    run(64, 1280, 11520, (1, 8, 1280), cold=cold)
