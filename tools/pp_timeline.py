#!/usr/bin/env python3
"""In-kernel timeline of k_gemm_pp (stamping variants of tools/bin/libdh_pp_variants.so): s_memtime of waves 0 (group 0) and 4
(group 1) of workgroup 0 at the seams of every segment -> cycles spent in: fragment reads issued, DMA issue, wait for LDS,
barrier, MFMA segment, second barrier.  DIFFHANDLES_LIB=tools/bin/libdh_pp_variants.so python3 tools/pp_timeline.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DH_DBG_PRETILED", "1")
import torch
from diffusionhandles_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
part = torch.empty(64 << 20, dtype=torch.float32, device=dev)

def run(M, N, K, conv, var):
    dt = torch.float16
    g = torch.Generator(device=dev).manual_seed(1)
    if conv:
        Bn, H, Cin = conv
        A = (torch.rand(Bn * H * H, Cin, generator=g, device=dev) * 2 - 1).to(dt); lda = Cin; geo = (H, H, Cin, H, H, 1, 0); mode = 1
    else:
        A = (torch.rand(M, K, generator=g, device=dev) * 2 - 1).to(dt); lda = K; geo = (0, 0, 0, 0, 0, 1, 0); mode = 0
    W = (torch.rand(N, K, generator=g, device=dev) * 2 - 1).to(dt)
    C = torch.empty(M, N, dtype=dt, device=dev)
    ts = torch.zeros(1024, dtype=torch.int64, device=dev)
    L.dh_dbg_gemm_family(2)
    for it in range(3):
        ts.zero_()
        L.dh_dbg_gemm_pp_variant(var | 1, P(ts))
        L.dh_dbg_gemm(0, P(A), lda, P(W), M, N, K, mode, *geo, P(None), P(None), 0, 1, P(None), N, P(C), N, 0, P(part), part.numel(), _lib.stream_ptr())
        torch.cuda.synchronize()
    L.dh_dbg_gemm_family(0); L.dh_dbg_gemm_pp_variant(-1, None)
    t = ts.cpu().numpy()
    seg = 1 if var & 4 else 2
    names = ["reads issued", "DMA issued", "vm wait", "lds wait", "barrier A", "MFMA", "(wait+) barrier B"]
    print(f"== M={M} N={N} K={K} {'conv' if conv else 'dense'} variant {var}: cycles per segment (median over the steady tiles), {K // 64} K tiles x {seg} segments")
    for grp in (0, 1):
        s = t[grp * 512:(grp + 1) * 512]
        n = int((s != 0).sum())
        s = s[:n]
        # layout: [0] tile start (persistent loop), [1] in front of the prologue, [2] after the first wait; then 6 stamps per segment;
        # then the loop end and the epilogue's stamps (ignored here: tools/pp_tile_timeline.py)
        s = s[1:]
        body = s[2:]
        nseg = min((K // 64) * (1 if var & 4 else 2), (len(body) - 1) // 6)      # (512 stamps per group: the first ~84 segments)
        import numpy as np
        d = np.diff(body[:nseg * 6 + 1].astype(np.int64)).reshape(nseg, 6)
        mid = d[nseg // 4: max(nseg // 4 + 1, nseg - 2)]
        med = np.median(mid, axis=0)
        print(f"  group {grp}: prologue {int(s[1] - s[0])} cyc; per segment: " + ", ".join(f"{nm} {int(v)}" for nm, v in zip(
            ["reads issued", "DMA issued", "waits (vm, lds)", "barrier A", "MFMA", "wait + barrier B"], med)) + f" | total {int(med.sum())}; whole loop {int(body[-1] - body[0])} cyc for {nseg} segments")

shapes = [(32768, 320, 2880, (8, 64, 320)), (8192, 640, 5760, (8, 32, 640)), (32768, 320, 320, None)]
if os.environ.get("DH_PP_SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["DH_PP_SHAPES"].split(",")]
for var in [int(v) for v in os.environ.get("DH_PP_VARS", "20,52").split(",")]:
    for s in shapes:
        run(*s, var)
