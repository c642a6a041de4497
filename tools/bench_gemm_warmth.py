#!/usr/bin/env python3
"""How much of a weight-streaming GEMM's time is the weights being HBM-cold?  Three regimes per shape, read from a
rocprofv3 kernel trace (k_gemm_dma + its split-K reduce): warm (back-to-back launches), cold (a 1 GiB streaming pass in between
evicts L2 and the Infinity Cache: the state in the step), prefetched (evict, then one pass over the tiled weights alone, then the GEMM).
  run:   DH_DBG_PRETILED=1 rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/bench_gemm_warmth.py run
  parse: python3 tools/bench_gemm_warmth.py parse DIR"""
import csv, ctypes, glob, os, sys
SHAPES = [(64, 1280, 11520, (1, 8, 1280)), (256, 1280, 11520, (1, 16, 1280)), (256, 1280, 1280, None), (64, 1280, 1280, None),
          (1024, 640, 5760, (1, 32, 640)), (1024, 640, 640, None)]
REPS = 8


def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from diffusionhandles_amd import _lib
    dev = torch.device("cuda:0")
    L = _lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    part = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    for M, N, K, conv in SHAPES:
        dt = torch.float16
        if conv:
            B, H, Cin = conv
            A = torch.randn(B * H * H, Cin, device=dev).to(dt); lda = Cin; geo = (H, H, Cin, H, H, 1, 0); mode = 1
        else:
            A = torch.randn(M, K, device=dev).to(dt); lda = K; geo = (0, 0, 0, 0, 0, 1, 0); mode = 0
        W = torch.randn(N, K, device=dev).to(dt)
        bias = torch.randn(N, device=dev)
        C = torch.empty(M, N, dtype=dt, device=dev)
        def call():
            L.dh_dbg_gemm(0, P(A), lda, P(W), M, N, K, mode, *geo, P(bias), P(None), 0, 1, P(None), N, P(C), N, 0, P(part), part.numel(), _lib.stream_ptr())
        call()                                   # tiles the weights into the hook's scratch buffer (once per matrix under DH_DBG_PRETILED=1)
        torch.cuda.synchronize()
        for regime in ("warm", "cold", "prefetched"):
            for _ in range(REPS):
                if regime != "warm":
                    junk.add_(1.0)
                if regime == "prefetched":
                    L.dh_dbg_touch_tiled(ctypes.c_size_t(N * K * 2), _lib.stream_ptr())
                call()
            torch.cuda.synchronize()


def parse(d):
    f = glob.glob(d + "/*/*kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    seq = []
    for r in rows:
        n = r["Kernel_Name"]
        t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "k_gemm_dma" in n:
            seq.append([t, 0.0])
        elif "k_splitk_reduce" in n and seq:
            seq[-1][1] += t
    i = 0
    for M, N, K, conv in SHAPES:
        i += 1                                    # the tiling call
        out = []
        for regime in ("warm", "cold", "prefetched"):
            seg = sorted(a for a, b in seq[i:i + REPS]); red = sorted(b for a, b in seq[i:i + REPS]); i += REPS
            out.append(f"{regime} {seg[len(seg)//2]:6.1f} (+{red[len(red)//2]:4.1f} reduce)")
        print(f"M={M:5d} N={N:5d} K={K:6d} {'conv ' if conv else 'dense'}: " + "   ".join(out) + "  us")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else parse(sys.argv[2])
