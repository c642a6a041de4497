#!/usr/bin/env python3
"""Randomised BIT-LEVEL sweep of the two staging forms of k_gemm_dma (round 6): the buffer-descriptor form (`BUF`, shipped) and the
address form of rounds 1-5 (dh_dbg_gemm_stage(0)) run the same tiles, the same K order and the same epilogue -- only how a tile gets
into LDS differs (scalar K cursor / tap offset + hardware zero fill against 64-bit addresses + a zero page) -- so every output must be
IDENTICAL, bit for bit.  Random shapes: ragged row counts from one tile to a few hundred, every column count the engines use, dense
with a column window of a wider matrix (lda > K), 3x3 stride-1 convolutions at every image size of the U-Net levels incl. several
images, bias / per-image vector / residual, with and without K split, the GEGLU epilogues, fp16 and bf16; k_gemm_pp is switched off
so that every case runs on k_gemm_dma.  GroupNorm-epilogue statistics are switched off in both arms (they do not depend on staging).
    python3 tools/fuzz_gemm_stage.py [cases] [seed]       exit code 1 on the first difference (the case is printed)"""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
DT = {torch.float16: 0, torch.bfloat16: 1}
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
part = torch.empty(48 << 20, dtype=torch.float32, device=dev)


def gemm(dtype, A, lda, W, M, N, K, mode, geo, bias, rowvec, rpb, R, split):
    C = torch.full((M, N), float("nan"), dtype=dtype, device=dev)
    _lib.check(L.dh_dbg_gemm(DT[dtype], P(A), lda, P(W), M, N, K, mode, *geo, P(bias), P(rowvec), rowvec.shape[1] if rowvec is not None else 0, rpb,
                             P(R), N, P(C), N, 0, P(part) if split else P(None), part.numel() if split else 0, _lib.stream_ptr()), "dh_dbg_gemm")
    return C


L.dh_dbg_gemm_family(1)                       # k_gemm_dma only
try:
    for ci in range(cases):
        dtype = rnd.choice([torch.float16, torch.bfloat16])
        g = torch.Generator(device=dev).manual_seed(seed * 100003 + ci)
        kind = rnd.choice(["dense", "dense", "window", "conv", "conv", "glu_fwd", "glu_bwd"])
        N = rnd.choice([64, 128, 192, 256, 320, 640, 960, 1280, 2560])
        K = 64 * rnd.randint(1, 40)
        case = dict(ci=ci, kind=kind, dtype=str(dtype))
        if kind == "conv":
            H = rnd.choice([8, 16, 24, 32, 64]); Bn = rnd.randint(1, 3 if H >= 32 else 6); Cin = 64 * rnd.randint(1, 6)
            M, K = Bn * H * H, 9 * Cin
            A = torch.randn(M, Cin, generator=g, device=dev).to(dtype); lda = Cin
            geo, mode, rpb = (H, H, Cin, H, H, 1, 0), 1, H * H
        elif kind == "window":                                   # A = a column window of a wider row-major matrix (the engine's q | k | v split)
            M = rnd.choice([rnd.randint(1, 300), rnd.randint(300, 5000)])
            wide = K + 64 * rnd.randint(1, 4); col = 64 * rnd.randint(0, (wide - K) // 64)
            full = torch.randn(M, wide, generator=g, device=dev).to(dtype)
            A = full[:, col:]; lda = wide
            geo, mode, rpb = (0, 0, 0, 0, 0, 1, 0), 0, max(1, M // 2)
        else:
            M = rnd.choice([rnd.randint(1, 300), rnd.randint(300, 5000), rnd.randint(5000, 20000)])
            A = torch.randn(M, K, generator=g, device=dev).to(dtype); lda = K
            geo, mode, rpb = (0, 0, 0, 0, 0, 1, 0), 0, rnd.choice([M, max(1, M // 2), 256, 4096])
        case.update(M=M, N=N, K=K)
        if kind in ("dense", "window", "conv"):
            W = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).to(dtype)
            bias = torch.randn(N, generator=g, device=dev) if rnd.random() < 0.7 else None
            R = torch.randn(M, N, generator=g, device=dev).to(dtype) if rnd.random() < 0.5 else None
            nimg = (M + rpb - 1) // rpb
            rowvec = torch.randn(nimg, N, generator=g, device=dev) if rnd.random() < 0.3 else None
            split = rnd.random() < 0.5
            case.update(bias=bias is not None, R=R is not None, rowvec=rowvec is not None, rpb=rpb, split=split)
            outs = []
            for stage in (1 | 4, 0 | 4):
                L.dh_dbg_gemm_stage(stage)
                outs.append(gemm(dtype, A, lda, W, M, N, K, mode, geo, bias, rowvec, rpb, R, split))
            same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
        else:
            Fd = rnd.choice([128, 256, 640, 1280])
            K = 64 * rnd.randint(1, 20)
            A = torch.randn(M, K, generator=g, device=dev).to(dtype)
            case.update(Fd=Fd, K=K)
            outs = []
            for stage in (1 | 4, 0 | 4):
                L.dh_dbg_gemm_stage(stage)
                if kind == "glu_fwd":
                    W = (torch.randn(2 * Fd, K, generator=torch.Generator(device=dev).manual_seed(ci), device=dev) / K ** 0.5).to(dtype)
                    pre = torch.empty(M, 2 * Fd, dtype=dtype, device=dev); y = torch.empty(M, Fd, dtype=dtype, device=dev)
                    _lib.check(L.dh_dbg_gemm_glu(DT[dtype], 0, P(A), K, P(W), M, 2 * Fd, K, P(None), P(pre), P(y), P(None), P(None), _lib.stream_ptr()), "glu fwd")
                    outs.append(torch.cat([pre, y], dim=1))
                else:
                    Wb = (torch.randn(Fd, K, generator=torch.Generator(device=dev).manual_seed(ci), device=dev) / K ** 0.5).to(dtype)
                    pre = torch.randn(M, 2 * Fd, generator=torch.Generator(device=dev).manual_seed(ci + 1), device=dev).to(dtype)
                    dx = torch.empty(M, 2 * Fd, dtype=dtype, device=dev)
                    _lib.check(L.dh_dbg_gemm_glu(DT[dtype], 1, P(A), K, P(Wb), M, Fd, K, P(None), P(None), P(None), P(pre), P(dx), _lib.stream_ptr()), "glu bwd")
                    outs.append(dx)
            same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
        if not same or not torch.isfinite(outs[0].float()).all():
            d = (outs[0].float() - outs[1].float()).abs()
            print(f"DIFFERENCE: {case}: {int((d > 0).sum())} elements differ, max |diff| {d.max().item():.4g}, finite {bool(torch.isfinite(outs[0].float()).all())}")
            sys.exit(1)
        if (ci + 1) % 50 == 0:
            print(f"{ci + 1} cases bit-identical", flush=True)
finally:
    L.dh_dbg_gemm_stage(1)
    L.dh_dbg_gemm_family(0)
print(f"{cases} cases, buffer-descriptor staging bit-identical to the address form (seed {seed})")
