#!/usr/bin/env python3
"""Randomised parity sweep of k_gemm_pp (csrc/gemm_pp.hip) against torch fp32 and against k_gemm_dma: random row counts (ragged,
from one tile to several rounds of persistent workgroups), column / K sizes the kernel carries, dense and 3x3 stride-1
convolution, every combination of bias / residual / per-image vector, fp16 and bf16, and both GEGLU epilogues.
    python3 tools/fuzz_gemm_pp.py [cases] [seed]       exit code 1 on the first mismatch (the case is printed)"""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from diffusionhandles_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
DT = {torch.float16: 0, torch.bfloat16: 1}
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
part = torch.empty(48 << 20, dtype=torch.float32, device=dev)


def gemm(dtype, A, lda, W, M, N, K, mode, geo, bias, rowvec, rpb, R, split):
    C = torch.empty(M, N, dtype=dtype, device=dev)
    _lib.check(L.dh_dbg_gemm(DT[dtype], P(A), lda, P(W), M, N, K, mode, *geo, P(bias), P(rowvec), rowvec.shape[1] if rowvec is not None else 0, rpb,
                             P(R), N, P(C), N, 0, P(part) if split else P(None), part.numel() if split else 0, _lib.stream_ptr()), "dh_dbg_gemm")
    return C


def check(got, ref, tol, what, case):
    err = (got.float() - ref.float()).abs()
    lim = tol + tol * ref.float().abs()
    bad = (err > lim).float().mean().item()
    if bad > 1e-4 or not torch.isfinite(got.float()).all():
        print(f"MISMATCH {what}: {case}: max err {err.max().item():.4g}, frac bad {bad:.3g}")
        sys.exit(1)


worst = 0.0
for ci in range(cases):
    dtype = rnd.choice([torch.float16, torch.bfloat16])
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    g = torch.Generator(device=dev).manual_seed(seed * 100003 + ci)
    kind = rnd.choice(["dense", "dense", "conv", "glu_fwd", "glu_bwd"])
    N = rnd.choice([128, 256, 320, 384, 640, 960, 1280])          # (the debug entry takes multiples of 64; 320 = two 160-column tiles)
    K = 64 * rnd.randint(1, 24)
    if kind == "conv":
        H = rnd.choice([8, 16, 24, 32, 48, 64]); Bn = rnd.randint(1, 12 if H <= 32 else 9); Cin = 64 * rnd.randint(1, 4)
        M, K = Bn * H * H, 9 * Cin
        x = torch.randn(Bn, Cin, H, H, generator=g, device=dev).to(dtype)
        w = (torch.randn(N, Cin, 3, 3, generator=g, device=dev) / (9 * Cin) ** 0.5).to(dtype)
        A = x.permute(0, 2, 3, 1).contiguous().reshape(M, Cin); lda = Cin
        W = w.permute(0, 2, 3, 1).reshape(N, K).contiguous()
        geo, mode = (H, H, Cin, H, H, 1, 0), 1
        rpb = H * H
    else:
        M = rnd.choice([rnd.randint(1, 600), rnd.randint(600, 9000), rnd.randint(9000, 70000)])
        A = torch.randn(M, K, generator=g, device=dev).to(dtype); lda = K
        geo, mode = (0, 0, 0, 0, 0, 1, 0), 0
        rpb = rnd.choice([M, max(1, M // 2), 256, 4096])
    case = dict(ci=ci, kind=kind, dtype=str(dtype), M=M, N=N, K=K)
    if kind in ("dense", "conv"):
        if kind == "dense":
            W = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).to(dtype)
        bias = torch.randn(N, generator=g, device=dev) if rnd.random() < 0.7 else None
        R = torch.randn(M, N, generator=g, device=dev).to(dtype) if rnd.random() < 0.5 else None
        nimg = (M + rpb - 1) // rpb
        rowvec = torch.randn(nimg, N, generator=g, device=dev) if rnd.random() < 0.3 else None
        split = rnd.random() < 0.5
        case.update(bias=bias is not None, R=R is not None, rowvec=rowvec is not None, rpb=rpb, split=split)
        ref = (F.conv2d(x.float(), w.float(), padding=1).permute(0, 2, 3, 1).reshape(M, N) if kind == "conv" else A.float() @ W.float().t())
        if bias is not None: ref = ref + bias
        if rowvec is not None: ref = ref + rowvec.repeat_interleave(rpb, dim=0)[:M]
        if R is not None: ref = ref + R.float()
        outs = {}
        for fam in (2, 1):
            L.dh_dbg_gemm_family(fam)
            outs[fam] = gemm(dtype, A, lda, W, M, N, K, mode, geo, bias, rowvec, rpb, R, split)
        L.dh_dbg_gemm_family(0)
        check(outs[2], ref, tol, "k_gemm_pp vs torch", case)
        check(outs[2], outs[1], 2 * tol, "k_gemm_pp vs k_gemm_dma", case)
        worst = max(worst, ((outs[2].float() - ref).abs() / (1 + ref.abs())).max().item())
    else:
        Fd = rnd.choice([128, 256, 640, 1280])                   # GEGLU width: the GEMM has 2 Fd (forward) / Fd (backward) columns
        if kind == "glu_fwd":
            W = (torch.randn(2 * Fd, K, generator=g, device=dev) / K ** 0.5).to(dtype)
            bias = torch.randn(2 * Fd, generator=g, device=dev) if rnd.random() < 0.7 else None
            save = rnd.random() < 0.5
            case.update(Fd=Fd, bias=bias is not None, save=save)
            res = {}
            for fam in (2, 1):
                L.dh_dbg_gemm_family(fam)
                pre = torch.empty(M, 2 * Fd, dtype=dtype, device=dev) if save else None
                y = torch.empty(M, Fd, dtype=dtype, device=dev)
                _lib.check(L.dh_dbg_gemm_glu(DT[dtype], 0, P(A), K, P(W), M, 2 * Fd, K, P(bias), P(pre), P(y), P(None), P(None), _lib.stream_ptr()), "glu fwd")
                res[fam] = (pre, y)
            L.dh_dbg_gemm_family(0)
            check(res[2][1], res[1][1], 2 * tol, "GEGLU forward y, k_gemm_pp vs k_gemm_dma", case)
            if save: check(res[2][0], res[1][0], 2 * tol, "GEGLU forward pre-activations", case)
        else:
            K2 = K
            A2 = torch.randn(M, K2, generator=g, device=dev).to(dtype)
            Wb = (torch.randn(Fd, K2, generator=g, device=dev) / K2 ** 0.5).to(dtype)
            pre = torch.randn(M, 2 * Fd, generator=g, device=dev).to(dtype)
            case.update(Fd=Fd)
            res = {}
            for fam in (2, 1):
                L.dh_dbg_gemm_family(fam)
                dx = torch.empty(M, 2 * Fd, dtype=dtype, device=dev)
                _lib.check(L.dh_dbg_gemm_glu(DT[dtype], 1, P(A2), K2, P(Wb), M, Fd, K2, P(None), P(None), P(None), P(pre), P(dx), _lib.stream_ptr()), "glu bwd")
                res[fam] = dx
            L.dh_dbg_gemm_family(0)
            check(res[2], res[1], 4 * tol, "GEGLU backward, k_gemm_pp vs k_gemm_dma", case)
    if (ci + 1) % 50 == 0:
        print(f"{ci + 1} cases ok", flush=True)
print(f"{cases} cases, no mismatch (seed {seed}); worst |err| / (1 + |ref|) of the plain cases {worst:.3g}")
