"""GPU parity of each U-Net kernel (through the debug C-ABI hooks) against plain PyTorch
fp32 references of the same op.  Tolerances are fp16/bf16-storage tolerances: the kernels
read 16-bit inputs, accumulate in f32 and round the output once."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {torch.float16: 0, torch.bfloat16: 1}


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def L():
    from diffusionhandles_amd import _lib
    return _lib


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def close(got, ref, rtol, atol, what=""):
    err = (got.float() - ref.float()).abs()
    tol = atol + rtol * ref.float().abs()
    bad = (err > tol).float().mean().item()
    assert bad < 1e-4, f"{what}: max err {err.max().item():.4g}, frac bad {bad:.3g}, ref max {ref.abs().max().item():.3g}"


def run_gemm(dtype, A, lda, W, M, N, K, mode=0, geo=(0, 0, 0, 0, 0, 1, 0), bias=None, rowvec=None, rpb=1, R=None, silu=0,
             split=True):
    lib = L().lib()
    C = torch.empty((M, N), dtype=dtype, device=dev())
    part = torch.empty(16 << 20, dtype=torch.float32, device=dev()) if split else None
    Hin, Win, Cin, Hout, Wout, stride, up = geo
    rc = lib.dh_dbg_gemm(DT[dtype], P(A), lda, P(W), M, N, K, mode, Hin, Win, Cin, Hout, Wout, stride, up, P(bias),
                         P(rowvec), rowvec.shape[1] if rowvec is not None else 0, rpb, P(R), N, P(C), N, silu, P(part),
                         part.numel() if part is not None else 0, L().stream_ptr())
    L().check(rc, "dh_dbg_gemm")
    return C


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(4096, 320, 320), (1024, 640, 2560), (256, 1280, 1280), (64, 1280, 1280),
                                     (77, 2560, 1024), (3, 1280, 320), (4096, 2560, 320), (200, 128, 6400),
                                     (28800, 960, 320), (28700, 320, 640),       # >= 224 row tiles: the eight-wave 128x320 tile (N = 3 x 320; ragged M)
                                     (8192, 640, 1280), (7300, 640, 320), (2048, 2560, 640)])    # 224..256 tiles of 128x160 (column tiles start inside a 64-row weight tile; ragged M)
def test_gemm_dense(dtype, M, N, K):
    g = torch.Generator(device=dev()).manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g, device=dev()).to(dtype)
    W = (torch.randn(N, K, generator=g, device=dev()) / K ** 0.5).to(dtype)
    bias = torch.randn(N, generator=g, device=dev())
    R = torch.randn(M, N, generator=g, device=dev()).to(dtype)
    ref = A.float() @ W.float().t() + bias + R.float()
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    for split in (True, False):
        C = run_gemm(dtype, A, K, W, M, N, K, bias=bias, R=R, split=split)
        close(C, ref, tol, tol, f"gemm {M}x{N}x{K} split={split}")
    # asymmetric operands catch a transposed output: also check silu + rowvec path
    rv = torch.randn(2, N, generator=g, device=dev())
    if M % 2 == 0:
        C = run_gemm(dtype, A, K, W, M, N, K, rowvec=rv, rpb=M // 2, silu=1)
        z = A.float() @ W.float().t() + rv.repeat_interleave(M // 2, dim=0)
        close(C, F.silu(z), tol, tol, "gemm silu/rowvec")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(4096, 960, 320), (1024, 640, 640), (256, 2560, 1280)])
@pytest.mark.parametrize("mean_over_std", [0.0, 3.0, 30.0])
def test_gemm_layernorm_fold(dtype, M, N, K, mean_over_std):
    """The LayerNorm folded into its consuming GEMM (A = the LayerNorm input, W * gamma, out = rstd (A W'^T - mean s) + t; the
    row sums come out of the K loop as f32 sums of the 16-bit inputs, variance = E[x^2] - mean^2) against LayerNorm followed by
    the GEMM in fp32, for rows whose mean is 0, 3 and 30 standard deviations away from zero: the one-pass variance could lose
    accuracy as |mean| / std grows (advisor, round 2).  Measured: it does not matter at 16-bit inputs -- rel-L2 2.6e-4 (fp16) /
    2.0e-3 (bf16) even at 30 standard deviations, the saved (mean, rstd) within 1e-3 / 2e-2 relative."""
    g = torch.Generator(device=dev()).manual_seed(M + N + K + int(mean_over_std))
    x = (torch.randn(M, K, generator=g, device=dev()) + mean_over_std * torch.randn(M, 1, generator=g, device=dev()).sign()).to(dtype)
    gamma = 1.0 + 0.1 * torch.randn(K, generator=g, device=dev())
    beta = 0.1 * torch.randn(K, generator=g, device=dev())
    W0 = (torch.randn(N, K, generator=g, device=dev()) / K ** 0.5).to(dtype)
    bias = torch.randn(N, generator=g, device=dev())
    Wf = (W0.float() * gamma).to(dtype)                                  # what k_fold_ln stores
    s = Wf.float().sum(dim=1).contiguous()
    t = (W0.float() @ beta + bias).contiguous()
    ref = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ W0.float().t() + bias
    C = torch.empty(M, N, dtype=dtype, device=dev())
    stats = torch.empty(M, 2, dtype=torch.float32, device=dev())
    lib = L().lib()
    L().check(lib.dh_dbg_gemm_lnfold(DT[dtype], P(x), K, P(Wf.contiguous()), M, N, K, P(s), P(t), P(stats), 1e-5, P(C), N,
                                     L().stream_ptr()), "dh_dbg_gemm_lnfold")
    mean, var = x.float().mean(dim=1), x.float().var(dim=1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    assert torch.allclose(stats[:, 0], mean, rtol=1e-3, atol=1e-4)
    assert torch.allclose(stats[:, 1], rstd, rtol=2e-2 if mean_over_std > 10 else 2e-3, atol=0)
    if mean_over_std <= 3.0:
        tol = 6e-3 if dtype == torch.float16 else 3e-2
        close(C, ref, tol, tol, f"LN-folded gemm {M}x{N}x{K} mean/std {mean_over_std}")
    else:       # mean * s is ~30x the result it is subtracted from: measured 2.6e-4 (fp16) / 2.0e-3 (bf16)
        err = ((C.float() - ref).norm() / ref.norm()).item()
        print(f"LN-folded gemm {dtype} {M}x{N}x{K} at |mean| = {mean_over_std} std: rel L2 {err:.3e}")
        assert err < (2e-3 if dtype == torch.float16 else 1e-2)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,Cin,Cout,H,stride,up", [(1, 320, 320, 64, 1, 0), (2, 640, 320, 32, 1, 0), (1, 320, 320, 64, 2, 0),
                                                     (1, 1280, 1280, 8, 1, 1), (2, 64, 128, 16, 1, 0), (1, 1920, 640, 32, 1, 0),
                                                     (2, 128, 128, 16, 2, 0), (1, 640, 640, 32, 1, 1),
                                                     (8, 320, 640, 32, 1, 0)])      # M = 8192, N = 640: the 128x160 tile
def test_conv3_forward_and_input_gradient(dtype, B, Cin, Cout, H, stride, up):
    g = torch.Generator(device=dev()).manual_seed(Cin + Cout + H + stride + up)
    x = torch.randn(B, Cin, H, H, generator=g, device=dev()).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, device=dev()) / (9 * Cin) ** 0.5).to(dtype)
    bias = torch.randn(Cout, generator=g, device=dev())
    xr = x.float().requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2.0, mode="nearest") if up else xr
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1)
    Ho = ref.shape[-1]
    wf = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()                      # [Cout][tap][Cin]
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    C = run_gemm(dtype, nhwc(x), Cin, wf, B * Ho * Ho, Cout, 9 * Cin, mode=1, geo=(H, H, Cin, Ho, Ho, stride, up), bias=bias)
    close(C.view(B, Ho, Ho, Cout), nhwc(ref), tol, tol, "conv fwd")
    # input gradient through the same kernel with flipped / transposed weights
    dy = torch.randn(ref.shape, generator=g, device=dev()).to(dtype)
    gref, = torch.autograd.grad(ref, xr, dy.float())
    wb = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).contiguous()           # [Cin][tap'][Cout]
    if up:
        hi = run_gemm(dtype, nhwc(dy), Cout, wb, B * Ho * Ho, Cin, 9 * Cout, mode=1, geo=(Ho, Ho, Cout, Ho, Ho, 1, 0))
        dx = torch.empty((B, H, H, Cin), dtype=dtype, device=dev())
        L().check(L().lib().dh_dbg_pool2x2(DT[dtype], P(hi), P(dx), B, H, H, Cin, 0, L().stream_ptr()))
    elif stride == 2:
        dx = run_gemm(dtype, nhwc(dy), Cout, wb, B * H * H, Cin, 9 * Cout, mode=2, geo=(Ho, Ho, Cout, H, H, 1, 0)).view(B, H, H, Cin)
    else:
        dx = run_gemm(dtype, nhwc(dy), Cout, wb, B * H * H, Cin, 9 * Cout, mode=1, geo=(H, H, Cout, H, H, 1, 0)).view(B, H, H, Cin)
    close(dx.reshape(B, H, H, Cin), nhwc(gref), tol, tol * gref.abs().max().item(), "conv dX")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,HW,C,silu", [(1, 4096, 320, 1), (2, 1024, 960, 1), (1, 64, 2560, 1), (2, 256, 1280, 0), (1, 4096, 64, 1),
                                          (1, 4096, 640, 1), (1, 9216, 320, 0), (3, 576, 1920, 1)])
def test_groupnorm(dtype, B, HW, C, silu):
    g = torch.Generator(device=dev()).manual_seed(C + HW)
    x = (torch.randn(B, HW, C, generator=g, device=dev()) * 2 + 0.5).to(dtype)
    gamma = 1 + 0.1 * torch.randn(C, generator=g, device=dev())
    beta = 0.1 * torch.randn(C, generator=g, device=dev())
    dy = torch.randn(B, HW, C, generator=g, device=dev()).to(dtype)
    acc0 = torch.randn(B, HW, C, generator=g, device=dev()).to(dtype)
    xr = x.float().requires_grad_(True)
    ref = F.group_norm(xr.transpose(1, 2), 32, gamma, beta, eps=1e-5).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    gref, = torch.autograd.grad(ref, xr, dy.float())
    y = torch.empty_like(x); dx = acc0.clone()
    stats = torch.empty(B * 32 * 2, dtype=torch.float32, device=dev()); scr = torch.empty(B * 32 * 32 * 3 + 64, dtype=torch.float32, device=dev())
    L().check(L().lib().dh_dbg_groupnorm(DT[dtype], P(x), P(gamma), P(beta), P(y), P(stats), P(dy), P(dx), P(scr), B, HW, C, 32,
                                         1e-5, silu, 1, L().stream_ptr()))
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    close(y, ref, tol, tol, "gn fwd")
    close(dx, gref + acc0.float(), tol, tol * 2, "gn bwd (accumulate)")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(4096, 320), (1024, 640), (77, 1280), (300, 64)])
def test_layernorm(dtype, rows, C):
    g = torch.Generator(device=dev()).manual_seed(C + rows)
    x = (torch.randn(rows, C, generator=g, device=dev()) * 1.5 - 0.3).to(dtype)
    gamma = 1 + 0.1 * torch.randn(C, generator=g, device=dev()); beta = 0.1 * torch.randn(C, generator=g, device=dev())
    dy = torch.randn(rows, C, generator=g, device=dev()).to(dtype); add = torch.randn(rows, C, generator=g, device=dev()).to(dtype)
    xr = x.float().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
    gref, = torch.autograd.grad(ref, xr, dy.float())
    y = torch.empty_like(x); dx = torch.empty_like(x); stats = torch.empty(rows * 2, dtype=torch.float32, device=dev())
    L().check(L().lib().dh_dbg_layernorm(DT[dtype], P(x), P(gamma), P(beta), P(y), P(stats), P(dy), P(add), P(dx), rows, C, 1e-5, L().stream_ptr()))
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    close(y, ref, tol, tol, "ln fwd")
    close(dx, gref + add.float(), tol, tol * 2, "ln bwd")


def glu_paired_index(Fd):
    """Stored column n' of a GEGLU pre-activation tensor [rows][2F] -> column of the torch layout (value | gate): every 32-column
    block holds the 16 value columns of outputs 16b .. 16b+15, then their 16 gate columns (csrc/unet_kernels.h glu_src_row)."""
    n = torch.arange(2 * Fd)
    return ((n >> 4) & 1) * Fd + 16 * (n >> 5) + (n & 15)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_geglu(dtype):
    g = torch.Generator(device=dev()).manual_seed(5)
    rows, Fd = 1000, 1280
    x = torch.randn(rows, 2 * Fd, generator=g, device=dev()).to(dtype); dy = torch.randn(rows, Fd, generator=g, device=dev()).to(dtype)
    xr = x.float().requires_grad_(True)
    h, gt = xr.chunk(2, dim=-1)
    ref = h * F.gelu(gt)
    gref, = torch.autograd.grad(ref, xr, dy.float())
    idx = glu_paired_index(Fd).to(dev())
    xp = x[:, idx].contiguous()                      # the engine keeps this tensor in the paired column order
    y = torch.empty(rows, Fd, dtype=dtype, device=dev()); dxp = torch.empty_like(xp)
    L().check(L().lib().dh_dbg_geglu(DT[dtype], P(xp), P(y), P(dy), P(dxp), rows, Fd, L().stream_ptr()))
    dx = torch.empty_like(dxp); dx[:, idx] = dxp
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    close(y, ref, tol, tol, "geglu fwd"); close(dx, gref, tol, tol, "geglu bwd")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,Fd,K", [(4096, 1280, 320), (1024, 2560, 640), (256, 5120, 1280), (64, 5120, 1280), (77, 256, 64),
                                      (8192, 1280, 320)])
def test_gemm_geglu_epilogues(dtype, M, Fd, K):
    """GEGLU in the GEMM epilogues (reference model/attention.py:345-400 FeedForward: ff.net.0 = GEGLU(dim, 4 dim), ff.net.2 =
    Linear): forward y = value * gelu(gate) out of the ff.net.0.proj GEMM (+ the paired-layout pre-activations when asked for),
    backward d_value | d_gate out of the input-gradient GEMM of ff.net.2, against torch fp32 autograd.  Shapes = the three tile
    families of the SD-2 levels (256x128, 128x128, 64x64), a ragged M and a batch-2 M."""
    g = torch.Generator(device=dev()).manual_seed(M + Fd + K)
    lib = L().lib()
    A = torch.randn(M, K, generator=g, device=dev()).to(dtype)
    W = (torch.randn(2 * Fd, K, generator=g, device=dev()) / K ** 0.5).to(dtype)
    bias = 0.5 * torch.randn(2 * Fd, generator=g, device=dev())
    idx = glu_paired_index(Fd).to(dev())
    Wp, bp = W[idx].contiguous(), bias[idx].contiguous()
    pre_ref = A.float() @ W.float().t() + bias
    pre16 = pre_ref.to(dtype).float()
    y_ref = pre16[:, :Fd] * F.gelu(pre16[:, Fd:])
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    for save in (True, False):
        pre = torch.zeros(M, 2 * Fd, dtype=dtype, device=dev()) if save else None
        y = torch.empty(M, Fd, dtype=dtype, device=dev())
        L().check(lib.dh_dbg_gemm_glu(DT[dtype], 0, P(A), K, P(Wp), M, 2 * Fd, K, P(bp), P(pre), P(y), P(None), P(None),
                                      L().stream_ptr()), "dh_dbg_gemm_glu fwd")
        # (the activation is computed from the ROUNDED pre-activations: a pre-activation one 16-bit ulp off moves y by ~|h| ulp)
        close(y, y_ref, 2 * tol, 2 * tol, f"geglu-epilogue y {M}x{Fd}x{K} save={save}")
        if save:
            nat = torch.empty_like(pre); nat[:, idx] = pre
            close(nat, pre_ref, tol, tol, "geglu-epilogue pre-activations")
    # backward: dy = dT W2^T comes out of the GEMM (A2 [M][K2] x Wb [F][K2]); x = saved pre-activations (paired)
    K2 = K
    A2 = torch.randn(M, K2, generator=g, device=dev()).to(dtype)
    Wb = (torch.randn(Fd, K2, generator=g, device=dev()) / K2 ** 0.5).to(dtype)
    x = torch.randn(M, 2 * Fd, generator=g, device=dev()).to(dtype)
    xr = x.float().requires_grad_(True)
    out = xr[:, :Fd] * F.gelu(xr[:, Fd:])
    dy = A2.float() @ Wb.float().t()
    gref, = torch.autograd.grad(out, xr, dy)
    xp = x[:, idx].contiguous()
    dxp = torch.zeros_like(xp)
    L().check(lib.dh_dbg_gemm_glu(DT[dtype], 1, P(A2), K2, P(Wb), M, Fd, K2, P(None), P(None), P(None), P(xp), P(dxp),
                                  L().stream_ptr()), "dh_dbg_gemm_glu bwd")
    dx = torch.empty_like(dxp); dx[:, idx] = dxp
    close(dx, gref, tol, tol, f"geglu-epilogue backward {M}x{Fd}x{K2}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 5, 1024, 1024), (2, 2, 256, 256), (1, 20, 64, 64), (2, 5, 1024, 77), (1, 10, 200, 77), (1, 5, 4096, 77), (1, 2, 4096, 4096), (1, 3, 1000, 1000), (1, 2, 1100, 600)])
def test_attention_forward_backward(dtype, B, H, Nq, Nk):
    g = torch.Generator(device=dev()).manual_seed(Nq + Nk + H)
    C = H * 64
    q = torch.randn(B, Nq, C, generator=g, device=dev()).to(dtype)
    k = torch.randn(B, Nk, C, generator=g, device=dev()).to(dtype)
    v = torch.randn(B, Nk, C, generator=g, device=dev()).to(dtype)
    # one spiked key per head forces a large running-max jump mid-stream (online softmax rescale)
    k[:, Nk // 2, :] *= 6.0
    do = torch.randn(B, Nq, C, generator=g, device=dev()).to(dtype)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    sp = lambda t, n: t.view(B, n, H, 64).transpose(1, 2)
    s = (sp(qr, Nq) @ sp(kr, Nk).transpose(-1, -2)) * 0.125
    ref = (torch.softmax(s, dim=-1) @ sp(vr, Nk)).transpose(1, 2).reshape(B, Nq, C)
    lse_ref = torch.logsumexp(s, dim=-1)
    gq, gk, gv = torch.autograd.grad(ref, (qr, kr, vr), do.float())
    o = torch.empty_like(q); lse = torch.empty(B, H, Nq, dtype=torch.float32, device=dev()); delta = torch.empty_like(lse)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    L().check(L().lib().dh_dbg_attention(DT[dtype], P(q), C, P(k), P(v), C, P(o), C, P(lse), P(do), P(delta), P(dq), P(dk), P(dv),
                                         B, H, Nq, Nk, L().stream_ptr()))
    tol = 5e-3 if dtype == torch.float16 else 3e-2
    close(o, ref, tol, tol, "attn fwd")
    close(lse, lse_ref, 1e-3, 2e-3, "attn lse")
    for got, r, nm in ((dq, gq, "dq"), (dk, gk, "dk"), (dv, gv, "dv")):
        close(got, r, 2 * tol, 2 * tol * max(1.0, r.abs().max().item()) * 0.2, "attn " + nm)


def test_cross_lane_helpers():
    """common.h's VALU cross-lane steps (DPP / v_permlane16_swap / v_permlane32_swap): every lane of the wave (aligned octet,
    lane pair l / l^32) holds the total; the 16-byte exchange of the wide epilogue stores hands the right halves over."""
    torch.manual_seed(5)
    for trial in range(4):
        x = torch.randn(64, device=dev()) * (10.0 ** trial)
        out = torch.zeros(320, device=dev())
        ex = torch.zeros(256, dtype=torch.int32, device=dev())
        L().check(L().lib().dh_dbg_lane_ops(P(x), P(out), P(ex), L().stream_ptr()), "lane ops")
        torch.cuda.synchronize()
        xs = x.double().cpu()
        o = out.cpu()
        tot = xs.sum().item()
        assert torch.all(o[:64] == o[0]) and abs(o[0].item() - tot) <= 1e-5 * xs.abs().sum().item()
        assert torch.all(o[64:128] == x.max().item())
        for g in range(8):
            seg = o[128 + 8 * g:136 + 8 * g]
            assert torch.all(seg == seg[0]) and abs(seg[0].item() - xs[8 * g:8 * g + 8].sum().item()) <= 1e-5 * xs.abs().sum().item()
        xc = x.cpu()
        assert torch.equal(o[192:224], xc[:32] + xc[32:]) and torch.equal(o[224:256], xc[:32] + xc[32:])
        assert torch.equal(o[256:288], torch.maximum(xc[:32], xc[32:])) and torch.equal(o[288:320], torch.maximum(xc[:32], xc[32:]))
        e = ex.cpu().view(64, 4)
        for lane in range(32):      # lower lane: {its a, the upper lane's a}; upper lane: {the lower lane's b, its b}
            up = lane + 32
            assert e[lane].tolist() == [4 * lane, 4 * lane + 1, 4 * up, 4 * up + 1]
            assert e[up].tolist() == [4 * lane + 2, 4 * lane + 3, 4 * up + 2, 4 * up + 3]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,HW,N,K,conv,expect", [
    (1, 4096, 320, 320, False, "epilogue"),      # 128x64 tiles; 10-channel groups straddle the 64-column tiles
    (1, 4096, 320, 2880, True, "epilogue"),      # the 64^2-level convolution (two K groups inside the workgroup)
    (2, 1024, 640, 640, False, "epilogue"),      # 64x64 tiles, two images
    (2, 1024, 256, 1152, True, "epilogue"),      # 8-channel groups (the narrowest the epilogue takes); 18 K tiles: not split
    (1, 256, 1280, 1280, False, "epilogue"),     # 40-channel groups
    (1, 1024, 640, 5760, True, "reduce"),        # K split: the reduce kernel leaves the statistics (unchanged path)
    (1, 4096, 64, 320, False, "kernel"),         # 2-channel groups: the statistics kernel
])
def test_gemm_groupnorm_statistics_by_producer(dtype, B, HW, N, K, conv, expect):
    """GEMM -> GroupNorm(+SiLU) as the engine's forward runs the pair (dh_dbg_gemm_groupnorm): whichever launch leaves the slice
    statistics -- the unsplit GEMM's own epilogue (round 6: two slices per row tile, groups that straddle column tiles), the split-K
    reduce, or the statistics kernel -- the published (mean, rstd) equal those of the GEMM's ROUNDED output computed in f64 by torch to
    1e-5 (mean: of the group's standard deviation; rstd: relative), and the normalised tensor is the GroupNorm of that output."""
    lib = L().lib()
    G, M = 32, B * HW
    H = int(round(HW ** 0.5))
    g = torch.Generator(device=dev()).manual_seed(N + K + HW)
    if conv:
        Cin = K // 9
        A = torch.randn(M, Cin, generator=g, device=dev()).to(dtype); lda = Cin; mode = 1
    else:
        Cin = 0
        A = torch.randn(M, K, generator=g, device=dev()).to(dtype); lda = K; mode = 0
    W = (torch.randn(N, K, generator=g, device=dev()) / K ** 0.5).to(dtype)
    bias = torch.randn(N, generator=g, device=dev()) * 3.0             # a DC offset per channel: the unshifted sums must survive it
    gamma = torch.randn(N, generator=g, device=dev()); beta = torch.randn(N, generator=g, device=dev())
    C = torch.empty(M, N, dtype=dtype, device=dev()); Y = torch.empty_like(C)
    stats = torch.zeros(B * G, 2, device=dev()); scratch = torch.zeros(1 << 20, device=dev())
    part = torch.empty(16 << 20, dtype=torch.float32, device=dev())
    have = ctypes.c_int(-1)
    L().check(lib.dh_dbg_gemm_groupnorm(DT[dtype], P(A), lda, P(W), M, N, K, mode, H, H, Cin, P(bias), P(C), P(part), part.numel(), HW, G,
                                        P(gamma), P(beta), 1e-5, 1, P(Y), P(stats), P(scratch), ctypes.byref(have), L().stream_ptr()),
              "dh_dbg_gemm_groupnorm")
    torch.cuda.synchronize()
    assert {"epilogue": have.value > 1, "reduce": have.value == 1, "kernel": have.value == 0}[expect], (expect, have.value)
    x = C.double().view(B, HW, G, N // G)
    mean = x.mean(dim=(1, 3)); var = x.var(dim=(1, 3), unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    got = stats.double().view(B, G, 2)
    em = float(((got[..., 0] - mean).abs() / var.sqrt()).max()); er = float(((got[..., 1] - rstd).abs() / rstd).max())
    print(f"{expect}: slices {have.value}, mean err / sigma {em:.2e}, rstd rel err {er:.2e}")
    assert em < 1e-5 and er < 1e-5, (em, er)
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(C.float().view(B, HW, N).permute(0, 2, 1), G, gamma, beta, 1e-5)).permute(0, 2, 1)
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    close(Y.view(B, HW, N), ref, tol, tol, "GroupNorm of the GEMM output")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,HW,N,K,conv,silu,expect", [
    (1, 4096, 320, 320, False, 0, "epilogue"),      # a transformer's norm in front of proj_in: dense, no activation, 128x64 tiles
    (1, 4096, 320, 2880, True, 1, "epilogue"),      # the 64^2-level input-gradient convolution (two K groups inside the workgroup)
    (2, 1024, 640, 640, False, 1, "epilogue"),      # 64x64 tiles, two images
    (2, 1024, 256, 1152, True, 1, "epilogue"),      # 8-channel groups (the narrowest the epilogue takes)
    (1, 256, 1280, 1280, False, 0, "epilogue"),     # 40-channel groups
    (1, 1024, 640, 5760, True, 1, "reduce"),        # K split: the reduce kernel leaves the statistics (unchanged path)
    (1, 4096, 64, 320, False, 1, "kernel"),         # 2-channel groups: the statistics kernel
])
def test_gemm_groupnorm_backward_statistics_by_producer(dtype, B, HW, N, K, conv, silu, expect):
    """input-gradient GEMM -> GroupNorm(+SiLU) backward as the engine's backward pass runs the pair (dh_dbg_gemm_groupnorm_bwd): the GEMM's
    output is dy of the GroupNorm; whichever launch leaves the backward slice statistics (sum d, sum d xhat) -- the unsplit GEMM's own
    epilogue (round 6), the split-K reduce, or the statistics kernel -- dx equals torch autograd's on the ROUNDED dy, and the three
    producers agree with one another far below that tolerance (the epilogue form is switched off through dh_dbg_gemm_stage bit 3)."""
    lib = L().lib()
    G, M = 32, B * HW
    H = int(round(HW ** 0.5))
    g = torch.Generator(device=dev()).manual_seed(N + K + HW + 7)
    if conv:
        Cin = K // 9
        A = torch.randn(M, Cin, generator=g, device=dev()).to(dtype); lda = Cin; mode = 1
    else:
        Cin = 0
        A = torch.randn(M, K, generator=g, device=dev()).to(dtype); lda = K; mode = 0
    W = (torch.randn(N, K, generator=g, device=dev()) / K ** 0.5).to(dtype)
    x = (torch.randn(M, N, generator=g, device=dev()) * 1.5 + torch.randn(N, generator=g, device=dev())).to(dtype)
    gamma = torch.randn(N, generator=g, device=dev()); beta = torch.randn(N, generator=g, device=dev())
    xd = x.double().view(B, HW, G, N // G)
    mean = xd.mean(dim=(1, 3)); rstd = (xd.var(dim=(1, 3), unbiased=False) + 1e-5).rsqrt()
    stats = torch.stack([mean, rstd], dim=-1).float().contiguous()
    part = torch.empty(16 << 20, dtype=torch.float32, device=dev())

    def run(stage):
        C = torch.empty(M, N, dtype=dtype, device=dev()); dx = torch.empty_like(C)
        scratch = torch.full((1 << 20,), float("nan"), device=dev())
        have = ctypes.c_int(-1)
        L().check(lib.dh_dbg_gemm_stage(stage), "stage")
        try:
            L().check(lib.dh_dbg_gemm_groupnorm_bwd(DT[dtype], P(A), lda, P(W), M, N, K, mode, H, H, Cin, P(C), P(part), part.numel(), HW, G,
                                                    P(x), P(gamma), P(beta), P(stats), silu, P(dx), P(scratch), ctypes.byref(have),
                                                    L().stream_ptr()), "dh_dbg_gemm_groupnorm_bwd")
            torch.cuda.synchronize()
        finally:
            L().check(lib.dh_dbg_gemm_stage(1), "stage")
        return C, dx, have.value

    C, dx, have = run(1)
    assert {"epilogue": have > 1, "reduce": have == 1, "kernel": have == 0}[expect], (expect, have)
    C0, dx0, have0 = run(1 | 8)            # the statistics kernel (or the reduce) instead of the epilogue
    assert have0 <= 1 and torch.equal(C, C0)
    xin = x.float().view(B, HW, N).permute(0, 2, 1).clone().requires_grad_(True)
    y = torch.nn.functional.group_norm(xin, G, gamma, beta, 1e-5)
    if silu:
        y = torch.nn.functional.silu(y)
    y.backward(C.float().view(B, HW, N).permute(0, 2, 1))
    ref = xin.grad.permute(0, 2, 1).reshape(M, N)
    scale = float(ref.abs().max())
    e_ref = float((dx.float() - ref).abs().max()) / scale
    e_ab = float((dx.float() - dx0.float()).abs().max()) / scale
    print(f"{expect}: slices {have}, dx vs autograd {e_ref:.2e}, vs the statistics kernel {e_ab:.2e} (of max |dx|)")
    tol = 3e-3 if dtype == torch.float16 else 2e-2
    assert e_ref < tol and e_ab <= (1e-3 if dtype == torch.float16 else 8e-3), (e_ref, e_ab)
