"""GPU parity of k_gemm_pp (csrc/gemm_pp.hip: the eight-wave ping-pong main loop of round 5) against plain PyTorch fp32
references of the same contraction, through the same debug hook as the k_gemm_dma tests.  dh_dbg_gemm_family(2) routes every
launch the kernel can carry to it (small shapes too); family 1 keeps k_gemm_dma, and the two must agree to 16-bit rounding on
the shapes the policy moves at batch 8 (same operands, same f32 accumulation, different summation order)."""
import pytest
import torch
import torch.nn.functional as F

from test_unet_kernels_gpu import DT, L, P, close, dev, run_gemm

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.fixture()
def family():
    lib = L().lib()

    def set_family(f):
        L().check(lib.dh_dbg_gemm_family(f), "dh_dbg_gemm_family")
    yield set_family
    lib.dh_dbg_gemm_family(0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(512, 320, 320),        # 256x160: two row tiles x two column tiles, 5 K tiles
                                     (256, 320, 64),         # a single K tile (prologue only)
                                     (300, 320, 128),        # ragged M: rows past M are zero-filled by the out-of-range offsets
                                     (100, 128, 192),        # 128x128, one ragged tile
                                     (640, 256, 1024),       # 256x128, 16 K tiles, 3 + 2 tiles
                                     (128, 640, 320),        # 128x160: four column tiles
                                     (1024, 960, 384)])      # six column tiles of 160: tiles starting inside a 64-row weight tile
def test_pp_dense_matches_torch(family, dtype, M, N, K):
    g = torch.Generator(device=dev()).manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g, device=dev()).to(dtype)
    W = (torch.randn(N, K, generator=g, device=dev()) / K ** 0.5).to(dtype)
    bias = torch.randn(N, generator=g, device=dev())
    R = torch.randn(M, N, generator=g, device=dev()).to(dtype)
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    family(2)
    close(run_gemm(dtype, A, K, W, M, N, K, bias=bias, R=R), A.float() @ W.float().t() + bias + R.float(), tol, tol, "pp bias+R")
    close(run_gemm(dtype, A, K, W, M, N, K), A.float() @ W.float().t(), tol, tol, "pp plain")
    close(run_gemm(dtype, A, K, W, M, N, K, bias=bias, split=False), A.float() @ W.float().t() + bias, tol, tol, "pp bias")
    # the per-image vector of a resnet's first convolution (time-embedding projection): row m belongs to image m // (M / 2)
    if M % 2 == 0:
        rv = torch.randn(2, N, generator=g, device=dev())
        ref = A.float() @ W.float().t() + bias + rv.repeat_interleave(M // 2, dim=0) + R.float()
        close(run_gemm(dtype, A, K, W, M, N, K, bias=bias, rowvec=rv, rpb=M // 2, R=R), ref, tol, tol, "pp bias + rowvec + R")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,Cin,Cout,H,stride,up", [(2, 64, 320, 16, 1, 0), (1, 128, 128, 32, 1, 0), (3, 64, 640, 8, 1, 0),
                                                     (2, 64, 320, 16, 2, 0), (1, 128, 320, 8, 1, 1), (2, 128, 128, 16, 2, 0)])
def test_pp_conv_forward_and_input_gradient(family, dtype, B, Cin, Cout, H, stride, up):
    """3x3 convolution (stride 1: the tap-shifted DMA with out-of-image taps as out-of-range offsets; stride 2 / nearest-2x
    source / transposed stride 2: the generic gather) and its input gradient through the flipped weights."""
    g = torch.Generator(device=dev()).manual_seed(Cin + Cout + H + stride + up)
    x = torch.randn(B, Cin, H, H, generator=g, device=dev()).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, device=dev()) / (9 * Cin) ** 0.5).to(dtype)
    bias = torch.randn(Cout, generator=g, device=dev())
    xr = x.float().requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2.0, mode="nearest") if up else xr
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1)
    Ho = ref.shape[-1]
    wf = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    family(2)
    C = run_gemm(dtype, nhwc(x), Cin, wf, B * Ho * Ho, Cout, 9 * Cin, mode=1, geo=(H, H, Cin, Ho, Ho, stride, up), bias=bias)
    close(C.view(B, Ho, Ho, Cout), nhwc(ref), tol, tol, "pp conv fwd")
    if Cin % 160 and Cin % 128:
        return                                   # the input gradient has N = Cin columns: not a k_gemm_pp tile width
    dy = torch.randn(ref.shape, generator=g, device=dev()).to(dtype)
    gref, = torch.autograd.grad(ref, xr, dy.float())
    wb = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).contiguous()
    if up:
        return
    if stride == 2:
        dx = run_gemm(dtype, nhwc(dy), Cout, wb, B * H * H, Cin, 9 * Cout, mode=2, geo=(Ho, Ho, Cout, H, H, 1, 0)).view(B, H, H, Cin)
    else:
        dx = run_gemm(dtype, nhwc(dy), Cout, wb, B * H * H, Cin, 9 * Cout, mode=1, geo=(H, H, Cout, H, H, 1, 0)).view(B, H, H, Cin)
    close(dx.reshape(B, H, H, Cin), nhwc(gref), tol, tol * gref.abs().max().item(), "pp conv dX")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_pp_policy_shapes_agree_with_k_gemm_dma(family, dtype):
    """The batch-8 shapes the policy moves to k_gemm_pp: the policy's own choice (family 0) against k_gemm_dma (family 1) and
    against torch, incl. split K over workgroups (M = 2048, N = 1280: f32 slabs + the reduce kernel)."""
    g = torch.Generator(device=dev()).manual_seed(5)
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    for (M, N, K, conv) in [(32768, 320, 2880, (8, 64, 320)), (8192, 640, 5760, (8, 32, 640)), (2048, 1280, 11520, (8, 16, 1280)),
                            (32768, 320, 1280, None), (8192, 640, 640, None), (32768, 960, 320, None)]:
        if conv:
            Bn, H, Cin = conv
            A = torch.randn(Bn * H * H, Cin, generator=g, device=dev()).to(dtype); lda = Cin
            geo, mode = (H, H, Cin, H, H, 1, 0), 1
        else:
            A = torch.randn(M, K, generator=g, device=dev()).to(dtype); lda = K
            geo, mode = (0, 0, 0, 0, 0, 1, 0), 0
        W = (torch.randn(N, K, generator=g, device=dev()) / K ** 0.5).to(dtype)
        bias = torch.randn(N, generator=g, device=dev())
        R = torch.randn(M, N, generator=g, device=dev()).to(dtype)
        outs = {}
        for fam in (0, 1):
            family(fam)
            outs[fam] = run_gemm(dtype, A, lda, W, M, N, K, mode=mode, geo=geo, bias=bias, R=R)
        close(outs[0], outs[1].float(), tol, tol, f"policy vs k_gemm_dma {M}x{N}x{K}")
        if conv:
            x = A.view(Bn, H, H, Cin).permute(0, 3, 1, 2).float()
            w4 = W.view(N, 3, 3, Cin).permute(0, 3, 1, 2).float()
            ref = nhwc(F.conv2d(x, w4, bias, padding=1)).reshape(M, N) + R.float()
        else:
            ref = A.float() @ W.float().t() + bias + R.float()
        close(outs[0], ref, tol, tol, f"policy vs torch {M}x{N}x{K}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,Fd,K", [(512, 256, 128), (300, 128, 64), (128, 384, 192), (2048, 1280, 320)])
def test_pp_geglu_epilogues(family, dtype, M, Fd, K):
    """The GEGLU epilogues of k_gemm_pp (forward: value * gelu(gate) out of the in-projection GEMM, with and without the saved
    pre-activations; backward: d_value | d_gate out of the input-gradient GEMM of the out-projection) -- the body of
    test_unet_kernels_gpu.test_gemm_geglu_epilogues routed to the ping-pong kernel (256x128 and 128x128 tiles, ragged M)."""
    import test_unet_kernels_gpu as TK
    family(2)
    TK.test_gemm_geglu_epilogues(dtype, M, Fd, K)
    # the policy's own shapes at batch 8 (family 0) agree with k_gemm_dma (family 1) bit-for-bit-close
    if M == 2048:
        g = torch.Generator(device=dev()).manual_seed(3)
        A = torch.randn(M, K, generator=g, device=dev()).to(dtype)
        W = (torch.randn(2 * Fd, K, generator=g, device=dev()) / K ** 0.5).to(dtype)
        ys = {}
        for fam in (0, 1):
            family(fam)
            y = torch.empty(M, Fd, dtype=dtype, device=dev())
            L().check(L().lib().dh_dbg_gemm_glu(DT[dtype], 0, P(A), K, P(W), M, 2 * Fd, K, P(None), P(None), P(y), P(None), P(None),
                                                L().stream_ptr()), "dh_dbg_gemm_glu")
            ys[fam] = y
        tol = 8e-3 if dtype == torch.float16 else 5e-2
        close(ys[0], ys[1].float(), tol, tol, "policy vs k_gemm_dma geglu")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_pp_persistent_workgroups_walk_several_tiles(family, dtype):
    """More work items than workgroup slots (256): a workgroup then walks several tiles, the stores of one tile in flight under
    the next tile's K loop.  Bit-identical to the one-workgroup-per-item launch; against torch; dense
    (ragged M, residual), a stride-1 convolution, and the GEGLU forward with saved pre-activations."""
    lib = L().lib()
    g = torch.Generator(device=dev()).manual_seed(11)
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    try:
        family(2)
        for (M, N, K) in [(40000, 640, 128), (33000, 384, 192)]:        # 157 x 4 = 628 tiles of 256 x 160; 129 x 3 = 387 of 256 x 128
            A = torch.randn(M, K, generator=g, device=dev()).to(dtype)
            W = (torch.randn(N, K, generator=g, device=dev()) / K ** 0.5).to(dtype)
            bias = torch.randn(N, generator=g, device=dev())
            R = torch.randn(M, N, generator=g, device=dev()).to(dtype)
            outs = []
            for persist in (0, 1):
                lib.dh_dbg_gemm_pp_persist(persist)
                outs.append(run_gemm(dtype, A, K, W, M, N, K, bias=bias, R=R, split=False))
            close(outs[1], A.float() @ W.float().t() + bias + R.float(), tol, tol, f"persistent dense {M}x{N}x{K}")
            assert torch.equal(outs[0], outs[1])
        # convolution: 9 images of 64 x 64, 64 -> 320 channels: 144 x 2 = 288 tiles
        B, Cin, Cout, H = 9, 64, 320, 64
        x = torch.randn(B, Cin, H, H, generator=g, device=dev()).to(dtype)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g, device=dev()) / (9 * Cin) ** 0.5).to(dtype)
        xa = nhwc(x).reshape(B * H * H, Cin)
        wk = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
        outs = []
        for persist in (0, 1):
            lib.dh_dbg_gemm_pp_persist(persist)
            outs.append(run_gemm(dtype, xa, Cin, wk, B * H * H, Cout, 9 * Cin, mode=1, geo=(H, H, Cin, H, H, 1, 0), split=False))
        ref = nhwc(F.conv2d(x.float(), w.float(), padding=1)).reshape(B * H * H, Cout)
        close(outs[1], ref, tol, tol, "persistent conv")
        assert torch.equal(outs[0], outs[1])
        # GEGLU forward: 16384 x (2 x 640): 64 x 10 = 640 tiles of 256 x 128
        M, Fd, K = 16384, 640, 64
        A = torch.randn(M, K, generator=g, device=dev()).to(dtype)
        W = (torch.randn(2 * Fd, K, generator=g, device=dev()) / K ** 0.5).to(dtype)
        res = []
        for persist in (0, 1):
            lib.dh_dbg_gemm_pp_persist(persist)
            pre = torch.empty(M, 2 * Fd, dtype=dtype, device=dev())
            y = torch.empty(M, Fd, dtype=dtype, device=dev())
            L().check(lib.dh_dbg_gemm_glu(DT[dtype], 0, P(A), K, P(W), M, 2 * Fd, K, P(None), P(pre), P(y), P(None), P(None), L().stream_ptr()),
                      "dh_dbg_gemm_glu")
            res.append((pre, y))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    finally:
        lib.dh_dbg_gemm_pp_persist(1)
