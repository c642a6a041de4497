"""k_conv_slab (csrc/conv_slab.hip, round 6): the stride-1 3x3 convolution of small grids with the source staged once per
64-channel chunk and loader waves, against a torch fp32 convolution of the same 16-bit operands and against the k_gemm_dma path
it replaces (dh_dbg_gemm_stage(1 | 8) switches the slab kernel off).  Every image width of the U-Net levels (64, 32, 16, and 96 of
the 768 x 768 configuration), batch 2 (tiles must not read across images), bias / per-image vector / residual epilogues, K split
over chunks (f32 slabs + the reduce kernels of gemm.hip), a column count that is not a multiple of 128."""
import pytest
import torch
import torch.nn.functional as F

from test_unet_kernels_gpu import DT, L, P, close, dev, nhwc

pytestmark = pytest.mark.gpu


def run_conv(dtype, x_nhwc, wf, B, H, Cin, Cout, bias=None, rowvec=None, R=None):
    lib = L().lib()
    M = B * H * H
    C = torch.empty((M, Cout), dtype=dtype, device=dev())
    part = torch.empty(16 << 20, dtype=torch.float32, device=dev())
    rc = lib.dh_dbg_gemm(DT[dtype], P(x_nhwc), Cin, P(wf), M, Cout, 9 * Cin, 1, H, H, Cin, H, H, 1, 0, P(bias), P(rowvec),
                         rowvec.shape[1] if rowvec is not None else 0, H * H, P(R), Cout, P(C), Cout, 0, P(part), part.numel(),
                         L().stream_ptr())
    L().check(rc, "dh_dbg_gemm")
    return C


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,Cin,Cout,H,epi", [
    (1, 320, 320, 64, "bias+rowvec"),          # resnet conv1 of the 64^2 level (5 chunks, no split)
    (1, 320, 320, 64, "bias+res"),             # conv2
    (2, 320, 320, 64, "bias"),                 # the CFG pass: two images
    (1, 640, 320, 64, "none"),                 # an input gradient (no bias)
    (1, 640, 640, 32, "bias+res"),             # 32^2 level: K split over chunks
    (2, 1280, 640, 32, "bias+rowvec"),
    (1, 1280, 1280, 16, "bias+res"),           # 16^2 level: 20 chunks, 6 splits
    (2, 128, 192, 16, "bias"),                 # two chunks, three column tiles, two images of two tiles each
    (1, 64, 64, 96, "bias+res"),               # 768 x 768: image width 96, ONE chunk
    (1, 192, 64, 96, "none"),
])
def test_conv_slab_matches_torch_and_the_tile_kernel(dtype, B, Cin, Cout, H, epi):
    g = torch.Generator(device=dev()).manual_seed(Cin + 3 * Cout + 7 * H + B)
    x = torch.randn(B, Cin, H, H, generator=g, device=dev()).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, device=dev()) / (9 * Cin) ** 0.5).to(dtype)
    bias = torch.randn(Cout, generator=g, device=dev()) if "bias" in epi else None
    rowvec = torch.randn(B, Cout, generator=g, device=dev()) if "rowvec" in epi else None
    R = torch.randn(B * H * H, Cout, generator=g, device=dev()).to(dtype) if "res" in epi else None
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    if rowvec is not None:
        ref = ref + rowvec[:, :, None, None]
    ref = nhwc(ref).reshape(B * H * H, Cout)
    if R is not None:
        ref = ref + R.float()
    wf = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    lib = L().lib()
    try:
        L().check(lib.dh_dbg_gemm_stage(1), "stage")
        slab = run_conv(dtype, nhwc(x), wf, B, H, Cin, Cout, bias, rowvec, R)
        L().check(lib.dh_dbg_gemm_stage(1 | 8), "stage")
        tile = run_conv(dtype, nhwc(x), wf, B, H, Cin, Cout, bias, rowvec, R)
    finally:
        lib.dh_dbg_gemm_stage(1)
    tol = 4e-3 if dtype == torch.float16 else 2.5e-2
    close(tile, ref, tol, tol, "k_gemm_dma conv")
    close(slab, ref, tol, tol, "k_conv_slab conv")
    # the two kernels round the same f32 sums taken in different orders: they agree far inside the torch tolerance
    close(slab, tile, tol / 2, tol / 2, "slab vs tile")
    # image borders and the seam between the two images are where a slab indexing error would sit: compare them exactly to torch's
    # tolerance row by row (first / last image row and column of every image)
    s4, r4 = slab.view(B, H, H, Cout).float(), ref.view(B, H, H, Cout)
    for a, b2 in ((s4[:, 0], r4[:, 0]), (s4[:, -1], r4[:, -1]), (s4[:, :, 0], r4[:, :, 0]), (s4[:, :, -1], r4[:, :, -1])):
        assert float((a - b2).abs().max()) <= tol * (1.0 + float(b2.abs().max())), "border rows / columns"
