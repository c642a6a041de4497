#!/usr/bin/env python3
"""Where does the error of the full-size null-text gradient come from?  (round-2 review: d loss / d uncond 3.2e-2 at the full
SD-2-depth size while d / d sample of the same engine is 1.8e-3.)

One backward of eps -> text embedding at the full size against the oracle's autograd (torch fp32 on the GPU), for a range of
cotangent amplitudes (the engine's backward is linear in the cotangent, so any difference between the rows is 16-bit range:
underflow into fp16 subnormals at small amplitudes, overflow at large ones) and both storage types.

  python tools/probe_text_grad.py            # prints one line per (dtype, log2 scale)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def main():
    from diffusionhandles_amd.unet import HipUNet
    from oracle import unet_torch as U
    dev = torch.device("cuda:0")
    ref = U.init_synthetic_(U.UNetTorch(U.SD2_DEPTH), seed=0).to(dev).eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.half().float())
            p.requires_grad_(False)
    g = torch.Generator(device=dev).manual_seed(31)
    text = torch.randn(1, 77, 1024, generator=g, device=dev)
    x = torch.randn(1, 5, 64, 64, generator=g, device=dev)
    d_eps = torch.randn(1, 4, 64, 64, generator=g, device=dev)
    d_eps = d_eps / d_eps.abs().max()                       # max |cotangent| = 1
    tq = text.clone().requires_grad_(True)
    xq = x.clone().requires_grad_(True)
    eps = ref(xq, torch.tensor(920, device=dev), encoder_hidden_states=tq)["sample"]
    gt, gx = torch.autograd.grad(eps, [tq, xq], d_eps)
    print(f"oracle: |d_text| rms {gt.pow(2).mean().sqrt().item():.3e} max {gt.abs().max().item():.3e}; |d_sample| rms "
          f"{gx.pow(2).mean().sqrt().item():.3e}")
    sd = ref.state_dict()
    for dtype in (torch.float16, torch.bfloat16):
        hip = HipUNet(dict(U.SD2_DEPTH, text_len=77), dtype=dtype, max_batch=1)
        hip.load_state_dict(sd)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            xs = x.permute(0, 2, 3, 1).contiguous()
            for lg in (-20, -16, -12, -8, -4, 0, 4, 8, 10, 12):
                s = 2.0 ** lg
                hip.forward(xs, 920.0, text.contiguous(), save_for_backward=True, want_acts=False)
                d = (d_eps * s).permute(0, 2, 3, 1).contiguous()
                dx, dt = hip.backward(None, d, want_sample_grad=True, want_text_grad=True)
                torch.cuda.synchronize()
                fin = bool(torch.isfinite(dt).all()) and bool(torch.isfinite(dx).all())
                print(f"{str(dtype):16s} max|d_eps| = 2^{lg:+3d}: d_text rel err {rel(dt / s, gt):.3e}  d_sample rel err "
                      f"{rel(dx.permute(0, 3, 1, 2)[:, :5] / s, gx):.3e}  finite {fin}", flush=True)
        del hip
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
