#!/usr/bin/env python3
"""Engine forward / backward-to-sample against the oracle's autograd for U-Nets with 5 and 4 input channels (use_depth false)."""
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
    for cin in (5, 4):
        for seed in (0, 1):
            cfg = dict(U.TINY, in_channels=cin)
            ref = U.init_synthetic_(U.UNetTorch(cfg), seed=seed).to(dev).eval()
            with torch.no_grad():
                for p in ref.parameters():
                    p.copy_(p.half().float())
                    p.requires_grad_(False)
            g = torch.Generator(device=dev).manual_seed(3)
            text = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g, device=dev)
            x = torch.randn(1, cin, 64, 64, generator=g, device=dev)
            xq = x.clone().requires_grad_(True)
            out = ref(xq, torch.tensor(920, device=dev), encoder_hidden_states=text, return_dict=False)
            d2 = torch.randn(out[6].shape, generator=g, device=dev).half().float()
            gx, = torch.autograd.grad(out[6], xq, d2)
            hip = HipUNet(dict(cfg, text_len=77), dtype=torch.float16, max_batch=2)
            hip.load_state_dict(ref.state_dict())
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                eps, acts = hip.forward(x.permute(0, 2, 3, 1).contiguous(), 920.0, text.contiguous(), save_for_backward=True,
                                        want_acts=[2], want_eps=False)
                d = d2.permute(0, 2, 3, 1).contiguous().half()
                dx, _ = hip.backward([None, None, d], None, want_sample_grad=True)
                torch.cuda.synchronize()
            print(f"in_channels {cin} seed {seed}: act2 rel err {rel(acts[2].permute(0, 3, 1, 2), out[6]):.3e}  d_sample rel err "
                  f"{rel(dx.permute(0, 3, 1, 2), gx):.3e}  per channel "
                  f"{[round(rel(dx[..., c], gx[:, c]), 4) for c in range(cin)]}", flush=True)


if __name__ == "__main__":
    main()
