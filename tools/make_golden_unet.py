#!/usr/bin/env python3
"""g12: pin oracle/unet_torch.py against the REFERENCE'S OWN vendored U-Net files
(/root/reference/diffhandles/model/{unet_2d_condition,unet_2d_blocks,transformer_2d,attention,
attention_processor}.py), imported and executed here on top of stand-ins for the diffusers-0.23
leaf primitives only (tools/diffusers_standins.py).

For each configuration: build the reference class with the published SD-2-depth config.json
at the given sizes, build the oracle, load the ORACLE's seeded state dict into the reference
model with strict=True (so the parameter naming is pinned too), run both on the same seeded
inputs in fp32 on CPU and assert the 7-tuple (eps, None x3, act0, act1, act2), d/d sample and
d/d encoder_hidden_states equal on the spot; then write tests/golden/g12_unet.npz (inputs are
re-creatable from the seeds; expected outputs are stored).  Runs only in the build container.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "tests", "golden")

import diffusers_standins as S  # noqa: E402
from oracle import unet_torch as U  # noqa: E402

# name -> (config, batch, latent h, latent w, text length)
MID = dict(in_channels=5, out_channels=4, block_out_channels=(64, 128, 256, 256), layers_per_block=2,
           heads=(1, 2, 4, 4), cross_attention_dim=96, norm_groups=32, sample_size=64)
CASES = {
    "tiny": (U.TINY, 1, 16, 16, 77),
    "tiny_b2_rect": (U.TINY, 2, 16, 24, 77),
    "mid": (MID, 2, 16, 16, 77),
    "mid_t5": (MID, 1, 8, 8, 5),
}


def inputs(cfg, b, h, w, n_text, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, cfg["in_channels"], h, w, generator=g)
    ctx = torch.randn(b, n_text, cfg["cross_attention_dim"], generator=g)
    t = int(torch.randint(0, 1000, (1,), generator=g))
    # cotangents: one per output, so the gradient test exercises every capture point
    ws = [torch.randn(s, generator=g) for s in out_shapes(cfg, b, h, w)]
    return x, ctx, t, ws


def out_shapes(cfg, b, h, w):
    ch = cfg["block_out_channels"]
    return [(b, cfg["out_channels"], h, w), (b, ch[3], h // 2, w // 2), (b, ch[1], h, w), (b, ch[0], h, w)]


def run(model, x, ctx, t, ws):
    x = x.clone().requires_grad_(True)
    ctx = ctx.clone().requires_grad_(True)
    out = model(x, t, encoder_hidden_states=ctx, return_dict=False)
    assert len(out) == 7 and out[1] is None and out[2] is None and out[3] is None
    ys = [out[0], out[4], out[5], out[6]]
    loss = sum((y * w).sum() for y, w in zip(ys, ws))
    gx, gc = torch.autograd.grad(loss, [x, ctx])
    return [y.detach() for y in ys], gx, gc


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(True)
    RefUNet = S.import_reference_unet()
    store = {}
    for ci, (name, (cfg, b, h, w, n_text)) in enumerate(CASES.items()):
        seed = 1200 + ci
        ref = RefUNet(**S.sd2_depth_kwargs(cfg)).eval()
        ora = U.init_synthetic_(U.UNetTorch(cfg), seed=12).eval()
        missing = ref.load_state_dict(ora.state_dict(), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        n_ref = sum(p.numel() for p in ref.parameters())
        assert n_ref == sum(p.numel() for p in ora.parameters())
        x, ctx, t, ws = inputs(cfg, b, h, w, n_text, seed=seed)
        ys_r, gx_r, gc_r = run(ref, x, ctx, t, ws)
        ys_o, gx_o, gc_o = run(ora, x, ctx, t, ws)
        worst = 0.0
        for a, o in zip(ys_r + [gx_r, gc_r], ys_o + [gx_o, gc_o]):
            assert a.shape == o.shape
            worst = max(worst, float((a - o).abs().max() / a.abs().max()))
        # same torch ops in (nearly) the same order; SDPA vs the explicit softmax product is the only re-association
        assert worst < 2e-5, (name, worst)
        # return_dict=True path
        with torch.no_grad():
            assert torch.equal(ref(x, t, encoder_hidden_states=ctx)["sample"], ys_r[0])
        print(f"g12 {name}: {n_ref} params, reference vs oracle max rel diff {worst:.2e}")
        for k, v in zip(("eps", "act0", "act1", "act2", "gx", "gc"), ys_r + [gx_r, gc_r]):
            store[f"{name}.{k}"] = v.numpy().astype(np.float32)
        store[f"{name}.meta"] = np.array([b, h, w, n_text, t, seed, n_ref], dtype=np.int64)
    # the full-size structure: parameter count and names of the reference class at the SD-2-depth config
    ref = RefUNet(**S.sd2_depth_kwargs(dict(U.SD2_DEPTH, sample_size=32)))
    names = sorted(ref.state_dict().keys())
    ora_names = sorted(U.UNetTorch(U.SD2_DEPTH).state_dict().keys())
    assert names == ora_names
    store["sd2.n_params"] = np.array(sum(p.numel() for p in ref.parameters()), dtype=np.int64)
    import hashlib
    store["sd2.names_sha"] = np.frombuffer(hashlib.sha256("\n".join(
        f"{k}:{tuple(v.shape)}" for k, v in sorted(ref.state_dict().items())).encode()).digest(), dtype=np.uint8)
    print("g12 sd2: reference class at the SD-2-depth config has", int(store["sd2.n_params"]), "parameters,",
          len(names), "tensors; names and shapes equal the oracle's")
    np.savez_compressed(os.path.join(OUT, "g12_unet.npz"), **store)
    print("wrote", os.path.join(OUT, "g12_unet.npz"))


if __name__ == "__main__":
    main()
