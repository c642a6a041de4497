"""GPU parity of the native U-Net engine (forward with activation capture, backward to the
sample and to the text embedding) against the oracle's plain-PyTorch fp32 U-Net with the
same (16-bit-rounded) weights.  Metric: relative L2 error per output tensor."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rel(got, ref):
    return ((got.float() - ref.float()).norm() / (ref.float().norm() + 1e-12)).item()


def build(cfg, dtype, max_batch, seed=0):
    from diffusionhandles_amd.unet import HipUNet
    from oracle import unet_torch as U
    ref = U.init_synthetic_(U.UNetTorch(cfg), seed=seed).to(dev()).eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(dtype).float())
    hip = HipUNet(dict(cfg, text_len=77), dtype=dtype, max_batch=max_batch)
    hip.load_state_dict(ref.state_dict())
    return ref, hip


def run_case(cfg, dtype, B, t, tol_f, tol_b, check_text=True):
    ref, hip = build(cfg, dtype, B)
    g = torch.Generator(device=dev()).manual_seed(11)
    S, D = cfg["sample_size"], cfg["cross_attention_dim"]
    sample = torch.randn(B, cfg["in_channels"], S, S, generator=g, device=dev())
    text = torch.randn(B, 77, D, generator=g, device=dev())
    xs = sample.clone().requires_grad_(True)
    xt = text.clone().requires_grad_(True)
    out = ref(xs, torch.tensor(t), xt, return_dict=False)
    eps, acts = hip.forward(sample.permute(0, 2, 3, 1).contiguous(), t, text, save_for_backward=True)
    errs = {"eps": rel(eps.permute(0, 3, 1, 2), out[0])}
    for k in range(3):
        errs[f"act{k}"] = rel(acts[k].permute(0, 3, 1, 2), out[4 + k])
    print("forward rel errors", errs)
    for k, v in errs.items():
        assert v < tol_f, f"forward {k}: measured rel-L2 {v:.3e} >= gate {tol_f:.1e} (all: {errs})"
    # backward: random cotangents on the three activations and on eps
    gs = [torch.randn(o.shape, generator=g, device=dev()) * 0.05 for o in (out[4], out[5], out[6])]
    ge = torch.randn(out[0].shape, generator=g, device=dev()) * 0.05
    gs16 = [x.to(dtype) for x in gs]
    loss = sum((o * x.float()).sum() for o, x in zip((out[4], out[5], out[6]), gs16)) + (out[0] * ge).sum()
    gx, gt = torch.autograd.grad(loss, (xs, xt), retain_graph=True)
    d_acts = [x.permute(0, 2, 3, 1).contiguous() for x in gs16]
    d_sample, d_text = hip.backward(d_acts, ge.permute(0, 2, 3, 1).contiguous(), True, check_text)
    e1 = rel(d_sample.permute(0, 3, 1, 2), gx)
    print("backward rel errors: d_sample", e1)
    assert e1 < tol_b, f"backward d_sample: measured rel-L2 {e1:.3e} >= gate {tol_b:.1e}"
    if check_text:
        e2 = rel(d_text, gt)
        print("d_text", e2)
        assert e2 < tol_b, f"backward d_text: measured rel-L2 {e2:.3e} >= gate {tol_b:.1e}"
    # activation-only cotangent (the guided-inference case: d_eps = None, only act2 seeded)
    loss2 = (out[6] * gs16[2].float()).sum()
    gx2, = torch.autograd.grad(loss2, xs)
    eps, acts = hip.forward(sample.permute(0, 2, 3, 1).contiguous(), t, text, save_for_backward=True)
    d2, _ = hip.backward([None, None, d_acts[2]], None, True, False)
    e3 = rel(d2.permute(0, 3, 1, 2), gx2)
    print("act2-only d_sample", e3)
    assert e3 < tol_b, f"backward (act2 only) d_sample: measured rel-L2 {e3:.3e} >= gate {tol_b:.1e}"
    # determinism
    d3, _ = hip.backward([None, None, d_acts[2]], None, True, False)
    assert torch.equal(d2, d3)


def test_engine_tiny_fp16():
    from oracle import unet_torch as U
    with torch.cuda.stream(torch.cuda.Stream()):      # created stream: exercises the hipGraph replay path
        run_case(U.TINY, torch.float16, 2, 480.0, 1e-2, 3e-2)


def test_engine_tiny_bf16():
    from oracle import unet_torch as U
    run_case(U.TINY, torch.bfloat16, 1, 20.0, 5e-2, 1.5e-1)


def test_engine_reference_call_signature():
    from oracle import unet_torch as U
    ref, hip = build(U.TINY, torch.float16, 2)
    g = torch.Generator(device=dev()).manual_seed(3)
    sample = torch.randn(2, 5, 64, 64, generator=g, device=dev())
    text = torch.randn(2, 77, 64, generator=g, device=dev())
    out = hip(sample, torch.tensor(980), encoder_hidden_states=text, return_dict=False)
    assert len(out) == 7 and out[1] is None and out[4].shape == (2, 128, 32, 32) and out[6].shape == (2, 64, 64, 64)
    r = ref(sample, torch.tensor(980), text, return_dict=False)
    assert rel(out[0], r[0]) < 1e-2
    assert rel(hip(sample, 980, encoder_hidden_states=text)["sample"], r[0]) < 1e-2


def test_engine_sd2_depth_full_size_fp16():
    """The full SD-2-depth configuration (865.7 M parameters), B=1, forward + backward."""
    from oracle import unet_torch as U
    names = None
    run_case(U.SD2_DEPTH, torch.float16, 1, 500.0, 4e-3, 6e-3)          # measured 1.2e-3 / 1.8e-3: gates <= 3x measured


def test_engine_sd2_depth_full_size_bf16_batch2():
    """BASELINE config 5 computes the U-Net in bf16: full configuration, B=2 (the CFG pass shape), bf16 tolerance."""
    from oracle import unet_torch as U
    run_case(U.SD2_DEPTH, torch.bfloat16, 2, 261.0, 3e-2, 4.5e-2, check_text=False)   # measured 1.1e-2 / 1.5e-2


def test_engine_sd2_depth_latent96_fp16():
    """768x768 images: 96x96 latents, so 9216 / 2304 / 576 / 144 rows per image (not powers of two) through the
    reciprocal index arithmetic, the GroupNorm slices and the split-K policy."""
    from oracle import unet_torch as U
    run_case(dict(U.SD2_DEPTH, sample_size=96), torch.float16, 1, 740.0, 4e-3, 6e-3, check_text=False)


def test_engine_sd2_depth_batch8_fp16():
    """Eight images per pass (the batched-edits mode): 256 row tiles per 64x64 layer, i.e. the 128x320 GEMM tile and the
    16-slice GroupNorm statistics, against the torch restatement."""
    from oracle import unet_torch as U
    run_case(U.SD2_DEPTH, torch.float16, 8, 120.0, 4e-3, 6e-3, check_text=False)


def test_engine_truncated_forward_matches_full():
    """want_eps=False stops the tape after the last requested activation: that activation and the
    gradient from it must be bit-identical to the full pass (same kernels, same order)."""
    from oracle import unet_torch as U
    cfg = U.TINY
    _, hip = build(cfg, torch.float16, 2)
    g = torch.Generator(device=dev()).manual_seed(5)
    S, D = cfg["sample_size"], cfg["cross_attention_dim"]
    x = torch.randn(2, S, S, cfg["in_channels"], generator=g, device=dev())
    text = torch.randn(2, 77, D, generator=g, device=dev())
    side = torch.cuda.Stream()
    for stream in (torch.cuda.current_stream(), side):          # eager (null stream) and hipGraph replay
        with torch.cuda.stream(stream):
            for _ in range(2):
                eps, full = hip.forward(x, 321.0, text, save_for_backward=True)
                d1 = (torch.randn(full[1].shape, generator=g, device=dev()) * 1e-2).to(torch.float16)
                ds_full, _ = hip.backward([None, d1, None], None)
                none_eps, part = hip.forward(x, 321.0, text, save_for_backward=True, want_acts=[1], want_eps=False)
                assert none_eps is None and part[0] is None and part[2] is None
                assert torch.equal(part[1], full[1])
                ds_part, _ = hip.backward([None, d1, None], None)
                assert torch.equal(ds_part, ds_full)
                with pytest.raises(RuntimeError):
                    hip.backward([None, None, d1.new_zeros(full[2].shape)], None)
        torch.cuda.synchronize()


def test_engine_graph_cache_keys_batch16():
    """Graph cache keys give every field its own bit range: at B = 16 (and 17) the batch bits used to alias the
    activation-mask bits of the backward key, so alternating two masks replayed the wrong graph.  Replay on a created
    stream must equal the eager (null-stream) result for every (batch, mask) pair, in alternation."""
    from oracle import unet_torch as U
    cfg = dict(U.TINY, sample_size=16)
    _, hip = build(cfg, torch.float16, 17)
    g = torch.Generator(device=dev()).manual_seed(7)
    side = torch.cuda.Stream()
    cases = []
    for B in (16, 17, 1):
        x = torch.randn(B, 16, 16, 5, generator=g, device=dev())
        text = torch.randn(B, 77, cfg["cross_attention_dim"], generator=g, device=dev())
        d = [(torch.randn((B,) + s, generator=g, device=dev()) * 1e-2).to(torch.float16) for s in hip.act_shapes]
        for mask in ([None, d[1], None], [None, d[1], d[2]], [d[0], None, None], [None, None, d[2]]):
            hip.forward(x, 300.0, text, save_for_backward=True)
            eager, _ = hip.backward(mask, None)
            cases.append((x, text, mask, eager.clone()))
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for rep in range(2):
            for x, text, mask, eager in cases:
                hip.forward(x, 300.0, text, save_for_backward=True)
                got, _ = hip.backward(mask, None)
                assert torch.equal(got, eager)
    torch.cuda.synchronize()


def test_engine_text_kv_cache_by_key():
    """dh_unet_set_text_key: forwards that name the same text reuse the hoisted K|V projections (bit-identical to
    recomputing them); a new key, another batch size or an unnamed forward in between recomputes."""
    from oracle import unet_torch as U
    cfg = dict(U.TINY, sample_size=16)
    _, hip = build(cfg, torch.float16, 2)
    g = torch.Generator(device=dev()).manual_seed(17)
    x = torch.randn(1, 16, 16, 5, generator=g, device=dev())
    x2 = torch.randn(1, 16, 16, 5, generator=g, device=dev())
    ta = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g, device=dev())
    tb = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g, device=dev())
    with torch.cuda.stream(torch.cuda.Stream()):
        ref_a = hip.forward(x, 100.0, ta)[0].clone()
        ref_a2 = hip.forward(x2, 100.0, ta)[0].clone()
        ref_b = hip.forward(x, 100.0, tb)[0].clone()
        assert torch.equal(hip.forward(x, 100.0, ta, text_key=5)[0], ref_a)           # miss: computes and names the projections
        assert torch.equal(hip.forward(x2, 100.0, ta, text_key=5)[0], ref_a2)         # hit
        assert torch.equal(hip.forward(x, 100.0, tb, text_key=6)[0], ref_b)           # new key: recomputed
        assert torch.equal(hip.forward(x, 100.0, ta)[0], ref_a)                       # unnamed forward overwrites the buffer ...
        assert torch.equal(hip.forward(x, 100.0, tb, text_key=6)[0], ref_b)           # ... so key 6 is recomputed, not stale
        d = (torch.randn((1,) + hip.act_shapes[2], generator=g, device=dev()) * 1e-2).to(torch.float16)
        hip.forward(x, 100.0, tb, save_for_backward=True, text_key=6)                 # hit, then a backward to the text through it
        ds_hit, dt_hit = hip.backward([None, None, d], None, True, True)
        hip.forward(x, 100.0, tb, save_for_backward=True)
        ds_ref, dt_ref = hip.backward([None, None, d], None, True, True)
        assert torch.equal(ds_hit, ds_ref) and torch.equal(dt_hit, dt_ref)
    torch.cuda.synchronize()


@pytest.mark.parametrize("cfg_name", ["TINY", "SD2_DEPTH"])
def test_forward_only_layout_is_bit_identical_and_smaller(cfg_name):
    """Round 5: a forward nobody differentiates shares the activation arena by liveness (dh_unet_config.max_diff_batch,
    csrc/unet_engine.cpp layout_tensors).  Engine A keeps a slot per tensor for every batch (max_batch = max_diff_batch = 4), engine
    B is sized for saved forwards of 2 and forward-only passes of 4: B's forward-only pass at batch 4 gives bit for bit what A's
    saved forward gives (same kernels, other addresses; the captured activations and eps included), with a smaller workspace; a
    saved forward + backward on B right after a forward-only pass (which overwrote the saved layout) still equals A's; a saved
    forward above max_diff_batch is refused."""
    from diffusionhandles_amd.unet import HipUNet
    from oracle import unet_torch as U
    cfg = getattr(U, cfg_name)
    ref, a = build(cfg, torch.float16, 4)
    b = HipUNet(dict(cfg, text_len=77), dtype=torch.float16, max_batch=4, max_diff_batch=2)
    b.load_state_dict(ref.state_dict())
    assert b.workspace_bytes() < 0.72 * a.workspace_bytes(), (b.workspace_bytes(), a.workspace_bytes())
    g = torch.Generator(device=dev()).manual_seed(17)
    S, D = cfg["sample_size"], cfg["cross_attention_dim"]
    x4 = torch.randn(4, S, S, cfg["in_channels"], generator=g, device=dev())
    txt4 = torch.randn(4, 77, D, generator=g, device=dev())
    with torch.cuda.stream(torch.cuda.Stream()):
        ea, aa = a.forward(x4, 300.0, txt4, save_for_backward=True)
        eb, ab = b.forward(x4, 300.0, txt4, save_for_backward=False)
        assert torch.equal(ea, eb) and all(torch.equal(p, q) for p, q in zip(aa, ab)), "forward-only layout changed the result"
        # truncated tape (no eps, one activation) and the in-place views
        _, ab2 = b.forward(x4, 300.0, txt4, save_for_backward=False, want_acts=[1], want_eps=False, inplace=True)
        assert torch.equal(ab2[1], aa[1]) and ab2[0] is None
        # a saved forward + backward at batch 2 on B after the forward-only pass, against A
        d2 = torch.randn((2,) + a.act_shapes[2], generator=g, device=dev()).half() * 0.05
        outs = []
        for eng in (a, b):
            eng.forward(x4[:2].contiguous(), 300.0, txt4[:2].contiguous(), save_for_backward=True, want_acts=[2], want_eps=False)
            ds, _ = eng.backward([None, None, d2], None, True, False)
            outs.append(ds.clone())
        assert torch.equal(outs[0], outs[1]), "backward after a forward-only pass differs"
        with pytest.raises(RuntimeError):
            b.forward(x4, 300.0, txt4, save_for_backward=True)
    print(cfg_name, "workspace bytes: slot per tensor at batch 4", a.workspace_bytes(), "-> saved 2 / forward-only 4", b.workspace_bytes())


def test_groupnorm_statistics_from_the_gemm_epilogue_match_the_statistics_kernel():
    """Round 6: an UNSPLIT k_gemm_dma launch whose output a GroupNorm consumes leaves that GroupNorm's slice statistics from its own
    epilogue (gemm.hip gn_epi: two "slices" per row tile, groups that straddle two 64-column tiles) instead of a k_gn_partial launch.
    The input-gradient GEMMs in front of a GroupNorm backward do the same for the BACKWARD statistics (gn_epi == 2, stage bit 3).
    Two engines on the same weights, one captured with both fusions (the shipped path), one with dh_dbg_gemm_stage(1 | 4 | 8) = statistics
    kernels as in round 5: a network whose groups are 8 / 16 channels wide (the fusion needs >= 8; the TINY net's 2-channel groups never
    take it) and whose GEMMs at the 32 x 32 / 16 x 16 levels are short enough not to split K.  The two differ by the f32 summation order
    of the statistics only (see the comment at the asserts for what can be asserted about two fp16 runs)."""
    from diffusionhandles_amd import _lib
    from diffusionhandles_amd.unet import HipUNet
    from oracle import unet_torch as U
    cfg = dict(in_channels=5, out_channels=4, block_out_channels=(256, 256, 512, 512), layers_per_block=2, heads=(4, 4, 8, 8),
               cross_attention_dim=64, norm_groups=32, sample_size=32)
    lib = _lib.lib()
    ref = U.init_synthetic_(U.UNetTorch(cfg), seed=5).to(dev()).eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.half().float())
    g = torch.Generator(device=dev()).manual_seed(23)
    B = 2
    sample = torch.randn(B, 5, 32, 32, generator=g, device=dev())
    text = torch.randn(B, 77, 64, generator=g, device=dev())
    d_act = [None, None, (torch.randn(B, 32, 32, 256, generator=g, device=dev()) * 0.05).half()]
    outs = {}
    try:
        for name, stage in (("epilogue", 1), ("kernel", 1 | 4 | 8)):
            _lib.check(lib.dh_dbg_gemm_stage(stage), "dh_dbg_gemm_stage")
            hip = HipUNet(dict(cfg, text_len=77), dtype=torch.float16, max_batch=B)
            hip.load_state_dict(ref.state_dict())
            with torch.cuda.stream(torch.cuda.Stream()):
                eps, acts = hip.forward(sample.permute(0, 2, 3, 1).contiguous(), 500.0, text, save_for_backward=True)
                d_sample, _ = hip.backward(d_act, None, True, False)
                torch.cuda.synchronize()
            outs[name] = (eps.clone(), [a.clone() for a in acts], d_sample.clone())
            del hip
    finally:
        lib.dh_dbg_gemm_stage(1)
    o = ref(sample, torch.tensor(500.0), text, return_dict=False)
    vs_oracle = {name: rel(outs[name][0].permute(0, 3, 1, 2), o[0]) for name in outs}
    e_eps = rel(outs["epilogue"][0], outs["kernel"][0])
    e_act = max(rel(a, b2) for a, b2 in zip(outs["epilogue"][1], outs["kernel"][1]))
    e_bwd = rel(outs["epilogue"][2], outs["kernel"][2])
    print(f"GroupNorm statistics from the GEMM epilogue vs the statistics kernel: eps {e_eps:.2e}, activations {e_act:.2e}, d_sample {e_bwd:.2e}; "
          f"each engine vs the fp32 oracle: {vs_oracle}")
    # Both engines sit at the fp16 noise floor of this network against the fp32 oracle, and so does their mutual difference: statistics that
    # differ in the last f32 bits flip 16-bit roundings in the first normalised tensor, and sixty layers later the two runs are two
    # realisations of the same rounding noise.  (The statistics themselves are held to 1e-5 by
    # tests/test_unet_kernels_gpu.py::test_gemm_groupnorm_statistics_by_producer.)  What is asserted: neither engine is further from the
    # oracle than the gate, the fused one is not worse than the other by more than a fifth, and their difference stays below twice the floor.
    floor = max(vs_oracle.values())
    assert floor < 5e-3, vs_oracle
    assert vs_oracle["epilogue"] < 1.2 * vs_oracle["kernel"] + 1e-4, vs_oracle
    assert e_eps < 2.0 * floor and e_act < 2.0 * floor and e_bwd < 3.0 * floor, (e_eps, e_act, e_bwd, floor)
