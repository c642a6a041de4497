"""BASELINE config 5 (768x768 edit, bf16 U-Net + f32 guidance backward) as a tested configuration.

What "f32 guidance backward" means in this build: the guidance energy and its gradient are computed
in f32 from the 16-bit activations, the cotangent handed to the engine is rounded ONCE to the engine dtype (bf16: scale 1),
every backward kernel accumulates in f32 (MFMA accumulators, GroupNorm / LayerNorm statistics, softmax / dS of the attention
backward), the latent gradient leaves the engine in f32 and the latent update is f32; only the tensors BETWEEN kernels are
bf16.  The tests below hold each piece of that chain at 96x96 latents against the f32 oracles.
"""
import numpy as np
import pytest
import torch

from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


def test_engine_sd2_depth_latent96_bf16():
    """Full SD-2-depth configuration at 96x96 latents in bf16, B = 1 and the CFG shape B = 2 (9216 / 2304 / 576 / 144 rows
    per image): forward and backward-to-sample against the torch fp32 restatement (measured 1.1e-2 / 1.4e-2)."""
    from oracle import unet_torch as U
    from test_unet_engine_gpu import run_case
    run_case(dict(U.SD2_DEPTH, sample_size=96), torch.bfloat16, 2, 381.0, 3e-2, 5e-2, check_text=False)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_energy_planned_at_grid_96_vs_oracle_autograd(dtype):
    """energy_and_grad_planned on 96x96 cells (768 px / 8) for the two guided layers' channel counts against the
    oracle's closed form differentiated by autograd in f32: loss within 2e-5, gradient within one rounding of the
    engine dtype (the kernel computes in f32 and rounds the cotangent once)."""
    from diffusionhandles_amd import losses as LS
    from oracle import depth_ref as D
    from oracle import guidance_ref as G
    depth, bg, mask = make_scene(768)
    ang, tr = TRANSFORMS[2]
    _, corr = D.transform_depth_pc(depth, bg, mask, rot_angle=ang, rot_axis=[0, 1, 0], translation=tr)
    pc = LS.process_correspondences(corr, 768, 0, grid=96, device=dev())
    cells = G.cells_from_correspondences(corr.numpy(), 768, 0, grid=96)
    assert int(np.asarray(pc["original_x"]).max()) > 64        # the edit does use cells beyond the 64-grid
    plan = LS.EnergyPlan(pc, 96, dev())
    gen = torch.Generator().manual_seed(12)
    scale = 256.0 if dtype == torch.float16 else 1.0
    for C in (320, 640):
        cur = torch.randn(96, 96, C, generator=gen).to(dtype)
        org = torch.randn(96, 96, C, generator=gen).to(dtype)
        for fw, bw in ((11.25, 1.875), (0.0, 3.75), (18.75, 0.0)):
            loss, grad = LS.energy_and_grad_planned(cur.to(dev()), org.to(dev()), plan, fw, bw, grad_scale=scale, want_loss=True)
            a = cur.float().permute(2, 0, 1).clone().requires_grad_(True)
            o = org.float().permute(2, 0, 1)
            ref = fw * G.foreground_energy(a, o, cells, 1, (96, 96)) + bw * G.background_energy(a, o, cells, 1, (96, 96))
            gr, = torch.autograd.grad(ref, a)
            assert abs(loss[0].item() - ref.item()) < 2e-5 * abs(ref.item()), (C, fw, bw, loss[0].item(), ref.item())
            got = grad.float().cpu().permute(2, 0, 1) / scale
            eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10          # one rounding to the 16-bit type
            assert (got - gr).abs().max() <= eps * gr.abs().max() + 1e-9, (C, fw, bw)
            assert grad.dtype == dtype and grad.shape == (96, 96, C)


def _rig768(dtype):
    from diffusionhandles_amd import conf as C
    from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
    from diffusionhandles_amd.unet import HipUNet
    from oracle import unet_torch as U
    cfg = dict(U.TINY, sample_size=96)
    ref = U.init_synthetic_(U.UNetTorch(cfg), seed=0).to(dev()).eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(dtype).float())
    hip = HipUNet(dict(cfg, text_len=77), dtype=dtype, max_batch=2)
    hip.load_state_dict(ref.state_dict())
    conf = C.load_default().guided_diffuser
    gd = GuidedStableDiffuser(conf, unet=hip, unet_config=dict(cfg, text_len=77), dtype=dtype).to(dev())
    return ref, gd, conf


@pytest.mark.parametrize("dtype,tol_up,tol_step", [(torch.bfloat16, 8e-2, 3e-2), (torch.float16, 3e-2, 1e-2)])
def test_guided_step_at_768_matches_oracle(dtype, tol_up, tol_step):
    """One guided-denoise step at 768x768 (TINY U-Net at 96x96 latents, the real re-projection, cells on the 96 grid) against
    oracle.loop_ref: the first latent update (pure guidance gradient through the engine's backward) and the latent after
    the step.  bf16: measured 4.0e-2 / 6.8e-3 (8 mantissa bits between kernels); fp16 for comparison 1.3e-2 / 8.8e-4."""
    from diffusionhandles_amd import depth_transform as DT
    from oracle import depth_ref as D
    from oracle import loop_ref as L
    ref, gd, conf = _rig768(dtype)
    depth, bg, mask = make_scene(768)
    dsp = D.normalize_depth(1.0 / depth)[0].to(dev())
    ang, tr = TRANSFORMS[2]
    disp_e, corr = DT.transform_depth(depth.to(dev()), bg.to(dev()), mask.to(dev()), gd.get_depth_intrinsics(), rot_angle=ang,
                                      rot_axis=torch.tensor([0.0, 1.0, 0.0]), translation=torch.tensor(tr))
    _, corr_o = D.transform_depth_pc(depth, bg, mask, rot_angle=ang, rot_axis=[0, 1, 0], translation=tr)
    assert np.array_equal(corr.numpy(), corr_o.numpy())                      # integer maps bit-exact at 768 too
    cond = gd._encode(["a sphere"])
    unc = gd._encode([""])[None].expand(50, -1, -1, -1).contiguous()
    noise = torch.randn(1, 4, 96, 96, generator=torch.Generator().manual_seed(2773)).to(dev())

    class OneStep(L.DDIM):
        def set_timesteps(self, n):
            super().set_timesteps(n)
            self.timesteps = self.timesteps[:1]
    acts_o, _, _, _ = L.initial_inference(ref, OneStep(), noise, dsp, unc, cond)
    assert acts_o[1].shape[-2:] == (96, 96) and acts_o[0].shape[-2:] == (48, 48)
    st = gd.prepare_guidance(disp_e, "a sphere", [a.expand(50, -1, -1, -1) for a in acts_o], corr)
    assert st.plan is not None and st.plan.grid == 96                        # the planned energy path runs at this grid
    gd.scheduler.set_timesteps(50)
    rec_p, rec_o = {}, {}
    with gd.on_stream():
        x = gd.guided_step(st, noise.permute(0, 2, 3, 1).contiguous(), 0, gd.scheduler.timesteps[0], unc[0], rec_p)
    L.guided_inference(ref, OneStep(), noise, disp_e, unc, cond, [a.float() for a in acts_o], corr.numpy(), conf, record=rec_o)
    up = rel(rec_p["opt"][0] - noise, rec_o["opt"][0] - noise)
    step = rel(x.permute(0, 3, 1, 2), rec_o["step"][0])
    print(f"{dtype}: first guidance update rel err {up:.3e} (norm {(rec_o['opt'][0] - noise).norm().item():.3e}), step rel err {step:.3e}")
    assert up < tol_up and step < tol_step
