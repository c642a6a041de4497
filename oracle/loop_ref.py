"""ORACLE (test infrastructure, never on the product path).

CPU (torch fp32 + autograd) restatement of the denoising loops:
  DDIMScheduler [ext: diffusers 0.23]           SURVEY.md section 8 a16
  init_depth                                    guided_stable_diffuser.py:110-127
  initial_inference                             guided_stable_diffuser.py:155-275
  guided_inference                              guided_stable_diffuser.py:291-488
  StableNullInverter.{next,prev}_step, ddim_loop, null_optimization, invert
                                                stable_null_inverter.py:25-181
Works with any U-Net callable returning the reference's 7-tuple.  Pinned against the
imported reference loops (run with the same stand-in U-Net) by tools/make_golden.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import guidance_ref as G

CFG_SCALE = 7.5
VAE_SCALE = 0.18215


class DDIM:
    """beta scaled_linear(.00085,.012), 1000 train steps, leading spacing, eta 0, eps-pred,
    set_alpha_to_one=False (final alpha = alphas_cumprod[0])."""

    def __init__(self, num_train=1000, beta_start=0.00085, beta_end=0.012):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_train = num_train
        self.set_timesteps(50)

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.num_train // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64))

    def alpha(self, t):
        return self.alphas_cumprod[t] if t >= 0 else self.final_alpha_cumprod

    def step(self, eps, t, x):
        """x_t -> x_{t-ratio}."""
        t = int(t)
        a_t, a_p = self.alpha(t), self.alpha(t - self.num_train // self.num_inference_steps)
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps

    def add_noise(self, x, noise, t):
        a = self.alphas_cumprod[int(t)]
        return a ** 0.5 * x + (1 - a) ** 0.5 * noise

    def invert_step(self, eps, t, x):
        """x_{t-ratio} -> x_t (the inverter's next_step)."""
        t = int(t)
        tp = min(t - self.num_train // self.num_inference_steps, 999)
        a_t, a_n = self.alpha(tp), self.alphas_cumprod[t]
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_n ** 0.5 * x0 + (1 - a_n) ** 0.5 * eps


def init_depth(disparity, size):
    d = F.interpolate(disparity, size=size, mode="bicubic", align_corners=False)
    lo = torch.amin(d, dim=[1, 2, 3], keepdim=True)
    hi = torch.amax(d, dim=[1, 2, 3], keepdim=True)
    return 2.0 * (d - lo) / (hi - lo) - 1.0


def _with_depth(x, depth64):
    """use_depth (guided_stable_diffuser.py:218, 244, 400, 454): the depth map is a fifth input channel, or absent."""
    return x if depth64 is None else torch.cat([x, depth64], dim=1)


def _eps_cfg(unet, x, depth64, t, uncond, cond):
    inp = _with_depth(torch.cat([x] * 2), None if depth64 is None else torch.cat([depth64] * 2))
    e = unet(inp, t, encoder_hidden_states=torch.cat([uncond.expand(*cond.shape), cond]), return_dict=False)[0]
    eu, ec = e.chunk(2)
    return eu + CFG_SCALE * (ec - eu)


@torch.no_grad()
def initial_inference(unet, sched, init_latents, disparity, uncond_list, cond, num_steps=50, seed=2773, use_depth=True):
    torch.manual_seed(seed)
    sched.set_timesteps(num_steps)
    s = unet.config.sample_size
    depth64 = init_depth(disparity, (s, s)) if use_depth else None
    if init_latents is None:
        noise = torch.randn([1, unet.config.in_channels - 1 if use_depth else unet.config.in_channels, s, s])
        init_latents = sched.add_noise(torch.zeros_like(noise), noise, sched.timesteps[0])
    x = init_latents
    acts = ([], [], [])
    for i, t in enumerate(sched.timesteps):
        out = unet(_with_depth(x, depth64), t, encoder_hidden_states=cond, return_dict=False)
        for k in range(3):
            acts[k].append(out[4 + k][0])
        x = sched.step(_eps_cfg(unet, x, depth64, t, uncond_list[i], cond), t, x)
    return [torch.stack(a) for a in acts], x, uncond_list, init_latents


def guided_inference(unet, sched, latents, disparity, uncond_list, cond, acts_orig, corr, conf,
                     fg_weight=None, bg_weight=None, record=None, steps=None):
    """conf: namespace with the 13 guided_diffuser keys.  Returns the final latents.
    steps (tests only): run just these step indices of the loop (the same body, teacher-forced from `latents`); None = all."""
    fg_weight = conf.fg_weight if fg_weight is None else fg_weight
    bg_weight = conf.bg_weight if bg_weight is None else bg_weight
    torch.manual_seed(conf.seed)
    sched.set_timesteps(conf.num_timesteps)
    s = unet.config.sample_size
    # the reference hard-codes a 64 x 64 cell grid (guided_stable_diffuser.py:526-535: img_res // 64, masks of (64, 64)), which
    # is its U-Net's latent size: it supports 512 x 512 only.  The grid here is the latent size, identical at 512 x 512 and
    # the consistent choice elsewhere (768 x 768: 96 x 96 cells of 8 px, the resolution of the guided activations).
    cells = G.cells_from_correspondences(corr, disparity.shape[-1], conf.bg_erosion, grid=s)
    depth64 = init_depth(disparity, (s, s)) if getattr(conf, "use_depth", True) else None
    x = latents
    for i, t in enumerate(sched.timesteps):
        if steps is not None and i not in steps:
            continue
        size = tuple(acts_orig[2][i].shape[-2:])
        it = 0
        while it < conf.num_optsteps and i < conf.guidance_max_step:
            with torch.enable_grad():
                x = x.detach().requires_grad_(True)
                out = unet(_with_depth(x, depth64), t, encoder_hidden_states=cond, return_dict=False)
                fgw, bgw = G.guidance_weights(i, it, fg_weight, bg_weight, conf.guidance_max_step,
                                              conf.guidance_schedule_type)
                loss = 0.0
                for k in range(3):
                    loss = loss + fgw[k] * G.foreground_energy(out[4 + k][0], acts_orig[k][i], cells,
                                                               conf.fg_patch_size, size)
                    loss = loss + bgw[k] * G.background_energy(out[4 + k][0], acts_orig[k][i], cells,
                                                               conf.bg_patch_size, size, conf.bg_loss_type)
                g = torch.autograd.grad(loss, [x])[0]
            x = (x - 0.1 * g).detach()
            if record is not None:
                record.setdefault("opt", []).append(x.clone())
            it += 1
        with torch.no_grad():
            x = sched.step(_eps_cfg(unet, x, depth64, t, uncond_list[i], cond), t, x)
        if record is not None:
            record.setdefault("step", []).append(x.clone())
    return x


def _eps_single(unet, x, depth64, t, ctx):
    return unet(torch.cat([x, depth64[0].view(1, 1, *depth64.shape[2:])], dim=1), t, encoder_hidden_states=ctx)["sample"]


@torch.no_grad()
def get_noise_pred(unet, sched, latents, depth64, t, context, guidance_scale=CFG_SCALE, is_forward=True):
    """Reference stable_null_inverter.py:55-70 (StableNullInverter.get_noise_pred): one classifier-free-guidance DDIM move --
    a B = 2 pass over context = [uncond | cond], guidance scale 1 and next_step on the way up (is_forward), the configured
    scale and prev_step on the way down."""
    inp = _with_depth(torch.cat([latents] * 2), None if depth64 is None else torch.cat([depth64] * 2))
    e = unet(inp, t, encoder_hidden_states=context)["sample"]
    eu, ec = e.chunk(2)
    w = 1.0 if is_forward else guidance_scale
    eps = eu + w * (ec - eu)
    return sched.invert_step(eps, t, latents) if is_forward else sched.step(eps, t, latents)


def null_text_inversion(unet, sched, latent0, disparity, uncond0, cond, num_inner_steps=5, eps0=1e-5,
                        num_steps=50, null_steps=None, record=None):
    """Returns (ddim_latents list[51], uncond [50,1,77,C]).  `record` (a list) receives per timestep a dict with the
    state the inner loop started from (cur, uncond) and what it did (losses, gradient of the first inner step)."""
    sched.set_timesteps(num_steps)
    s = unet.config.sample_size
    depth64 = init_depth(disparity, (s, s))
    lat = [latent0]
    x = latent0.clone()
    with torch.no_grad():
        for i in range(num_steps):
            t = sched.timesteps[num_steps - 1 - i]
            x = sched.invert_step(_eps_single(unet, x, depth64, t, cond), t, x)
            lat.append(x)
    unc = uncond0
    out = []
    cur = lat[-1]
    for i in range(num_steps if null_steps is None else null_steps):
        unc = unc.clone().detach().requires_grad_(True)
        opt = torch.optim.Adam([unc], lr=1e-2 * (1.0 - i / 100.0))
        target = lat[len(lat) - i - 2]
        t = sched.timesteps[i]
        with torch.no_grad():
            e_c = _eps_single(unet, cur, depth64, t, cond)
        if record is not None:
            record.append(dict(cur=cur.clone(), uncond=unc.detach().clone(), target=target, loss=[], grad=[]))
        for j in range(num_inner_steps):
            e_u = _eps_single(unet, cur, depth64, t, unc)
            rec = sched.step(e_u + CFG_SCALE * (e_c - e_u), t, cur)
            loss = F.mse_loss(rec, target)
            opt.zero_grad()
            loss.backward()
            if record is not None:
                record[-1]["loss"].append(loss.item())
                record[-1]["grad"].append(unc.grad.detach().clone())
            opt.step()
            if loss.item() < eps0 + i * 2e-5:
                break
        if record is not None:
            record[-1]["uncond_out"] = unc.detach().clone()
        out.append(unc[:1].detach())
        with torch.no_grad():
            cur = sched.step(_eps_cfg(unet, cur, depth64, t, unc.detach(), cond), t, cur)
    return lat, torch.stack(out, dim=0)
