"""GuidedStableDiffuser on the native MI355X engine.

Same public surface as the reference class (guided_stable_diffuser.py:22-610): `.unet .vae
.text_encoder .tokenizer .scheduler .conf .device`, `to`, `get_image_shape`,
`get_feature_shape`, `init_prompt`, `init_depth`, `get_depth_intrinsics`,
`initial_inference`, `guided_inference`, `decode_latent_image`, `encode_latent_image`,
`process_correspondences`, `get_timesteps`, `prepare_extra_step_kwargs`, and
`StepGuidanceWeightSchedule`.

What differs underneath: the U-Net forward, the guidance energy with its gradient, the
backward-to-latent, the latent update and the CFG/DDIM step are explicit calls into the HIP
library (no autograd graph, no per-iteration host sync), activations stay channels-last
16-bit, and zero-weight energy terms are skipped (exact: 0 * finite = 0).
N = 0 / N = 1 correspondences are defined (fg term skipped / handled) instead of NaN / TypeError.
"""
import inspect
import os

import numpy as np
import torch

from . import _lib
from .guided_diffuser import GuidedDiffuser
from .losses import (EnergyPlan, energy_and_grad, energy_and_grad_planned,
                     process_correspondences as _process_correspondences)
from .scheduler import DDIMScheduler
from .unet import HipUNet, SD2_DEPTH

CFG_SCALE = 7.5


class GuidanceWeightSchedule:
    def __call__(self, denoising_step: int, optimization_step: int):
        return [1.0] * 3, [1.0] * 3


class StepGuidanceWeightSchedule(GuidanceWeightSchedule):
    """Piecewise-constant lookup: last entry with step <= query, per-layer product
    (guided_stable_diffuser.py:622-665)."""

    def __init__(self, denoising_steps, optimization_steps):
        for steps in (denoising_steps, optimization_steps):
            if not all(len(f) == len(b) for _, f, b in steps):
                raise ValueError("Number of foreground and background weights do not match.")
        if len(denoising_steps[0][1]) != len(optimization_steps[0][1]):
            raise ValueError("Number of denoising and optimization weights do not match.")
        self.denoising_steps = sorted(denoising_steps, key=lambda s: s[0])
        self.optimization_steps = sorted(optimization_steps, key=lambda s: s[0])

    @staticmethod
    def _lookup(table, q):
        hit = None
        for step, f, b in table:
            if q >= step:
                hit = (f, b)
        return hit

    def __call__(self, denoising_step: int, optimization_step: int):
        d = self._lookup(self.denoising_steps, denoising_step)
        o = self._lookup(self.optimization_steps, optimization_step)
        if d is None or o is None:
            raise ValueError(f"Could not find weights for denoising step {denoising_step} and optimization step "
                             f"{optimization_step}.")
        return [x * y for x, y in zip(d[0], o[0])], [x * y for x, y in zip(d[1], o[1])]


def build_weight_schedule(fg_weight, bg_weight, max_step, kind):
    """fg/bg_weight are the user-facing values; the x30 happens here (guided_stable_diffuser.py:336-373)."""
    wf, wb = fg_weight * 30, bg_weight * 30
    if kind == "constant":
        ff, fb = np.linspace(wf, wf, max_step), np.linspace(wb, wb, max_step)
    elif kind == "linear":
        ff, fb = np.linspace(wf, 0.0, max_step), np.linspace(wb, 0.0, max_step)
    elif kind == "quadratic":
        ff, fb = np.linspace(np.sqrt(wf), 0.0, max_step) ** 2, np.linspace(np.sqrt(wb), 0.0, max_step) ** 2
    else:
        raise ValueError(f"Unknown guidance schedule type: {kind}")
    pattern = [([0.0, 0.0, 7.5], [0.0, 0.0, 1.5]), ([0.0, 5.0, 0.0], [0.0, 1.5, 0.0]), ([0.0, 5.0, 7.5], [0.0, 1.5, 1.5])]
    den = []
    for t in range(max_step):
        pf, pb = pattern[t % 3]
        den.append((t, (np.array(pf) * ff[t]).tolist(), (np.array(pb) * fb[t]).tolist()))
    den.append((max_step, [0.0] * 3, [0.0] * 3))
    opt = [(0, [2.5] * 3, [1.25] * 3), (1, [1.25] * 3, [2.5] * 3), (2, [1.25] * 3, [1.25] * 3), (3, [2.5] * 3, [2.5] * 3)]
    return StepGuidanceWeightSchedule(den, opt)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


class GuidedStableDiffuser(GuidedDiffuser):
    _text_keys = 0

    def __init__(self, conf, unet=None, vae=None, text_encoder=None, tokenizer=None, dtype=torch.float16,
                 unet_config=None, max_batch=2, synthetic_seed=0):
        """max_batch sizes the engine this diffuser builds in .to(device) (ignored when `unet` is handed in): the largest batch of
        ANY U-Net pass.  Contract (since round 5): K batched edits need max_batch >= 2 K (their CFG pass runs [uncond | cond] for
        every edit; guided_step_batch / guided_inference_batch / guided_inference_streams raise RuntimeError otherwise, lanes are
        never silently re-sized), and a forward that is SAVED for a backward pass -- unet.forward(save_for_backward=True), i.e.
        the optimisation passes and null-text inversion -- may be at most max_batch // 2 wide (the engine's max_diff_batch:
        only those passes keep every activation, which is what halves the arenas); a wider saved forward fails with the
        engine's "exceeds max_diff_batch" error.  Build a HipUNet yourself (max_diff_batch=...) for any other split."""
        super().__init__(conf=conf)
        self.scheduler = DDIMScheduler()
        self.dtype = dtype
        self._unet_config = dict(SD2_DEPTH if unet_config is None else unet_config)
        self._max_batch = max(2, int(max_batch))
        self._synthetic_seed = synthetic_seed
        self.unet = unet            # built lazily in .to(device): the engine lives on a GPU
        self.vae = vae
        self.text_encoder = text_encoder
        self.tokenizer = tokenizer
        self.device = torch.device("cpu")
        # fp16 needs the guidance gradient scaled through the backward pass
        self.grad_scale = 256.0 if dtype == torch.float16 else 1.0
        # guided_step drives the engine through its own I/O buffers (no device copies / torch.cat around the passes); False
        # routes the same kernels through caller-owned tensors (bit-identical; kept for tools/ab_inplace.py)
        self._inplace_io = True

    # ---- plumbing -----------------------------------------------------------------------
    def to(self, device=None):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("GuidedStableDiffuser runs on an MI355X HIP device only (no CPU fallback)")
        _lib.require_gpu()
        with torch.cuda.device(device):
            if self.unet is None:
                # the optimisation passes (saved for their backward) run half the batch of the CFG pass: only they size the arenas
                self.unet = HipUNet(self._unet_config, dtype=self.dtype, max_batch=self._max_batch,
                                    max_diff_batch=max(1, self._max_batch // 2), device=device)
                wdir = os.environ.get("DIFFHANDLES_UNET_SAFETENSORS")
                if wdir:
                    from safetensors.torch import load_file
                    self.unet.load_state_dict(load_file(wdir))
                else:
                    self.unet.init_synthetic(self._synthetic_seed)     # no checkpoint offline: seeded weights
            from .synthetic import SyntheticTextEncoder, SyntheticTokenizer, SyntheticVAE
            from .vae import AutoencoderKL, build_text_encoder
            # Checkpoints named through the environment run on the ENGINE's kernels (csrc/vae_engine.cpp, csrc/text_engine.cpp):
            # the PyTorch-ROCm modules of vae.py only carry the parameters (and serve as the tests' oracles).  "sd" / "sd2"
            # name those torch modules with random weights (test oracles), "sd-native" / "sd2-native" the native engines with
            # random weights; with nothing given the cheap deterministic stand-ins of synthetic.py are used (no checkpoints offline).
            if self.tokenizer is None and os.environ.get("DIFFHANDLES_TOKENIZER_DIR"):
                from transformers import CLIPTokenizer
                self.tokenizer = CLIPTokenizer.from_pretrained(os.environ["DIFFHANDLES_TOKENIZER_DIR"])
            text_native = isinstance(self.text_encoder, str) and self.text_encoder.endswith("-native")
            if self.text_encoder is None and os.environ.get("DIFFHANDLES_TEXT_ENCODER_DIR"):
                self.text_encoder = build_text_encoder(os.environ["DIFFHANDLES_TEXT_ENCODER_DIR"])
                text_native = True
            elif isinstance(self.text_encoder, str):
                # (manual_seed also re-seeds every device generator, lazily: the fork must cover the current device as well, or a
                #  caller that draws noise on the device after building the diffuser gets a different stream)
                with torch.random.fork_rng(devices=[device]):      # random weights, but the SAME in every process (seeded like the U-Net's)
                    torch.manual_seed(1000 + self._synthetic_seed)
                    self.text_encoder = build_text_encoder()       # the SD-2 text configuration
            if text_native:
                from .vae import HipTextEncoder
                tc = self.text_encoder.config.to_dict()
                keys = ("hidden_size", "num_attention_heads", "num_hidden_layers", "intermediate_size", "max_position_embeddings",
                        "layer_norm_eps", "hidden_act", "vocab_size")
                self.text_encoder = HipTextEncoder({k: tc[k] for k in keys if k in tc}, self.dtype, max_batch=2,
                                                   device=device).load_state_dict(self.text_encoder.state_dict())
            from .vae import NativeDecodeVAE
            if self.vae is None and os.environ.get("DIFFHANDLES_VAE_SAFETENSORS"):
                self.vae = NativeDecodeVAE(AutoencoderKL.from_safetensors(os.environ["DIFFHANDLES_VAE_SAFETENSORS"]),
                                           self._unet_config["sample_size"], self.dtype)
            elif isinstance(self.vae, str):
                native = self.vae == "sd-native"
                with torch.random.fork_rng(devices=[device]):      # (two runs of a driver must decode with the same random VAE)
                    torch.manual_seed(2000 + self._synthetic_seed)
                    self.vae = AutoencoderKL()
                if native:
                    self.vae = NativeDecodeVAE(self.vae, self._unet_config["sample_size"], self.dtype)
            if self.tokenizer is None:
                self.tokenizer = SyntheticTokenizer()
            if self.text_encoder is None:
                self.text_encoder = SyntheticTextEncoder(dim=self._unet_config["cross_attention_dim"])
            if self.vae is None:
                self.vae = SyntheticVAE()
            self.text_encoder = self.text_encoder.to(device)
            self.vae = self.vae.to(device)
        self.device = device
        # a created (non-default) stream: the engine replays its passes as hipGraphs, which the legacy
        # default stream cannot capture
        self._stream = torch.cuda.Stream(device=device)
        return self

    class _on_stream:
        """Run a block on the diffuser's stream, ordered after / before the caller's current stream."""

        def __init__(self, gd):
            self.gd = gd

        def __enter__(self):
            self.outer = torch.cuda.current_stream(self.gd.device)
            self.gd._stream.wait_stream(self.outer)
            self.ctx = torch.cuda.stream(self.gd._stream)
            self.ctx.__enter__()

        def __exit__(self, *a):
            self.ctx.__exit__(*a)
            self.outer.wait_stream(self.gd._stream)

    def on_stream(self):
        return GuidedStableDiffuser._on_stream(self)

    def get_image_shape(self):
        f = self.get_feature_shape()
        s = 2 ** (len(self.vae.config.block_out_channels) - 1)
        return (f[0] * s, f[1] * s, 3)

    def get_feature_shape(self):
        hw = self.unet.sample_size
        hw = (hw, hw) if isinstance(hw, int) else hw
        return (hw[0], hw[1], self.unet.config.out_channels)

    def _encode(self, texts):
        ids = self.tokenizer(texts, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                             return_tensors="pt")
        return self.text_encoder(ids.input_ids.to(self.device))[0].float()

    @torch.no_grad()
    def init_prompt(self, prompt: str):
        return torch.cat([self._encode([""]), self._encode([prompt])])

    @torch.no_grad()
    def init_depth(self, depth):
        h, w = self.get_feature_shape()[:2]
        depth = torch.nn.functional.interpolate(depth, size=(h, w), mode="bicubic", align_corners=False)
        lo = torch.amin(depth, dim=[1, 2, 3], keepdim=True)
        hi = torch.amax(depth, dim=[1, 2, 3], keepdim=True)
        return 2.0 * (depth - lo) / (hi - lo) - 1.0

    @staticmethod
    def get_depth_intrinsics(device=None):
        f = 1.0 / np.tan(0.5 * 55.0 * (np.pi / 180.0))
        return torch.tensor([[f, 0, 0], [0, f, 0], [0, 0, 1]], dtype=torch.float32, device=device)

    def encode_latent_image(self, image):
        raise NotImplementedError

    def decode_latent_image(self, latent_image):
        image = self.vae.decode(latent_image / self.vae.config.scaling_factor, return_dict=False)[0]
        return (image / 2 + 0.5).clamp(0, 1)

    def process_correspondences(self, correspondences, img_res, bg_erosion=0):
        return _process_correspondences(correspondences, img_res, bg_erosion, grid=self.unet.sample_size,
                                        device=self.device)

    def get_timesteps(self, num_inference_steps, strength):
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start * self.scheduler.order:], num_inference_steps - t_start

    def prepare_extra_step_kwargs(self, generator, eta):
        keys = set(inspect.signature(self.ddim_step).parameters.keys())
        kw = {}
        if "eta" in keys:
            kw["eta"] = eta
        if "generator" in keys:
            kw["generator"] = generator
        return kw

    # ---- native loop pieces --------------------------------------------------------------
    def ddim_step(self, x, eps_u, eps_c, t, scale=CFG_SCALE, eta=0.0, generator=None):
        """x, eps: [B,H,W,4] f32 channels-last.  CFG combine + DDIM step in one kernel."""
        a_t, a_p = self.scheduler.step_alphas(t)
        out = torch.empty_like(x)
        L = _lib.lib()
        _lib.check(L.dh_ddim_cfg_step(_lib.ptr(out), _lib.ptr(x), _lib.ptr(eps_u), _lib.ptr(eps_c), float(scale), a_t,
                                      a_p, x.numel(), _lib.stream_ptr()), "dh_ddim_cfg_step")
        return out

    def _unet_input(self, x, depth_nhwc, reps=1):
        """x [1,H,W,4], depth [1,H,W,1] -> [reps,H,W,5]"""
        s = torch.cat([x, depth_nhwc], dim=-1) if self.conf.use_depth else x
        return s.expand(reps, -1, -1, -1).contiguous() if reps > 1 else s.contiguous()

    def _cfg_eps(self, x, depth_nhwc, t, uncond, cond, want_acts=False, inplace=False):
        """(eps_uncond, eps_cond[, activations]) of the B=2 classifier-free-guidance pass.  inplace: the input is packed into
        the engine's buffer by one launch and eps comes back as views of the engine's output (valid until the next pass)."""
        text = torch.cat([uncond.reshape(1, *cond.shape[1:]).to(self.device, torch.float32), cond]).contiguous()
        if inplace and x.shape[0] == 1:
            sample = self.unet.stage_sample(x, depth_nhwc if self.conf.use_depth else None, 2)
        else:
            sample = self._unet_input(x, depth_nhwc, 2)
        eps, acts = self.unet.forward(sample, float(t), text, save_for_backward=False, want_acts=want_acts, inplace=inplace)
        return (eps[0:1], eps[1:2], acts) if want_acts else (eps[0:1], eps[1:2])

    # ---- reference API ----------------------------------------------------------------------
    @torch.no_grad()
    def initial_inference(self, init_latents, depth, uncond_embeddings, prompt):
        """Returns (activations [3 x [T,C,h,w]], latents [1,4,H,W], uncond_embeddings, init_latents)."""
        with self.on_stream():
            return self._initial_inference(init_latents, depth, uncond_embeddings, prompt)

    def _initial_inference(self, init_latents, depth, uncond_embeddings, prompt):
        torch.manual_seed(self.conf.seed)
        self.scheduler.set_timesteps(self.conf.num_timesteps, device=self.device)
        timesteps, _ = self.get_timesteps(self.conf.num_timesteps, 1.0)
        depth_nhwc = _nhwc(self.init_depth(depth.to(self.device, torch.float32))) if self.conf.use_depth else None
        cond = self._encode([prompt]).contiguous()
        if uncond_embeddings is None:
            uncond_embeddings = self._encode([""])[None].expand(len(timesteps), -1, -1, -1)
        s = self.unet.sample_size
        if init_latents is None:
            nlat = self.unet.config.in_channels - 1 if self.conf.use_depth else self.unet.config.in_channels
            noise = torch.randn([1, nlat, s, s], dtype=torch.float32).to(self.device)
            init_latents = self.scheduler.add_noise(torch.zeros_like(noise), noise, timesteps[0])
        x = _nhwc(init_latents.to(self.device, torch.float32))
        T = len(timesteps)
        store = [torch.empty((T,) + shp, dtype=self.dtype, device=self.device) for shp in self.unet.act_shapes]
        for t_idx, t in enumerate(timesteps):
            # the reference runs a cond-only B=1 pass for the activations and then the B=2 CFG pass (guided_stable_diffuser.py
            # :222-257: three samples per step); the conditional half of the CFG pass has exactly the inputs of the B=1 pass, so
            # the activations are captured there (two samples per step).  Same values up to the engine's batch-dependent tile
            # selection (16-bit: < 1e-2 against the oracle's B=1 activations, tests/test_loops_gpu.py).
            eu, ec, acts = self._cfg_eps(x, depth_nhwc, t, uncond_embeddings[t_idx], cond, want_acts=True)
            for k in range(3):
                store[k][t_idx].copy_(acts[k][1])
            x = self.ddim_step(x, eu, ec, t)
        activations = [a.permute(0, 3, 1, 2) for a in store]      # [T,C,h,w] views of channels-last storage
        return activations, x.permute(0, 3, 1, 2), uncond_embeddings, init_latents

    def prepare_guidance(self, depth, prompt, activations_orig, correspondences, fg_weight=None, bg_weight=None, orig=None,
                         cond=None):
        """Everything of guided_inference that is constant over the denoising loop.  `orig`: the channels-last copies of
        the original activations of another guidance state of the SAME image (K edits of one image share them); `cond`: the
        prompt embedding when the caller already has it (lanes: the text tower is not run concurrently with itself)."""
        from types import SimpleNamespace
        fg_weight = self.conf.fg_weight if fg_weight is None else fg_weight
        bg_weight = self.conf.bg_weight if bg_weight is None else bg_weight
        st = SimpleNamespace()
        st.pc = self.process_correspondences(correspondences, img_res=depth.shape[-1], bg_erosion=self.conf.bg_erosion)
        st.depth_nhwc = _nhwc(self.init_depth(depth.to(self.device, torch.float32))) if self.conf.use_depth else None
        st.cond = self._encode([prompt]).contiguous() if cond is None else cond
        GuidedStableDiffuser._text_keys += 1
        st.cond_key = GuidedStableDiffuser._text_keys          # names st.cond for the engine's text K|V cache
        st.schedule = build_weight_schedule(fg_weight, bg_weight, self.conf.guidance_max_step,
                                            self.conf.guidance_schedule_type)
        # original activations as channels-last engine-dtype storage [T,h,w,C]
        st.orig = orig if orig is not None else \
            [a.to(self.device).permute(0, 2, 3, 1).to(self.dtype).contiguous() for a in activations_orig]
        st.size = (st.orig[2].shape[1], st.orig[2].shape[2])
        st.n_pairs = len(st.pc["original_x"])
        # default configuration: the energy runs through a per-edit plan (CSR + background flags built once)
        st.plan = None
        if (self.conf.fg_patch_size == 1 and self.conf.bg_loss_type == "global_avg"
                and all(o.shape[1] == st.size[0] and o.shape[2] == st.size[1] for o in st.orig[1:])):
            st.plan = EnergyPlan(st.pc, st.size[0], self.device)
        return st

    def _energy_grad(self, st, k, act, t_idx, fgw, bgw, out=None):
        """d(energy of layer k)/d(act) * grad_scale, act [h,w,C] channels-last; `out`: where to write it."""
        if st.plan is not None and act.shape[0] == st.plan.grid and act.shape[1] == st.plan.grid:
            return energy_and_grad_planned(act, st.orig[k][t_idx], st.plan, fgw, bgw, grad_scale=self.grad_scale, out=out)[1]
        return energy_and_grad(act, st.orig[k][t_idx], st.pc, fgw, bgw, self.conf.fg_patch_size, self.conf.bg_patch_size,
                               st.size, self.conf.bg_loss_type, grad_scale=self.grad_scale, out=out)[1]

    @staticmethod
    def _latent_buffer(st, x, iteration):
        """The latent of optimisation iteration `iteration`: two buffers per guidance state, used alternately (the update reads
        the previous iteration's buffer or the caller's tensor and writes the other one; the DDIM step that follows allocates its
        own output, so nothing the caller holds is ever overwritten) -- no allocation inside the step loop."""
        bufs = st.__dict__.setdefault("_xbuf", [None, None])
        k = iteration & 1
        if bufs[k] is None or bufs[k].shape != x.shape or bufs[k].device != x.device:
            bufs[k] = torch.empty_like(x)
        return bufs[k]

    def guided_step(self, st, x, t_idx, t, uncond, record=None, images=None):
        """One guided-denoise step (guided_stable_diffuser.py:377-479): up to num_optsteps x
        {U-Net forward, energy + gradient, backward-to-latent, latent update}, then the CFG
        forward (B=2) and the DDIM step.  x: [1,H,W,4] f32 channels-last.  `images` (save_denoising_steps): this
        timestep's list, which receives the decoded image after the optimisation loop and after the DDIM step
        (reference :385-386, 446-448, 476-478)."""
        L = _lib.lib()
        iteration = 0
        while iteration < self.conf.num_optsteps and t_idx < self.conf.guidance_max_step:
            fgw, bgw = st.schedule(t_idx, iteration)
            active = [k for k in range(3) if (fgw[k] != 0.0 and st.n_pairs > 0) or bgw[k] != 0.0]
            if active:
                # no copies either side of the engine: the input is packed into the engine's buffer by one launch, the
                # energy kernels read the captured activations where the engine left them and write their cotangents where
                # its backward pass starts from, the latent update reads d(sample) in place (its first 4 of 5 channels)
                ip = self._inplace_io
                sample = self.unet.stage_sample(x, st.depth_nhwc if self.conf.use_depth else None, 1) if ip else \
                    self._unet_input(x, st.depth_nhwc)
                _, acts = self.unet.forward(sample, float(t), st.cond, save_for_backward=True, want_acts=active, want_eps=False,
                                            text_key=st.cond_key, inplace=ip)
                d_acts = [None, None, None]
                for k in active:
                    d_acts[k] = self.unet.io_view("act_grad", k)[:1] if ip else torch.empty_like(acts[k])
                    self._energy_grad(st, k, acts[k][0], t_idx, fgw[k], bgw[k], out=d_acts[k][0])
                d_sample, _ = self.unet.backward(d_acts, None, want_sample_grad=True, want_text_grad=False, inplace=ip)
                x_new = self._latent_buffer(st, x, iteration)
                _lib.check(L.dh_latent_update_strided(_lib.ptr(x_new), _lib.ptr(x), _lib.ptr(d_sample), d_sample.shape[-1],
                                                      x.shape[-1], 0.1, self.grad_scale, x.numel() // x.shape[-1],
                                                      _lib.stream_ptr()), "dh_latent_update_strided")
                x = x_new
            if record is not None:
                record.setdefault("opt", []).append(x.permute(0, 3, 1, 2).clone())
            iteration += 1
        if images is not None:
            images.append(self.decode_latent_image(x.permute(0, 3, 1, 2)).cpu())
        eu, ec = self._cfg_eps(x, st.depth_nhwc, t, uncond, st.cond, inplace=self._inplace_io)      # (views: consumed by the step right here)
        x = self.ddim_step(x, eu, ec, t)
        if images is not None:
            images.append(self.decode_latent_image(x.permute(0, 3, 1, 2)).cpu())
        return x

    # ---- batched edits: K transforms of ONE image identity in one U-Net batch (BASELINE config 3) ------
    def guided_step_batch(self, sts, x, t_idx, t, uncond):
        """x: [K,H,W,4]; sts: K guidance states (same prompt / original activations, different edits)."""
        L = _lib.lib()
        K = x.shape[0]
        depth = torch.cat([st.depth_nhwc for st in sts], dim=0) if self.conf.use_depth else None
        cond = sts[0].cond.expand(K, -1, -1).contiguous()
        iteration = 0
        while iteration < self.conf.num_optsteps and t_idx < self.conf.guidance_max_step:
            fgw, bgw = sts[0].schedule(t_idx, iteration)
            active = [k for k in range(3) if fgw[k] != 0.0 or bgw[k] != 0.0]
            if active:
                # the same in-place engine I/O as guided_step: one pack launch for the K inputs, every edit's energy gradient
                # written straight into its slice of the engine's cotangent buffer, d(sample) read in place
                sample = self.unet.stage_sample(x, depth, K)
                _, acts = self.unet.forward(sample, float(t), cond, save_for_backward=True, want_acts=active, want_eps=False,
                                            text_key=sts[0].cond_key, inplace=True)
                d_acts = [None, None, None]
                for k in active:
                    d_acts[k] = self.unet.io_view("act_grad", k)[:K]
                    for e, st in enumerate(sts):
                        fw = fgw[k] if st.n_pairs > 0 else 0.0
                        self._energy_grad(st, k, acts[k][e], t_idx, fw, bgw[k], out=d_acts[k][e])
                d_sample, _ = self.unet.backward(d_acts, None, want_sample_grad=True, want_text_grad=False, inplace=True)
                x_new = self._latent_buffer(sts[0], x, iteration)
                _lib.check(L.dh_latent_update_strided(_lib.ptr(x_new), _lib.ptr(x), _lib.ptr(d_sample), d_sample.shape[-1],
                                                      x.shape[-1], 0.1, self.grad_scale, x.numel() // x.shape[-1],
                                                      _lib.stream_ptr()), "dh_latent_update_strided")
                x = x_new
            iteration += 1
        sample2 = self.unet.stage_sample(x, depth, 2 * K)          # the K edits twice: unconditional and conditional halves
        unc = uncond.reshape(1, *cond.shape[1:]).to(self.device, torch.float32).expand(K, -1, -1)
        text2 = torch.cat([unc, cond], dim=0).contiguous()
        eps, _ = self.unet.forward(sample2, float(t), text2, save_for_backward=False, want_acts=False, inplace=True)
        return self.ddim_step(x, eps[:K], eps[K:], t)

    def _batch_edit_steps(self, latents, depths, uncond_embeddings, prompt, activations_orig, correspondences_list,
                          fg_weight=None, bg_weight=None, cond=None, orig=None):
        """The batched edit as a generator: yields after the preparation and after every denoising step (everything is only
        ENQUEUED on the current stream by then), returns the final latents [K,4,H,W] through StopIteration.  One body for the
        one-stream call below and for the lanes of guided_inference_batch_lanes."""
        K = len(depths)
        if self.unet.max_batch < 2 * K:
            raise RuntimeError(f"engine max_batch {self.unet.max_batch} < 2*K = {2 * K}")
        torch.manual_seed(self.conf.seed)
        self.scheduler.set_timesteps(self.conf.num_timesteps, device=self.device)
        timesteps, _ = self.get_timesteps(self.conf.num_timesteps, 1.0)
        sts = []
        for d, c in zip(depths, correspondences_list):
            sts.append(self.prepare_guidance(d, prompt, activations_orig, c, fg_weight, bg_weight,
                                             orig=sts[0].orig if sts else orig, cond=sts[0].cond if sts else cond))
        x = _nhwc(latents.to(self.device, torch.float32)).expand(K, -1, -1, -1).contiguous()
        yield None
        for t_idx, t in enumerate(timesteps):
            x = self.guided_step_batch(sts, x, t_idx, t, uncond_embeddings[t_idx])
            yield None
        return x.permute(0, 3, 1, 2)

    def guided_inference_batch(self, latents, depths, uncond_embeddings, prompt, activations_orig, correspondences_list,
                               fg_weight=None, bg_weight=None):
        """K edits of one image at once.  depths: list of K edited disparities [1,1,H,W]; correspondences_list:
        K [N_k,4] tensors.  Needs an engine built with max_batch >= 2K.  Returns images [K,3,H,W]."""
        with torch.no_grad(), self.on_stream():
            gen = self._batch_edit_steps(latents, depths, uncond_embeddings, prompt, activations_orig, correspondences_list,
                                         fg_weight, bg_weight)
            try:
                while True:
                    next(gen)
            except StopIteration as done:
                self.last_latents = done.value
            return self.decode_latent_image(self.last_latents)

    # ---- lanes: concurrent edit streams on ONE copy of the weights --------------------------------------------------------
    def fork(self, max_batch=None):
        """A lane of this diffuser: the same configuration, VAE, text tower and U-Net WEIGHTS (HipUNet.share: resident once),
        with its own engine arenas, hipGraphs, stream and scheduler, so that a second edit can run next to the first one in the
        same process.  A single edit's passes are thousands of dependent launches of at most a few hundred workgroups: two
        independent edits interleave on the chip (two processes on one GPU measured 1.4x the steps/s of one,
        profiles/r03_bench_gloo2_one_gpu.json); lanes give that without a second copy of the weights."""
        import copy
        if self.unet is None:
            raise RuntimeError("fork() needs the diffuser on its device first (.to(device))")
        lane = copy.copy(self)
        lane.unet = self.unet.share(max_batch)
        lane._stream = torch.cuda.Stream(device=self.device)
        lane.scheduler = DDIMScheduler()
        lane._root = getattr(self, "_root", self)
        return lane

    def lanes(self, n, max_batch=None):
        """[self, fork, ...]: n lanes on this diffuser's weights.  ONE growing list of forks per max_batch: lanes(2) is a prefix
        of lanes(3), so a process that serves 8, 16 and 24 edits at streams = 3 holds two forks, not three sets (a fork owns a
        full engine arena: 28.4 GB at max_batch 16 in round 4, about half of it since the forward-only layout of round 5).  release_lanes() destroys them."""
        n = int(n)
        if n < 1:
            raise ValueError(f"lanes(n): n must be >= 1, got {n}")
        forks = self.__dict__.setdefault("_lane_forks", {}).setdefault(max_batch, [])
        while len(forks) < n - 1:
            forks.append(self.fork(max_batch))
        return [self] + forks[:n - 1]

    def lane_arena_bytes(self):
        """HBM held by the forks of this diffuser (all max_batch keys), for services that budget it."""
        return sum(f.unet.workspace_bytes() for forks in self.__dict__.get("_lane_forks", {}).values() for f in forks)

    def release_lanes(self):
        """Destroy every fork's engine (arenas, graphs) and forget them; the root diffuser and the shared weights stay."""
        for forks in self.__dict__.pop("_lane_forks", {}).values():
            for f in forks:
                f.unet.close()

    def _decode_serialized(self, latents):
        """VAE decode of a lane's result on the ROOT diffuser's decode stream: the decoder engine has one activation arena, so
        the lanes' decodes run one after the other (in host issue order) while the other lane keeps denoising."""
        root = getattr(self, "_root", self)
        if not hasattr(root, "_decode_stream"):
            root._decode_stream = torch.cuda.Stream(device=root.device)
        cur = torch.cuda.current_stream(self.device)
        root._decode_stream.wait_stream(cur)
        with torch.cuda.stream(root._decode_stream):
            img = root.decode_latent_image(latents)
        latents.record_stream(root._decode_stream)
        cur.wait_stream(root._decode_stream)
        return img

    @staticmethod
    def run_lanes(lanes, jobs):
        """Drive generator jobs on lanes from ONE host thread.  jobs: callables lane -> generator (a *_edit_steps body); job i
        runs on lanes[i % len(lanes)], a lane's jobs one after the other.  The host only enqueues: it advances every lane by one
        yield in turn, so the lanes' streams always hold work and the GPU interleaves them; nothing synchronises the lanes with
        each other.  Returns the generators' return values in job order (tensors produced on the lanes' streams; the caller's
        current stream is made to wait for every lane before this returns)."""
        import contextlib
        on_device = all(getattr(ln, "_stream", None) is not None for ln in lanes)      # (False: the scheduling alone, CPU tests)
        outer = torch.cuda.current_stream(lanes[0].device) if on_device else None
        for ln in lanes:
            if on_device:
                ln._stream.wait_stream(outer)
        queues = [[(i, job) for i, job in enumerate(jobs) if i % len(lanes) == li] for li in range(len(lanes))]
        running = [None] * len(lanes)
        results = [None] * len(jobs)
        with torch.no_grad():
            while any(q for q in queues) or any(r is not None for r in running):
                for li, ln in enumerate(lanes):
                    with (torch.cuda.stream(ln._stream) if on_device else contextlib.nullcontext()):
                        if running[li] is None and queues[li]:
                            i, job = queues[li].pop(0)
                            running[li] = (i, job(ln))
                        if running[li] is None:
                            continue
                        i, gen = running[li]
                        try:
                            next(gen)
                        except StopIteration as done:
                            results[i] = done.value
                            running[li] = None
        for ln in lanes:
            if on_device:
                outer.wait_stream(ln._stream)
        return results

    def guided_inference_batch_lanes(self, latents, chunks, uncond_embeddings, prompt, activations_orig, streams=2,
                                     fg_weight=None, bg_weight=None):
        """Batched edits of one image on `streams` concurrent lanes.  chunks: list of (depths, correspondences_list), each the
        argument pair of one guided_inference_batch call; chunk i runs on lane i % streams.  Every lane executes exactly the
        passes the one-stream call executes for its chunk (same batch, same kernels, private arenas), so the images are
        bit-identical to guided_inference_batch chunk by chunk.  Returns one image tensor [K_i,3,H,W] per chunk."""
        kmax = max(len(d) for d, _ in chunks)
        if self.unet.max_batch < 2 * kmax:
            # lane 0 is THIS diffuser: it could not run its chunk, and forks sized 2 * kmax would be made for nothing
            raise RuntimeError(f"engine max_batch {self.unet.max_batch} < 2*K = {2 * kmax}: build the diffuser with max_batch >= {2 * kmax}")
        lanes = self.lanes(min(int(streams), len(chunks)))
        with torch.no_grad(), self.on_stream():
            cond = self._encode([prompt]).contiguous()
            orig = [a.to(self.device).permute(0, 2, 3, 1).to(self.dtype).contiguous() for a in activations_orig]

            def make(depths, corrs):
                def job(lane):
                    def body():
                        lat = yield from lane._batch_edit_steps(latents, depths, uncond_embeddings, prompt, activations_orig,
                                                                corrs, fg_weight, bg_weight, cond=cond, orig=orig)
                        return lane._decode_serialized(lat)
                    return body()
                return job
            return GuidedStableDiffuser.run_lanes(lanes, [make(d, c) for d, c in chunks])

    def _edit_steps(self, latents, depth, uncond_embeddings, prompt, activations_orig, correspondences, fg_weight=None,
                    bg_weight=None, cond=None, orig=None):
        """One edit as a generator (see _batch_edit_steps): yields after the preparation and after every denoising step,
        returns the final latents [1,4,H,W]."""
        torch.manual_seed(self.conf.seed)
        self.scheduler.set_timesteps(self.conf.num_timesteps, device=self.device)
        timesteps, _ = self.get_timesteps(self.conf.num_timesteps, 1.0)
        st = self.prepare_guidance(depth, prompt, activations_orig, correspondences, fg_weight, bg_weight, orig=orig, cond=cond)
        x = _nhwc(latents.to(self.device, torch.float32))
        yield None
        for t_idx, t in enumerate(timesteps):
            x = self.guided_step(st, x, t_idx, t, uncond_embeddings[t_idx])
            yield None
        return x.permute(0, 3, 1, 2)

    def guided_inference_lanes(self, latents, edits, uncond_embeddings, prompt, activations_orig, streams=2, fg_weight=None,
                               bg_weight=None):
        """Single (B = 1) edits of one image on `streams` concurrent lanes.  edits: list of (depth, correspondences), the
        argument pair of guided_inference; edit i runs on lane i % streams with exactly the passes guided_inference runs, so
        the images are bit-identical to the one-stream calls.  Returns a list of images [1,3,H,W]."""
        lanes = self.lanes(min(int(streams), len(edits)))
        with torch.no_grad(), self.on_stream():
            cond = self._encode([prompt]).contiguous()
            orig = [a.to(self.device).permute(0, 2, 3, 1).to(self.dtype).contiguous() for a in activations_orig]

            def make(depth, corr):
                def job(lane):
                    def body():
                        lat = yield from lane._edit_steps(latents, depth, uncond_embeddings, prompt, activations_orig, corr,
                                                          fg_weight, bg_weight, cond=cond, orig=orig)
                        return lane._decode_serialized(lat)
                    return body()
                return job
            return GuidedStableDiffuser.run_lanes(lanes, [make(d, c) for d, c in edits])

    def guided_inference(self, latents, depth, uncond_embeddings, prompt, activations_orig, correspondences,
                         fg_weight=None, bg_weight=None, save_denoising_steps=False, record=None):
        with torch.no_grad(), self.on_stream():
            torch.manual_seed(self.conf.seed)
            self.scheduler.set_timesteps(self.conf.num_timesteps, device=self.device)
            timesteps, _ = self.get_timesteps(self.conf.num_timesteps, 1.0)
            st = self.prepare_guidance(depth, prompt, activations_orig, correspondences, fg_weight, bg_weight)
            denoising_steps = {"opt": [], "post-opt": []} if save_denoising_steps else None
            x = _nhwc(latents.to(self.device, torch.float32))
            for t_idx, t in enumerate(timesteps):
                if save_denoising_steps:      # one list per timestep in 'opt': [after the optimisation, after the DDIM step];
                    denoising_steps["opt"].append([])         # 'post-opt' stays empty, as in the reference
                x = self.guided_step(st, x, t_idx, t, uncond_embeddings[t_idx], record,
                                     denoising_steps["opt"][-1] if save_denoising_steps else None)
                if record is not None:
                    record.setdefault("step", []).append(x.permute(0, 3, 1, 2).clone())
            self.last_latents = x.permute(0, 3, 1, 2)
            image = self.decode_latent_image(self.last_latents)
        return (image, denoising_steps) if save_denoising_steps else image
