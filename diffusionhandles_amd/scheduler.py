"""DDIM schedule used by the loops (diffusers-0.23 DDIMScheduler restated [ext]; constructed by
the reference at guided_stable_diffuser.py:31-32 with beta_start .00085, beta_end .012,
scaled_linear, clip_sample False, set_alpha_to_one False; leading spacing, offset 0, eta 0).

Only scalars live here; the per-element update runs in the HIP library (dh_ddim_cfg_step).
"""
from types import SimpleNamespace

import numpy as np
import torch


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps)
        self.num_inference_steps = None
        self.timesteps = None
        self.set_timesteps(50)

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def alpha(self, t):
        t = int(t)
        return float(self.alphas_cumprod[t]) if t >= 0 else float(self.final_alpha_cumprod)

    def step_alphas(self, t):
        """(alpha_t, alpha_prev) of the denoising step x_t -> x_{t - ratio}."""
        t = int(t)
        return self.alpha(t), self.alpha(t - self.config.num_train_timesteps // self.num_inference_steps)

    def inversion_alphas(self, t):
        """(alpha_from, alpha_to) of the inversion step x_{t - ratio} -> x_t (next_step)."""
        t = int(t)
        tp = min(t - self.config.num_train_timesteps // self.num_inference_steps, 999)
        return self.alpha(tp), self.alpha(t)

    def add_noise(self, x, noise, t):
        a = self.alphas_cumprod[int(t)]
        return a ** 0.5 * x + (1 - a) ** 0.5 * noise
