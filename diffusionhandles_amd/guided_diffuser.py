"""Abstract diffuser interface (the reference's plugin seam, guided_diffuser.py:5-35)."""
import torch


class GuidedDiffuser:
    def __init__(self, conf):
        self.conf = conf

    def to(self, device: torch.device = None, dtype: torch.dtype = None):
        raise NotImplementedError

    @staticmethod
    def get_depth_intrinsics(device: torch.device = None):
        raise NotImplementedError

    def encode_latent_image(self, image):
        raise NotImplementedError

    def decode_latent_image(self, latent_image):
        raise NotImplementedError

    def initial_inference(self, init_latents, depth, uncond_embeddings, prompt):
        raise NotImplementedError

    def guided_inference(self, latents, depth, uncond_embeddings, prompt, activations_orig, correspondences,
                         save_denoising_steps=False):
        raise NotImplementedError
