"""DiffusionHandles facade (reference diffusion_handles.py:15-166) on the native MI355X path."""
import torch

from . import conf as _conf
from .depth_transform import laplacian_depth_blend, normalize_depth, transform_depth
from .guided_stable_diffuser import GuidedStableDiffuser
from .stable_null_inverter import StableNullInverter


class DiffusionHandles:
    def __init__(self, conf=None, **diffuser_kwargs):
        if conf is None:
            conf = _conf.load_default()
        self.conf = _conf.Conf.wrap(conf) if isinstance(conf, dict) else conf
        self.diffuser = GuidedStableDiffuser(conf=self.conf.guided_diffuser, **diffuser_kwargs)
        self.inverter = StableNullInverter(self.diffuser)
        self.device = torch.device("cpu")

    def to(self, device=None):
        self.diffuser.to(device=device)
        self.inverter.to(device=device)
        self.device = torch.device(device)
        return self

    def invert_input_image(self, img, depth, prompt):
        """-> (null_text_emb [T,1,77,D], init_noise [1,4,h,w])"""
        disparity = normalize_depth(1.0 / depth)
        _, init_noise, null_text_emb = self.inverter.invert(target_img=img, depth=disparity, prompt=prompt,
                                                            num_inner_steps=5, verbose=False)
        return null_text_emb, init_noise

    def generate_input_image(self, depth, prompt, null_text_emb=None, init_noise=None):
        """-> (null_text_emb, init_noise, activations [3], latent_image)"""
        disparity = normalize_depth(1.0 / depth)
        with torch.no_grad():
            activations, latent_image, null_text_emb, init_noise = self.diffuser.initial_inference(
                init_latents=init_noise, depth=disparity, uncond_embeddings=null_text_emb, prompt=prompt)
        return null_text_emb, init_noise, activations, latent_image

    def set_foreground(self, depth, fg_mask, bg_depth):
        """Background depth = input depth with the hole of the (15x cross-dilated) foreground mask in-filled
        from the background depth's Laplacian (reference diffusion_handles.py:90-111)."""
        return laplacian_depth_blend(depth, bg_depth, fg_mask, dilate_iterations=15)

    def transform_foreground_batch(self, depth, prompt, fg_mask, bg_depth, null_text_emb, init_noise, activations,
                                   transforms, fg_weight=None, bg_weight=None, use_input_depth_normalization=False,
                                   streams=1, batch=None):
        """K edits of one image in batched passes (not in the reference; BASELINE config 3).
        transforms: list of (rot_angle_deg, rot_axis[3], translation[3]).  Returns (images [K,3,H,W], [K disparities]).
        streams > 1: the K edits are cut into chunks of `batch` (default ceil(K / streams)) and the chunks run on `streams`
        concurrent lanes that share the U-Net weights (GuidedStableDiffuser.fork); batch = 1 runs single (B = 1) edits on the
        lanes.  Images are bit-identical to the one-stream result at the same batch."""
        from .depth_transform import reproject_edits
        K = len(transforms)
        with torch.no_grad():
            edits = reproject_edits(depth, bg_depth, fg_mask, self.diffuser.get_depth_intrinsics(device=depth.device),
                                    transforms, use_input_depth_normalization, device_correspondences=True)
            per = K if batch is None and streams <= 1 else int(batch or -(-K // max(1, int(streams))))
            if streams <= 1 and per >= K:
                imgs = self.diffuser.guided_inference_batch(init_noise, [d for d, _ in edits], null_text_emb, prompt,
                                                            activations, [c for _, c in edits], fg_weight, bg_weight)
            elif per <= 1:
                imgs = torch.cat(self.diffuser.guided_inference_lanes(init_noise, edits, null_text_emb, prompt, activations,
                                                                      max(1, int(streams)), fg_weight, bg_weight))
            else:
                chunks = [([d for d, _ in edits[i:i + per]], [c for _, c in edits[i:i + per]]) for i in range(0, K, per)]
                imgs = torch.cat(self.diffuser.guided_inference_batch_lanes(init_noise, chunks, null_text_emb, prompt,
                                                                            activations, max(1, int(streams)), fg_weight,
                                                                            bg_weight))
        return imgs, [d for d, _ in edits]

    def transform_foreground(self, depth, prompt, fg_mask, bg_depth, null_text_emb, init_noise, activations,
                             rot_angle=None, rot_axis=None, translation=None, fg_weight=None, bg_weight=None,
                             use_input_depth_normalization=False):
        with torch.no_grad():
            edited_disparity, correspondences = transform_depth(
                depth=depth, bg_depth=bg_depth, fg_mask=fg_mask,
                intrinsics=self.diffuser.get_depth_intrinsics(device=depth.device),
                rot_angle=rot_angle, rot_axis=rot_axis, translation=translation,
                use_input_depth_normalization=use_input_depth_normalization,
                depth_transform_mode=self.conf.depth_transform_mode)
            results = self.diffuser.guided_inference(
                latents=init_noise, depth=edited_disparity, uncond_embeddings=null_text_emb, prompt=prompt,
                activations_orig=activations, correspondences=correspondences, fg_weight=fg_weight,
                bg_weight=bg_weight, save_denoising_steps=self.conf.guided_diffuser.save_denoising_steps)
        if self.conf.guided_diffuser.save_denoising_steps:
            edited_img, denoising_steps = results
            return edited_img, edited_disparity, denoising_steps
        return results, edited_disparity
