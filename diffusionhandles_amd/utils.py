"""Correspondence packing helpers (reference: diffhandles/utils.py:111-117)."""
import torch


def pack_correspondences(original_x, original_y, transformed_x, transformed_y):
    """Four [N] tensors -> [N,4] (ox, oy, tx, ty)."""
    return torch.stack((original_x, original_y, transformed_x, transformed_y), dim=-1)


def unpack_correspondences(correspondences):
    """[N,4] -> four [N,1] tensors."""
    return torch.split(correspondences, 1, dim=-1)
