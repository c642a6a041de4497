"""Multi-GPU driver helpers: independent edits shard embarrassingly.

One process per GPU (torchrun / torch.distributed env), weights replicated, round-robin
assignment of (image, transform) work items, NO collective on the data path; results are
gathered with one all_gather_object at the very end (control plane only).
"""
import os

import torch


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_edits(items, rank=None, world=None):
    """Round-robin slice of the work list for this rank (stable, covers every item exactly once)."""
    if rank is None or world is None:
        rank, world = rank_world()
    return [it for i, it in enumerate(items) if i % world == rank]


def gather_results(local_results, group=None):
    """Collect per-rank result lists on every rank, restoring the global round-robin order."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return list(local_results)
    world = dist.get_world_size(group)
    buckets = [None] * world
    dist.all_gather_object(buckets, list(local_results), group=group)
    out, i = [], 0
    while any(i < len(b) for b in buckets):
        for b in buckets:
            if i < len(b):
                out.append(b[i])
        i += 1
    return out


def run_edits(dh, image_identity, edits, depth, fg_mask, bg_depth, prompt):
    """This rank's share of `edits` (list of dicts with rot_angle / rot_axis / translation) on one image
    identity (null_text_emb, init_noise, activations); returns [(global_index, image, disparity)]."""
    rank, world = rank_world()
    null_text, noise, acts = image_identity
    out = []
    for gi, e in shard_edits(list(enumerate(edits)), rank, world):
        img, disp = dh.transform_foreground(depth, prompt, fg_mask, bg_depth, null_text, noise, acts,
                                            rot_angle=e.get("rot_angle"), rot_axis=e.get("rot_axis"),
                                            translation=e.get("translation"))
        out.append((gi, img.cpu(), disp.cpu()))
    return out
