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


def broadcast_identity(identity, src=0, device=None, group=None):
    """One-shot hand-over of a per-image identity (null_text_emb, init_noise, activations[3]) from rank `src` to every
    rank, so that only one rank pays inversion + initial inference when an image's edits are spread over GPUs (SURVEY
    section 8e: 0.53 GB fp16 + 15.8 MB + 64 KB per image).  Off the per-edit path: one broadcast per tensor (RCCL on the
    GPU, gloo on CPU tensors); ranks other than `src` pass identity=None.  Without a process group it is the identity."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return identity
    rank = dist.get_rank(group)
    meta = [None]
    if rank == src:
        null_text, noise, acts = identity
        tensors = [null_text, noise] + list(acts)
        meta[0] = [(tuple(t.shape), str(t.dtype).replace("torch.", "")) for t in tensors]
    dist.broadcast_object_list(meta, src=src, group=group)
    # gloo (CPU tests; the one-GPU rehearsal of the multi-rank drivers) moves host tensors; RCCL moves them GPU to GPU
    wire = device if dist.get_backend(group) != "gloo" else torch.device("cpu")
    out = []
    for i, (shape, dt) in enumerate(meta[0]):
        if rank == src:
            t = tensors[i].contiguous()          # activations are channels-last views: the layout ranks agree on is [T,C,h,w] contiguous
            if wire is not None:
                t = t.to(wire)
        else:
            t = torch.empty(shape, dtype=getattr(torch, dt), device=wire)
        dist.broadcast(t, src=src, group=group)
        out.append(t if device is None else t.to(device))
    return out[0], out[1], out[2:]


def run_edits(dh, image_identity, edits, depth, fg_mask, bg_depth, prompt, batch=8, streams=1):
    """This rank's share of `edits` (list of dicts with rot_angle / rot_axis / translation) on one image identity
    (null_text_emb, init_noise, activations), executed `batch` edits at a time as ONE batched pass
    (DiffusionHandles.transform_foreground_batch; BASELINE config 4: 64 edits, 8 per GPU).  batch <= 1 runs them one by
    one through transform_foreground.  streams > 1: the rank's chunks of `batch` edits run on that many concurrent lanes of
    one process (two engine arenas and streams on ONE copy of the weights; images bit-identical to streams = 1 at the same
    batch).  Returns [(global_index, image [3,H,W] cpu, disparity [1,1,H,W] cpu)]; no collective is involved."""
    rank, world = rank_world()
    null_text, noise, acts = image_identity
    mine = shard_edits(list(enumerate(edits)), rank, world)
    out = []
    if streams > 1 and mine:
        tfs = [(e.get("rot_angle"), e.get("rot_axis"), e.get("translation")) for _, e in mine]
        imgs, disps = dh.transform_foreground_batch(depth, prompt, fg_mask, bg_depth, null_text, noise, acts, tfs,
                                                    streams=streams, batch=max(1, batch))
        return [(gi, imgs[k].cpu(), disps[k].cpu()) for k, (gi, _) in enumerate(mine)]
    if batch <= 1:
        for gi, e in mine:
            img, disp = dh.transform_foreground(depth, prompt, fg_mask, bg_depth, null_text, noise, acts,
                                                rot_angle=e.get("rot_angle"), rot_axis=e.get("rot_axis"),
                                                translation=e.get("translation"))
            out.append((gi, img[0].cpu(), disp.cpu()))
        return out
    for b0 in range(0, len(mine), batch):
        chunk = mine[b0:b0 + batch]
        tfs = [(e.get("rot_angle"), e.get("rot_axis"), e.get("translation")) for _, e in chunk]
        imgs, disps = dh.transform_foreground_batch(depth, prompt, fg_mask, bg_depth, null_text, noise, acts, tfs)
        for k, (gi, _) in enumerate(chunk):
            out.append((gi, imgs[k].cpu(), disps[k].cpu()))
    return out
