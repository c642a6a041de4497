"""Synthetic inputs for tests and benchmarks (no weights / EXR files needed).

The analytic scene of SURVEY.md section 8d: a receding background plane with a sphere in
front of it; the foreground mask is the sphere's silhouette.
"""
import numpy as np
import torch

# the 6-edit transform set (angle deg about +y, translation) + two more for batch-8
TRANSFORMS = [
    (0.0, (0.0, 0.0, 0.0)),
    (15.0, (0.0, 0.0, 0.0)),
    (30.0, (0.0, 0.0, 0.0)),
    (60.0, (0.0, 0.0, 0.0)),
    (0.0, (0.3, 0.0, 0.2)),
    (0.0, (-1.0, 0.0, 0.0)),
    (-20.0, (0.5, 0.0, -0.3)),
    (45.0, (-0.4, 0.1, 0.4)),
]


def make_scene(res=512):
    """Returns depth, bg_depth, fg_mask as [1,1,res,res] float32 CPU tensors."""
    h = res
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(h, dtype=np.float64), indexing="ij")
    bg = 4.0 + 1.5 * (h - 1 - yy) / h
    cy, cx, rad = h / 2.0, 300.0 * h / 512.0, 110.0 * h / 512.0
    r2 = ((yy - cy) ** 2 + (xx - cx) ** 2) / (rad * rad)
    mask = r2 < 1.0
    sphere = 2.6 - 0.6 * np.sqrt(np.clip(1.0 - r2, 0.0, None))
    depth = np.where(mask, sphere, bg)
    t = lambda a: torch.from_numpy(a.astype(np.float32))[None, None].contiguous()
    return t(depth), t(bg), t(mask.astype(np.float32))


def make_image(res=512, seed=7):
    """A smooth synthetic RGB image in [0,1], [1,3,res,res]."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, 16, 16, generator=g)
    return torch.nn.functional.interpolate(low, size=(res, res), mode="bicubic", align_corners=False).clamp(0, 1)


# ---- stand-ins for the components whose weights are not available offline ------------------
class SyntheticTokenizer:
    """Whitespace tokenizer with a hashed vocabulary; same call surface as CLIPTokenizer."""
    model_max_length = 77

    def __call__(self, texts, padding=None, max_length=None, truncation=None, return_tensors=None):
        L = max_length or self.model_max_length
        ids = torch.zeros((len(texts), L), dtype=torch.int64)
        for i, t in enumerate(texts):
            toks = [1] + [2 + (sum(w.encode()) * 2654435761 % 49000) for w in t.lower().split()][: L - 2] + [0]
            ids[i, : len(toks)] = torch.tensor(toks)
        from types import SimpleNamespace
        return SimpleNamespace(input_ids=ids)


class SyntheticTextEncoder(torch.nn.Module):
    """Seeded embedding table + positional term -> [B,77,dim] (stand-in for CLIPTextModel)."""

    def __init__(self, dim=1024, vocab=49408, seed=1):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.emb = torch.nn.Parameter(torch.randn(vocab, dim, generator=g), requires_grad=False)
        self.pos = torch.nn.Parameter(0.1 * torch.randn(77, dim, generator=g), requires_grad=False)

    def forward(self, input_ids):
        return (self.emb[input_ids] + self.pos[None, : input_ids.shape[1]],)


class SyntheticVAE(torch.nn.Module):
    """Deterministic 8x linear autoencoder stand-in (avg-pool encode, nearest decode) exposing the
    AutoencoderKL call surface the loops touch (.config, .encode()['latent_dist'].mean, .decode())."""

    def __init__(self):
        super().__init__()
        from types import SimpleNamespace
        self.config = SimpleNamespace(scaling_factor=0.18215, block_out_channels=(128, 256, 512, 512))

    def encode(self, x):
        from types import SimpleNamespace
        z = torch.nn.functional.avg_pool2d(x, 8)
        z = torch.cat([z, z.mean(dim=1, keepdim=True)], dim=1)
        return {"latent_dist": SimpleNamespace(mean=z)}

    def decode(self, z, return_dict=True):
        img = torch.nn.functional.interpolate(z[:, :3], scale_factor=8.0, mode="nearest")
        return (img,) if return_dict is False else {"sample": img}
