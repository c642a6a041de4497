"""ORACLE (test infrastructure, never on the product path).

Plain-PyTorch fp32 restatement of the SD-2-depth U-Net forward with activation capture:
the configuration the reference instantiates through diffusers 0.23 [ext: diffusers is not
vendored in /root/reference and not installed -> parity with diffusers itself is unpinned;
the structure follows SURVEY.md section 8 a3 and the patched model files
model/unet_2d_condition.py:809-1198, model/unet_2d_blocks.py, model/transformer_2d.py:242-444,
model/attention.py:219-342, model/attention_processor.py:1178-1262].

Parameter names follow the diffusers state-dict convention so real weights can be loaded.
Returns the reference's 7-tuple (eps, None, None, None, act0, act1, act2) when
return_dict=False and {'sample': eps} otherwise.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

SD2_DEPTH = dict(in_channels=5, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, heads=(5, 10, 20, 20), cross_attention_dim=1024,
                 norm_groups=32, sample_size=64)

TINY = dict(in_channels=5, out_channels=4, block_out_channels=(64, 128, 128, 128),
            layers_per_block=2, heads=(1, 2, 2, 2), cross_attention_dim=64,
            norm_groups=32, sample_size=64)


def sinusoid(t, dim):
    """Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    ang = t.float()[:, None] * freq[None]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


class Resnet(nn.Module):
    def __init__(self, cin, cout, temb, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-5)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-5)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class Attn(nn.Module):
    def __init__(self, dim, heads, kv_dim):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(kv_dim, dim, bias=False)
        self.to_v = nn.Linear(kv_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim)])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        b, n, c = x.shape
        hd = c // self.heads
        q = self.to_q(x).view(b, n, self.heads, hd).transpose(1, 2)
        k = self.to_k(ctx).view(b, -1, self.heads, hd).transpose(1, 2)
        v = self.to_v(ctx).view(b, -1, self.heads, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
        o = torch.softmax(s, dim=-1) @ v
        return self.to_out[0](o.transpose(1, 2).reshape(b, n, c))


class GEGLU(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Linear(dim, dim * 8)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)


class FF(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class TBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attn(dim, heads, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attn(dim, heads, ctx_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FF(dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        return x + self.ff(self.norm3(x))


class Transformer2D(nn.Module):
    def __init__(self, dim, heads, ctx_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList([TBlock(dim, heads, ctx_dim)])
        self.proj_out = nn.Linear(dim, dim)

    def forward(self, x, ctx):
        b, c, h, w = x.shape
        t = self.norm(x).permute(0, 2, 3, 1).reshape(b, h * w, c)
        t = self.proj_in(t)
        for blk in self.transformer_blocks:
            t = blk(t, ctx)
        t = self.proj_out(t)
        return t.reshape(b, h, w, c).permute(0, 3, 1, 2) + x


class Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, n, heads, ctx, groups, attn, down):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(cin if i == 0 else cout, cout, temb, groups) for i in range(n)])
        if attn:
            self.attentions = nn.ModuleList([Transformer2D(cout, heads, ctx, groups) for _ in range(n)])
        self.has_attn = attn
        if down:
            self.downsamplers = nn.ModuleList([Down(cout)])
        self.has_down = down

    def forward(self, x, temb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.has_down:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, c, temb, heads, ctx, groups):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(c, c, temb, groups), Resnet(c, c, temb, groups)])
        self.attentions = nn.ModuleList([Transformer2D(c, heads, ctx, groups)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, cprev, cout, temb, n, heads, ctx, groups, attn, up):
        super().__init__()
        rs = []
        for i in range(n):
            skip = cin if i == n - 1 else cout
            rin = cprev if i == 0 else cout
            rs.append(Resnet(rin + skip, cout, temb, groups))
        self.resnets = nn.ModuleList(rs)
        if attn:
            self.attentions = nn.ModuleList([Transformer2D(cout, heads, ctx, groups) for _ in range(n)])
        self.has_attn = attn
        if up:
            self.upsamplers = nn.ModuleList([Up(cout)])
        self.has_up = up

    def forward(self, x, skips, temb, ctx):
        for i, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips.pop()], dim=1), temb)
            if self.has_attn:
                x = self.attentions[i](x, ctx)
        if self.has_up:
            x = self.upsamplers[0](x)
        return x


class UNetTorch(nn.Module):
    def __init__(self, cfg=None, save_activations=True):
        super().__init__()
        cfg = dict(SD2_DEPTH if cfg is None else cfg)
        self.cfg = cfg
        ch = cfg["block_out_channels"]
        n, g, ctx, heads = cfg["layers_per_block"], cfg["norm_groups"], cfg["cross_attention_dim"], cfg["heads"]
        temb = ch[0] * 4
        self.config = SimpleNamespace(in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                                      sample_size=cfg["sample_size"])
        self.sample_size = cfg["sample_size"]
        self.save_activations = save_activations
        self.conv_in = nn.Conv2d(cfg["in_channels"], ch[0], 3, padding=1)
        self.time_embedding = nn.ModuleDict(dict(linear_1=nn.Linear(ch[0], temb), linear_2=nn.Linear(temb, temb)))
        L = len(ch)
        self.down_blocks = nn.ModuleList([
            DownBlock(ch[max(i - 1, 0)], ch[i], temb, n, heads[i], ctx, g, attn=i < L - 1, down=i < L - 1)
            for i in range(L)])
        self.mid_block = MidBlock(ch[-1], temb, heads[-1], ctx, g)
        rch, rheads = list(reversed(ch)), list(reversed(heads))
        self.up_blocks = nn.ModuleList([
            UpBlock(rch[min(i + 1, L - 1)], rch[max(i - 1, 0)], rch[i], temb, n + 1, rheads[i], ctx, g,
                    attn=i > 0, up=i < L - 1)
            for i in range(L)])
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=1e-5)
        self.conv_out = nn.Conv2d(ch[0], cfg["out_channels"], 3, padding=1)

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None, return_dict=True):
        t = torch.as_tensor(timestep, device=sample.device).reshape(-1).expand(sample.shape[0])
        e = sinusoid(t, self.cfg["block_out_channels"][0]).to(sample.dtype)
        temb = self.time_embedding["linear_2"](F.silu(self.time_embedding["linear_1"](e)))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, temb, encoder_hidden_states)
        acts = []
        for blk in self.up_blocks:
            x = blk(x, skips, temb, encoder_hidden_states)
            if blk.has_attn:
                acts.append(x)
        eps = self.conv_out(F.silu(self.conv_norm_out(x)))
        if not return_dict:
            if self.save_activations:
                return (eps, None, None, None, acts[0], acts[1], acts[2])
            return (eps, None, None, None, None, None, None)
        return {"sample": eps}


def init_synthetic_(model, seed=0, scale=None):
    """Seeded synthetic weights: N(0, s^2) with s = 1/sqrt(fan_in) for matrices (keeps
    activations O(1)), norm gains 1 + 0.1 N, biases 0.02 N.  Deterministic given the seed
    and the parameter iteration order (sorted by name)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
            if p.dim() >= 2:
                fan_in = p[0].numel()
                s = (1.0 / math.sqrt(fan_in)) if scale is None else scale
                p.copy_(torch.randn(p.shape, generator=g) * s)
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    return model
