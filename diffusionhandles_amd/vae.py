"""SD-2 AutoencoderKL and CLIP text tower in plain PyTorch-ROCm (once per image / edit,
"keep in PyTorch-ROCm first").

The reference takes both from third-party packages that are not in its tree: `diffusers.AutoencoderKL`
(`guided_stable_diffuser.py:9,29`, used at `:93-108, 481-483` and `stable_null_inverter.py:72-110`) and
`transformers.CLIPTextModel` (`:8,34-35`).  diffusers is not installed here, so the VAE is restated from its
published structure with **diffusers' state-dict key names** (a real `vae/diffusion_pytorch_model.safetensors`
loads with `load_state_dict`); parity is unpinned (no weights, no diffusers).  The text tower is
transformers' own `CLIPTextModel` built from the SD-2 configuration (random init offline, `from_pretrained`
when a checkpoint directory is given).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

SD_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
              layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
SD2_TEXT = dict(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23,
                num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5,
                projection_dim=512, pad_token_id=1, bos_token_id=0, eos_token_id=2)


class _Resnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return h + (x if self.conv_shortcut is None else self.conv_shortcut(x))


class _Attention(nn.Module):
    """Single-head spatial self-attention of the VAE mid block (diffusers Attention, residual connection)."""

    def __init__(self, ch, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Identity()])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        o = F.scaled_dot_product_attention(self.to_q(t)[:, None], self.to_k(t)[:, None], self.to_v(t)[:, None])[:, 0]
        return x + self.to_out[0](o).transpose(1, 2).reshape(b, c, h, w)


class _Mid(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.attentions = nn.ModuleList([_Attention(ch, groups)])
        self.resnets = nn.ModuleList([_Resnet(ch, ch, groups), _Resnet(ch, ch, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Down(nn.Module):
    def __init__(self, cin, cout, n, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.downsamplers = nn.ModuleList([nn.ModuleDict(dict(conv=nn.Conv2d(cout, cout, 3, stride=2)))]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0]["conv"](F.pad(x, (0, 1, 0, 1)))      # diffusers pads right/bottom, padding=0
        return x


class _Up(nn.Module):
    def __init__(self, cin, cout, n, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.upsamplers = nn.ModuleList([nn.ModuleDict(dict(conv=nn.Conv2d(cout, cout, 3, padding=1)))]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0]["conv"](F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return x


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        ch, g = c["block_out_channels"], c["norm_num_groups"]
        self.conv_in = nn.Conv2d(c["in_channels"], ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([_Down(ch[max(i - 1, 0)], ch[i], c["layers_per_block"], g, i < len(ch) - 1)
                                          for i in range(len(ch))])
        self.mid_block = _Mid(ch[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * c["latent_channels"], 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class _Decoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        ch, g = list(reversed(c["block_out_channels"])), c["norm_num_groups"]
        self.conv_in = nn.Conv2d(c["latent_channels"], ch[0], 3, padding=1)
        self.mid_block = _Mid(ch[0], g)
        self.up_blocks = nn.ModuleList([_Up(ch[max(i - 1, 0)], ch[i], c["layers_per_block"] + 1, g, i < len(ch) - 1)
                                        for i in range(len(ch))])
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], c["out_channels"], 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class _Gaussian:
    """diffusers DiagonalGaussianDistribution: `.mean`, `.sample(generator)`, `.mode()`."""

    def __init__(self, moments):
        self.mean, self.logvar = moments.chunk(2, dim=1)
        self.logvar = self.logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(self.mean.shape, generator=generator, device=self.mean.device,
                                                   dtype=self.mean.dtype)

    def mode(self):
        return self.mean


class _Out(dict):
    """dict with attribute access, like diffusers' BaseOutput (`out['sample']`, `out.sample`, `out[0]`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        return list(self.values())[k] if isinstance(k, int) else dict.__getitem__(self, k)


class AutoencoderKL(nn.Module):
    """`vae.encode(img)['latent_dist'].mean`, `vae.decode(z)['sample']`, `vae.config.scaling_factor` as the loops use
    them (guided_stable_diffuser.py:93-108, 481-483; stable_null_inverter.py:72-110)."""

    def __init__(self, config=None):
        super().__init__()
        c = dict(SD_VAE, **(config or {}))
        self.config = SimpleNamespace(**c)
        self.encoder, self.decoder = _Encoder(c), _Decoder(c)
        self.quant_conv = nn.Conv2d(2 * c["latent_channels"], 2 * c["latent_channels"], 1)
        self.post_quant_conv = nn.Conv2d(c["latent_channels"], c["latent_channels"], 1)

    def encode(self, x, return_dict=True):
        dist = _Gaussian(self.quant_conv(self.encoder(x)))
        return _Out(latent_dist=dist) if return_dict else (dist,)

    def decode(self, z, return_dict=True):
        img = self.decoder(self.post_quant_conv(z))
        return _Out(sample=img) if return_dict else (img,)

    @classmethod
    def from_safetensors(cls, path, config=None):
        from safetensors.torch import load_file
        m = cls(config)
        m.load_state_dict(load_file(path))
        return m


def build_text_encoder(pretrained_dir=None, config=None):
    """transformers.CLIPTextModel with the SD-2 text configuration (`text_encoder(ids)[0]` -> [B,77,1024])."""
    from transformers import CLIPTextConfig, CLIPTextModel
    if pretrained_dir is not None:
        return CLIPTextModel.from_pretrained(pretrained_dir)
    return CLIPTextModel(CLIPTextConfig(**dict(SD2_TEXT, **(config or {})))).eval()


class HipVAEDecoder:
    """The decoder half of `AutoencoderKL` on the native engine kernels (csrc/vae_engine.cpp): MFMA implicit-GEMM
    convolutions, the engine's GroupNorm, the single 512-dim attention head as two GEMMs around a row softmax.
    `decode(z)` has AutoencoderKL.decode's call surface (z [B,4,h,w] already divided by the scaling factor); the 1x1
    `post_quant_conv` (a 4x4 matrix per latent pixel) stays a torch op.  Weights come from an AutoencoderKL state dict."""

    _CREATE = "dh_vae_decoder_create"

    def __init__(self, config=None, latent_size=64, dtype=torch.float16, device=None):
        import ctypes
        from . import _lib
        _lib.require_gpu()
        c = dict(SD_VAE, **(config or {}))
        self.config = SimpleNamespace(**c)
        self.dtype, self.latent_size = dtype, int(latent_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        cfg = _lib.VAEConfig()
        cfg.latent_channels, cfg.out_channels = c["latent_channels"], c["out_channels"]
        for i in range(4):
            cfg.block_out_channels[i] = c["block_out_channels"][i]
        cfg.layers_per_block, cfg.norm_groups = c["layers_per_block"], c["norm_num_groups"]
        cfg.latent_size, cfg.dtype = self.latent_size, _lib.DTYPE_CODE[dtype]
        self._L = _lib.lib()
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(getattr(self._L, self._CREATE)(ctypes.byref(cfg), ctypes.byref(h)), self._CREATE)
        self._h = h
        self._pq_w = self._pq_b = None
        self._table = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.dh_vae_decoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def param_table(self):
        import ctypes
        from . import _lib
        if self._table is None:
            tab = []
            for i in range(self._L.dh_vae_decoder_num_params(self._h)):
                name, nd, shp = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int64 * 4)()
                _lib.check(self._L.dh_vae_decoder_param_info(self._h, i, ctypes.byref(name), ctypes.byref(nd), shp))
                tab.append((name.value.decode(), tuple(int(shp[k]) for k in range(nd.value))))
            self._table = tab
        return self._table

    def load_state_dict(self, sd):
        """sd: an AutoencoderKL state dict (diffusers names); the `decoder.*` and `post_quant_conv.*` entries are used."""
        from . import _lib
        st = _lib.stream_ptr()
        for i, (name, shape) in enumerate(self.param_table()):
            t = sd[name].detach()
            if t.dim() == 4 and len(shape) == 2:          # a Linear stored as a 1x1 convolution (older checkpoints)
                t = t.reshape(shape)
            if tuple(t.shape) != shape:
                raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
            t = t.to(self.device, torch.float32).contiguous()
            _lib.check(self._L.dh_vae_decoder_load_param(self._h, i, _lib.ptr(t), st), f"load {name}")
        self._pq_w = sd["post_quant_conv.weight"].detach().to(self.device, torch.float32).reshape(
            self.config.latent_channels, self.config.latent_channels).contiguous()
        self._pq_b = sd["post_quant_conv.bias"].detach().to(self.device, torch.float32).contiguous()
        torch.cuda.synchronize(self.device)
        return self

    def workspace_bytes(self):
        return int(self._L.dh_vae_decoder_bytes(self._h))

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        from . import _lib
        B, C, h, w = z.shape
        if h != self.latent_size or w != self.latent_size:
            raise ValueError(f"decoder built for {self.latent_size}x{self.latent_size} latents, got {h}x{w}")
        zl = z.to(self.device, torch.float32).permute(0, 2, 3, 1)                       # channels-last
        zl = (zl @ self._pq_w.t() + self._pq_b).contiguous()                           # post_quant_conv (1x1)
        img = torch.empty((B, 8 * h, 8 * w, self.config.out_channels), dtype=torch.float32, device=self.device)
        _lib.check(self._L.dh_vae_decoder_decode(self._h, _lib.ptr(zl), B, _lib.ptr(img), _lib.stream_ptr()),
                   "dh_vae_decoder_decode")
        img = img.permute(0, 3, 1, 2)
        return _Out(sample=img) if return_dict else (img,)


class HipVAEEncoder(HipVAEDecoder):
    """The encoder half of `AutoencoderKL` on the same kernels (csrc/vae_engine.cpp, `dh_vae_encoder_*`): `encode(x)` has
    AutoencoderKL.encode's call surface (x [B,3,H,W] in [-1, 1]) and returns the latent distribution; the 1x1 `quant_conv`
    (an 8x8 matrix per latent pixel) stays a torch op.  Weights: the `encoder.*` / `quant_conv.*` entries of the state dict."""

    _CREATE = "dh_vae_encoder_create"

    def load_state_dict(self, sd):
        from . import _lib
        st = _lib.stream_ptr()
        for i, (name, shape) in enumerate(self.param_table()):
            t = sd[name].detach()
            if t.dim() == 4 and len(shape) == 2:
                t = t.reshape(shape)
            if tuple(t.shape) != shape:
                raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
            t = t.to(self.device, torch.float32).contiguous()
            _lib.check(self._L.dh_vae_decoder_load_param(self._h, i, _lib.ptr(t), st), f"load {name}")
        m = 2 * self.config.latent_channels
        self._q_w = sd["quant_conv.weight"].detach().to(self.device, torch.float32).reshape(m, m).contiguous()
        self._q_b = sd["quant_conv.bias"].detach().to(self.device, torch.float32).contiguous()
        torch.cuda.synchronize(self.device)
        return self

    def decode(self, z, return_dict=True):
        raise NotImplementedError("this is the encoder half")

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        from . import _lib
        B, C, H, W = x.shape
        if H != 8 * self.latent_size or W != 8 * self.latent_size:
            raise ValueError(f"encoder built for {8 * self.latent_size}x{8 * self.latent_size} images, got {H}x{W}")
        xl = x.to(self.device, torch.float32).permute(0, 2, 3, 1).contiguous()          # channels-last
        m = 2 * self.config.latent_channels
        mom = torch.empty((B, self.latent_size, self.latent_size, m), dtype=torch.float32, device=self.device)
        _lib.check(self._L.dh_vae_encoder_encode(self._h, _lib.ptr(xl), B, _lib.ptr(mom), _lib.stream_ptr()),
                   "dh_vae_encoder_encode")
        mom = (mom @ self._q_w.t() + self._q_b).permute(0, 3, 1, 2).contiguous()         # quant_conv (1x1)
        dist = _Gaussian(mom)
        return _Out(latent_dist=dist) if return_dict else (dist,)


class NativeDecodeVAE(nn.Module):
    """AutoencoderKL whose `decode` AND `encode` run on the native engine kernels (the module keeps the parameters)."""

    def __init__(self, vae, latent_size=64, dtype=torch.float16):
        super().__init__()
        self.vae, self.config = vae, vae.config
        self._latent_size, self._dtype, self._dec, self._enc = latent_size, dtype, None, None

    def to(self, device):
        self.vae = self.vae.to(device)
        self._device = torch.device(device)
        if self._device.type == "cuda":
            self._dec = HipVAEDecoder(vars(self.vae.config), self._latent_size, self._dtype, device).load_state_dict(
                self.vae.state_dict())
            self._enc = None            # built on the first encode(): only the inversion of an input image needs it
        return self

    def _native_size(self, h, w):
        return self._dec is not None and h == self._latent_size and w == self._latent_size

    def encode(self, x, return_dict=True):
        if not self._native_size(x.shape[-2] // 8, x.shape[-1] // 8) or x.shape[-1] % 8 or x.shape[-2] % 8:
            return self.vae.encode(x, return_dict)          # another resolution than the engines were built for (or not on a GPU yet)
        if self._enc is None:
            self._enc = HipVAEEncoder(vars(self.vae.config), self._latent_size, self._dtype, self._device).load_state_dict(
                self.vae.state_dict())
        return self._enc.encode(x, return_dict)

    def decode(self, z, return_dict=True):
        if not self._native_size(z.shape[-2], z.shape[-1]):
            return self.vae.decode(z, return_dict)          # the same fallback as encode
        return self._dec.decode(z, return_dict)


class HipTextEncoder:
    """transformers' `CLIPTextModel` call surface (`enc(ids)[0]` -> last_hidden_state [B,77,hidden]) with the transformer on the
    native engine kernels (csrc/text_engine.cpp): LayerNorm, MFMA GEMMs, the causal flash-attention forward, erf-GELU.  The token
    and position embedding tables stay torch tensors (a gather); weights come from a CLIPTextModel state dict."""

    def __init__(self, config=None, dtype=torch.float16, max_batch=2, device=None):
        import ctypes
        from . import _lib
        _lib.require_gpu()
        c = dict(SD2_TEXT, **(config or {}))
        if c.get("hidden_act", "gelu") != "gelu":
            raise NotImplementedError("the native text tower implements the erf GELU of the SD-2 text encoder")
        self.config = SimpleNamespace(**c)
        self.dtype = dtype
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        cfg = _lib.TextConfig()
        cfg.hidden, cfg.heads, cfg.layers = c["hidden_size"], c["num_attention_heads"], c["num_hidden_layers"]
        cfg.intermediate, cfg.max_tokens, cfg.max_batch = c["intermediate_size"], c["max_position_embeddings"], int(max_batch)
        cfg.eps, cfg.dtype = float(c.get("layer_norm_eps", 1e-5)), _lib.DTYPE_CODE[dtype]
        self._L = _lib.lib()
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._L.dh_text_encoder_create(ctypes.byref(cfg), ctypes.byref(h)), "dh_text_encoder_create")
        self._h = h
        self._tok = self._pos = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.dh_text_encoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def load_state_dict(self, sd):
        import ctypes
        from . import _lib
        st = _lib.stream_ptr()

        def get(key):       # published checkpoints carry the "text_model." prefix; newer transformers state dicts do not
            return sd[key] if key in sd else sd[key[len("text_model."):]]

        for i in range(self._L.dh_text_encoder_num_params(self._h)):
            name, nd, shp = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int64 * 2)()
            _lib.check(self._L.dh_text_encoder_param_info(self._h, i, ctypes.byref(name), ctypes.byref(nd), shp))
            key, shape = name.value.decode(), tuple(int(shp[k]) for k in range(nd.value))
            t = get(key).detach()
            if tuple(t.shape) != shape:
                raise ValueError(f"{key}: expected shape {shape}, got {tuple(t.shape)}")
            t = t.to(self.device, torch.float32).contiguous()
            _lib.check(self._L.dh_text_encoder_load_param(self._h, i, _lib.ptr(t), st), f"load {key}")
        self._tok = get("text_model.embeddings.token_embedding.weight").detach().to(self.device, torch.float32)
        self._pos = get("text_model.embeddings.position_embedding.weight").detach().to(self.device, torch.float32)
        torch.cuda.synchronize(self.device)
        return self

    def workspace_bytes(self):
        return int(self._L.dh_text_encoder_bytes(self._h))

    def to(self, device):
        return self

    def eval(self):
        return self

    @torch.no_grad()
    def __call__(self, input_ids, attention_mask=None):
        from . import _lib
        if attention_mask is not None:
            raise NotImplementedError("the reference calls the text encoder without a padding mask")
        ids = input_ids.to(self.device)
        B, T = ids.shape
        emb = (self._tok[ids] + self._pos[:T]).contiguous()
        out = torch.empty_like(emb)
        _lib.check(self._L.dh_text_encoder_encode(self._h, _lib.ptr(emb), B, T, _lib.ptr(out), _lib.stream_ptr()),
                   "dh_text_encoder_encode")
        return (out,)
