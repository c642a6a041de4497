"""Stand-ins for the diffusers-0.23 PRIMITIVES the reference's vendored U-Net files import
(`/root/reference/diffhandles/model/*.py`), so that those files run here AS THE REFERENCE'S
OWN CODE (diffusers is neither vendored nor installed).

Golden-generator infrastructure (tools/make_golden*.py only): the product and the oracle
never import this.  What is restated here -- and therefore stays "[ext] parity unpinned" -- is
exactly the leaf set: ResnetBlock2D, Upsample2D, Downsample2D, GEGLU, Timesteps /
TimestepEmbedding, get_activation, LoRACompatibleLinear / Conv, ModelMixin, ConfigMixin /
register_to_config.  Everything above the leaves (block wiring, skip order, capture points,
transformer block, attention processor, GroupNorm eps 1e-6, rescale_output_factor, the
7-tuple) is executed from the reference's files.
"""
import functools
import inspect
import math
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kw):
        d = dict(getattr(self, "_internal_dict", {}))
        d.update(kw)
        object.__setattr__(self, "_internal_dict", FrozenDict(d))


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        params = [(n, p.default) for i, (n, p) in enumerate(inspect.signature(init).parameters.items()) if i > 0]
        cfg = {n: a for a, (n, _) in zip(args, params)}
        for n, default in params:
            if n not in cfg:
                cfg[n] = kwargs.get(n, default)
        self.register_to_config(**cfg)
        init(self, *args, **kwargs)
    return inner


class ModelMixin(nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype


class LoRACompatibleLinear(nn.Linear):
    lora_layer = None

    def forward(self, x, scale=1.0):
        return super().forward(x)


class LoRACompatibleConv(nn.Conv2d):
    lora_layer = None

    def forward(self, x, scale=1.0):
        return super().forward(x)


def get_activation(name):
    name = name.lower()
    if name in ("swish", "silu"):
        return nn.SiLU()
    if name == "mish":
        return nn.Mish()
    if name == "gelu":
        return nn.GELU()
    if name == "relu":
        return nn.ReLU()
    raise ValueError(name)


class GEGLU(nn.Module):
    """diffusers.models.activations.GEGLU: proj to 2*dim_out, value * gelu_erf(gate)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = LoRACompatibleLinear(dim_in, dim_out * 2)

    def forward(self, x, scale=1.0):
        h, gate = self.proj(x, scale).chunk(2, dim=-1)
        return h * F.gelu(gate)


class Timesteps(nn.Module):
    """diffusers.models.embeddings.Timesteps / get_timestep_embedding (scale 1, max_period 1e4)."""

    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
        emb = torch.exp(exponent / (half - self.shift))
        emb = timesteps[:, None].float() * emb[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        super().__init__()
        assert post_act_fn is None and cond_proj_dim is None
        self.linear_1 = LoRACompatibleLinear(in_channels, time_embed_dim)
        self.act = get_activation(act_fn)
        self.linear_2 = LoRACompatibleLinear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


class ResnetBlock2D(nn.Module):
    """diffusers.models.resnet.ResnetBlock2D, the `default` time-embedding norm, no up/down."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", skip_time_act=False,
                 time_embedding_norm="default", kernel=None, output_scale_factor=1.0, use_in_shortcut=None,
                 up=False, down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        super().__init__()
        assert time_embedding_norm == "default" and not up and not down and kernel is None
        out_channels = in_channels if out_channels is None else out_channels
        self.output_scale_factor = output_scale_factor
        self.skip_time_act = skip_time_act
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = LoRACompatibleConv(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = LoRACompatibleLinear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups_out or groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        c2 = conv_2d_out_channels or out_channels
        self.conv2 = LoRACompatibleConv(out_channels, c2, kernel_size=3, stride=1, padding=1)
        self.nonlinearity = get_activation(non_linearity)
        use_in_shortcut = in_channels != c2 if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = LoRACompatibleConv(in_channels, c2, kernel_size=1, stride=1, padding=0,
                                                bias=conv_shortcut_bias) if use_in_shortcut else None

    def forward(self, input_tensor, temb, scale=1.0):
        h = self.conv1(self.nonlinearity(self.norm1(input_tensor)), scale)
        if self.time_emb_proj is not None:
            if not self.skip_time_act:
                temb = self.nonlinearity(temb)
            temb = self.time_emb_proj(temb, scale)[:, :, None, None]
        if temb is not None:
            h = h + temb
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))), scale)
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor, scale)
        return (input_tensor + h) / self.output_scale_factor


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        super().__init__()
        assert use_conv and not use_conv_transpose and name == "conv"
        self.conv = LoRACompatibleConv(channels, out_channels or channels, 3, padding=1)

    def forward(self, x, output_size=None, scale=1.0):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x, scale)


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv and padding == 1
        self.conv = LoRACompatibleConv(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, x, scale=1.0):
        return self.conv(x, scale)


class BaseOutput:
    """diffusers.utils.BaseOutput as far as callers use it: a dataclass readable by key or index."""

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return tuple(getattr(self, f) for f in self.__dataclass_fields__)[k]


class _Unused:
    """Placeholder for names the SD-2-depth configuration never instantiates."""

    def __init__(self, *a, **k):
        raise NotImplementedError("diffusers primitive outside the SD-2-depth configuration")


class _Logger:
    def info(self, *a, **k):
        pass
    warning = warn = debug = error = info


def install():
    """Register the stand-in `diffusers.*` modules in sys.modules (idempotent, additive)."""
    def mod(name, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            sys.modules[name] = m
        m.__dict__.update(attrs)
        return m

    mod("diffusers")
    mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config,
        FrozenDict=FrozenDict)
    mod("diffusers.loaders", UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {}))
    mod("diffusers.utils", USE_PEFT_BACKEND=False, BaseOutput=BaseOutput, deprecate=lambda *a, **k: None,
        logging=types.SimpleNamespace(get_logger=lambda name=None: _Logger()),
        scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None,
        is_torch_version=lambda op, v: True)
    mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    tu = mod("diffusers.utils.torch_utils", maybe_allow_in_graph=lambda cls: cls, apply_freeu=_Unused)
    if not hasattr(tu, "randn_tensor"):
        tu.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(
            shape, generator=generator, dtype=dtype)
    mod("diffusers.models")
    mod("diffusers.models.activations", get_activation=get_activation, GEGLU=GEGLU, GELU=_Unused,
        ApproximateGELU=_Unused)
    mod("diffusers.models.embeddings", Timesteps=Timesteps, TimestepEmbedding=TimestepEmbedding,
        **{n: _Unused for n in ("GaussianFourierProjection", "ImageHintTimeEmbedding", "ImageProjection",
                                "ImageTimeEmbedding", "PositionNet", "TextImageProjection",
                                "TextImageTimeEmbedding", "TextTimeEmbedding", "SinusoidalPositionalEmbedding",
                                "ImagePositionalEmbeddings", "CaptionProjection", "PatchEmbed")})
    mod("diffusers.models.lora", LoRACompatibleLinear=LoRACompatibleLinear, LoRACompatibleConv=LoRACompatibleConv,
        LoRALinearLayer=_Unused)
    mod("diffusers.models.normalization", AdaLayerNorm=_Unused, AdaLayerNormZero=_Unused,
        AdaLayerNormSingle=_Unused, AdaGroupNorm=_Unused)
    mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    mod("diffusers.models.attention_processor", Attention=_Unused, AttnAddedKVProcessor=_Unused,
        AttnAddedKVProcessor2_0=_Unused)
    mod("diffusers.models.dual_transformer_2d", DualTransformer2DModel=_Unused)
    mod("diffusers.models.resnet", ResnetBlock2D=ResnetBlock2D, Upsample2D=Upsample2D, Downsample2D=Downsample2D,
        **{n: _Unused for n in ("FirDownsample2D", "FirUpsample2D", "KDownsample2D", "KUpsample2D")})


def import_reference_unet(ref_root="/root/reference"):
    """Import the reference's own model/unet_2d_condition.py (and the four files it pulls in)
    on top of the stand-ins; returns its UNet2DConditionModel class."""
    import importlib
    import os
    install()
    if "diffhandles" not in sys.modules:
        pkg = types.ModuleType("diffhandles")
        pkg.__path__ = [os.path.join(ref_root, "diffhandles")]
        sys.modules["diffhandles"] = pkg
    for name in ("diffhandles.model", "diffhandles.model.unet_2d_condition"):
        m = sys.modules.get(name)
        if m is not None and not getattr(m, "__file__", None):
            del sys.modules[name]          # drop a placeholder stub, import the real file
    m = types.ModuleType("diffhandles.model")
    m.__path__ = [os.path.join(ref_root, "diffhandles", "model")]
    sys.modules["diffhandles.model"] = m
    return importlib.import_module("diffhandles.model.unet_2d_condition").UNet2DConditionModel


def sd2_depth_kwargs(cfg):
    """Constructor arguments of the reference class for an oracle config dict (oracle/unet_torch.py):
    the published stable-diffusion-2-depth unet/config.json with the sizes taken from `cfg`."""
    return dict(sample_size=cfg["sample_size"], in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
                down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
                block_out_channels=tuple(cfg["block_out_channels"]), layers_per_block=cfg["layers_per_block"],
                downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=cfg["norm_groups"],
                norm_eps=1e-5, cross_attention_dim=cfg["cross_attention_dim"],
                attention_head_dim=tuple(cfg["heads"]), dual_cross_attention=False, use_linear_projection=True,
                only_cross_attention=False, upcast_attention=False, save_activations=True)
