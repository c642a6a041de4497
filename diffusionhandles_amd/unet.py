"""Python handle of the native SD-2-depth U-Net engine (csrc/unet_engine.cpp).

`HipUNet` mirrors the call surface the reference uses on its patched diffusers U-Net
(model/unet_2d_condition.py:809-1198): `unet(sample, t, encoder_hidden_states=...,
return_dict=False)` returns the 7-tuple `(eps, None, None, None, act0, act1, act2)`;
with `return_dict=True` it returns `{'sample': eps}`.  On top of that it exposes the
explicit `forward` / `backward` pair the guided loop drives (no autograd graph).
Activations are channels-last 16-bit tensors; the [B,C,h,w] tensors handed out are views.
"""
import ctypes
import math
from types import SimpleNamespace

import torch

from . import _lib

SD2_DEPTH = dict(in_channels=5, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, heads=(5, 10, 20, 20), cross_attention_dim=1024,
                 norm_groups=32, sample_size=64, text_len=77)


class HipUNet:
    def __init__(self, cfg=None, dtype=torch.float16, max_batch=2, device=None, max_diff_batch=None):
        """max_batch: largest batch of any forward; max_diff_batch (default: max_batch): largest batch of a forward that is
        saved for a backward pass.  Only the latter sizes the activation / gradient arenas: a forward nobody differentiates
        shares the activation arena by liveness (batched edits of K transforms: max_batch = 2 K for the CFG pass,
        max_diff_batch = K for the optimisation passes)."""
        _lib.require_gpu()
        self.cfg = dict(SD2_DEPTH if cfg is None else cfg)
        self.cfg.setdefault("text_len", 77)
        self.dtype = dtype
        self.max_batch = int(max_batch)
        self.max_diff_batch = self.max_batch if not max_diff_batch else max(1, min(int(max_diff_batch), self.max_batch))
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        c = _lib.UNetConfig()
        c.in_channels, c.out_channels, c.n_levels = self.cfg["in_channels"], self.cfg["out_channels"], 4
        for i in range(4):
            c.block_out_channels[i] = self.cfg["block_out_channels"][i]
            c.heads[i] = self.cfg["heads"][i]
        c.layers_per_block = self.cfg["layers_per_block"]
        c.cross_attention_dim = self.cfg["cross_attention_dim"]
        c.norm_groups = self.cfg["norm_groups"]
        c.sample_size = self.cfg["sample_size"]
        c.text_len = self.cfg["text_len"]
        c.max_batch = self.max_batch
        c.max_diff_batch = self.max_diff_batch
        c.dtype = _lib.DTYPE_CODE[dtype]
        self._L = _lib.lib()
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._L.dh_unet_create(ctypes.byref(c), ctypes.byref(h)), "dh_unet_create")
        self._h = h
        self.config = SimpleNamespace(in_channels=c.in_channels, out_channels=c.out_channels,
                                      sample_size=c.sample_size)
        self.sample_size = c.sample_size
        self._table = None
        ch, s = self.cfg["block_out_channels"], c.sample_size
        self.act_shapes = [(s // 2, s // 2, ch[2]), (s, s, ch[1]), (s, s, ch[0])]   # (h, w, C) of act0..2
        self._saved_batch = 0
        self._text_key = 0

    def close(self):
        """Destroy the engine now (its arenas go back to the device); the object is unusable afterwards."""
        if getattr(self, "_h", None):
            self._L.dh_unet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def share(self, max_batch=None):
        """A second engine that reads THIS engine's weights (resident once) and owns everything a pass writes
        (dh_unet_create_shared): two of them on two streams run two edits concurrently in one process.  Load the parameters
        first; loading into either engine afterwards raises.  The returned handle keeps this one alive."""
        mb = self.max_batch if max_batch is None else int(max_batch)
        child = object.__new__(HipUNet)
        child.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_h", "_views", "_table")})
        child.max_batch = mb
        child.max_diff_batch = self.max_diff_batch if mb == self.max_batch else mb
        child._table = self._table
        child._saved_batch = 0
        child._text_key = 0
        child._parent = self
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._L.dh_unet_create_shared(self._h, mb, _lib.stream_ptr(), ctypes.byref(h)), "dh_unet_create_shared")
        child._h = h
        return child

    # ---- parameters ---------------------------------------------------------------------
    def param_table(self):
        if self._table is None:
            n = self._L.dh_unet_num_params(self._h)
            tab = []
            for i in range(n):
                name = ctypes.c_char_p()
                nd = ctypes.c_int()
                shp = (ctypes.c_int64 * 4)()
                _lib.check(self._L.dh_unet_param_info(self._h, i, ctypes.byref(name), ctypes.byref(nd), shp))
                tab.append((name.value.decode(), tuple(int(shp[k]) for k in range(nd.value))))
            self._table = tab
        return self._table

    def load_state_dict(self, sd, strict=True):
        """sd: diffusers-named tensors in torch layout (any float dtype / device)."""
        names = {n for n, _ in self.param_table()}
        missing = names - set(sd.keys())
        if strict and missing:
            raise KeyError(f"missing parameters: {sorted(missing)[:5]} ... ({len(missing)})")
        st = _lib.stream_ptr()
        for i, (name, shape) in enumerate(self.param_table()):
            if name not in sd:
                continue
            t = sd[name].detach()
            if tuple(t.shape) != shape:
                raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
            t = t.to(self.device, torch.float32).contiguous()
            _lib.check(self._L.dh_unet_load_param(self._h, i, _lib.ptr(t), st), f"load {name}")
        torch.cuda.synchronize(self.device)

    def init_synthetic(self, seed=0):
        """Seeded random weights of the exact architecture (no checkpoint needed):
        matrices N(0, 1/fan_in), norm gains 1 + 0.1 N, other vectors 0.02 N."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        st = _lib.stream_ptr()
        for i, (name, shape) in enumerate(self.param_table()):
            if len(shape) >= 2:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                t = torch.randn(shape, generator=g, device=self.device) * (1.0 / math.sqrt(fan_in))
            elif "norm" in name and name.endswith("weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=self.device)
            else:
                t = 0.02 * torch.randn(shape, generator=g, device=self.device)
            t = t.float().contiguous()
            _lib.check(self._L.dh_unet_load_param(self._h, i, _lib.ptr(t), st), f"load {name}")
        torch.cuda.synchronize(self.device)

    def weight_bytes(self):
        return int(self._L.dh_unet_weight_bytes(self._h))

    def workspace_bytes(self):
        return int(self._L.dh_unet_workspace_bytes(self._h))

    # ---- compute ------------------------------------------------------------------------
    def forward(self, sample_nhwc, timestep, text, save_for_backward=False, want_acts=True, want_eps=True, text_key=0,
                inplace=False):
        """sample_nhwc [B,H,W,Cin] f32, text [B,L,D] f32 (device, contiguous).
        Returns eps [B,H,W,Cout] f32 and a list of 3 channels-last activations [B,h,w,C].
        want_acts may be a collection of activation indices; with want_eps=False the engine stops after
        the last requested activation (eps is None, the other activations are None).
        text_key != 0 names the content of `text`: consecutive forwards with the same key and batch reuse the text K|V
        projections (the caller changes the key when the embedding changes).
        inplace: eps and the activations are returned as VIEWS of the engine's buffers (no device copies; valid until the
        next pass) instead of fresh tensors."""
        B = sample_nhwc.shape[0]
        s = self.sample_size
        eps = None
        if want_eps:
            eps = self.io_view("eps")[:B] if inplace else \
                torch.empty((B, s, s, self.cfg["out_channels"]), dtype=torch.float32, device=self.device)
        acts = None
        arr = None
        if want_acts:
            idx = range(3) if want_acts is True else set(want_acts)
            acts = [(self.io_view("act", i)[:B] if inplace else torch.empty((B,) + shp, dtype=self.dtype, device=self.device))
                    if i in idx else None for i, shp in enumerate(self.act_shapes)]
            arr = (ctypes.c_void_p * 3)(*[a.data_ptr() if a is not None else None for a in acts])
        if int(text_key) != self._text_key:           # (0 = unnamed text, the engine's default)
            _lib.check(self._L.dh_unet_set_text_key(self._h, int(text_key)), "dh_unet_set_text_key")
            self._text_key = int(text_key)
        _lib.check(self._L.dh_unet_forward(self._h, _lib.ptr(sample_nhwc), float(timestep), _lib.ptr(text), B,
                                           1 if save_for_backward else 0, _lib.ptr(eps), arr, _lib.stream_ptr()),
                   "dh_unet_forward")
        self._saved_batch = B if save_for_backward else 0
        return eps, acts

    def backward(self, d_acts=None, d_eps=None, want_sample_grad=True, want_text_grad=False, inplace=False):
        """Gradients of the last saved forward.  d_acts: list of 3 (or None entries) channels-last
        tensors in the engine dtype; d_eps [B,H,W,Cout] f32.  Returns (d_sample, d_text)."""
        B = self._saved_batch
        if B == 0:
            raise RuntimeError("backward() needs a forward(save_for_backward=True) first")
        s = self.sample_size
        arr = None
        if d_acts is not None:
            ptrs = []
            for a, shp in zip(d_acts, self.act_shapes):
                if a is None:
                    ptrs.append(None)
                else:
                    assert a.dtype == self.dtype and a.is_contiguous() and tuple(a.shape) == (B,) + shp
                    ptrs.append(a.data_ptr())
            arr = (ctypes.c_void_p * 3)(*ptrs)
        d_sample = None
        if want_sample_grad:      # inplace: a view of the engine's buffer (valid until the next backward)
            d_sample = self.io_view("dsample")[:B] if inplace else \
                torch.empty((B, s, s, self.cfg["in_channels"]), dtype=torch.float32, device=self.device)
        d_text = torch.empty((B, self.cfg["text_len"], self.cfg["cross_attention_dim"]), dtype=torch.float32,
                             device=self.device) if want_text_grad else None
        _lib.check(self._L.dh_unet_backward(self._h, arr, _lib.ptr(d_eps), _lib.ptr(d_sample), _lib.ptr(d_text),
                                            _lib.stream_ptr()), "dh_unet_backward")
        return d_sample, d_text

    # ---- the engine's own I/O buffers as tensors (no copies either side of a pass) -------------------------------------
    class _Raw:
        """A device buffer described through __cuda_array_interface__ so that torch can view it without owning it."""

        def __init__(self, ptr, shape, typestr):
            self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)

    def io_view(self, which, index=0):
        """Tensor view of one of the engine's fixed buffers (dh_unet_io_ptr): 'sample' [maxB,H,W,Cin] f32, 'eps' [maxB,H,W,Cout]
        f32, 'dsample' like 'sample', 'act' / 'act_grad' (index 0..2) [maxB,h,w,C] in the engine dtype.  The engine owns the
        memory (it lives as long as this object); contents are valid until the next pass that writes them.  Passing such a
        view (or its leading batch items) to forward / backward makes the corresponding device copy disappear."""
        if not hasattr(self, "_views"):
            self._views = {}
        key = (which, index)
        if key not in self._views:
            code = dict(sample=0, text=1, eps=2, act=3, act_grad=4, dsample=5, dtext=6)[which]
            p, nb = ctypes.c_void_p(), ctypes.c_size_t()
            _lib.check(self._L.dh_unet_io_ptr(self._h, code, index, ctypes.byref(p), ctypes.byref(nb)), "dh_unet_io_ptr")
            s, mb = self.sample_size, self.max_batch
            if which in ("act", "act_grad"):
                shape = (mb if which == "act" else self.max_diff_batch,) + self.act_shapes[index]
                t = torch.as_tensor(HipUNet._Raw(p.value, shape, "<i2"), device=self.device).view(self.dtype)
            else:
                shape = {"sample": (mb, s, s, self.cfg["in_channels"]), "dsample": (mb, s, s, self.cfg["in_channels"]),
                         "eps": (mb, s, s, self.cfg["out_channels"]),
                         "text": (mb, self.cfg["text_len"], self.cfg["cross_attention_dim"]),
                         "dtext": (mb, self.cfg["text_len"], self.cfg["cross_attention_dim"])}[which]
                t = torch.as_tensor(HipUNet._Raw(p.value, shape, "<f4"), device=self.device)
            assert t.data_ptr() == p.value and t.numel() * t.element_size() == nb.value
            self._views[key] = t
        return self._views[key]

    def stage_sample(self, latent_nhwc, depth_nhwc, batch):
        """The U-Net input cat([latents, depth], channel) for `batch` items written straight into the engine's input buffer
        (one launch; latents / depth given once are broadcast).  Returns the view to pass to forward()."""
        dst = self.io_view("sample")
        cl = latent_nhwc.shape[-1]
        cd = depth_nhwc.shape[-1] if depth_nhwc is not None else 0
        if cl + cd != self.cfg["in_channels"] or batch > self.max_batch:
            raise ValueError("stage_sample: channels / batch do not match the engine")
        pixels = self.sample_size * self.sample_size
        _lib.check(self._L.dh_pack_sample(_lib.ptr(dst), _lib.ptr(latent_nhwc), latent_nhwc.shape[0], cl, _lib.ptr(depth_nhwc),
                                          depth_nhwc.shape[0] if depth_nhwc is not None else 1, cd, batch, pixels,
                                          _lib.stream_ptr()), "dh_pack_sample")
        return dst[:batch]

    def stats(self):
        f, b, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(self._L.dh_unet_stats(self._h, ctypes.byref(f), ctypes.byref(b), ctypes.byref(n)))
        return dict(flops_fwd=f.value, flops_bwd=b.value, ops=n.value)

    # reference-style call (NCHW in, 7-tuple out)
    def __call__(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None, return_dict=True):
        x = sample.detach().to(self.device, torch.float32).permute(0, 2, 3, 1).contiguous()
        txt = encoder_hidden_states.detach().to(self.device, torch.float32).contiguous()
        t = float(timestep.item()) if isinstance(timestep, torch.Tensor) else float(timestep)
        eps, acts = self.forward(x, t, txt, save_for_backward=False, want_acts=not return_dict)
        eps = eps.permute(0, 3, 1, 2)
        if return_dict:
            return {"sample": eps}
        a = [t_.permute(0, 3, 1, 2) for t_ in acts]
        return (eps, None, None, None, a[0], a[1], a[2])
