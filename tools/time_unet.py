#!/usr/bin/env python3
"""Quick timing of the native U-Net at the full SD-2-depth size (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd.unet import HipUNet

from diffusionhandles_amd.unet import SD2_DEPTH
dt = torch.bfloat16 if os.environ.get("DH_DTYPE") == "bf16" else torch.float16
LAT = int(os.environ.get("DH_LATENT", "64"))           # 96 = the 768x768 configuration
BATCHES = tuple(int(b) for b in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 2)
# (DH_GEMM_FAMILY / DH_PP_GLU / DH_PP_PERSIST: the GEMM dispatch switches are read by diffusionhandles_amd/_lib.py at load time)
u = HipUNet(dict(SD2_DEPTH, sample_size=LAT), dtype=dt, max_batch=max(BATCHES))
u.init_synthetic(0)
print("weights GB", u.weight_bytes() / 1e9, "workspace GB", u.workspace_bytes() / 1e9)
dev = u.device
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream()
torch.cuda.set_stream(side)
for B in BATCHES:
    x = torch.randn(B, LAT, LAT, 5, generator=g, device=dev)
    txt = torch.randn(B, 77, 1024, generator=g, device=dev)
    da = [None, torch.randn((B,) + u.act_shapes[1], generator=g, device=dev).to(dt) * 1e-2,
          torch.randn((B,) + u.act_shapes[2], generator=g, device=dev).to(dt) * 1e-2]
    for _ in range(3):
        u.forward(x, 500.0, txt, save_for_backward=True)
        u.backward(da, None)
    torch.cuda.synchronize()
    n = 10
    t0 = time.time()
    for _ in range(n):
        u.forward(x, 500.0, txt, save_for_backward=True)
    torch.cuda.synchronize()
    tf = (time.time() - t0) / n
    t0 = time.time()
    for _ in range(n):
        u.backward(da, None)
    torch.cuda.synchronize()
    tb = (time.time() - t0) / n
    s = u.stats()
    print(f"B={B}: fwd {tf*1e3:.2f} ms ({s['flops_fwd']/tf/1e12:.1f} TF/s)  bwd {tb*1e3:.2f} ms ({s['flops_bwd']/tb/1e12:.1f} TF/s)  ops {s['ops']}")
    # algorithmic HBM bytes of the pass's GEMM launches (every operand once: dh_gemm_profile_bytes) -- the denominator of the
    # wasted-traffic ratio whose numerator is the PMC traffic of this same launch mix (tools/lab.sh pmc-traffic)
    if os.environ.get("DH_ALG_BYTES"):
        import ctypes
        from diffusionhandles_amd import _lib
        L = _lib.lib()
        _lib.check(L.dh_gemm_profile_begin())
        u.forward(x, 500.0, txt, save_for_backward=True)
        u.backward(da, None)
        ms, nl, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        _lib.check(L.dh_gemm_profile_end(ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(fl)))
        _lib.check(L.dh_gemm_profile_bytes(ctypes.byref(by)))
        print(f"ALG B={B}: k_gemm_dma launches {nl.value}, algorithmic bytes per launch {by.value / max(1, nl.value):.0f}, "
              f"flops per launch {fl.value / max(1, nl.value):.4g}, event time per launch {ms.value * 1e3 / max(1, nl.value):.2f} us")

