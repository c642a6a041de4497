#!/usr/bin/env python3
"""s_memtime stamps of wave 0 of block (0,0,0) of k_attn_fwd (library built with -DDH_ATTN_STAMP): where one block's time goes.
   DIFFHANDLES_LIB=tools/bin/libdh_stamp.so python3 tools/attn_timeline.py [N] [H]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, C, dt = 1, H * 64, torch.float16
q = torch.randn(B, N, C, device=dev).to(dt); k = torch.randn(B, N, C, device=dev).to(dt); v = torch.randn(B, N, C, device=dev).to(dt)
o = torch.empty_like(q); lse = torch.empty(B, H, N, dtype=torch.float32, device=dev); delta = torch.empty_like(lse)
fn = L._lib.dh_dbg_attn_stamps if hasattr(L, "_lib") else ctypes.CDLL(os.environ["DIFFHANDLES_LIB"]).dh_dbg_attn_stamps
ts = (ctypes.c_ulonglong * 8)()
names = ["start -> loop (Q fragments, first DMA)", "first tile (incl. its landing)", "tiles 1..", "merge of the key groups", "store", ]
for rep in range(4):
    L.dh_dbg_attention(0, P(q), C, P(k), P(v), C, P(o), C, P(lse), P(None), P(delta), P(None), P(None), P(None), B, H, N, N, _lib.stream_ptr())
    torch.cuda.synchronize()
    assert fn(ts) == 0
    t = [int(x) for x in ts]
    d = [t[i + 1] - t[i] for i in range(5)]
    print(f"N={N} H={H} rep {rep}: total {t[5] - t[0]} ticks (100 MHz: {(t[5] - t[0]) / 100:.1f} us): " + "; ".join(f"{n} {x}" for n, x in zip(names, d)))
