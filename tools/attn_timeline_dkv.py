#!/usr/bin/env python3
"""s_memtime stamps of wave 0 of block (0,0,0) of k_attn_bwd_dkv (library built with -DDH_ATTN_STAMP: DH_DEFS=-DDH_ATTN_STAMP
DH_NAME=libdh_stamp.so tools/lab.sh build-tuning): where one query tile's time goes inside the dK/dV loop.
   DIFFHANDLES_LIB=tools/bin/libdh_stamp.so python3 tools/attn_timeline_dkv.py [N] [H] [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionhandles_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
C, dt = H * 64, torch.float16
q, k, v, do = (torch.randn(B, N, C, device=dev).to(dt) for _ in range(4))
o = torch.empty_like(q); lse = torch.empty(B, H, N, dtype=torch.float32, device=dev); delta = torch.empty_like(lse)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
fn = ctypes.CDLL(os.environ["DIFFHANDLES_LIB"]).dh_dbg_attn_stamps_dkv
ts = (ctypes.c_ulonglong * 16)()
for rep in range(4):
    L.dh_dbg_attention(0, P(q), C, P(k), P(v), C, P(o), C, P(lse), P(do), P(delta), P(dq), P(dk), P(dv), B, H, N, N, _lib.stream_ptr())
    torch.cuda.synchronize()
    assert fn(ts) == 0
    t = [int(x) for x in ts]
    if os.environ.get("DH_TIMELINE_KERNEL") == "dq":      # library built with -DDH_ATTN_STAMP -DDH_ATTN_STAMP_DQ: the stamps sit in k_attn_bwd_dq
        print(f"dq N={N} H={H} B={B} rep {rep}: loop total {t[7] - t[0]} ticks; key tile 2: wait at barrier 1 {t[2] - t[1]}, wait for the prefetched K / V {t[8] - t[2]}, "
              f"commit {t[3] - t[8]}, barrier 2 {t[4] - t[3]}, next fetch issue {t[9] - t[4]}, first 32-key half (S, dP, exp, dQ) {t[5] - t[9]}, second half {t[6] - t[5]}, whole tile {t[6] - t[1]}")
        continue
    print(f"N={N} H={H} B={B} rep {rep}: loop total {t[7] - t[0]} ticks; tile 2: wait at barrier 1 {t[2] - t[1]}, commit phase {t[3] - t[2]} (= wait for the prefetched tile {t[8] - t[2]} + ds_write of Q / dO {t[9] - t[8]} + stats and next fetch issue {t[3] - t[9]}), "
          f"barrier 2 {t[4] - t[3]}, first 32-query half (S, dP, exp, dV, dK) {t[5] - t[4]}, second half + next fetch issue {t[6] - t[5]}, whole tile {t[6] - t[1]}")
