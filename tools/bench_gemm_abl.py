#!/usr/bin/env python3
"""Ablation timings of k_gemm_dma (DH_GEMM_ABLATE=0/1/2 in separate processes) on B=1 U-Net shapes."""
import os, subprocess, sys
code = r'''
import sys, os
sys.path.insert(0, os.getcwd())
import tools.bench_gemm as bg
'''
for abl in ("0", "1", "2"):
    env = dict(os.environ, DH_GEMM_ABLATE=abl, DH_DBG_PRETILED="1", DH_SHAPES="b1")
    print("ABLATE", abl, flush=True)
    subprocess.run([sys.executable, "tools/bench_gemm.py"], env=env)
