#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per launch of the k_gemm_dma kernels from the two tools/lab.sh pmc-traffic passes
(gpurun_out/pmc/{FETCH_SIZE,WRITE_SIZE}.tsv: kernel name, launches, counter sum in KiB) -> profiles-style JSON."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import GEMM_SOURCES, gemm_sources_sha  # noqa: E402

out = {}
by_kernel = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    n = v = 0.0
    for line in open(f"{sys.argv[1]}/{c}.tsv"):
        name, cnt, val = line.rstrip("\n").split("\t")
        by_kernel.setdefault(name.split("(")[0][:60], {})[c] = (int(cnt), float(val))
        if "k_gemm_dma" in name or "k_gemm_pp" in name:
            n += int(cnt)
            v += float(val)
    out[c] = (n, v)
launches = out["FETCH_SIZE"][0]
f = out["FETCH_SIZE"][1] / launches
w = out["WRITE_SIZE"][1] / out["WRITE_SIZE"][0]
print(json.dumps({
    "kernel": "dh::k_gemm_dma + dh::k_gemm_pp (all instantiations)",
    "workload": f"SD-2-depth U-Net fwd+bwd, B={__import__('os').environ.get('DH_PMC_BATCH', '1')}, 64x64 latent, fp16 (tools/time_unet.py, 13 iterations)",
    "launches": int(launches), "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w,
    "fetch_correction": "x2 (gfx950 FETCH_SIZE under-count, MI355X_MICROARCH.md HBM section)",
    "traffic_bytes_per_launch": (2 * f + w) * 1024,
    "sources_sha256": gemm_sources_sha(),
    "sources": "sha256 over diffusionhandles_amd/csrc/{" + ", ".join(GEMM_SOURCES) + "} (bench.py gemm_sources_sha): bench.py reports this "
               "figure only while the tree's GEMM sources hash to the same value",
    "command": "tools/lab.sh pmc-traffic: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/time_unet.py 1 ; "
               "same with --pmc WRITE_SIZE (separate passes); tools/pmc_summarise.py gpurun_out/pmc"}, indent=1))
