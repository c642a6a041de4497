#!/usr/bin/env python3
"""Copy what `bash tools/lab.sh evidence` left under gpurun_out/ (scratch) into profiles/ (tracked), named per round.

    python tools/publish_evidence.py r06 [--tests gpurun_out/r06/gpu_tests_final.txt]

bench_n1.json, bench_under_rocprof.json, rocprofv3_kernel_stats.csv and the step_* breakdowns come from gpurun_out/final/, the SQ
counter summary and the PMC traffic figure from gpurun_out/pmc/.  The traffic JSON gets the algorithmic bytes per launch (from
gemm_algorithmic_bytes.txt, the same launch mix), the ratio, and keeps the SHA-256 of the GEMM sources tools/pmc_summarise.py
stamped into it: bench.py reports the figure only while the tree's sources still hash to that value, which this script checks.
"""
import argparse
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NOTE = ("not reduced in round 6: the excess over the algorithmic bytes is (i) the A tile of the N = 320 / 640 / 1280 layers staged once per "
        "64-column tile (L2 / memory-side cache re-reads: FETCH_SIZE counts them), (ii) the nine taps of the implicit im2col, (iii) the f32 "
        "split-K slabs; DESIGN.md 5.2-5.3: none of it is what bounds a B = 1 launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("round")
    ap.add_argument("--tests", default=None, help="pytest -m gpu log to publish as <round>_gpu_tests.txt")
    ap.add_argument("--traffic-only", action="store_true",
                    help="only the PMC traffic JSON (tools/lab.sh evidence runs this on the GPU box BEFORE the bench line is taken, so "
                         "that the line's roofline.traffic is the figure collected in the same call)")
    args = ap.parse_args()
    fin, pmc, prof = (os.path.join(ROOT, p) for p in ("gpurun_out/final", "gpurun_out/pmc", "profiles"))
    r = args.round
    for name in () if args.traffic_only else ("bench_n1.json", "bench_under_rocprof.json", "rocprofv3_kernel_stats.csv", "step_kernel_types.txt",
                 "step_breakdown_by_grid.txt", "step_by_level.txt", "step_gaps.txt"):
        shutil.copy(os.path.join(fin, name), os.path.join(prof, f"{r}_{name}"))
    if not args.traffic_only:
        shutil.copy(os.path.join(pmc, "sq_summary.txt"), os.path.join(prof, f"{r}_pmc_gemm_sq.txt"))
    d = json.load(open(os.path.join(pmc, "gemm_traffic.json")))
    alg = open(os.path.join(pmc, "gemm_algorithmic_bytes.txt")).read().strip().splitlines()[0]
    a = float(re.search(r"algorithmic bytes per launch (\d+)", alg).group(1))
    d["algorithmic_bytes_per_launch"] = a
    d["algorithmic_note"] = alg
    d["traffic_over_algorithmic"] = round(d["traffic_bytes_per_launch"] / a, 3)
    d["note"] = NOTE
    json.dump(d, open(os.path.join(prof, f"{r}_pmc_gemm_traffic.json"), "w"), indent=1)
    if args.tests:
        shutil.copy(os.path.join(ROOT, args.tests), os.path.join(prof, f"{r}_gpu_tests.txt"))
    import bench
    got = bench.committed_traffic()
    print("traffic / algorithmic:", d["traffic_over_algorithmic"], "| bench.committed_traffic():", got)
    if got[0] is None:
        raise SystemExit("the published traffic figure is STALE against the tree's GEMM sources: re-collect (tools/lab.sh evidence)")
    if args.traffic_only:
        return
    b = json.load(open(os.path.join(prof, f"{r}_bench_n1.json")))
    print("bench:", b["value"], b["unit"], "| launches per step:", open(os.path.join(prof, f"{r}_step_kernel_types.txt")).readline().strip())


if __name__ == "__main__":
    main()
