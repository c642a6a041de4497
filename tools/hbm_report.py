#!/usr/bin/env python3
"""Achieved GB/s of the HBM-bound kernels from a rocprofv3 --kernel-trace CSV.

  hbm_report.py energy <kernel_trace.csv> <bench_energy.json>     # per kernel and layer (C): avg us, bytes touched, GB/s, % of 8 TB/s
  hbm_report.py reproject <kernel_trace.csv> <res> <n_fg> <K>     # per kernel of the batched K-edit re-projection
Output: CSV on stdout (committed under profiles/)."""
import collections
import csv
import json
import sys

PEAK = 8000.0


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]),
                     int(r["Grid_Size_Z"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    return rows


def short(name):
    """k_xxx out of a demangled ('void dh::k_xxx<...>(...)') or mangled ('_ZN2dh10k_xxxIDF16_EEv...') kernel name."""
    import re
    m = re.search(r"k_[a-z0-9_]+", name)
    return m.group(0) if m else name.split("(")[0]


def energy(trace, info_path):
    info = json.loads(open(info_path).read().strip().splitlines()[-1])
    rows = load(trace)
    G2 = info["grid"] ** 2
    print("kernel,C,launches,avg_us,bytes_touched,GBps,frac_of_8TBps")
    for layer in info["layers"]:
        C = layer["C"]
        tot_us = 0.0
        for k, nbytes in layer["kernels"].items():
            # the launch geometry tells the layers apart: k_colsum_q (C / 64, 4 quarters, 2 lists) workgroups (rounds 1-4: k_colsum16 (ceil(C/8*128/256), 2) and
            # k_global_diff C / 64), k_energy_grad G2 / (256 / (C/8))
            want = {"k_colsum16": (-(-(C // 8) * 128 // 256), 2), "k_global_diff": (-(-C // 64), 1), "k_colsum_q": (-(-C // 64), 4),
                    "k_energy_grad": (-(-G2 // (256 // (C // 8))), 1)}[k]
            us = [d for n, gx, gy, gz, d in rows if short(n) == k and (gx, gy) == want]
            us = us[len(us) // 5:]                 # drop the cold first fifth
            if not us:
                continue
            avg = sum(us) / len(us)
            tot_us += avg
            print(f"{k},{C},{len(us)},{avg:.2f},{nbytes},{nbytes / avg / 1e3:.1f},{nbytes / avg / 1e3 / PEAK:.4f}")
        a = layer["algorithmic_bytes"]
        print(f"evaluation ({len(layer['kernels'])} dependent launches; algorithmic bytes = read cur + read orig + write grad),{C},,{tot_us:.2f},{a},"
              f"{a / tot_us / 1e3:.1f},{a / tot_us / 1e3 / PEAK:.4f}")


def reproject(trace, res, n_fg, K):
    rows = load(trace)
    px = res * res
    npts = (px + n_fg) * K
    touched = {      # bytes each kernel of the batched call touches (~9 MB per edit at 512^2 in total)
        "k_points": npts * (4 + 4 + 8 + 8),            # depth read, pixel id + f64 key written, one 64-bit atomicMin per point
        "k_resolve": npts * (4 + 8 + 8 + 4),           # pixel, key, z-buffer word read, owner atomicMin
        "k_pixels": px * K * (8 + 4 + 4 + 1 + 4),      # z-buffer + owner read, zmap / raw mask / disparity written
        "k_morph": px * K * 2,
        "k_normalize": px * K * 8,
    }
    agg = collections.defaultdict(list)
    for n, gx, gy, gz, d in rows:
        agg[short(n)].append(d)
    print("kernel,launches,avg_us,total_us_per_call,bytes_touched,GBps,frac_of_8TBps")
    calls = max(1, len(agg.get("k_points", [1])))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if not k.startswith("k_"):
            continue
        avg = sum(v) / len(v)
        b = touched.get(k)
        gb = f"{b / avg / 1e3:.1f},{b / avg / 1e3 / PEAK:.4f}" if b else ","
        print(f"{k},{len(v)},{avg:.2f},{sum(v) / calls:.1f},{b or ''},{gb}")


if __name__ == "__main__":
    if sys.argv[1] == "energy":
        energy(sys.argv[2], sys.argv[3])
    else:
        reproject(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
