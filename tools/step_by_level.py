#!/usr/bin/env python3
"""Share of a guided step's kernel time by U-Net resolution level (B = 1: 64^2 / 32^2 / 16^2 / 8^2 latents), inferred from
the (kernel, grid) lines of tools/step_breakdown.py: a GEMM's rows = grid.x * tile rows (template argument), a LayerNorm's
rows = 4 * grid.x, the element-wise kernels by element count, attention by its query tiles.
usage: step_by_level.py profiles/r02_step_breakdown_by_grid.txt"""
import collections
import re
import sys

LEVEL = {4096: "64^2", 1024: "32^2", 256: "16^2", 64: "8^2"}


def level_of(name, gx, gy, gz):
    m = re.search(r"k_gemm_dmaID(?:F16_|F16b)Li(\d+)ELi(\d+)E", name)
    if m:
        bm = int(m.group(1))
        rows = gx * bm
        if "Li1ELi0E" in name or True:
            pass
        for r in (8192, 4096, 2048, 1024, 512, 256, 128, 64):
            if rows >= r:
                rows = r
                break
        return LEVEL.get(rows if rows <= 4096 else rows // 2, f"B=2 {LEVEL.get(rows // 2, rows)}")
    if "k_ln_" in name:
        rows = gx * 4
        return LEVEL.get(rows, LEVEL.get(rows // 2, str(rows)))
    if "k_attn" in name:
        return {43: "64^2", 16: "32^2", 8: "16^2 / 8^2", 11: "32^2", 3: "16^2", 2: "8^2", 1: "8^2"}.get(gx, f"attn {gx}")
    if "splitk_reduce" in name:
        if "gn" in name:
            return {32: "64^2 / 32^2 (32 slices)", 16: "16^2 / 8^2"}.get(gx, str(gx))
        return "split-K reduce (plain)"
    return "other"


def main():
    agg = collections.defaultdict(float)
    tot = 0.0
    for line in open(sys.argv[1]):
        m = re.match(r"\d*(\S+)\s+grid=\((\d+),(\d+),(\d+)\)\s+n=\s*(\d+)\s+total\s+([\d.]+) us", line)
        if not m:
            continue
        name, gx, gy, gz, n, us = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), float(m.group(6))
        kind = "gemm" if "k_gemm" in name else ("attention" if "k_attn" in name else ("layernorm" if "k_ln_" in name else
               ("reduce" if "splitk" in name else "rest")))
        agg[(kind, level_of(name, gx, gy, gz))] += us
        tot += us
    print(f"kernel time in the listing: {tot:.0f} us")
    for (kind, lvl), us in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print(f"{kind:10s} {lvl:28s} {us:8.0f} us {100 * us / tot:5.1f} %")


if __name__ == "__main__":
    main()
