#!/bin/bash
# SQ counter passes on one B=1 U-Net forward+backward mix (tools/time_unet.py 1): MFMA busy, LDS bank conflicts, waits,
# aggregated per kernel family.  Run through gpurun; the summary lands in gpurun_out/pmc/sq_summary.txt
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/sq1 -- python3 $R/tools/time_unet.py 1 > /tmp/sq1.log 2>&1; echo "pass1 rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/sq2 -- python3 $R/tools/time_unet.py 1 > /tmp/sq2.log 2>&1; echo "pass2 rc=$?"
python3 - <<'PY' > $R/gpurun_out/pmc/sq_summary.txt
import csv, glob, collections, re
def fam(n):
    if "k_gemm_dma" in n:
        m = re.search(r"Li(\d+)ELi(\d+)E", n)
        return f"k_gemm_dma {m.group(1)}x{m.group(2)}" if m else "k_gemm_dma"
    m = re.search(r"k_[a-z0-9_]+", n)
    return m.group(0) if m else n[:30]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for d in ("/tmp/sq1", "/tmp/sq2"):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = fam(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if d == "/tmp/sq1" and r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
print("# rocprofv3 --pmc (two passes of 8 SQ counters) --kernel-trace -- python3 tools/time_unet.py 1   (B=1 U-Net forward+backward, 13 iterations; sums over all launches)")
print("# raw sums per kernel family; lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; insts per MFMA instruction; wait = SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY")
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))
for k, c in rows[:22]:
    mf = max(1.0, c.get("SQ_INSTS_MFMA", 0))
    ldsc = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0))
    has = c.get("SQ_INSTS_MFMA", 0) > 0
    per = f"per MFMA: valu {c.get('SQ_INSTS_VALU',0)/mf:5.1f} salu {c.get('SQ_INSTS_SALU',0)/mf:5.1f} lds {c.get('SQ_INSTS_LDS',0)/mf:4.1f}" if has else "no MFMA"
    print(f"{k:26s} launches {cnt[k]:5d} SQ_BUSY_CYCLES {c.get('SQ_BUSY_CYCLES',0):.3e} SQ_VALU_MFMA_BUSY_CYCLES {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3e} SQ_WAVE_CYCLES {c.get('SQ_WAVE_CYCLES',0):.3e} "
          f"lds_conflict {ldsc:5.3f} {per} wait {c.get('SQ_WAIT_INST_ANY',0)/max(1.0,c.get('SQ_ACTIVE_INST_ANY',0)):5.2f}")
PY
cat $R/gpurun_out/pmc/sq_summary.txt | head -30
