#!/bin/bash
# SQ counter passes on the attention kernels at N = 4096, H = 5, B = 1 (tools/bench_attn.py, 23 forward + 23 forward/backward
# launches): where the wave cycles go.  usage: tools/pmc_attn.sh [library]   -> gpurun_out/pmc_attn/<name>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB=${1:-diffusionhandles_amd/libdiffhandles_hip.so}
NAME=$(basename $LIB .so)
export DIFFHANDLES_LIB=$R/$LIB DH_ATTN_CFGS=n4096 TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/pmc_attn
rm -rf /tmp/pa1 /tmp/pa2 /tmp/pa3
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pa1 -- python3 $R/tools/bench_attn.py > /tmp/pa1.log 2>&1; echo "pass1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d /tmp/pa2 -- python3 $R/tools/bench_attn.py > /tmp/pa2.log 2>&1; echo "pass2 rc=$?"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d /tmp/pa3 -- python3 $R/tools/bench_attn.py > /tmp/pa3.log 2>&1; echo "pass3 rc=$?"
python3 - $NAME <<'PY' > $R/gpurun_out/pmc_attn/$NAME.txt
import csv, glob, collections, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for d in ("/tmp/pa1", "/tmp/pa2", "/tmp/pa3"):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_attn[a-z_]+", r["Kernel_Name"])
            if not m: continue
            agg[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"]); n[m.group(0)][r["Counter_Name"]] += 1
print(f"# {sys.argv[1]}: rocprofv3 --pmc (three passes) -- tools/bench_attn.py at B=1 H=5 N=4096 d=64 fp16; per-launch averages")
for k, c in sorted(agg.items()):
    print(k)
    for name in sorted(c):
        print(f"   {name:28s} {c[name] / n[k][name]:14.0f}")
PY
cat $R/gpurun_out/pmc_attn/$NAME.txt
