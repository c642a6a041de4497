#!/bin/bash
# One driver for the measurement / experiment recipes of this repo (round 4: the ~20 one-off shell scripts of rounds 1-3
# folded into functions; the Python benchmarks they call stay in tools/).  Run on the GPU box through gpurun, from the repo root:
#   gpurun -- 'bash tools/lab.sh <recipe> [args]'        bash tools/lab.sh list   prints the recipes
# Every recipe writes under gpurun_out/; what is judged is copied into profiles/ by hand (named per round).
# Profilers go in front of ONE process only (rocprofv3 ... -- python3 ...): never in front of a --gpus N launcher.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
# ab: same-box A/B of library builds on any command: lab.sh ab "<command>" libA.so libB.so ... (each run twice, interleaved)
recipe_ab() { (
CMD=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib bash -c "$CMD" 2>&1 | grep -E "B=|steps/s|value" | cut -c1-160
  done
done
) }
# ab-libs: same-box A/B of library builds: U-Net passes at B = 1, 2, 8, the 96x96 latent (bf16), the batch-8 GEMM shapes
recipe_ab_libs() { (
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib python3 tools/time_unet.py 1,2,8 2>&1 | grep "^B="
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib DH_LATENT=96 DH_DTYPE=bf16 python3 tools/time_unet.py 1 2>&1 | grep "^B=" | sed 's/^/L96 /'
  done
done
for lib in "$@"; do
  echo "== $lib"
  DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib DH_DBG_PRETILED=1 DH_SHAPES=b8 python3 tools/bench_gemm.py 2>&1 | grep "^M="
done
) }
# ab-attn: same-box A/B of attention builds (tools/bench_attn.py)
recipe_ab_attn() { (
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib python3 tools/bench_attn.py 2>&1 | grep "^B="
  done
done
) }
# ab-step: same-box A/B of library builds on the guided step (tools/ab_inplace.py timing, in-place I/O on)
recipe_ab_step() { (
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib"
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["x"]
exec(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools", "ab_inplace.py")).read().split("finals = {}")[0])
gd._inplace_io = True
run(6)
a, _ = run(76)
print(f"guided steps/s {a:.2f}")
PY
  done
done
) }
# ablate-gemm: k_gemm_dma ablations (0 full, 1 staging only, 2 compute only) on the batch-8 shapes; needs the tuning build
recipe_ablate_gemm() { (
cd $GRAFT_REPO_ROOT
for abl in 0 1 2; do
  echo "ABLATE $abl"
  DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/tools/bin/libdiffhandles_hip_tuning.so DH_DBG_PRETILED=1 DH_SHAPES=${1:-b8} DH_GEMM_ABLATE=$abl python3 tools/bench_gemm.py 2>&1 | grep "^M="
done
) }
# ablate-attn: builds tools/bin/libdh_attn_<n>.so with attention.hip under -DDH_ATTN_ABL=n (timing-only ablations; CPU container)
recipe_ablate_attn() { (
set -e
ROOT=$R
SRC=$ROOT/diffusionhandles_amd/csrc
make -C "$SRC" -j8 >/dev/null
mkdir -p "$ROOT/tools/bin"
OBJS=$(ls "$SRC"/*.o | grep -v attention.o)
for n in ${@:-0 1 4 5}; do
  ( cd "$SRC" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDH_ATTN_ABL=$n -c attention.hip -o /tmp/dh_attn_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/libdh_attn_$n.so" $OBJS /tmp/dh_attn_$n.o ) &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
ls -la "$ROOT"/tools/bin/libdh_attn_*.so
) }
# build-pp-variants: the library with every k_gemm_pp main-loop variant (gemm_pp.hip under -DDH_PP_VARIANTS; CPU container) -> tools/bin/libdh_pp_variants.so
recipe_build_pp_variants() { (
set -e
ROOT=$R
SRC=$ROOT/diffusionhandles_amd/csrc
make -C "$SRC" -j8 >/dev/null
mkdir -p "$ROOT/tools/bin"
OBJS=$(ls "$SRC"/*.o | grep -v gemm_pp.o)
( cd "$SRC" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-c++20-extensions -DDH_PP_VARIANTS -c gemm_pp.hip -o /tmp/dh_gemm_pp_variants.o )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/libdh_pp_variants.so" $OBJS /tmp/dh_gemm_pp_variants.o
ls -la "$ROOT/tools/bin/libdh_pp_variants.so"
) }
# build-tuning: tuning build of the library (-DDH_TUNING) into tools/bin/ (CPU container)
recipe_build_tuning() { (
set -e
ROOT=$R
SRC=${DH_SRC:-$ROOT/diffusionhandles_amd/csrc}          # DH_SRC: another checkout's csrc (A/B of two source states on one box)
NAME=${DH_NAME:-libdiffhandles_hip_tuning.so}
OBJ=${TMPDIR:-/tmp}/dh_obj_$NAME
mkdir -p "$OBJ" "$ROOT/tools/bin"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 ${DH_DEFS--DDH_TUNING} -Wno-unused-function -Wno-unused-result -Wno-c++20-extensions"
pids=()
for f in api.cpp geometry.hip mesh.hip cells.hip energy.hip loop_ops.hip gemm.hip gemm_pp.hip attention.hip unet_kernels.hip unet_engine.cpp vae_engine.cpp text_engine.cpp debug_api.cpp; do
  extra=""; case $f in geometry.hip|mesh.hip) extra="-ffp-contract=off";; esac
  ( cd "$SRC" && $HIPCC $FLAGS $extra -x hip -c $f -o "$OBJ/${f%.*}.o" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/$NAME" "$OBJ"/*.o
echo built "$ROOT/tools/bin/$NAME"
) }
# gemm-warmth: GEMM duration with weights already in the caches vs cold (tools/bench_gemm_warmth.py under rocprofv3)
recipe_gemm_warmth() { (
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp DH_DBG_PRETILED=1
rm -rf /tmp/warmth; rocprofv3 --kernel-trace --output-format csv -d /tmp/warmth -- python3 tools/bench_gemm_warmth.py run > /tmp/warmth.log 2>&1
tail -3 /tmp/warmth.log | cut -c1-200
python3 tools/bench_gemm_warmth.py parse /tmp/warmth | tee gpurun_out/gemm_warmth.txt
) }
# hbm: rocprofv3 kernel traces of the HBM-bound pieces (guidance energy, K=8 re-projection) -> GB/s per kernel
recipe_hbm() { (
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/hbm
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for res in 512 768; do
  rm -rf /tmp/prof_e$res /tmp/prof_r$res
  DH_RES=$res rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e$res -- python3 $R/tools/bench_energy.py 2>/dev/null | tail -1 > $O/energy_$res.json
  python3 $R/tools/hbm_report.py energy $(ls /tmp/prof_e$res/*/*kernel_trace.csv | head -1) $O/energy_$res.json > $O/energy_${res}_kernel_gbps.csv
  DH_RES=$res rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r$res -- python3 $R/tools/bench_reproject.py 2>/dev/null | tail -1 > $O/reproject_$res.txt
  cp $(ls /tmp/prof_r$res/*/*kernel_stats.csv | head -1) $O/reproject_k8_${res}_kernel_stats.csv
  NFG=$(python3 -c "import sys; sys.path.insert(0,'$R'); from diffusionhandles_amd.synthetic import make_scene; print(int(make_scene($res)[2].sum()))")
  python3 $R/tools/hbm_report.py reproject $(ls /tmp/prof_r$res/*/*kernel_trace.csv | head -1) $res $NFG 8 > $O/reproject_k8_${res}_kernel_gbps.csv
done
cat $O/energy_512_kernel_gbps.csv $O/reproject_k8_512_kernel_gbps.csv $O/reproject_512.txt
) }
# pmc-traffic: FETCH_SIZE / WRITE_SIZE passes over tools/time_unet.py 1 -> gpurun_out/pmc/*.tsv (then tools/pmc_summarise.py)
recipe_pmc_traffic() { (
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $GRAFT_REPO_ROOT/tools/time_unet.py ${DH_PMC_BATCH:-1} > /tmp/pmc_$c.log 2>&1
  echo "$c rc=$?"; tail -2 /tmp/pmc_$c.log
  ls /tmp/pmc_$c/*/ | head
done
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  f=$(ls /tmp/pmc_$c/*/*counter_collection.csv | head -1)
  head -3 $f
  python3 - $f $c > $GRAFT_REPO_ROOT/gpurun_out/pmc/$c.tsv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    n = r["Kernel_Name"]
    agg[n][0] += 1; agg[n][1] += float(r["Counter_Value"])
for n, v in agg.items():
    print(f"{n}\t{v[0]}\t{v[1]}")
PY
  wc -l $GRAFT_REPO_ROOT/gpurun_out/pmc/$c.tsv
done
) }
# pmc-sq: SQ counter passes (MFMA busy, LDS conflicts, waits) per kernel family over tools/time_unet.py 1
recipe_pmc_sq() { (
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/sq1 -- python3 $R/tools/time_unet.py ${DH_PMC_BATCH:-1} > /tmp/sq1.log 2>&1; echo "pass1 rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/sq2 -- python3 $R/tools/time_unet.py ${DH_PMC_BATCH:-1} > /tmp/sq2.log 2>&1; echo "pass2 rc=$?"
python3 - <<'PY' > $R/gpurun_out/pmc/sq_summary.txt
import csv, glob, collections, re
def fam(n):
    if "k_gemm_dma" in n or "k_gemm_pp" in n:
        m = re.search(r"Li(\d+)ELi(\d+)E", n)
        nm = "k_gemm_pp" if "k_gemm_pp" in n else "k_gemm_dma"
        return f"{nm} {m.group(1)}x{m.group(2)}" if m else nm
    m = re.search(r"k_[a-z0-9_]+", n)
    return m.group(0) if m else n[:30]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
dur = collections.defaultdict(float)          # kernel time (ns) of the launches of pass 1, from its kernel trace
for d in ("/tmp/sq1", "/tmp/sq2"):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = fam(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if d == "/tmp/sq1" and r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for f in glob.glob("/tmp/sq1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[fam(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
import os
print("# rocprofv3 --pmc (two passes of 8 SQ counters) --kernel-trace -- python3 tools/time_unet.py " + os.environ.get("DH_PMC_BATCH", "1") + "   (U-Net forward+backward at that batch, 13 iterations; sums over all launches)")
print("# raw sums per kernel family; lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; insts per MFMA instruction; wait = SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY")
print("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x 2.4 GHz): share of the chip's matrix-pipe cycles the family's launches kept busy")
print("#   (kernel time = sum of the launches' durations in the same pass's kernel trace: counter collection serialises and slows launches, so")
print("#    this is a LOWER bound of the utilisation in the un-profiled step); mfma_busy/launch in cycles = 32 x MFMA instructions (32x32x16; k_gemm_pp: 16 x its 16x16x32 instructions)")
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))
for k, c in rows[:22]:
    mf = max(1.0, c.get("SQ_INSTS_MFMA", 0))
    ldsc = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0))
    has = c.get("SQ_INSTS_MFMA", 0) > 0
    per = f"per MFMA: valu {c.get('SQ_INSTS_VALU',0)/mf:5.1f} salu {c.get('SQ_INSTS_SALU',0)/mf:5.1f} lds {c.get('SQ_INSTS_LDS',0)/mf:4.1f}" if has else "no MFMA"
    util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * dur[k] * 2.4) if dur.get(k) else float("nan")
    print(f"{k:26s} launches {cnt[k]:5d} avg {dur.get(k, 0) / max(1, cnt[k]) / 1e3:7.2f} us mfma_util {util:6.3f} SQ_BUSY_CYCLES {c.get('SQ_BUSY_CYCLES',0):.3e} SQ_VALU_MFMA_BUSY_CYCLES {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3e} SQ_WAVE_CYCLES {c.get('SQ_WAVE_CYCLES',0):.3e} "
          f"lds_conflict {ldsc:5.3f} {per} wait {c.get('SQ_WAIT_INST_ANY',0)/max(1.0,c.get('SQ_ACTIVE_INST_ANY',0)):5.2f}")
PY
cat $R/gpurun_out/pmc/sq_summary.txt | head -30
) }
# pmc-attn: SQ counter passes on the attention kernels at N = 4096 (tools/bench_attn.py): lab.sh pmc-attn [library]
recipe_pmc_attn() { (
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
) }
# prof-b8: kernel accounting of the batch-8 pass and of the 96x96-latent bf16 pass
recipe_prof_b8() { (
cd "$R"
export TMPDIR=/tmp
mkdir -p gpurun_out/b8 gpurun_out/l96
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b8 -- python3 tools/time_unet.py 8 > gpurun_out/b8/log.txt 2>&1
f=$(ls gpurun_out/b8/*/*kernel_trace.csv | head -1)
python3 tools/trace_by_grid.py $f 500 > gpurun_out/b8/by_grid.txt
python3 - "$f" > gpurun_out/b8/by_type.txt <<'PY'
import csv,sys,re,collections
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    m=re.search(r"k_[a-z0-9_]+",r["Kernel_Name"]); nm=m.group(0) if m else r["Kernel_Name"][:40]
    agg[nm][0]+=1; agg[nm][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in agg.values())
for k,v in sorted(agg.items(),key=lambda kv:-kv[1][1])[:30]: print(f"{k:28s} n={v[0]:6d} total {v[1]:10.1f} avg {v[1]/v[0]:7.1f} {100*v[1]/tot:5.1f}%")
PY
rm -f $f
DH_LATENT=96 DH_DTYPE=bf16 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/l96 -- python3 tools/time_unet.py 1 > gpurun_out/l96/log.txt 2>&1
f=$(ls gpurun_out/l96/*/*kernel_trace.csv | head -1)
python3 tools/trace_by_grid.py $f 300 > gpurun_out/l96/by_grid.txt
rm -f $f
tail -n 3 gpurun_out/b8/log.txt; tail -n 3 gpurun_out/l96/log.txt
) }
# prof-invert: kernel accounting of the per-image phase (null-text inversion + initial inference)
recipe_prof_invert() { (
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/invert
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/invert -- python3 tools/time_invert.py > gpurun_out/invert/log.txt 2>&1
f=$(ls gpurun_out/invert/*/*kernel_trace.csv | head -1)
python3 tools/trace_by_grid.py $f 20000 > gpurun_out/invert/by_grid.txt
python3 - "$f" > gpurun_out/invert/by_type.txt <<'PY'
import csv,sys,re,collections
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    m=re.search(r"k_[a-z0-9_]+",r["Kernel_Name"]); nm=m.group(0) if m else r["Kernel_Name"][:40]
    agg[nm][0]+=1; agg[nm][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in agg.values())
print(f"total busy {tot/1e3:.1f} ms")
for k,v in sorted(agg.items(),key=lambda kv:-kv[1][1])[:30]: print(f"{k:28s} n={v[0]:7d} total {v[1]/1e3:9.1f} ms avg {v[1]/v[0]:7.1f} us {100*v[1]/tot:5.1f}%")
PY
rm -f $f
grep "^rep" gpurun_out/invert/log.txt
) }
# scan-small-grids: every launch shape of the guided steps with its workgroup count (finds launches that leave the chip idle)
recipe_scan_small_grids() { (
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/scan; rocprofv3 --kernel-trace --output-format csv -d /tmp/scan -- python3 bench.py --no-phases --no-res768 --no-cpu-baseline --batch-edits 0 > /dev/null 2>&1
f=$(ls /tmp/scan/*/*kernel_trace.csv | head -1)
python3 - $f <<'PY' > gpurun_out/scan_small_grids.txt
import csv, sys, re, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    m = re.search(r"(k_[a-z0-9_]+)(I[A-Za-z0-9_]*E)?", n)
    short = (m.group(1) + (m.group(2) or "")) if m else n[:50]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    k = (short[:64], g, wg)
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total {tot/1e3:.1f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    waves = k[1] * (k[2] // 64)
    if waves < 1024 and v[1] / tot > 0.001:
        print(f"{100*v[1]/tot:5.2f}%  n={v[0]:6d} avg {v[1]/v[0]:7.2f} us  workgroups {k[1]:5d} x {k[2]:4d} threads = {waves:5d} waves  {k[0]}")
PY
cat gpurun_out/scan_small_grids.txt | head -50
) }
# sweep-knobs: A/B of dispatch knobs of the tuning build on the guided step of bench.py
recipe_sweep_knobs() { (
# lab.sh sweep-knobs ["KNOB=v [KNOB2=w]" ...]: every point once between two baselines; no arguments = the standing list
cd ${GRAFT_REPO_ROOT:-$PWD}
export DIFFHANDLES_LIB=${DIFFHANDLES_LIB:-$PWD/tools/bin/libdiffhandles_hip_tuning.so}
run() { echo "== $*: $(env $@ timeout 200 python bench.py --no-time-edit --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
if [ $# -eq 0 ]; then
  set -- DH_BIG_TILES=96 DH_BIG_TILES=200 DH_SPLITK_MINKT=28 DH_SPLITK_MINKT=40 DH_SPLITK_TARGET=240 DH_SPLITK_TILES=128 \
         DH_GN_SLICES=24 DH_GN_SLICES=48 DH_ATTN_KS=2 DH_ATTN_KS=4 DH_ATTN_QW=2 DH_ATTN_QW=4
fi
run X=0
n=0
for point in "$@"; do
  run $point
  n=$((n + 1)); if [ $((n % 8)) -eq 0 ]; then run X=0; fi
done
run X=0
) }
# sweep-modes: per-shape tile / split-K search on the batch-8 and 96x96-latent GEMM shapes
recipe_sweep_modes() { (
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/tools/bin/libdiffhandles_hip_tuning.so
mkdir -p gpurun_out/sweep
# lab.sh sweep-modes [b8] [l96] [b1]: which passes to take the shapes from (default b8 l96)
[ $# -eq 0 ] && set -- b8 l96
for mode in "$@"; do
  case $mode in
    b8) DH_GEMM_LOG=1 python3 tools/time_unet.py 8 2> gpurun_out/sweep/gemmlog_b8.txt | grep "^B="; export DH_SWEEP_BATCH=8 ;;
    l96) DH_GEMM_LOG=1 DH_LATENT=96 python3 tools/time_unet.py 1 2> gpurun_out/sweep/gemmlog_l96.txt | grep "^B="; export DH_SWEEP_BATCH=1 ;;
    b1) DH_GEMM_LOG=1 python3 tools/time_unet.py 1 2> gpurun_out/sweep/gemmlog_b1.txt | grep "^B="; export DH_SWEEP_BATCH=1 ;;
    *) echo "unknown pass $mode"; exit 2 ;;
  esac
  rm -rf gpurun_out/sweep/trace_$mode
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sweep/trace_$mode -- python3 tools/sweep_gemm_shapes.py run gpurun_out/sweep/gemmlog_$mode.txt gpurun_out/sweep/manifest_$mode.json > gpurun_out/sweep/run_$mode.log 2>&1
  f=$(ls gpurun_out/sweep/trace_$mode/*/*kernel_trace.csv | head -1)
  python3 tools/sweep_gemm_shapes.py parse gpurun_out/sweep/manifest_$mode.json $f > gpurun_out/sweep/result_$mode.txt 2>&1
  rm -rf gpurun_out/sweep/trace_$mode
  tail -4 gpurun_out/sweep/result_$mode.txt
done
) }
# ubench-xcd-trace: kernel durations of the reader kernel of tools/ubench_xcd.hip by chunk size and reader shift
recipe_ubench_xcd_trace() { (
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/xcdtr; rocprofv3 --kernel-trace --output-format csv -d /tmp/xcdtr -- tools/bin/ubench_xcd > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/xcdtr/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rd = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'k_r' in r['Kernel_Name']]
i = 0
for kb in (4, 16, 64, 128):
    for sr in (0, 8, 1, 3, 128):
        seg = sorted(rd[i + 5:i + 55]); i += 55
        print(f"chunk {kb:4d} KB ({kb*256/1024:5.1f} MB total) shift {sr:3d} ({'same' if sr % 8 == 0 else 'other'} XCD): reader kernel median {seg[len(seg)//2]:6.2f} us  min {seg[0]:6.2f}")
PY
) }
# refresh: re-collect the judged measurements (bench line, rocprofv3 kernel stats of bench.py, step breakdowns) into gpurun_out/final/
recipe_refresh() { (
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_n1.json
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --no-phases --no-res768 --no-cpu-baseline --batch-edits 0 2>/dev/null | tail -1 > $O/bench_under_rocprof.json
cd $R
T=$(ls /tmp/prof_bench/*/*kernel_trace.csv | head -1)
S=$(ls /tmp/prof_bench/*/*kernel_stats.csv | head -1)
cp $S $O/rocprofv3_kernel_stats.csv
python tools/step_types.py $T 3 > $O/step_kernel_types.txt
python tools/step_breakdown.py $T 3 > $O/step_breakdown_by_grid.txt
python tools/step_gaps.py $T > $O/step_gaps.txt
head -3 $O/bench_n1.json | cut -c1-400
head -12 $O/step_kernel_types.txt
) }
# evidence: everything the round's profiles/ files come from, in one call (bench line, kernel trace + step breakdowns, SQ counters per
# GEMM tile family, FETCH / WRITE traffic): gpurun_out/final, gpurun_out/pmc
recipe_evidence() { (
  cd "$R"
  # counters first: the traffic figure goes into profiles/ of THIS copy before the bench line is taken, so that the line's
  # roofline.traffic is the figure of the same call (bench.py reads profiles/<round>_pmc_gemm_traffic.json and refuses a stale one)
  recipe_pmc_sq
  recipe_pmc_traffic
  python3 tools/pmc_summarise.py gpurun_out/pmc > gpurun_out/pmc/gemm_traffic.json
  DH_ALG_BYTES=1 python3 tools/time_unet.py 1 2>&1 | grep "^ALG" > gpurun_out/pmc/gemm_algorithmic_bytes.txt
  python3 tools/publish_evidence.py "${1:-r06}" --traffic-only
  recipe_refresh
  python3 tools/step_by_level.py gpurun_out/final/step_breakdown_by_grid.txt > gpurun_out/final/step_by_level.txt
  cat gpurun_out/pmc/gemm_traffic.json
) }

# same-box A/B of DH_GEMM_STAGE values (gemm.hip dh_dbg_gemm_stage: bit 0 descriptor staging, bit 2 = no GroupNorm forward statistics
# in the GEMM epilogue, bit 3 = no backward ones): U-Net passes at B = 1, 2 and the bench line, each value twice, interleaved
recipe_ab_stage() {
  local out=gpurun_out/ab_stage.txt
  mkdir -p gpurun_out; : > "$out"
  [ $# -ge 2 ] || { echo "usage: lab.sh ab-stage <stage A> <stage B> ..."; return 2; }
  for rep in 1 2; do for st in "$@"; do
    echo "== stage $st" >> "$out"
    DH_GEMM_STAGE=$st timeout 200 python tools/time_unet.py 1,2 2>&1 | grep -E "^B=" >> "$out"
  done; done
  echo "# bench.py: steps/s, GEMM avg launch us (HIP events), GEMM launches per step, batched edit-steps/s, 768^2 bf16 steps/s" >> "$out"
  for rep in 1 2; do for st in "$@"; do
    echo "== stage $st" >> "$out"
    DH_GEMM_STAGE=$st timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches_per_step'], d['batched_edits']['edit_steps_per_s'], d['res768_bf16']['value'])" >> "$out"
  done; done
  cat "$out"
}

case "${1:-list}" in
  ab) shift; recipe_ab "$@" ;;
  ab-libs) shift; recipe_ab_libs "$@" ;;
  ab-attn) shift; recipe_ab_attn "$@" ;;
  ab-step) shift; recipe_ab_step "$@" ;;
  ab-stage) shift; recipe_ab_stage "$@" ;;
  ablate-gemm) shift; recipe_ablate_gemm "$@" ;;
  ablate-attn) shift; recipe_ablate_attn "$@" ;;
  build-tuning) shift; recipe_build_tuning "$@" ;;
  build-pp-variants) shift; recipe_build_pp_variants "$@" ;;
  gemm-warmth) shift; recipe_gemm_warmth "$@" ;;
  hbm) shift; recipe_hbm "$@" ;;
  pmc-traffic) shift; recipe_pmc_traffic "$@" ;;
  pmc-sq) shift; recipe_pmc_sq "$@" ;;
  pmc-attn) shift; recipe_pmc_attn "$@" ;;
  prof-b8) shift; recipe_prof_b8 "$@" ;;
  prof-invert) shift; recipe_prof_invert "$@" ;;
  scan-small-grids) shift; recipe_scan_small_grids "$@" ;;
  sweep-knobs) shift; recipe_sweep_knobs "$@" ;;
  sweep-modes) shift; recipe_sweep_modes "$@" ;;
  ubench-xcd-trace) shift; recipe_ubench_xcd_trace "$@" ;;
  refresh) shift; recipe_refresh "$@" ;;
  evidence) shift; recipe_evidence "$@" ;;
  list|*) cat <<'EOT'
recipes:
  ab                 same-box A/B of library builds on any command: lab.sh ab "<command>" libA.so libB.so ... (each run twice, interleaved)
  ab-libs            same-box A/B of library builds: U-Net passes at B = 1, 2, 8, the 96x96 latent (bf16), the batch-8 GEMM shapes
  ab-attn            same-box A/B of attention builds (tools/bench_attn.py)
  ab-step            same-box A/B of library builds on the guided step (tools/ab_inplace.py timing, in-place I/O on)
  ab-stage           same-box A/B of DH_GEMM_STAGE values (staging form, GroupNorm statistics in the GEMM epilogues): lab.sh ab-stage 9 1
  ablate-gemm        k_gemm_dma ablations (0 full, 1 staging only, 2 compute only) on the batch-8 shapes; needs the tuning build
  ablate-attn        builds tools/bin/libdh_attn_<n>.so with attention.hip under -DDH_ATTN_ABL=n (timing-only ablations; CPU container)
  build-tuning       tuning build of the library (-DDH_TUNING) into tools/bin/ (CPU container)
  build-pp-variants  the library with every k_gemm_pp main-loop variant (-DDH_PP_VARIANTS) into tools/bin/libdh_pp_variants.so (CPU container)
  gemm-warmth        GEMM duration with weights already in the caches vs cold (tools/bench_gemm_warmth.py under rocprofv3)
  hbm                rocprofv3 kernel traces of the HBM-bound pieces (guidance energy, K=8 re-projection) -> GB/s per kernel
  pmc-traffic        FETCH_SIZE / WRITE_SIZE passes over tools/time_unet.py 1 -> gpurun_out/pmc/*.tsv (then tools/pmc_summarise.py)
  pmc-sq             SQ counter passes (MFMA busy, LDS conflicts, waits) per kernel family over tools/time_unet.py 1
  pmc-attn           SQ counter passes on the attention kernels at N = 4096 (tools/bench_attn.py): lab.sh pmc-attn [library]
  prof-b8            kernel accounting of the batch-8 pass and of the 96x96-latent bf16 pass
  prof-invert        kernel accounting of the per-image phase (null-text inversion + initial inference)
  scan-small-grids   every launch shape of the guided steps with its workgroup count (finds launches that leave the chip idle)
  sweep-knobs        A/B of dispatch knobs of the tuning build on the guided step of bench.py
  sweep-modes        per-shape tile / split-K search on the batch-8 and 96x96-latent GEMM shapes
  ubench-xcd-trace   kernel durations of the reader kernel of tools/ubench_xcd.hip by chunk size and reader shift
  refresh            re-collect the judged measurements (bench line, rocprofv3 kernel stats of bench.py, step breakdowns) into gpurun_out/final/
  evidence           refresh + pmc-sq + pmc-traffic + summaries: everything a round's profiles/ files come from
EOT
  ;;
esac
