cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $GRAFT_REPO_ROOT/tools/time_unet.py 1 > /tmp/pmc_$c.log 2>&1
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
