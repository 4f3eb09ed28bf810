#!/bin/bash
# L2 behaviour of the gather assembly of the explicit S (k_pair_items_h) on the Final shape: hits, misses, requests
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_pair
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$n -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sparse-schur --solver sparse_schur --steps 1 --warmup 1 > /dev/null 2> $OUT/$n.err
  f=$(ls $OUT/$n/*/*counter_collection.csv | head -1)
  python3 - "$f" <<PY
import csv, sys, collections
tot=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0][-30:]
    if "pair_items" not in k and "row_h" not in k and "sp_update" not in k: continue
    tot[(k, r["Counter_Name"])]+=float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])]+=1
for key in sorted(tot): print("%-32s %-34s %14.0f per call" % (key[0], key[1], tot[key]/cnt[key]))
PY
  rm -rf $OUT/$n
done
