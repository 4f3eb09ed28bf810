#!/bin/bash
# FETCH_SIZE of the kernels of one SPARSE_SCHUR solve on the Final shape (k_pair_items, k_sp_update, ...)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sparse
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sparse-schur --solver sparse_schur $BENCH_ARGS --steps 1 --warmup 1 > /dev/null 2> $OUT/fetch.err
cp $(ls $OUT/fetch/*/*counter_collection.csv | head -1) $OUT/fetch.csv && rm -rf $OUT/fetch
python3 - <<PY
import csv, collections
tot=collections.defaultdict(float); cnt=collections.Counter(); dur=collections.defaultdict(float)
for r in csv.DictReader(open("$OUT/fetch.csv")):
    if r["Counter_Name"]!="FETCH_SIZE": continue
    k=r["Kernel_Name"].split("(")[0][-40:]
    tot[k]+=float(r["Counter_Value"]); cnt[k]+=1; dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
for k in sorted(tot, key=lambda k:-tot[k])[:12]:
    print("%-42s calls %5d  fetch(x2 corrected) %8.2f GB/call  %8.3f ms/call" % (k, cnt[k], 2*tot[k]*1024/cnt[k]/1e9, dur[k]/cnt[k]/1e6))
PY
