#!/bin/bash
# kernel breakdown of the dynamic-size path: tools/embed_timing.py with the embedding switched off, under rocprofv3
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_generic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CX_NO_EMBEDDING=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 $GRAFT_REPO_ROOT/tools/embed_timing.py > $OUT/out.txt 2> $OUT/err.txt
cp $(ls $OUT/run/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv && rm -rf $OUT/run
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/kernel_stats.csv")))[:16]:
    print("%-70s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
