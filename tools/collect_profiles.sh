#!/bin/bash
# Round-N profile collection on the GPU box (run from the repo root through gpurun):
#   rocprofv3 kernel-trace + stats of the bench commands DESIGN.md quotes, and the two PMC passes (FETCH_SIZE,
#   WRITE_SIZE; separate runs, no trace domains besides --kernel-trace) that profiles/make_traffic.py turns into bytes.
# Everything lands under gpurun_out/prof_rNN/; copy what is to be kept into profiles/.
set -o pipefail
R=${CX_ROUND:-04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sparse-schur --no-boundary "$@" > $OUT/$name.json 2> $OUT/$name.err
  cp $(ls $OUT/$name/*/*kernel_stats.csv | head -1) $OUT/${name}_kernel_stats.csv
  echo "$name done" 
}
run final13682 --steps 3 --warmup 1
run final13682_sparse_schur --solver sparse_schur --steps 2 --warmup 1
run final13682_sparse_schur_mixed --solver sparse_schur --mixed --steps 2 --warmup 1
run final13682_cluster_tridiagonal --eta 1e-2 --preconditioner cluster_tridiagonal --steps 2 --warmup 1
run dubrovnik356_dense_schur --workload dubrovnik356 --solver dense_schur --steps 5 --warmup 2
run ladybug49 --workload ladybug49 --steps 20 --warmup 3
run ladybug49_cgnr --workload ladybug49 --solver cgnr --steps 20 --warmup 3
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sparse-schur --no-boundary --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_$c.err
  cp $(ls $OUT/pmc_$c/*/*counter_collection.csv | head -1) $OUT/pmc_$c.csv
  echo "pmc $c done"
done
# keep the merged output small: drop the raw traces
rm -rf $OUT/final13682 $OUT/final13682_sparse_schur $OUT/final13682_sparse_schur_mixed $OUT/final13682_cluster_tridiagonal $OUT/dubrovnik356_dense_schur $OUT/ladybug49 $OUT/ladybug49_cgnr $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
