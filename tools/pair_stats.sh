#!/bin/bash
# kernel stats of one SPARSE_SCHUR bench run (Final shape): the explicit-S assembly kernels
OUT=gpurun_out/pair_stats_$1; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CX_PAIR_CAM_MAJOR=$1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT -o run -- python3 $GRAFT_REPO_ROOT/bench.py --solver sparse_schur --no-cpu-baseline --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/err.txt || exit 1
cd $GRAFT_REPO_ROOT
grep -E "k_pair_items|k_row_h|k_map_rows|k_sp_assemble" $(ls $OUT/*kernel_stats.csv $OUT/*/*kernel_stats.csv 2>/dev/null | head -1) | cut -c1-60,150-260
