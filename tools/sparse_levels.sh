#!/bin/bash
# Per-level timing of the tile-sparse factorisation: kernel trace of one bench run joined with the plan's per-level statistics
# (CX_SPARSE_PLAN_STATS).  BENCH_ARGS="--mixed" tools/sparse_levels.sh <window> [workload] -> gpurun_out/levels_<window>_<workload>.txt
W=${1:-1}; WL=${2:-final13682}
OUT=gpurun_out/levels_${W}_${WL}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CX_SPARSE_WINDOW=$W CX_SPARSE_PLAN_STATS=1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --solver sparse_schur --no-cpu-baseline --steps 1 --warmup 1 $BENCH_ARGS > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/err.txt || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/sparse_levels.py $OUT > $OUT.txt && tail -25 $OUT.txt
