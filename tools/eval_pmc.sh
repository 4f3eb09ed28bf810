#!/bin/bash
# Instruction-issue counters of k_bal_evaluate (tools/eval_ab.py under rocprofv3 --pmc, kernel trace only).
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/eval_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
pass() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/eval_ab.py > $OUT/$name.json 2> $OUT/$name.err
  cp $(ls $OUT/$name/*/*counter_collection.csv | head -1) $OUT/$name.csv && rm -rf $OUT/$name
  echo "$name done"
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS
pass sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
