#!/bin/bash
# Registers and scratch of every kernel of one translation unit (gfx950), from the compiler's own metadata:
#   tools/kernel_regs.sh ceres-solver-ceres-solver_amd/csrc/cx_eval.hip [filter]
# prints: kernel name, scratch bytes per lane, SGPRs, VGPRs.  No GPU needed.
set -e
SRC=$(realpath "$1")
FILTER=${2:-.}
EXTRA=""
case "$SRC" in *cx_eval.hip) EXTRA="-mllvm -disable-machine-licm" ;; esac
TMP=$(mktemp -d)
cd "$TMP"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $EXTRA -x hip -c "$SRC" --save-temps -o unit.o > /dev/null 2>&1
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|name):" ./*gfx950.s | paste - - - - | awk '{print $2, "scratch", $4, "sgpr", $6, "vgpr", $8}' | grep -E "$FILTER" || true
cd /
rm -rf "$TMP"
