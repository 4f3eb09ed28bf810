#!/bin/bash
# SPARSE_SCHUR on the Final shape and on the second scene under "VAR=value ... | bench flags" settings, one per argument, e.g.
#   tools/sparse_mixed_ab.sh "|" "|--mixed" "CX_SPARSE_F32_LDS=0|--mixed" "|--mixed --refinements 1" "|--refinements 1"
# (default: those five).  Prints solve ms and the phases.
[ $# -eq 0 ] && set -- "|" "|--mixed" "CX_SPARSE_F32_LDS=0|--mixed" "|--mixed --refinements 1" "|--refinements 1"
for wl in final13682 final13682_revisit; do
  for cfg in "$@"; do
    envs=${cfg%%|*}; flags=${cfg#*|}
    env $envs python bench.py --workload $wl --solver sparse_schur --no-cpu-baseline --steps 3 --warmup 1 $flags > gpurun_out/mixed_ab.json 2> gpurun_out/mixed_ab.err || { tail -5 gpurun_out/mixed_ab.err; exit 1; }
    python - "$wl" "$cfg" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/mixed_ab.json") if l.startswith('{')][-1])
print(sys.argv[1], "[%s]" % sys.argv[2], "solve %.2f ms" % d["value"], {k: round(v, 2) for k, v in d["phases_ms_per_solve"].items() if k.endswith("_ms") and v}, flush=True)
PY
  done
done
