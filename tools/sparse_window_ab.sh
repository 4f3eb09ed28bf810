#!/bin/bash
# A/B of the schedule of the tile-sparse factorisation: bench.py --solver sparse_schur on the Final shape and on the second scene under
# "VAR=value ..." settings, one per argument; prints solve ms and the phases.  Run on the GPU box:
#   tools/sparse_window_ab.sh "CX_SPARSE_WINDOW=1 CX_SPARSE_WAVE_TARGETS=0" "CX_SPARSE_WINDOW=4"
i=0
for cfg in "$@"; do
  i=$((i+1))
  for wl in final13682 final13682_revisit; do
    env $cfg python bench.py --workload $wl --solver sparse_schur --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/win_${i}_${wl}.json 2> gpurun_out/win_${i}_${wl}.err || exit 1
    python - "$cfg" "$wl" gpurun_out/win_${i}_${wl}.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith('{')][-1])
print(sys.argv[1], "|", sys.argv[2], "solve %.2f ms" % d["value"], {k: round(v, 2) for k, v in d["phases_ms_per_solve"].items() if k.endswith("_ms") and v}, flush=True)
PY
  done
done
