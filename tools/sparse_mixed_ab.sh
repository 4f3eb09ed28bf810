#!/bin/bash
# SPARSE_SCHUR on the Final shape and on the second scene: double precision, single precision factor (use_mixed_precision_solves)
# without and with refinement steps.  Prints solve ms, the phases and the error of the step against the double precision one.
for wl in final13682 final13682_revisit; do
  for cfg in "" "--mixed" "--mixed --refinements 1" "--mixed --refinements 2" "--refinements 1"; do
    python bench.py --workload $wl --solver sparse_schur --no-cpu-baseline --steps 3 --warmup 1 $cfg > gpurun_out/mixed_ab.json 2> gpurun_out/mixed_ab.err || { tail -5 gpurun_out/mixed_ab.err; exit 1; }
    python - "$wl" "$cfg" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/mixed_ab.json") if l.startswith('{')][-1])
print(sys.argv[1], "[%s]" % sys.argv[2], "solve %.2f ms" % d["value"], {k: round(v, 2) for k, v in d["phases_ms_per_solve"].items() if k.endswith("_ms") and v}, flush=True)
PY
  done
done
