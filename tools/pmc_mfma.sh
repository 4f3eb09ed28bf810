#!/bin/bash
# Matrix-core counters of the update kernels of one SPARSE_SCHUR solve (Final shape): SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES and
# GRBM_GUI_ACTIVE per launch, for the double and the single precision pool.  PMC pass with --kernel-trace only.
#   tools/pmc_mfma.sh            -> gpurun_out/pmc_mfma_{f64,f32}.txt
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export CX_SPARSE_PLAN_STATS=1
for mode in f64 f32; do
  flags=""; [ $mode = f32 ] && flags="--mixed"
  rm -rf $OUT/pmc_mfma_$mode
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_$mode -o run -- python3 $GRAFT_REPO_ROOT/bench.py --solver sparse_schur --no-cpu-baseline --steps 1 --warmup 1 $flags > /dev/null 2> $OUT/pmc_mfma_$mode.err || exit 1
  python3 - $OUT/pmc_mfma_$mode $OUT/pmc_mfma_$mode.err > $OUT/pmc_mfma_$mode.txt <<'PY'
import csv, glob, re, sys, collections
d, err = sys.argv[1], sys.argv[2]
stats = [tuple(int(x) for x in m.groups()) for m in (re.match(r"cxsp level (\d+) rows (\d+) panels (\d+) targets (\d+) products (\d+) longest (\d+)", l) for l in open(err)) if m]
L = max(s[0] for s in stats) + 1
products = {s[0]: s[4] for s in stats[-L:]}
rows = list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))
per = collections.OrderedDict()
for r in rows:
    if "k_sp_update" not in r["Kernel_Name"]:
        continue
    k = r["Dispatch_Id"]
    e = per.setdefault(k, {"t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
launches = list(per.values())
# the last solve's update launches, in level order: the levels that have targets
levels = [l for l in range(L) if products.get(l, 0) > 0]
launches = launches[-len(levels):]
print("level products dur_us ns/product clock_GHz mfma_busy_cycles/MFMA mfma_busy_fraction_of_SIMD_cycles")
tot = collections.Counter()
for l, e in zip(levels, launches):
    dur = (e["t1"] - e["t0"]) * 1e-9
    clock = e["GRBM_GUI_ACTIVE"] / 8 / dur
    n_mfma = products[l] * 256.0                       # 4 wavefronts x 64 instructions per product
    busy = e["SQ_VALU_MFMA_BUSY_CYCLES"]
    simd_cycles = e["GRBM_GUI_ACTIVE"] / 8 * 1024      # 256 CUs x 4 SIMDs
    if products[l] >= 5000:
        print("%4d %8d %8.1f %7.2f %6.2f %8.1f %6.3f" % (l, products[l], dur * 1e6, dur * 1e9 / products[l], clock * 1e-9, busy / n_mfma, busy / simd_cycles))
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
        tot[k] += e[k]
    tot["dur"] += dur; tot["mfma"] += n_mfma
print("all update launches: %.2f ms, mean clock %.2f GHz, %.1f busy cycles per MFMA, busy fraction %.3f" % (tot["dur"] * 1e3, tot["GRBM_GUI_ACTIVE"] / 8 / tot["dur"] * 1e-9, tot["SQ_VALU_MFMA_BUSY_CYCLES"] / tot["mfma"], tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (tot["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
  tail -4 $OUT/pmc_mfma_$mode.txt
  rm -rf $OUT/pmc_mfma_$mode
done
