#!/bin/bash
# The end-to-end runs of profiles/rNN_bundle_adjuster_runs.log (whole minimisations on the device, cx_minimize).
set -o pipefail
R=${CX_ROUND:-02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r${R}_bundle_adjuster_runs.log
BA="python $GRAFT_REPO_ROOT/ceres-solver-ceres-solver_amd/examples/bundle_adjuster.py"
: > $OUT
run() { echo "### bundle_adjuster.py $*" >> $OUT; $BA "$@" >> $OUT 2>&1; }
run --preset final13682 --num_iterations 8 --preconditioner jacobi
run --preset final13682 --num_iterations 8 --preconditioner cluster_jacobi
run --preset final13682 --num_iterations 8 --preconditioner cluster_tridiagonal
run --preset final13682 --num_iterations 8 --preconditioner schur_jacobi --explicit_schur_complement
run --preset final13682 --num_iterations 8 --linear_solver sparse_schur
run --preset dubrovnik356 --linear_solver dense_schur
if [ "$R" != "02" ]; then
  # round 3: the same minimisation with a ScaleColumns pass after every evaluation (A/B of the folded Jacobi scaling), on the
  # second scene, with the programs of --robustify / --use_quaternions, and on four logical shards behind one set of handles
  echo "### CX_NO_FUSED_SCALING=1 bundle_adjuster.py --preset final13682 --num_iterations 8 --linear_solver sparse_schur" >> $OUT
  CX_NO_FUSED_SCALING=1 $BA --preset final13682 --num_iterations 8 --linear_solver sparse_schur >> $OUT 2>&1
  run --preset final13682_revisit --num_iterations 8 --linear_solver sparse_schur
  run --preset final13682_revisit --num_iterations 8 --preconditioner jacobi
  run --preset final13682 --num_iterations 8 --linear_solver sparse_schur --robustify
  run --preset final13682 --num_iterations 8 --linear_solver sparse_schur --use_quaternions
  run --preset final13682 --num_iterations 8 --preconditioner jacobi --devices 0,0,0,0
  run --preset final13682 --num_iterations 8 --linear_solver sparse_schur --devices 0,0,0,0
fi
if [ "$R" != "02" ] && [ "$R" != "03" ]; then
  # round 4: CGNR (operator without memsets, set-up in two passes over J)
  run --preset synthetic10M --num_iterations 8 --linear_solver cgnr
fi
grep -n "^###\|^Time\|Minimizer iterations" $OUT
