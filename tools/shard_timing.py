#!/usr/bin/env python3
"""What ONE rank of an N-rank run computes, timed on one GPU without the exchange step: shard 0 of N (points
partitioned by observation count, all cameras) of the Final-13682 workload, ITERATIVE_SCHUR + JACOBI, a fixed number of CG
iterations.  The per-rank compute time of an N-GPU solve; what it leaves out is the all-reduce of 9 C doubles per iteration
(0.99 MB) and the fused set-up collective (5.9 MB) -- see DESIGN.md section 5.  N = 8-GPU hardware is not available to the
build; this is the measured half of the scaling model, not a scaling measurement."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import bench
cx = bench.load_cx()
ctx = cx.Context(0)
full = cx.bal.make_preset("final13682")
out = []
for n in (1, 2, 4, 8):
    bounds = cx.bal.partition_points(full, n)
    prob = full if n == 1 else cx.bal.shard(full, int(bounds[0]), int(bounds[1]))
    ev, A, b, D, cost, eval_ms, upd = bench.lm_prepare_device(cx, ctx, prob)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=prob.num_points,
                  max_num_iterations=20, min_num_iterations=20)
    x = ctx.empty(A.num_cols)
    for _ in range(2):
        S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.0, x=x)
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        _, summ = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.0, x=x)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    tm = S.timing()
    rec = {"ranks": n, "residual_blocks_of_rank_0": int(prob.num_observations), "cg_iterations": int(summ.num_iterations),
           "solve_ms": ms, "eliminate_ms": tm["eliminate_ms"], "reduced_solve_ms": tm["reduced_solve_ms"],
           "back_substitute_ms": tm["back_substitute_ms"],
           "cg_ms_per_iteration": tm["reduced_solve_ms"] / max(1, int(summ.num_iterations)),
           "jacobian_eval_ms": upd["jacobian_eval_ms"], "scale_columns_ms": upd["scale_columns_ms"]}
    out.append(rec)
    print(json.dumps(rec), flush=True)
    S.close(); ev.close()
base = out[0]
for r in out:
    r["compute_speedup_vs_1"] = base["solve_ms"] / r["solve_ms"]
print(json.dumps({"summary": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in out]}))
