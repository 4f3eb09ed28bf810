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

# ---- SPARSE_SCHUR with the distributed tile-sparse factorisation (round 3): rank 0 of N = 1, 2, 4, 8 alone on the GPU.
# The exchange step is replaced by a callback that leaves the buffers as they are, EXCEPT the one-time presence exchange of
# the structure, which gets the true union of all ranks' S cells (computed here from the whole problem) -- so rank 0 holds the
# real plan of an N-rank run (its own subtrees + the replicated top) and does a rank's real work: eliminating its points,
# assembling, factoring its subtrees, factoring the top, the triangular solves.  The numbers it computes are NOT the solution
# (the other ranks' contributions are missing); the times are a rank's compute times, without the collectives.
import ctypes
if "--sparse-schur" in sys.argv:
    C, P = full.num_cameras, full.num_points
    bs_full, _ = cx.bal.build_structure(full)
    cr, cc, _, _ = cx.binding.schur_pair_lists_host(bs_full, P)
    presence = np.zeros(C * C + 1)
    presence[cr.astype(np.int64) * C + cc] = 1.0
    lib = cx.load_library()
    sparse_out = []
    for n in (1, 2, 4, 8):
        c2 = cx.Context(0)
        if n > 1:
            def raw_cb(dptr, count, _user, _c=c2):
                if count == C * C + 1:
                    lib.cx_memcpy_h2d(_c._h, ctypes.c_void_p(dptr), presence.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(8 * (C * C + 1)))
                return 0
            c2._raw_cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)(raw_cb)
            assert lib.cx_context_set_comm_callback(c2._h, 0, n, c2._raw_cb, None) == 0
        bounds = cx.bal.partition_points(full, n)
        prob = full if n == 1 else cx.bal.shard(full, int(bounds[0]), int(bounds[1]))
        ev, A, b, D, cost, eval_ms, upd = bench.lm_prepare_device(cx, c2, prob)
        S = cx.Solver(c2, type=cx.SPARSE_SCHUR, num_eliminate_blocks=prob.num_points)
        x = c2.empty(A.num_cols)
        S.solve(A, b, D, x=x)
        reps = []
        for _ in range(3):
            S.solve(A, b, D, x=x)
            reps.append(S.timing())
        tm = {k: float(np.median([r[k] for r in reps])) for k in reps[0]}
        dist = cx.binding.sparse_cholesky_distribution_host(C, cr, cc, n)
        rec = {"solver": "sparse_schur", "ranks": n, "residual_blocks_of_rank_0": int(prob.num_observations),
               "eliminate_ms": tm["eliminate_ms"], "reduced_solve_ms": tm["reduced_solve_ms"], "back_substitute_ms": tm["back_substitute_ms"],
               "device_total_ms": tm["total_ms"], "collectives_per_solve": tm["allreduce_calls"], "collective_bytes_per_solve": tm["allreduce_bytes"],
               "tile_pair_updates_own_rank_0": int(dist["updates_per_rank"][0]), "tile_pair_updates_replicated": int(dist["updates_replicated"]),
               "tiles_replicated": int(dist["tiles_replicated"])}
        sparse_out.append(rec)
        print(json.dumps(rec), flush=True)
        S.close(); ev.close(); c2.close()
    print(json.dumps({"sparse_schur_summary": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in sparse_out]}))
