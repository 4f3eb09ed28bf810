#!/usr/bin/env python3
"""Benchmark of the LM inner linear solve on MI355X (BASELINE.json metric:
linear-solve ms/iter + J SpMV GB/s on BAL final-13682).

One "step" = one LinearSolver::Solve exactly as LevenbergMarquardtStrategy::
ComputeStep issues it at an LM iteration (levenberg_marquardt_strategy.cc:69-156):
ITERATIVE_SCHUR + JACOBI, q_tolerance = eta = 0.1, r_tolerance = -1,
max_num_iterations = 500, on the Jacobi-scaled Jacobian of a synthetic problem with
the public BAL Final-13682 header sizes (no BAL file exists offline).  Inputs
(J values, residuals, D) are resident in HBM before the timed region.

N > 1: one rank per GPU, launched by torch.distributed.run -- by the caller, or, when
`python bench.py --gpus N` is started plainly (no WORLD_SIZE in the environment), by this
script itself as a child process before anything here touches a GPU.  Points (and with
them residual blocks) are sharded over the ranks, camera-space sums go through RCCL
all-reduce inside the library (strong scaling of the fixed problem).

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# the host driver of this pool supports dmabuf IPC only: RCCL between processes needs this before the runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _load(name, path, search=None):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=search)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_cx():
    pkg = os.path.join(ROOT, "ceres-solver-ceres-solver_amd")
    return _load("cxschur", os.path.join(pkg, "__init__.py"), [pkg])


def load_oracle():
    return _load("orc", os.path.join(ROOT, "oracle", "orc.py"))


HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBPS = 6290.0          # measured float4 copy
MIN_LM_DIAGONAL, MAX_LM_DIAGONAL = 1e-6, 1e32   # solver.h:294-295
INITIAL_RADIUS = 1e4            # solver.h initial_trust_region_radius
ETA = 0.1                       # overwritten from --eta in main()


def lm_prepare_device(cx, ctx, prob):
    """What TrustRegionMinimizer / LevenbergMarquardtStrategy do before the first
    linear solve, all on the device: evaluate r and J, Jacobi-scale J
    (trust_region_minimizer.cc:263-279), build the LM diagonal D
    (levenberg_marquardt_strategy.cc:79-95).  Returns (evaluator, A, b, D, cost)."""
    ev = cx.Evaluator(ctx, prob)
    A = ev.jacobian()
    P, C = prob.num_points, prob.num_cameras
    ncols = 3 * P + 9 * C
    state = ctx.to_device(prob.state())
    res = ctx.empty(2 * prob.num_observations)
    cost, _, _ = ev.evaluate(state, residuals=res, gradient=None, want_jacobian=True)
    eval_ms = ev.last_kernel_ms
    sq = ctx.empty(ncols)

    def colnorm():
        A.squared_column_norm(sq)
        if ctx.num_ranks > 1:
            ctx.allreduce_sum(sq, offset=3 * P, count=9 * C)
        ctx.synchronize()
        return sq.to_host()

    scale = 1.0 / (1.0 + np.sqrt(colnorm()))
    dscale = ctx.to_device(scale)
    A.scale_columns(dscale)
    # What an LM iteration spends on producing the values the solve reads: the evaluation and the column scaling,
    # repeated on the same state (same J every time).  Both kernels also keep the camera-major copy of the F cells
    # current, which in round 1 was a separate pass (k_permute_ft) charged to the solve.
    ev.set_emit_camera_major(False)      # ScaleColumns follows every evaluation here, as in TrustRegionMinimizer
    eval_t, scale_t = [], []
    for _ in range(3):
        ev.evaluate(state, residuals=res, gradient=None, want_jacobian=True)
        eval_t.append(ev.last_kernel_ms)
        ctx.synchronize()
        t0 = time.perf_counter()
        A.scale_columns(dscale)
        ctx.synchronize()
        scale_t.append((time.perf_counter() - t0) * 1e3)
    round2 = {"jacobian_eval_ms": float(np.median(eval_t)), "scale_columns_ms": float(np.median(scale_t)),
              "camera_major_copy": "separate k_permute_ft pass inside the solve (CX_NO_FT_EMIT)" if os.environ.get("CX_NO_FT_EMIT")
              else "written by k_scale_239 / k_bal_evaluate"}
    # Round 3, what cx_minimize and the adapter's set_fuse_jacobi_scaling do from the second LM iteration on: the scaling
    # vector of iteration 0 is registered with the evaluator, which writes J diag(scale) itself (bit for bit the values of
    # evaluate -> ScaleColumns) and rebuilds the camera-major copy by a gather pass inside the same timed region -- no
    # ScaleColumns pass.  The values it leaves are the ones the timed solves below read.
    ev.set_column_scale(dscale)
    fused_t = []
    for _ in range(4):
        ev.evaluate(state, residuals=res, gradient=None, want_jacobian=True)
        fused_t.append(ev.last_kernel_ms)
    ev.set_column_scale(None)
    update = {"jacobian_eval_ms": float(np.median(fused_t[1:])), "scale_columns_ms": 0.0,
              "flow": "scaled evaluation (cx_evaluator_set_column_scale) + gather pass for the camera-major copy; no ScaleColumns",
              "round2_flow": round2}
    diag = np.clip(colnorm(), MIN_LM_DIAGONAL, MAX_LM_DIAGONAL)
    D = ctx.to_device(np.sqrt(diag / INITIAL_RADIUS))
    return ev, A, res, D, cost, eval_ms, update


def cpu_baseline(cx, prob, solver_kw, threads, fraction, solver="iterative_schur"):
    """Oracle (CPU restatement, kind 'port'), same LM preparation and solve, on the first `fraction` of the
    points (1.0 = the whole workload; smaller samples use the sharding rule of the multi-GPU path and
    are scaled by the residual-block ratio).  solver = "sparse_schur": SparseSchurComplementSolver with the oracle's
    block-sparse Cholesky in SuiteSparse's place (oracle/orc_sparse_chol.h) -- the north star's host SPARSE_SCHUR
    denominator; its factorisation does not scale with a point sample, so that baseline is timed on the whole workload."""
    orc = load_oracle()
    flags = orc.use_native_build()     # -O3 -march=native, compiled on the host that is being timed
    orc.lib()
    orc.set_num_threads(threads)
    P = prob.num_points
    hi = max(1, min(P, int(round(P * fraction))))
    sub = prob if hi == P else cx.bal.shard(prob, 0, hi)
    bs, order = cx.bal.build_structure(sub)
    _, res, _, vals = orc.bal_evaluate(bs, sub.num_cameras, sub.num_points, sub.camera_index, sub.point_index,
                                       sub.observations, order, sub.state(), want_gradient=False)
    scale = 1.0 / (1.0 + np.sqrt(orc.squared_column_norm(bs, vals)))
    vals = orc.scale_columns(bs, vals, scale)
    diag = np.clip(orc.squared_column_norm(bs, vals), MIN_LM_DIAGONAL, MAX_LM_DIAGONAL)
    # cameras unseen by the sample keep only the clamp value; the solve stays well posed
    D = np.sqrt(diag / INITIAL_RADIUS)
    if solver == "sparse_schur":
        o = orc.make_options(type=orc.SPARSE_SCHUR, num_eliminate_blocks=sub.num_points)
        t0 = time.time()
        x, s = orc.solve(bs, vals, res, D, o)
        wall = time.time() - t0
        numeric_s = orc.last_solve_seconds()
        st = orc.sparse_schur_stats()
        return {
            "value": numeric_s * 1e3, "unit": "ms", "cores": threads, "kind": "port", "compiler_flags": flags,
            "phases_s": {k: st[k] for k in ("eliminate_s", "factor_s", "solve_s")}, "analysis_s": st["analyze_s"],
            "factor_gflops": st["factor_flops"] / max(st["factor_s"], 1e-9) * 1e-9,
            "sample": "oracle SPARSE_SCHUR (eliminator into block-sparse S of %d cells, own nested-dissection block "
                      "Cholesky: %d blocks in L, %.3g flop, %d elimination-tree heights) on points [0,%d) of the workload "
                      "(%d of %d residual blocks): %.0f ms numeric, %.0f ms with structure analysis" % (
                          int(st["s_cells"]), int(st["factor_blocks"]), st["factor_flops"], int(st["etree_heights"]), hi,
                          sub.num_observations, prob.num_observations, numeric_s * 1e3, wall * 1e3),
        }
    o = orc.make_options(type=orc.ITERATIVE_SCHUR, preconditioner_type=orc.JACOBI,
                         num_eliminate_blocks=sub.num_points, max_num_iterations=solver_kw["max_num_iterations"])
    t0 = time.time()
    x, s = orc.solve(bs, vals, res, D, o, r_tolerance=-1.0, q_tolerance=ETA)
    wall = time.time() - t0
    numeric_s = orc.last_solve_seconds()
    ratio = prob.num_observations / sub.num_observations
    return {
        "value": numeric_s * 1e3 * ratio,
        "unit": "ms",
        "cores": threads,
        "kind": "port",
        "compiler_flags": flags,
        "sample": "oracle ITERATIVE_SCHUR+JACOBI on points [0,%d) of the workload (%d of %d residual blocks, all "
                  "cameras): %.0f ms for %d CG iterations (%.0f ms incl. structure set-up), scaled by the residual-"
                  "block ratio %.2f" % (hi, sub.num_observations, prob.num_observations, numeric_s * 1e3,
                                        s.num_iterations, wall * 1e3, ratio),
    }


def boundary_block(cx, ctx, prob, solver_kw, iterations):
    """What an LM iteration costs THROUGH the Evaluator / SparseMatrix / LinearSolver boundary: the unmodified call
    sequence of TrustRegionMinimizer with HOST vectors (ceres-solver-ceres-solver_amd/boundary.py, the ctypes twin of
    host/test_host_adapter --time) on one shard and on four logical shards behind the front, next to the same
    iterations of the device-resident loop (cx_minimize).  The adapters' three opt-ins are on (fused Jacobi scaling,
    residual aliasing, zeroed product target) and the loop's vectors are registered with the HIP runtime on first sight
    (CxRegisterCallerArrays: the loop owns their lifetime and releases the registrations before they are freed)."""
    n, m = 3 * prob.num_points + 9 * prob.num_cameras, 2 * prob.num_observations
    block = {"workload": "LM iterations 1..%d from the start point, ITERATIVE_SCHUR + JACOBI, eta = %g" % (iterations, ETA),
             "unmodified_caller_floor": {
                 "h2d_bytes": 8 * 4 * n, "d2h_bytes": 8 * (3 * n + 2 * m),
                 "what": "state, D, step, candidate state up; residuals, gradient, diag(J'J), step, model residuals down -- every one "
                         "read or written in HOST memory by TrustRegionMinimizer / LevenbergMarquardtStrategy between the calls"}}
    for label, devices in (("one_shard", None), ("four_logical_shards", [0, 0, 0, 0])):
        c = ctx if devices is None else cx.Context(devices=devices)
        loop = cx.boundary.BoundaryLoop(c, prob, solver_kw, eta=ETA, register_arrays=1)   # its vectors live until close()
        rep = loop.run(iterations)
        loop.close()                         # registrations released, policy back to "never"
        ev = cx.Evaluator(c, prob)
        S = cx.Solver(c, **solver_kw)
        opts = cx.binding.minimizer_options(max_num_iterations=iterations, eta=ETA)
        cx.minimize(ev, S, prob.state(), opts)
        _, msum, its = cx.minimize(ev, S, prob.state(), opts)
        S.close()
        ev.close()
        resident = [it["iteration_ms"] for it in its[1:iterations + 1]]
        per = []
        for k, r in enumerate(rep["per_iteration"]):
            per.append({"through_interfaces_ms": r["through_interfaces_ms"], "resident_ms": resident[k] if k < len(resident) else None,
                        "cg_iterations": r["cg_iterations"], "calls_ms": r["calls_ms"], "caller_numpy_ms": r["caller_ms"],
                        "h2d_bytes": r["h2d_bytes"], "d2h_bytes": r["d2h_bytes"], "h2d_ms": r["h2d_ms"], "d2h_ms": r["d2h_ms"]})
        both = [(p["through_interfaces_ms"], p["resident_ms"]) for p in per if p["resident_ms"]]
        block[label] = {
            "lm_iteration_through_interfaces_ms": rep["lm_iteration_through_interfaces_ms"],
            "resident_lm_iteration_ms": float(np.mean(resident)) if resident else None,
            "boundary_overhead_ms": float(np.mean([a - b for a, b in both])) if both else None,
            "first_iteration_ratio": both[0][0] / both[0][1] if both else None,
            "h2d_bytes": rep["h2d_bytes"], "d2h_bytes": rep["d2h_bytes"], "transfer_ms": rep["transfer_ms"],
            "h2d_GBps": rep["h2d_GBps"], "d2h_GBps": rep["d2h_GBps"], "registered_fraction": rep["registered_fraction"],
            "per_iteration": per, "costs": rep["costs"], "resident_costs": [it["cost"] for it in its[:iterations + 1]],
        }
        if devices is not None:
            c.close()
    return block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="final13682", choices=sorted(load_cx().bal.PRESETS))
    ap.add_argument("--solver", default="iterative_schur", choices=["iterative_schur", "dense_schur", "sparse_schur", "cgnr"],
                    help="linear solver of the step (the headline metric uses iterative_schur)")
    ap.add_argument("--preconditioner", default="jacobi",
                    choices=["jacobi", "schur_jacobi", "identity", "cluster_jacobi", "cluster_tridiagonal"])
    ap.add_argument("--mixed", action="store_true", help="CG products on fp32 copies of the J values: CGNR (BASELINE config 5) or ITERATIVE_SCHUR; "
                                                     "fp64 accumulation, vectors, set-up and back substitution")
    ap.add_argument("--refinements", type=int, default=0,
                    help="max_num_refinement_iterations (solver.h:587-590): with --solver dense_schur / sparse_schur, fp64 refinement steps "
                         "after the reduced solve; with --mixed there the factorisation is single precision (fp32 tile pool)")
    ap.add_argument("--explicit-schur", action="store_true",
                    help="ITERATIVE_SCHUR on the explicitly computed block-sparse S (Solver::Options::"
                         "use_explicit_schur_complement, solver.h:518-540); needs --preconditioner schur_jacobi")
    ap.add_argument("--eta", type=float, default=0.1,
                    help="q_tolerance of the inexact step: Solver::Options::eta default 0.1 (solver.h:628); "
                         "bundle_adjuster's flag default is 1e-2 (bundle_adjuster.cc:114)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-sample-fraction", type=float, default=1.0,
                    help="part of the points the CPU baseline solves (1.0: the whole workload, ~10 s on 16 threads)")
    ap.add_argument("--no-boundary", action="store_true",
                    help="skip the `boundary` block (the LM iteration through the interfaces with host vectors, 1 and 4 shards)")
    ap.add_argument("--boundary-iterations", type=int, default=3)
    ap.add_argument("--no-sparse-schur", action="store_true",
                    help="skip the SPARSE_SCHUR solve + its CPU baseline that the default Final-13682 line carries (north star: "
                         ">= 10x lower linear-solve ms than host SPARSE_SCHUR)")
    args = ap.parse_args()
    global ETA
    ETA = args.eta

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: start the N ranks as fresh child processes (nothing in this process has touched a GPU yet --
        # no torch import, no HIP call) and relay rank 0's JSON line, which the children print to the inherited stdout
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch  # first, so that libcxschur shares torch's HIP runtime instance
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    cx = load_cx()
    if os.environ.get("CX_BENCH_FORCE_DEVICE"):       # rehearsal of the N > 1 path on a box with one GPU
        local_rank = int(os.environ["CX_BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only.  (Gloo announces its connections on STDOUT -- "[Gloo] Rank 0 is connected to 1 peer ranks" -- and
        # stdout is where the ONE JSON line goes: file descriptor 1 points at stderr while the process group forms.)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    ctx = cx.Context(local_rank)
    if world > 1 and os.environ.get("CX_BENCH_TRANSPORT") == "gloo":
        # rehearsal only (several ranks sharing one GPU, where RCCL refuses to form a communicator): the
        # exchange step goes through the library's callback transport and a host-staged gloo all-reduce
        def _allreduce(a):
            dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM)
        ctx.set_comm_callback(rank, world, _allreduce)
    elif world > 1:
        ids = [cx.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.set_comm(rank, world, ids[0])                                     # data plane: RCCL over xGMI

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    t_gen = time.time()
    full = cx.bal.make_preset(args.workload)
    C, P, O = full.num_cameras, full.num_points, full.num_observations
    if world > 1:
        bounds = cx.bal.partition_points(full, world)
        prob = cx.bal.shard(full, int(bounds[rank]), int(bounds[rank + 1]))
    else:
        prob = full
    t_gen = time.time() - t_gen

    ev, A, b, D, cost, eval_ms, values_update = lm_prepare_device(cx, ctx, prob)
    stype = {"iterative_schur": cx.ITERATIVE_SCHUR, "dense_schur": cx.DENSE_SCHUR, "sparse_schur": cx.SPARSE_SCHUR,
             "cgnr": cx.CGNR}[args.solver]
    ptype = {"jacobi": cx.JACOBI, "schur_jacobi": cx.SCHUR_JACOBI, "identity": cx.IDENTITY, "cluster_jacobi": cx.CLUSTER_JACOBI,
             "cluster_tridiagonal": cx.CLUSTER_TRIDIAGONAL}[args.preconditioner]
    solver_kw = dict(type=stype, preconditioner_type=ptype, num_eliminate_blocks=prob.num_points,
                     max_num_iterations=500, min_num_iterations=0, residual_reset_period=10,
                     use_mixed_precision_solves=1 if args.mixed else 0, max_num_refinement_iterations=args.refinements,
                     use_explicit_schur_complement=1 if args.explicit_schur else 0)
    S = cx.Solver(ctx, **solver_kw)
    x = ctx.empty(A.num_cols)

    for _ in range(args.warmup):
        _, summ = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=ETA, x=x)
    barrier()
    t0 = time.perf_counter()
    kstats = {}
    phases = {}
    sampled_steps = 0
    separate_permute = bool(os.environ.get("CX_NO_FT_EMIT"))
    for _ in range(args.steps):
        # Every LM iteration hands the solver freshly evaluated and scaled values.  The kernels that produce them
        # (k_bal_evaluate, k_scale_239) now write the camera-major copy of the F cells as well, so the solve finds
        # it current -- exactly the state lm_prepare_device leaves.  With CX_NO_FT_EMIT=1 (round-1 behaviour, kept
        # for A/B runs) that copy is rebuilt by a separate pass inside every solve, so it is invalidated here.
        if separate_permute:
            A.values_changed()
        _, summ = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=ETA, x=x)
        tm = S.timing()
        if tm.pop("sampled", 1.0) == 0.0:
            # a launch-bound solver takes phase times and kernel samples every 16th solve only (cxschur.h at
            # cx_solver_last_timing): count the solves that took them
            phases["total_ms"] = phases.get("total_ms", 0.0) + tm["total_ms"]
            continue
        sampled_steps += 1
        for k in S.kernel_stats():
            e = kstats.setdefault(k["name"], [0.0, 0, 0])
            e[0] += k["sampled_ms"]
            e[1] += k["sampled_launches"]
            e[2] += k["launches"]
        for n, v in tm.items():
            phases[n] = phases.get(n, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if sampled_steps == 0:
        # none of the timed solves of a launch-bound solver took phase times / kernel samples: one more solve, outside the
        # timed region, that does
        S.sample_next()
        S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=ETA, x=x)
        tm = S.timing()
        tm.pop("sampled", None)
        tm.pop("total_ms", None)
        for k in S.kernel_stats():
            kstats[k["name"]] = [k["sampled_ms"], k["sampled_launches"], k["launches"]]
        for n, v in tm.items():
            phases[n] = v
        sampled_steps = 1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    ms_per_step = elapsed * 1e3 / max(1, args.steps)

    # ---- per-rank view of the timed solves (what a scaling curve is explained with): this rank's share of the
    # residual blocks, its device time in the two S x kernels, and the exchange step -- collectives per solve, their
    # payload, and the DEVICE time inside them (event pairs around every ncclAllReduce on the solve's stream)
    nsteps = max(1, args.steps)
    if sampled_steps != args.steps:   # phase sums cover the sampled solves only: scale them to "per timed solve"
        for k in list(phases):
            if k != "total_ms":
                phases[k] *= nsteps / max(1, sampled_steps)
    mine = {"rank": rank, "residual_blocks": int(prob.num_observations), "points": int(prob.num_points),
            "solve_ms": phases.get("total_ms", 0.0) / nsteps,
            "eliminate_ms": phases.get("eliminate_ms", 0.0) / nsteps, "reduced_solve_ms": phases.get("reduced_solve_ms", 0.0) / nsteps,
            "kernel_avg_ms": {k: v[0] / max(1, v[1]) for k, v in kstats.items()},
            "allreduce_device_ms_per_solve": phases.get("allreduce_ms", 0.0) / nsteps,
            "allreduce_host_ms_per_solve": phases.get("allreduce_host_ms", 0.0) / nsteps,
            "collectives_per_solve": phases.get("allreduce_calls", 0.0) / nsteps,
            "bytes_per_collective": (phases.get("allreduce_bytes", 0.0) / max(1.0, phases.get("allreduce_calls", 0.0)))}
    per_rank = [mine]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered

    # ---- the same solve at bundle_adjuster's tighter default (eta = 1e-2, bundle_adjuster.cc:114): many more
    # CG iterations, so this is the per-iteration cost of S x; informational, outside the timed region
    tight = None
    if args.solver not in ("dense_schur", "sparse_schur") and args.eta != 0.01:
        S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.01, x=x)
        barrier()
        t1 = time.perf_counter()
        _, summ_t = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.01, x=x)
        barrier()
        t_tight = (time.perf_counter() - t1) * 1e3
        tm = S.timing()
        tight = {"eta": 0.01, "ms": t_tight, "cg_iterations": int(summ_t.num_iterations),
                 "cg_ms_per_iteration": tm["reduced_solve_ms"] / max(1, int(summ_t.num_iterations))}

    explicit_info = None
    if args.explicit_schur:
        cr, _ = cx.binding.schur_sparse_structure(A)
        explicit_info = {"cells": int(cr.size), "bytes_per_s_times_x": int(648 * (2 * cr.size - C))}

    # ---- J SpMV GB/s (block_sparse_matrix.cc:239-349 replacement), this rank's shard
    Ol, Pl = prob.num_observations, prob.num_points
    n_c, n_r = 3 * Pl + 9 * C, 2 * Ol
    xv = ctx.to_device(np.random.default_rng(1).standard_normal(n_c))
    yv = ctx.zeros(n_r)
    zv = ctx.to_device(np.random.default_rng(2).standard_normal(n_r))
    cv = ctx.zeros(n_c)
    right_ms, left_ms = [], []
    for i in range(12):
        A.right_multiply(xv, yv)
        right_ms.append(A.last_kernel_ms)
        A.left_multiply(zv, cv)
        left_ms.append(A.last_kernel_ms)
    right_ms, left_ms = float(np.median(right_ms[2:])), float(np.median(left_ms[2:]))
    right_bytes = 240.0 * Ol + 8.0 * n_c       # 8 nnz + 16 O + 8 n_c + 16 n_r
    left_bytes = 224.0 * Ol + 16.0 * n_c       # 8 nnz + 16 O + 8 n_r + 16 n_c
    spmv = {
        "right_gbps": right_bytes / right_ms / 1e6, "left_gbps": left_bytes / left_ms / 1e6,
        "right_ms": right_ms, "left_ms": left_ms,
        "right_frac_of_8TBps": right_bytes / right_ms / 1e6 / HBM_PEAK_GBPS,
        "left_frac_of_8TBps": left_bytes / left_ms / 1e6 / HBM_PEAK_GBPS,
        "scope": "per GPU (this rank's shard)",
    }

    # ---- roofline of the dominant kernel: the chunk pass of S x
    dom = "k_chunk_pass<0>"
    roof = None
    if dom in kstats and kstats[dom][1] > 0:
        avg_ms = kstats[dom][0] / kstats[dom][1]
        # algorithmic bytes per launch: E+F values 192 B, camera id 4 B, t' written 16 B per residual
        # block; (E'E)^-1 72 B per point; x_f 72 B per camera  (kernel table)
        abytes = 212.0 * Ol + 72.0 * Pl + 72.0 * C
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1:   # the PMC passes were collected on the unsharded workload
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get(dom)
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": dom, "achieved": abytes / avg_ms / 1e6, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": abytes / avg_ms / 1e6 / HBM_PEAK_GBPS,
                "frac_of_measured_copy_peak": abytes / avg_ms / 1e6 / HBM_COPY_GBPS,
                "avg_launch_ms": avg_ms, "launches_sampled": kstats[dom][1], "algorithmic_bytes_per_launch": abytes,
                "traffic": traffic,
                "traffic_source": "profiles/traffic.json (PMC passes of a builder run of this command under rocprofv3 --pmc; "
                                  "not measured in this run)" if traffic is not None else None}

    # ---- the north star's own comparison, in the same run: one SPARSE_SCHUR solve of the same system (exact Newton step;
    # tile-sparse level-scheduled Cholesky in CHOLMOD's place) and, on rank 0 below, the oracle's SPARSE_SCHUR beside it
    sparse = None
    if (world == 1 and args.workload == "final13682" and args.solver == "iterative_schur" and args.preconditioner == "jacobi"
            and not args.mixed and not args.explicit_schur and not args.no_sparse_schur):
        SS = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=prob.num_points)
        SS.solve(A, b, D, x=x)           # first solve: structure analysis (pair lists, dissection, plan)
        reps = []
        for _ in range(3):
            barrier()
            t1 = time.perf_counter()
            _, summ_s = SS.solve(A, b, D, x=x)
            barrier()
            reps.append((time.perf_counter() - t1) * 1e3)
        tm = SS.timing()
        sparse = {"ms": float(np.median(reps)), "termination": int(summ_s.termination_type),
                  "phases_ms": {k: tm[k] for k in ("eliminate_ms", "reduced_solve_ms", "back_substitute_ms", "total_ms")}}
        SS.close()
        # ... and with use_mixed_precision_solves (solver.h:572-585): S factored in single precision, no refinement
        SM = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=prob.num_points, use_mixed_precision_solves=1)
        SM.solve(A, b, D, x=x)
        reps = []
        for _ in range(3):
            barrier()
            t1 = time.perf_counter()
            SM.solve(A, b, D, x=x)
            barrier()
            reps.append((time.perf_counter() - t1) * 1e3)
        sparse["single_precision_factor_ms"] = float(np.median(reps))
        SM.close()

    boundary = None
    if (world == 1 and args.workload == "final13682" and args.solver == "iterative_schur" and args.preconditioner == "jacobi"
            and not args.mixed and not args.explicit_schur and not args.no_boundary):
        boundary = boundary_block(cx, ctx, full, solver_kw, args.boundary_iterations)

    out = None
    if rank == 0:
        cpu = None
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        if world == 1 and not args.no_cpu_baseline and args.solver == "iterative_schur" and args.preconditioner == "jacobi":
            cpu = cpu_baseline(cx, full, solver_kw, threads, args.cpu_sample_fraction)
        elif world == 1 and not args.no_cpu_baseline and args.solver == "sparse_schur":
            cpu = cpu_baseline(cx, full, solver_kw, threads, args.cpu_sample_fraction, solver="sparse_schur")
        if sparse is not None and not args.no_cpu_baseline:
            sparse["cpu_baseline"] = cpu_baseline(cx, full, solver_kw, threads, 1.0, solver="sparse_schur")
            sparse["speedup_vs_cpu_baseline"] = sparse["cpu_baseline"]["value"] / sparse["ms"]
        out = {
            "metric": "linear_solve_ms_per_iter", "value": ms_per_step, "unit": "ms",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": ("f64" if not args.mixed else "f64 (S factored in fp32, %d fp64 refinement steps)" % args.refinements if args.solver in ("dense_schur", "sparse_schur") else "f64 (J values stored fp32)"), "data": "synthetic",
            "config": {"workload": "%s: %s%s + %s, q_tol=%g, synthetic BAL-shaped J "
                                   "(%d cameras, %d points, %d residual blocks)" % (args.workload, args.solver.upper(),
                                                                                   " (explicit S)" if args.explicit_schur else "",
                                                                                   args.preconditioner.upper(), ETA, C, P, O),
                       "cameras": C, "points": P, "residual_blocks": O, "cg_iterations": int(summ.num_iterations),
                       "termination": int(summ.termination_type), "initial_cost": cost,
                       "sharding": "points over %d rank(s), RCCL all-reduce of camera-space sums" % world},
            "phases_ms_per_solve": {k: v / max(1, args.steps) for k, v in phases.items()},
            "per_rank": per_rank,
            "load_balance": {"max_over_mean_residual_blocks": max(r["residual_blocks"] for r in per_rank) * len(per_rank) /
                             float(sum(r["residual_blocks"] for r in per_rank)),
                             "max_over_mean_solve_ms": max(r["solve_ms"] for r in per_rank) * len(per_rank) /
                             max(1e-12, float(sum(r["solve_ms"] for r in per_rank)))},
            "exchange": {"transport": "gloo callback (rehearsal)" if os.environ.get("CX_BENCH_TRANSPORT") == "gloo" else
                         ("RCCL ncclAllReduce(sum, fp64) over xGMI" if world > 1 else "none (one rank)"),
                         "collectives_per_solve": per_rank[0]["collectives_per_solve"],
                         "bytes_per_collective": per_rank[0]["bytes_per_collective"],
                         "device_ms_per_solve_max_over_ranks": max(r["allreduce_device_ms_per_solve"] for r in per_rank)},
            "at_bundle_adjuster_eta": tight,
            "explicit_s": explicit_info,
            "jacobian_eval_ms": eval_ms,
            "values_update_per_lm_iteration": values_update,
            # what one LM iteration costs on the device: the evaluation of r and J, the Jacobi scaling of J (both also
            # keep the camera-major copy of F current, work that round 1 did inside the solve) and the timed solve
            "lm_iteration_ms": values_update["jacobian_eval_ms"] + values_update["scale_columns_ms"] + ms_per_step,
            "sparse_schur": sparse,
            "boundary": boundary,
            "spmv": spmv,
            "kernels": {k: {"avg_ms": v[0] / max(1, v[1]), "launches_per_solve": v[2] / max(1, sampled_steps)} for k, v in kstats.items()},
            "roofline": roof,
            "cpu_baseline": cpu,
            "device": ctx.name,
            "problem_generation_s": t_gen,
        }
        print(json.dumps(out), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
