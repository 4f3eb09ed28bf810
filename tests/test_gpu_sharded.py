"""The sharded (N > 1) path of the HIP library with two ranks sharing the one GPU of the test
box.  RCCL refuses two ranks on one device, so the exchange step uses the library's callback
transport with a host-staged gloo all-reduce; everything else (sharded solvers, evaluator,
per-rank kernels) is the production code.  Each rank must reproduce its slice of the unsharded
solve."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, results_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest
    cx, orc = conftest.cx, conftest.orc
    orc.lib()
    orc.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)

    def allreduce(a):
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    ctx = cx.Context(0)
    ctx.set_comm_callback(rank, WORLD, allreduce)
    full = cx.bal.make_bal_like(14, 900, 4200, seed=6)
    C, P = full.num_cameras, full.num_points
    bs_full, order_full = cx.bal.build_structure(full)
    bounds = cx.bal.partition_points(full, WORLD)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    sub = cx.bal.shard(full, lo, hi)
    # reference: unsharded oracle
    cost_f, res_f, grad_f, vals_f = orc.bal_evaluate(bs_full, C, P, full.camera_index, full.point_index,
                                                     full.observations, order_full, full.state())
    rng = np.random.default_rng(3)
    D_full = rng.uniform(0.5, 2.0, 3 * P + 9 * C) * 1e-2 * np.sqrt(np.abs(vals_f).mean())
    D = np.concatenate([D_full[3 * lo:3 * hi], D_full[3 * P:]])
    # sharded device evaluation: cost and camera gradient are summed over the ranks
    ev = cx.Evaluator(ctx, sub)
    cost, res, grad = ev.evaluate(sub.state())
    out = []
    out.append(("cost", abs(cost - cost_f) <= 1e-11 * cost_f))
    out.append(("grad_cam", float(np.abs(grad[3 * sub.num_points:] - grad_f[3 * P:]).max() / np.abs(grad_f).max()) < 1e-10))
    A = ev.jacobian()
    for stype, pre, explicit in (("ITERATIVE_SCHUR", "JACOBI", 0), ("ITERATIVE_SCHUR", "SCHUR_JACOBI", 0),
                                 ("ITERATIVE_SCHUR", "SCHUR_POWER_SERIES_EXPANSION", 0), ("CGNR", "JACOBI", 0),
                                 ("DENSE_SCHUR", "IDENTITY", 0), ("ITERATIVE_SCHUR", "SCHUR_JACOBI", 1),
                                 ("ITERATIVE_SCHUR", "CLUSTER_JACOBI", 0), ("ITERATIVE_SCHUR", "CLUSTER_TRIDIAGONAL", 0)):
        o_full = orc.make_options(type=getattr(orc, stype), preconditioner_type=getattr(orc, pre),
                                  num_eliminate_blocks=P, max_num_iterations=300, use_explicit_schur_complement=explicit)
        x_full, s_full = orc.solve(bs_full, vals_f, res_f, D_full, o_full, r_tolerance=-1.0, q_tolerance=0.1)
        S = cx.Solver(ctx, type=getattr(cx, stype), preconditioner_type=getattr(cx, pre),
                      num_eliminate_blocks=sub.num_points, max_num_iterations=300, use_explicit_schur_complement=explicit)
        x, s = S.solve(A, res, D, r_tolerance=-1.0, q_tolerance=0.1)
        expect = np.concatenate([x_full[3 * lo:3 * hi], x_full[3 * P:]])
        err = float(np.abs(x - expect).max() / np.abs(expect).max())
        out.append((stype + "+" + pre + ("+explicit" if explicit else ""), s.termination_type == s_full.termination_type and
                    s.num_iterations == s_full.num_iterations and err < 1e-8, err, s.num_iterations, s_full.num_iterations))
        if (stype, pre, explicit) == ("ITERATIVE_SCHUR", "JACOBI", 0):
            # what the first real multi-GPU run will be analysed with: collectives per solve, their payload and the time
            # in them.  One fused set-up collective (rhs + the 45 distinct values of every 9x9 block = 54 C doubles), one
            # of 9 C doubles per S x (an extra one in every residual_reset_period-th iteration).
            tm = S.timing()
            it = s.num_iterations
            calls = 2 + it + it // 10      # + the one-word health agreement at the start of the solve
            ok = (tm["allreduce_calls"] == calls and tm["allreduce_bytes"] == 8.0 * (1 + 54 * C + 9 * C * (calls - 2)) and
                  tm["allreduce_host_ms"] > 0.0 and tm["allreduce_ms"] == 0.0)   # callback transport: host time only
            out.append(("exchange_stats", ok, tm["allreduce_calls"], tm["allreduce_bytes"], tm["allreduce_host_ms"]))
        S.close()
    # sharded trust-region loop: every rank must walk the same iterations as the unsharded oracle loop
    mo = cx.binding.minimizer_options(max_num_iterations=6)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=sub.num_points)
    x_min, summ, its = cx.binding.minimize(ev, S, sub.state(), mo)
    so = orc.make_options(type=orc.ITERATIVE_SCHUR, preconditioner_type=orc.JACOBI, num_eliminate_blocks=P)
    x_ref, summ_r, its_r = orc.minimize_bal(C, P, full.camera_index, full.point_index, full.observations, full.state(),
                                            so, orc.minimizer_options(max_num_iterations=6))
    expect = np.concatenate([x_ref[3 * lo:3 * hi], x_ref[3 * P:]])
    same_path = len(its) == len(its_r) and all(
        a["step_is_successful"] == b["step_is_successful"] and abs(a["cost"] - b["cost"]) <= 1e-5 * b["cost"] and
        abs(a["gradient_max_norm"] - b["gradient_max_norm"]) <= 1e-3 * its_r[0]["gradient_max_norm"] and
        abs(a["step_norm"] - b["step_norm"]) <= 1e-4 * max(1.0, b["step_norm"])
        for a, b in zip(its, its_r))
    err = float(np.abs(x_min - expect).max() / np.abs(expect).max())
    out.append(("minimize", same_path and summ["termination_type"] == summ_r["termination_type"] and err < 1e-4, err,
                len(its), len(its_r)))
    S.close()
    # SPARSE_SCHUR on several ranks: the tile-sparse factorisation on the union of the ranks' S cells (plan from a dense
    # presence exchange, cell values summed over the ranks, factorisation replicated).  700 cameras take that path by
    # themselves (>= 512); the 14-camera problem is forced onto it.
    mid = cx.bal.make_bal_like(700, 5000, 26000, seed=8)
    Cm, Pm = mid.num_cameras, mid.num_points
    bs_mid, order_mid = cx.bal.build_structure(mid)
    bm = cx.bal.partition_points(mid, WORLD)
    lo_m, hi_m = int(bm[rank]), int(bm[rank + 1])
    mid_sub = cx.bal.shard(mid, lo_m, hi_m)
    _, res_m, _, vals_m = orc.bal_evaluate(bs_mid, Cm, Pm, mid.camera_index, mid.point_index, mid.observations, order_mid,
                                           mid.state())
    D_mid = np.random.default_rng(4).uniform(0.5, 2.0, 3 * Pm + 9 * Cm) * 1e-2 * np.sqrt(np.abs(vals_m).mean())
    o_mid = orc.make_options(type=orc.SPARSE_SCHUR, num_eliminate_blocks=Pm)
    x_mid, s_mid = orc.solve(bs_mid, vals_m, res_m, D_mid, o_mid, r_tolerance=-1.0, q_tolerance=0.1)
    ev_m = cx.Evaluator(ctx, mid_sub)
    _, res_sub, _ = ev_m.evaluate(mid_sub.state())
    Sm = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=mid_sub.num_points)
    xm, sm = Sm.solve(ev_m.jacobian(), res_sub, np.concatenate([D_mid[3 * lo_m:3 * hi_m], D_mid[3 * Pm:]]),
                      r_tolerance=-1.0, q_tolerance=0.1)
    expect = np.concatenate([x_mid[3 * lo_m:3 * hi_m], x_mid[3 * Pm:]])
    err = float(np.abs(xm - expect).max() / np.abs(expect).max())
    first_calls = Sm.timing()["allreduce_calls"]      # agreement + structure exchange, then per solve: right-hand side, cell values, the replicated tiles at the split, the solution
    xm2, _ = Sm.solve(ev_m.jacobian(), res_sub, np.concatenate([D_mid[3 * lo_m:3 * hi_m], D_mid[3 * Pm:]]),
                      r_tolerance=-1.0, q_tolerance=0.1)
    tm = Sm.timing()
    out.append(("sharded_sparse_schur_700", sm.termination_type == s_mid.termination_type and err < 1e-8 and
                first_calls == 7 and tm["allreduce_calls"] == 5 and np.array_equal(xm, xm2), err, first_calls,
                tm["allreduce_calls"], tm["allreduce_bytes"]))
    Sm.close()
    ev_m.close()
    os.environ["CX_SPARSE_CHOLESKY"] = "1"
    o_full = orc.make_options(type=orc.SPARSE_SCHUR, num_eliminate_blocks=P)
    x_full, s_full = orc.solve(bs_full, vals_f, res_f, D_full, o_full, r_tolerance=-1.0, q_tolerance=0.1)
    Sf = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=sub.num_points)
    _, res, _ = ev.evaluate(sub.state())   # (the minimisation above left the Jacobian of its last state in A)
    xs, ss = Sf.solve(A, res, D, r_tolerance=-1.0, q_tolerance=0.1)
    del os.environ["CX_SPARSE_CHOLESKY"]
    expect = np.concatenate([x_full[3 * lo:3 * hi], x_full[3 * P:]])
    err = float(np.abs(xs - expect).max() / np.abs(expect).max())
    out.append(("sharded_sparse_schur_forced", ss.termination_type == s_full.termination_type and err < 1e-8, err))
    Sf.close()
    # DENSE_SCHUR with 6 200 cameras would be a 25 GB matrix per rank, all-reduced: refused with a message, not attempted
    big = cx.bal.make_bal_like(6200, 7000, 21000, seed=9)
    bb = cx.bal.partition_points(big, WORLD)
    big_sub = cx.bal.shard(big, int(bb[rank]), int(bb[rank + 1]))
    bs_big, _ = cx.bal.build_structure(big_sub)
    Ab = cx.Matrix(ctx, bs_big, big_sub.num_points)
    Ab.set_values(cx.bal.random_jacobian_values(big_sub.num_observations, 1))
    Sb = cx.Solver(ctx, type=cx.DENSE_SCHUR, num_eliminate_blocks=big_sub.num_points)
    try:
        Sb.solve(Ab, np.ones(Ab.num_rows), np.ones(Ab.num_cols))
        out.append(("sharded_dense_schur_refused", False, "no error"))
    except cx.binding.CxError as e:
        out.append(("sharded_dense_schur_refused", "does not fit" in str(e), str(e)[:80]))
    Sb.close()
    Ab.close()
    with open(os.path.join(results_dir, "rank%d.txt" % rank), "w") as f:
        for o in out:
            f.write(repr(o) + "\n")
    dist.barrier()
    ev.close()
    ctx.close()
    dist.destroy_process_group()


def test_two_ranks_one_gpu(tmp_path, oracle):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    for rank in range(WORLD):
        lines = open(tmp_path / ("rank%d.txt" % rank)).read().strip().splitlines()
        assert len(lines) == 15
        for line in lines:
            rec = eval(line)
            assert rec[1], line
