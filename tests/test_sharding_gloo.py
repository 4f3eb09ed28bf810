"""N > 1 path on CPU: world size 2 over gloo.  Each rank takes a contiguous range of points
(cx_partition_points / bal.partition_points), runs the oracle's sharded solve with a gloo
all-reduce for the camera-space sums -- the exchange step the GPU library performs with
RCCL -- and must reproduce the unsharded solve.  Also checks the shard bookkeeping the
bench uses (bal.shard + build_structure)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

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

    def allreduce(a):  # a: numpy view of the oracle's buffer, summed in place
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    full = cx.bal.make_bal_like(14, 400, 1900, seed=6)
    C, P = full.num_cameras, full.num_points
    bs_full, order_full = cx.bal.build_structure(full)
    # host helper and C ABI helper agree on the partition
    bounds = cx.bal.partition_points(full, WORLD)
    assert np.array_equal(bounds, cx.partition_points(bs_full, P, WORLD))
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    sub = cx.bal.shard(full, lo, hi)
    bs, order = cx.bal.build_structure(sub)
    # a shard's rows are exactly the full matrix' rows of its points, in the same order
    rows_full = np.flatnonzero((bs_full.cells["block_id"][0::2] >= lo) & (bs_full.cells["block_id"][0::2] < hi))
    assert np.array_equal(full.camera_index[order_full][rows_full], sub.camera_index[order])

    cost_f, res_f, grad_f, vals_f = orc.bal_evaluate(bs_full, C, P, full.camera_index, full.point_index,
                                                     full.observations, order_full, full.state())
    cost_s, res_s, grad_s, vals_s = orc.bal_evaluate(bs, C, sub.num_points, sub.camera_index, sub.point_index,
                                                     sub.observations, order, sub.state())
    # evaluation is local: values and residuals are slices of the full ones
    assert np.array_equal(res_s, res_f.reshape(-1, 2)[rows_full].ravel())
    assert np.array_equal(vals_s[:6 * rows_full.size], vals_f[:6 * full.num_observations].reshape(-1, 6)[rows_full].ravel())
    # cost and the camera part of the gradient are sums over the ranks
    c = np.array([cost_s])
    allreduce(c)
    assert abs(c[0] - cost_f) <= 1e-12 * cost_f
    gc = grad_s[3 * sub.num_points:].copy()
    allreduce(gc)
    np.testing.assert_allclose(gc, grad_f[3 * P:], rtol=1e-11, atol=1e-9)

    rng = np.random.default_rng(3)
    D_full = rng.uniform(0.5, 2.0, 3 * P + 9 * C) * 1e-2 * np.sqrt(np.abs(vals_f).mean())
    D = np.concatenate([D_full[3 * lo:3 * hi], D_full[3 * P:]])
    outcomes = []
    for stype, pre, explicit in (("ITERATIVE_SCHUR", "JACOBI", 0), ("ITERATIVE_SCHUR", "SCHUR_JACOBI", 0), ("CGNR", "JACOBI", 0),
                                 ("DENSE_SCHUR", "IDENTITY", 0), ("ITERATIVE_SCHUR", "SCHUR_JACOBI", 1),
                                 ("ITERATIVE_SCHUR", "CLUSTER_JACOBI", 0), ("ITERATIVE_SCHUR", "CLUSTER_TRIDIAGONAL", 0)):
        o_full = orc.make_options(type=getattr(orc, stype), preconditioner_type=getattr(orc, pre),
                                  num_eliminate_blocks=P, max_num_iterations=300, use_explicit_schur_complement=explicit)
        x_full, s_full = orc.solve(bs_full, vals_f, res_f, D_full, o_full, r_tolerance=-1.0, q_tolerance=0.1)
        o = orc.make_options(type=getattr(orc, stype), preconditioner_type=getattr(orc, pre),
                             num_eliminate_blocks=sub.num_points, max_num_iterations=300,
                             use_explicit_schur_complement=explicit)
        x, s = orc.solve(bs, vals_s, res_s, D, o, r_tolerance=-1.0, q_tolerance=0.1, allreduce=allreduce)
        expect = np.concatenate([x_full[3 * lo:3 * hi], x_full[3 * P:]])
        err = np.abs(x - expect).max() / np.abs(expect).max()
        outcomes.append((stype, pre, s.termination_type == s_full.termination_type,
                         s.num_iterations == s_full.num_iterations, float(err)))
    with open(os.path.join(results_dir, "rank%d.txt" % rank), "w") as f:
        for o in outcomes:
            f.write(repr(o) + "\n")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_solves_match_unsharded(tmp_path, oracle):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    for rank in range(WORLD):
        lines = open(tmp_path / ("rank%d.txt" % rank)).read().strip().splitlines()
        assert len(lines) == 7
        for line in lines:
            stype, pre, same_term, same_iters, err = eval(line)
            assert same_term and same_iters, line
            assert err < 1e-8, line
