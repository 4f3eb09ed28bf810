"""Several shards behind ONE set of handles in one process (cx_context_create_multi, csrc/cx_multi.hip): what a
Solver::Solve caller sees -- whole vectors in, whole vectors out -- while the per-rank sharded solvers run underneath.
The test box has one GPU, so the shards are logical shards on device 0 with the in-process exchange step; everything
else (partitioning, scatter / gather, the sharded solvers, evaluator and trust-region loop of every shard) is the
production code.  Checked against the UNSHARDED oracle, like the single-device parity tests."""
import numpy as np
import pytest

from conftest import cx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    prob = cx.bal.make_bal_like(14, 900, 4200, seed=6)
    bs, order = cx.bal.build_structure(prob)
    return prob, bs, order


def _oracle_system(orc, prob, bs, order):
    C, P = prob.num_cameras, prob.num_points
    cost, res, grad, vals = orc.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations, order,
                                             prob.state())
    D = np.random.default_rng(3).uniform(0.5, 2.0, 3 * P + 9 * C) * 1e-2 * np.sqrt(np.abs(vals).mean())
    return cost, res, grad, vals, D


@pytest.mark.parametrize("shards", [1, 2, 3, 4])
def test_evaluator_and_products_through_the_front(small, oracle, shards):
    prob, bs, order = small
    orc = oracle
    cost_r, res_r, grad_r, vals_r, _ = _oracle_system(orc, prob, bs, order)
    ctx = cx.Context(devices=[0] * shards)
    assert ctx.num_shards == shards
    ev = cx.Evaluator(ctx, prob)
    # integer structure: the front's row order is the reference's residual-block order
    assert np.array_equal(ev.row_of_observation(), np.argsort(order))
    cost, res, grad = ev.evaluate(prob.state())
    A = ev.jacobian(bs)
    vals = A.get_values()
    assert abs(cost - cost_r) <= 1e-11 * cost_r
    assert np.abs(res - res_r).max() <= 1e-11 * np.abs(res_r).max()
    assert np.abs(vals - vals_r).max() <= 1e-11 * np.abs(vals_r).max()
    assert np.abs(grad - grad_r).max() <= 1e-10 * np.abs(grad_r).max()
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(A.num_cols), rng.standard_normal(A.num_rows)
    y0, x0 = rng.standard_normal(A.num_rows), rng.standard_normal(A.num_cols)
    assert np.abs(A.right_multiply(x, y0) - (y0 + orc.right_multiply(bs, vals, x))).max() <= 1e-12 * np.abs(y0).max() * 1e3
    jt = orc.left_multiply(bs, vals, y)
    assert np.abs(A.left_multiply(y, x0) - (x0 + jt)).max() <= 1e-12 * np.abs(jt).max()
    sq = orc.squared_column_norm(bs, vals)
    assert np.abs(A.squared_column_norm() - sq).max() <= 1e-12 * sq.max()
    scale = 1.0 / (1.0 + np.sqrt(sq))
    A.scale_columns(scale)
    assert np.abs(A.get_values() - orc.scale_columns(bs, vals, scale)).max() <= 1e-13 * np.abs(vals).max()
    # a host BlockSparseMatrix handed to the front (cx_matrix_create + set_values): the value array is cut and
    # reassembled in the reference layout
    B = cx.Matrix(ctx, bs, prob.num_points)
    B.set_values(vals_r)
    assert np.array_equal(B.get_values(), vals_r)
    assert np.abs(B.right_multiply(x) - orc.right_multiply(bs, vals_r, x)).max() <= 1e-12 * np.abs(vals_r).max() * 1e3
    B.close()
    ev.close()
    ctx.close()


SOLVERS = [("ITERATIVE_SCHUR", "JACOBI", 0), ("ITERATIVE_SCHUR", "SCHUR_JACOBI", 0),
           ("ITERATIVE_SCHUR", "SCHUR_POWER_SERIES_EXPANSION", 0), ("CGNR", "JACOBI", 0), ("DENSE_SCHUR", "IDENTITY", 0),
           ("SPARSE_SCHUR", "IDENTITY", 0), ("ITERATIVE_SCHUR", "SCHUR_JACOBI", 1), ("ITERATIVE_SCHUR", "CLUSTER_JACOBI", 0),
           ("ITERATIVE_SCHUR", "CLUSTER_TRIDIAGONAL", 0)]


@pytest.mark.parametrize("shards", [2, 4])
def test_solvers_through_the_front_match_the_unsharded_oracle(small, oracle, shards):
    prob, bs, order = small
    orc = oracle
    C, P = prob.num_cameras, prob.num_points
    _, res_r, _, vals_r, D = _oracle_system(orc, prob, bs, order)
    ctx = cx.Context(devices=[0] * shards)
    ev = cx.Evaluator(ctx, prob)
    _, res, _ = ev.evaluate(prob.state())
    A = ev.jacobian(bs)
    for stype, pre, explicit in SOLVERS:
        o = orc.make_options(type=getattr(orc, stype), preconditioner_type=getattr(orc, pre), num_eliminate_blocks=P,
                             max_num_iterations=300, use_explicit_schur_complement=explicit)
        x_r, s_r = orc.solve(bs, vals_r, res_r, D, o, r_tolerance=-1.0, q_tolerance=0.1)
        S = cx.Solver(ctx, type=getattr(cx, stype), preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P,
                      max_num_iterations=300, use_explicit_schur_complement=explicit)
        x, s = S.solve(A, res, D, r_tolerance=-1.0, q_tolerance=0.1)
        assert s.termination_type == s_r.termination_type and s.num_iterations == s_r.num_iterations, (stype, pre, s.message)
        assert np.all(np.isfinite(x))
        assert np.abs(x - x_r).max() <= 1e-8 * np.abs(x_r).max(), (stype, pre)
        if (stype, pre) == ("ITERATIVE_SCHUR", "JACOBI"):
            # the exchange step as DESIGN.md section 5 predicts: the one-word health agreement at the start of the solve, one
            # fused set-up collective of 54 C doubles, then 9 C doubles per S x (one more in every residual_reset_period-th
            # iteration)
            tm, it = S.timing(), s.num_iterations
            calls = 2 + it + it // 10
            assert tm["allreduce_calls"] == calls and tm["allreduce_bytes"] == 8.0 * (1 + 54 * C + 9 * C * (calls - 2))
            x2, _ = S.solve(A, res, D, r_tolerance=-1.0, q_tolerance=0.1)
            assert np.array_equal(x, x2)                       # in-process sum in rank order: repeatable to the bit
        S.close()
    ev.close()
    ctx.close()


@pytest.mark.parametrize("shards", [2, 4])
def test_minimizer_through_the_front(small, oracle, shards):
    """cx_minimize on fronts: every shard runs the device-resident trust-region loop on its points, the front scatters
    the state and gathers the minimum-cost iterate; same iterations, flags and radii as the unsharded oracle loop."""
    prob, bs, order = small
    orc = oracle
    C, P = prob.num_cameras, prob.num_points
    ctx = cx.Context(devices=[0] * shards)
    ev = cx.Evaluator(ctx, prob)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=P)
    x, summ, its = cx.binding.minimize(ev, S, prob.state(), cx.binding.minimizer_options(max_num_iterations=6))
    so = orc.make_options(type=orc.ITERATIVE_SCHUR, preconditioner_type=orc.JACOBI, num_eliminate_blocks=P)
    x_r, summ_r, its_r = orc.minimize_bal(C, P, prob.camera_index, prob.point_index, prob.observations, prob.state(), so,
                                          orc.minimizer_options(max_num_iterations=6))
    assert len(its) == len(its_r) and summ["termination_type"] == summ_r["termination_type"]
    for a, b in zip(its, its_r):
        assert a["step_is_successful"] == b["step_is_successful"]
        assert abs(a["cost"] - b["cost"]) <= 1e-5 * b["cost"]
        assert abs(a["trust_region_radius"] - b["trust_region_radius"]) <= 1e-9 * b["trust_region_radius"]
    assert np.abs(x - x_r).max() <= 1e-4 * np.abs(x_r).max()
    S.close()
    ev.close()
    ctx.close()


def test_sparse_schur_700_cameras_through_the_front(oracle):
    """SPARSE_SCHUR on several shards: the tile-sparse factorisation planned from the union of the shards' S cells."""
    orc = oracle
    mid = cx.bal.make_bal_like(700, 5000, 26000, seed=8)
    C, P = mid.num_cameras, mid.num_points
    bs, order = cx.bal.build_structure(mid)
    _, res_r, _, vals_r = orc.bal_evaluate(bs, C, P, mid.camera_index, mid.point_index, mid.observations, order, mid.state())
    D = np.random.default_rng(4).uniform(0.5, 2.0, 3 * P + 9 * C) * 1e-2 * np.sqrt(np.abs(vals_r).mean())
    x_r, s_r = orc.solve(bs, vals_r, res_r, D, orc.make_options(type=orc.SPARSE_SCHUR, num_eliminate_blocks=P))
    ctx = cx.Context(devices=[0, 0, 0])
    ev = cx.Evaluator(ctx, mid)
    _, res, _ = ev.evaluate(mid.state())
    S = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P)
    x, s = S.solve(ev.jacobian(bs), res, D)
    assert s.termination_type == s_r.termination_type
    assert np.abs(x - x_r).max() <= 1e-8 * np.abs(x_r).max()
    x_again, _ = S.solve(ev.jacobian(bs), res, D)
    plain_calls = S.timing()["allreduce_calls"]       # per solve on the distributed plan (the first solve also exchanged the structure)
    assert np.array_equal(x, x_again)
    # use_mixed_precision_solves on shards: the float tile pool under the distributed factorisation (the exchange of the
    # replicated tiles converts to double and back); with refinement steps the factor is kept whole on every shard (the plan is
    # rebuilt: stored-factor sweeps need it) and the step comes back to double precision
    S32 = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P, use_mixed_precision_solves=1)
    x32, s32 = S32.solve(ev.jacobian(bs), res, D)
    e32 = np.abs(x32 - x_r).max() / np.abs(x_r).max()
    assert s32.termination_type == cx.SUCCESS and 1e-10 < e32 < 1e-3, e32
    S32r = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P, use_mixed_precision_solves=1, max_num_refinement_iterations=3)
    x32r, s32r = S32r.solve(ev.jacobian(bs), res, D)
    assert s32r.termination_type == cx.SUCCESS and np.abs(x32r - x_r).max() <= max(1e-8, 1e-3 * e32) * np.abs(x_r).max()
    refined_calls = S32r.timing()["allreduce_calls"]
    # round 4: refinement no longer forces the factor whole onto every shard -- the stored-factor sweeps run on the distributed
    # factor (per step: the implicit product's sum, the replicated rows' incoming partial sums at the split, the solution)
    assert refined_calls == plain_calls + 3 * 3, (refined_calls, plain_calls)
    x2, s2 = S.solve(ev.jacobian(bs), res, D)
    assert S.timing()["allreduce_calls"] == plain_calls and np.array_equal(x2, x_again)
    # the fp64 factor refined on shards: the step of the unrefined solve to rounding
    Sr = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P, max_num_refinement_iterations=2)
    xr, sr = Sr.solve(ev.jacobian(bs), res, D)
    assert sr.termination_type == cx.SUCCESS and np.abs(xr - x_r).max() <= 1e-8 * np.abs(x_r).max()
    xr2, _ = Sr.solve(ev.jacobian(bs), res, D)
    assert np.array_equal(xr, xr2)
    Sr.close()
    S32.close()
    S32r.close()
    S.close()
    ev.close()
    ctx.close()


def test_tile_sparse_visibility_preconditioner_on_shards(oracle):
    """with CX_VISIBILITY_SPARSE=1 the tile-sparse plan of a visibility preconditioner on a sharded
    matrix has to come from the block pairs of ALL shards -- a plan from a shard's own cells differs from shard to
    shard while the tile pool is summed entry by entry.  Iteration counts equal to the unsharded oracle's."""
    import os
    orc = oracle
    prob = cx.bal.make_bal_like(60, 1500, 7000, seed=12)
    C, P = prob.num_cameras, prob.num_points
    bs, order = cx.bal.build_structure(prob)
    _, res_r, _, vals_r = orc.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations, order, prob.state())
    D = np.random.default_rng(5).uniform(0.5, 2.0, 3 * P + 9 * C) * 1e-2 * np.sqrt(np.abs(vals_r).mean())
    os.environ["CX_VISIBILITY_SPARSE"] = "1"
    try:
        for pre in ("CLUSTER_JACOBI", "CLUSTER_TRIDIAGONAL"):
            o = orc.make_options(type=orc.ITERATIVE_SCHUR, preconditioner_type=getattr(orc, pre), num_eliminate_blocks=P,
                                 max_num_iterations=300)
            x_r, s_r = orc.solve(bs, vals_r, res_r, D, o, r_tolerance=-1.0, q_tolerance=1e-3)
            for shards in (2, 3):
                ctx = cx.Context(devices=[0] * shards)
                ev = cx.Evaluator(ctx, prob)
                _, res, _ = ev.evaluate(prob.state())
                S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P,
                              max_num_iterations=300)
                x, s = S.solve(ev.jacobian(bs), res, D, r_tolerance=-1.0, q_tolerance=1e-3)
                assert s.termination_type == s_r.termination_type and s.num_iterations == s_r.num_iterations, (pre, shards, s.message)
                assert np.abs(x - x_r).max() <= 1e-8 * np.abs(x_r).max()
                S.close()
                ev.close()
                ctx.close()
    finally:
        del os.environ["CX_VISIBILITY_SPARSE"]


def test_front_refuses_what_it_cannot_shard(small):
    prob, bs, order = small
    ctx = cx.Context(devices=[0, 0])
    with pytest.raises(cx.binding.CxError, match="num_eliminate_blocks must be positive"):
        cx.Matrix(ctx, bs, 0)
    ev = cx.Evaluator(ctx, prob)
    dev = ctx.to_device(prob.state())
    with pytest.raises(cx.binding.CxError, match="host vectors"):
        ev.evaluate(dev, residuals=ctx.empty(2 * prob.num_observations), gradient=None)
    # more shards than points hold observations for
    tiny = cx.bal.make_bal_like(3, 4, 9, seed=1)
    ctx8 = cx.Context(devices=[0] * 8)
    with pytest.raises(cx.binding.CxError, match="fewer shards"):
        cx.Evaluator(ctx8, tiny)
    ctx8.close()
    ev.close()
    ctx.close()


@pytest.mark.parametrize("shards,solver", [(2, "ITERATIVE_SCHUR"), (4, "ITERATIVE_SCHUR"), (3, "SPARSE_SCHUR"), (4, "CGNR")])
def test_a_shard_failing_mid_solve_releases_the_others(small, shards, solver):
    """VERDICT r3 item 2 / ADVICE medium: one logical shard drops out right before its k-th collective (cx_debug_inject_failure)
    while the others are blocked in the exchange step (the in-process transport blocks in a rendezvous).  Every shard must
    come back -- within seconds, with the failing shard's error -- and the context must be usable for the next solve, which
    gives the bits of an undisturbed one."""
    import time
    prob, bs, order = small
    P = prob.num_points
    ctx = cx.Context(devices=[0] * shards)
    ev = cx.Evaluator(ctx, prob)
    _, res, _ = ev.evaluate(prob.state())
    A = ev.jacobian(bs)
    D = np.full(A.num_cols, 1e-2)
    S = cx.Solver(ctx, type=getattr(cx, solver), preconditioner_type=cx.JACOBI, num_eliminate_blocks=P if solver != "CGNR" else 0,
                  max_num_iterations=200)
    x0, s0 = S.solve(A, res, D, r_tolerance=-1.0, q_tolerance=1e-3)
    assert s0.termination_type == cx.SUCCESS
    collectives = int(S.timing()["allreduce_calls"])
    assert collectives >= 2
    for victim, nth in ((shards - 1, 0), (0, 1), (shards // 2, collectives // 2), (shards - 1, collectives - 1)):
        ctx.inject_failure(victim, nth)
        t0 = time.perf_counter()
        with pytest.raises(cx.CxError) as err:
            S.solve(A, res, D, r_tolerance=-1.0, q_tolerance=1e-3)
        assert time.perf_counter() - t0 < 5.0
        assert "shard %d" % victim in str(err.value) and "injected failure" in str(err.value), str(err.value)
        ctx.inject_failure(victim, -1)
        x1, s1 = S.solve(A, res, D, r_tolerance=-1.0, q_tolerance=1e-3)
        assert s1.termination_type == cx.SUCCESS and s1.num_iterations == s0.num_iterations
        assert np.array_equal(x1, x0), (victim, nth)
    # the evaluator's sums (cost, camera part of the gradient) go through the same exchange step
    ctx.inject_failure(0, 0)
    with pytest.raises(cx.CxError):
        ev.evaluate(prob.state())
    ctx.inject_failure(0, -1)
    cost, _, _ = ev.evaluate(prob.state())
    assert np.isfinite(cost)
    S.close()
    ev.close()
    ctx.close()


def test_rccl_wait_has_a_deadline_and_aborts_the_communicator():
    """The waits of a context with an RCCL communicator poll with a deadline: a stream that makes no progress (here: a host
    callback that sleeps; in production: a collective whose peer never arrives) turns into CX_ERR_COMM after the deadline,
    the communicator is aborted (ncclCommAbort) and later sharded work is refused -- never a hang.  One GPU: the
    communicator has one rank, formed by real RCCL; `nranks` > 1 cannot be formed on this box, so the bounded wait is
    reached through cx_synchronize on the communicator's context with the rank count the library reports."""
    import time
    ctx = cx.Context(0)
    ctx.set_comm(0, 1, cx.Context.unique_id())      # real ncclCommInitRank, one rank
    d = ctx.to_device(np.arange(8.0))
    ctx.allreduce_sum(d)
    ctx.synchronize()
    assert np.array_equal(d.to_host(), np.arange(8.0))
    cx.binding._check(ctx.lib.cx_debug_force_rank_count(ctx._h, 2))   # from here on the waits are the sharded ones
    ctx.set_comm_timeout(0.3)
    ctx.stall_stream(1500)
    t0 = time.perf_counter()
    with pytest.raises(cx.CxError) as err:
        ctx.synchronize()
    waited = time.perf_counter() - t0
    # (the deadline fired at 0.3 s -- otherwise the wait would have SUCCEEDED when the callback ended; after aborting the
    # communicator the library gives the stream a bounded time to drain, which here is the rest of the 1.5 s sleep)
    assert 0.25 < waited < 3.0, waited
    assert "communicator aborted" in str(err.value)
    with pytest.raises(cx.CxError) as err2:
        ctx.allreduce_sum(d)
    assert "aborted" in str(err2.value)
    ctx.close()
