"""BASELINE.json's full size (Final-13682 shape: 13 682 cameras, 4.46 M points, 29.0 M residual blocks), where no
CPU oracle finishes in test time: size-independent properties of the domain instead -- adjointness and linearity
of the J products, a checksum of the evaluator, the normal equations satisfied by a converged solve, and two
independent implementations of S x (implicit chunk / camera passes vs explicitly assembled block-sparse S)
driving CG to the same solution."""
import numpy as np
import pytest

from conftest import cx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def final():
    ctx = cx.Context(0)
    prob = cx.bal.make_preset("final13682")
    ev = cx.Evaluator(ctx, prob)
    state = ctx.to_device(prob.state())
    res = ctx.empty(2 * prob.num_observations)
    cost, _, _ = ev.evaluate(state, residuals=res, gradient=None, want_jacobian=True)
    yield ctx, prob, ev, ev.jacobian(), res, cost
    ev.close()
    ctx.close()


def test_evaluator_checksum(final):
    ctx, prob, ev, A, res, cost = final
    r = res.to_host()
    assert np.all(np.isfinite(r))
    assert abs(cost - 0.5 * float(r @ r)) <= 1e-12 * cost            # cost is the checksum of the residuals
    cost2, _, _ = ev.evaluate(ctx.to_device(prob.state()), residuals=None, gradient=None, want_jacobian=False)
    assert abs(cost2 - cost) <= 1e-12 * cost                          # value-only kernel agrees with the Jet kernel


def test_products_adjoint_and_linear(final):
    ctx, prob, ev, A, res, cost = final
    rng = np.random.default_rng(0)
    n_c, n_r = A.num_cols, A.num_rows
    x1, x2 = rng.standard_normal(n_c), rng.standard_normal(n_c)
    y = rng.standard_normal(n_r)
    jx1 = A.right_multiply(x1)
    jty = A.left_multiply(y)
    lhs, rhs = float(jx1 @ y), float(x1 @ jty)
    assert abs(lhs - rhs) <= 1e-11 * (np.linalg.norm(jx1) * np.linalg.norm(y))      # <J x, y> = <x, J' y>
    jx2 = A.right_multiply(x2)
    comb = A.right_multiply(2.5 * x1 + x2)
    assert np.abs(comb - (2.5 * jx1 + jx2)).max() <= 1e-12 * np.abs(comb).max()       # linearity
    # column norms are the diagonal of J'J: e_j' J' J e_j for a few columns
    sq = A.squared_column_norm()
    for j in (0, 17, 3 * prob.num_points + 5, n_c - 1):
        e = np.zeros(n_c)
        e[j] = 1.0
        col = A.right_multiply(e)
        assert abs(float(col @ col) - sq[j]) <= 1e-12 * max(sq[j], 1e-300)


def test_two_implementations_of_schur_cg_agree_and_solve_the_normal_equations(final):
    ctx, prob, ev, A, res, cost = final
    P, C = prob.num_points, prob.num_cameras
    b = res.to_host()
    sq = A.squared_column_norm()
    D = np.sqrt(np.clip(sq, 1e-6, 1e32) / 1e2)          # a well-regularised LM system (radius 100)
    kw = dict(type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.SCHUR_JACOBI, num_eliminate_blocks=P, max_num_iterations=500)
    S_impl = cx.Solver(ctx, **kw)
    S_expl = cx.Solver(ctx, use_explicit_schur_complement=1, **kw)
    x1, s1 = S_impl.solve(A, b, D, r_tolerance=1e-11, q_tolerance=0.0)
    x2, s2 = S_expl.solve(A, b, D, r_tolerance=1e-11, q_tolerance=0.0)
    assert s1.termination_type == cx.SUCCESS and s2.termination_type == cx.SUCCESS, (s1.message, s2.message)
    # (the two operators round differently -- the implicit one adds 29 M row terms per product, the explicit one 81-entry cells
    # summed once -- so at r_tolerance = 1e-11 their last iterations differ by a few per cent; the solutions are compared below)
    assert abs(s1.num_iterations - s2.num_iterations) <= 0.1 * s1.num_iterations
    scale = np.abs(x1).max()
    assert np.abs(x1 - x2).max() <= 1e-7 * scale
    # normal equations (J'J + D^2) x = J'b through the plain products (a third code path)
    g = A.left_multiply(A.right_multiply(x1) - b) + D * D * x1
    jtb = A.left_multiply(b)
    assert np.linalg.norm(g) <= 1e-8 * np.linalg.norm(jtb)
    # determinism at full size: same solve, same bits
    x1b, _ = S_impl.solve(A, b, D, r_tolerance=1e-11, q_tolerance=0.0)
    assert np.array_equal(x1, x1b)
    S_impl.close()
    S_expl.close()


@pytest.mark.parametrize("pre", ["CLUSTER_JACOBI", "CLUSTER_TRIDIAGONAL"])
def test_visibility_preconditioners_solve_the_normal_equations(final, pre):
    """Final-13682 shape, 890 visibility clusters (CLUSTER_TRIDIAGONAL: one forest path of 3 849 block steps): the
    preconditioned CG converges to the solution of the normal equations, in fewer iterations than SCHUR_JACOBI, and
    twice to the same bits.  (The oracle's dense Cholesky of the 123 138^2 preconditioner is out of reach: a
    size-independent property instead.)"""
    ctx, prob, ev, A, res, cost = final
    P = prob.num_points
    b = res.to_host()
    sq = A.squared_column_norm()
    D = np.sqrt(np.clip(sq, 1e-6, 1e32) / 1e2)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P, max_num_iterations=500)
    x, s = S.solve(A, b, D, r_tolerance=1e-11, q_tolerance=0.0)
    assert s.termination_type == cx.SUCCESS, s.message
    g = A.left_multiply(A.right_multiply(x) - b) + D * D * x
    assert np.linalg.norm(g) <= 1e-8 * np.linalg.norm(A.left_multiply(b))
    J = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.SCHUR_JACOBI, num_eliminate_blocks=P, max_num_iterations=500)
    xj, sj = J.solve(A, b, D, r_tolerance=1e-11, q_tolerance=0.0)
    assert s.num_iterations < sj.num_iterations, (s.num_iterations, sj.num_iterations)
    assert np.abs(x - xj).max() <= 1e-7 * np.abs(xj).max()
    x2, _ = S.solve(A, b, D, r_tolerance=1e-11, q_tolerance=0.0)
    assert np.array_equal(x, x2)
    J.close()
    S.close()


def test_sparse_schur_solves_the_normal_equations(final):
    """Final-13682 shape, SPARSE_SCHUR through the tile-sparse Cholesky (nested dissection, 190 levels, 75 k tiles; the
    oracle's block-sparse factorisation of the same matrix takes 13 s and is timed in bench.py, not run here): the direct
    solve satisfies the normal equations through the plain products, agrees with the converged CG solve, and twice gives
    the same bits."""
    ctx, prob, ev, A, res, cost = final
    P = prob.num_points
    b = res.to_host()
    sq = A.squared_column_norm()
    D = np.sqrt(np.clip(sq, 1e-6, 1e32) / 1e2)
    S = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P)
    x, s = S.solve(A, b, D)
    assert s.termination_type == cx.SUCCESS, s.message
    g = A.left_multiply(A.right_multiply(x) - b) + D * D * x
    assert np.linalg.norm(g) <= 1e-9 * np.linalg.norm(A.left_multiply(b))
    J = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.SCHUR_JACOBI, num_eliminate_blocks=P, max_num_iterations=500)
    xj, sj = J.solve(A, b, D, r_tolerance=1e-11, q_tolerance=0.0)
    assert sj.termination_type == cx.SUCCESS
    assert np.abs(x - xj).max() <= 1e-7 * np.abs(x).max()
    x2, _ = S.solve(A, b, D)
    assert np.array_equal(x, x2)
    J.close()
    S.close()


def test_mixed_precision_schur_cg_at_full_size(final):
    """use_mixed_precision_solves on the Final-13682 shape (the `--mixed` line of DESIGN.md section 7): S x inside CG streams
    fp32 copies of the 29 M cells; against the fp64 solve the LM-style truncated step takes the same number of iterations
    (to one) and differs at the level of the fp32 rounding of J, not more."""
    ctx, prob, ev, A, res, cost = final
    P = prob.num_points
    b = res.to_host()
    sq = A.squared_column_norm()
    D = np.sqrt(np.clip(sq, 1e-6, 1e32) / 1e2)
    kw = dict(type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=P, max_num_iterations=500)
    S64 = cx.Solver(ctx, **kw)
    Smix = cx.Solver(ctx, use_mixed_precision_solves=1, **kw)
    for q in (0.1, 1e-2):
        x64, s64 = S64.solve(A, b, D, r_tolerance=-1.0, q_tolerance=q)
        xm, sm = Smix.solve(A, b, D, r_tolerance=-1.0, q_tolerance=q)
        assert sm.termination_type == cx.SUCCESS and s64.termination_type == cx.SUCCESS
        assert abs(sm.num_iterations - s64.num_iterations) <= 1, (sm.num_iterations, s64.num_iterations)
        err = np.abs(xm - x64).max() / np.abs(x64).max()
        assert 0.0 < err < (1e-4 if sm.num_iterations == s64.num_iterations else 1e-2), err
    S64.close()
    Smix.close()


@pytest.mark.parametrize("stype", ["ITERATIVE_SCHUR", "SPARSE_SCHUR"])
def test_minimizer_at_full_size_descends_and_repeats(final, stype):
    """cx_minimize (TrustRegionMinimizer + LevenbergMarquardtStrategy on the device) on the Final-13682 shape: three LM
    iterations from the perturbed start, with the exact (SPARSE_SCHUR) and with the truncated (ITERATIVE_SCHUR, eta = 0.1)
    step, reduce the cost in every step, by more than a factor of 100 in all, and a second run walks the same path to the
    last bit (nothing in the loop depends on atomics or timing)."""
    ctx, prob, ev, A, res, cost = final
    P = prob.num_points
    kw = dict(type=getattr(cx, stype), num_eliminate_blocks=P, max_num_iterations=500)
    if stype == "ITERATIVE_SCHUR":
        kw["preconditioner_type"] = cx.JACOBI
    runs = []
    for _ in range(2):
        S = cx.Solver(ctx, **kw)
        x, summ, its = cx.binding.minimize(ev, S, prob.state(), cx.binding.minimizer_options(max_num_iterations=3))
        S.close()
        runs.append((x, [it["cost"] for it in its], [it["step_is_successful"] for it in its]))
    x, costs, ok = runs[0]
    assert len(costs) == 4 and all(ok[1:])
    assert all(costs[i + 1] < costs[i] for i in range(3)), costs
    assert costs[-1] < 1e-2 * costs[0]
    assert costs == runs[1][1] and np.array_equal(x, runs[1][0])
    ev.evaluate(prob.state())   # leave the fixture's Jacobian as the other tests expect it


# ----------------------------------------------------------------------- config 4 on shards (BASELINE: "sharded over 8 x MI355X")
def _lm_diagonal(A):
    return np.sqrt(np.clip(A.squared_column_norm(), 1e-6, 1e32) / 1e4)      # levenberg_marquardt_strategy.cc:81-98, radius 1e4


@pytest.mark.parametrize("shards", [2, 4, 8])
def test_config_4_final13682_on_shards_equals_the_unsharded_solve(final, shards):
    """BASELINE config 4 at its full size on 2 / 4 / 8 shards (logical shards of the one GPU of the test box, in-process
    exchange step; partitioning, scatter / gather and the per-rank sharded solvers are the production code): the front's
    solve equals the unsharded DEVICE solve -- the same CG iteration count, x to 1e-8 -- at LM's eta = 0.1 and at
    bundle_adjuster's 1e-2; the exchange step is what DESIGN.md section 5 predicts (1 + it + it // 10 collectives: 54 C
    doubles once, then 9 C each); the cut balances residual blocks to 2 %."""
    ctx, prob, ev, A, res, cost = final
    C, P, O = prob.num_cameras, prob.num_points, prob.num_observations
    b = res.to_host()
    D = _lm_diagonal(A)
    kw = dict(type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=P, max_num_iterations=500)
    S1 = cx.Solver(ctx, **kw)
    mctx = cx.Context(devices=[0] * shards)
    mev = cx.Evaluator(mctx, prob)
    mcost, mres, mgrad = mev.evaluate(prob.state())
    assert abs(mcost - cost) <= 1e-12 * cost
    assert np.array_equal(mres, b)                                   # the same kernel on the same rows: the same bits
    MA = mev.jacobian()
    eb, rb = MA.shard_layout()
    assert eb.size == shards + 1 and eb[0] == 0 and eb[-1] == P and rb[-1] == O
    blocks = np.diff(rb)
    assert blocks.max() * shards / blocks.sum() < 1.02               # load_balance.max_over_mean_residual_blocks
    assert np.abs(_lm_diagonal(MA) - D).max() <= 1e-12 * D.max()
    g1 = A.left_multiply(b)
    assert np.abs(mgrad - g1).max() <= 1e-11 * np.abs(g1).max()       # J'r: camera part summed over the shards
    MS = cx.Solver(mctx, **kw)
    for q in (0.1, 0.01):
        x1, s1 = S1.solve(A, b, D, r_tolerance=-1.0, q_tolerance=q)
        xm, sm = MS.solve(MA, mres, D, r_tolerance=-1.0, q_tolerance=q)
        assert sm.termination_type == s1.termination_type == cx.SUCCESS, (sm.message, s1.message)
        assert sm.num_iterations == s1.num_iterations, (shards, q, sm.num_iterations, s1.num_iterations)
        assert np.abs(xm - x1).max() <= 1e-8 * np.abs(x1).max()
        tm, it = MS.timing(), sm.num_iterations
        calls = 2 + it + it // 10      # the one-word health agreement first, then set-up, then one per S x
        assert tm["allreduce_calls"] == calls and tm["allreduce_bytes"] == 8.0 * (1 + 54 * C + 9 * C * (calls - 2))
    MS.close()
    S1.close()
    mev.close()
    mctx.close()


def test_config_4_final13682_sparse_schur_on_shards(final):
    """Sharded SPARSE_SCHUR at 13 682 cameras (4 shards): the plan comes from the union of the shards' S cells (one dense
    C x C presence exchange, 1.5 GB per shard), cell values and right-hand side are summed over the shards, the
    factorisation is replicated -- the step equals the unsharded tile-sparse solve to 1e-8."""
    ctx, prob, ev, A, res, cost = final
    P = prob.num_points
    b = res.to_host()
    D = _lm_diagonal(A)
    S1 = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P)
    x1, s1 = S1.solve(A, b, D)
    S1.close()
    mctx = cx.Context(devices=[0] * 4)
    mev = cx.Evaluator(mctx, prob)
    _, mres, _ = mev.evaluate(prob.state(), want_gradient=False)
    MS = cx.Solver(mctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P)
    xm, sm = MS.solve(mev.jacobian(), mres, D)
    first = MS.timing()
    xm2, _ = MS.solve(mev.jacobian(), mres, D)
    tm = MS.timing()
    assert sm.termination_type == s1.termination_type == cx.SUCCESS, (sm.message, s1.message)
    assert np.abs(xm - x1).max() <= 1e-8 * np.abs(x1).max()
    assert np.array_equal(xm, xm2)
    assert first["allreduce_calls"] == 7 and tm["allreduce_calls"] == 5      # plan agreement + presence once; per solve: health agreement, values, rhs, replicated tiles, solution
    # round 4: the cell values travel by owner -- a reduce-scatter over the ranks' balanced cell lists (counted at half its
    # input: what it moves in all-reduce terms) instead of the all-reduce of all 0.98 M cells to every rank: 0.97 -> 0.66 GB
    # per solve at 4 shards (0.33 GB of it the replicated tiles of the distributed factorisation)
    assert tm["allreduce_bytes"] <= 0.70e9, tm["allreduce_bytes"]
    MS.close()
    # use_mixed_precision_solves on the shards at this size: the float tile pool under the distributed factorisation, unrefined
    # (single precision error, the same bits as the unsharded float solve's up to the summation order) and with two refinement
    # steps (the factor then stays whole on every shard; back to the double precision step)
    S32 = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P, use_mixed_precision_solves=1)
    x32, s32 = S32.solve(A, b, D)
    S32.close()
    MS32 = cx.Solver(mctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P, use_mixed_precision_solves=1)
    xm32, sm32 = MS32.solve(mev.jacobian(), mres, D)
    MS32.close()
    scale = np.abs(x1).max()
    e1, em = np.abs(x32 - x1).max() / scale, np.abs(xm32 - x1).max() / scale
    assert s32.termination_type == sm32.termination_type == cx.SUCCESS
    assert 1e-10 < e1 < 1e-2 and 1e-10 < em < 1e-2 and em < 10 * e1 + 1e-7, (e1, em)
    MS32r = cx.Solver(mctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P, use_mixed_precision_solves=1, max_num_refinement_iterations=2)
    xm32r, sm32r = MS32r.solve(mev.jacobian(), mres, D)
    MS32r.close()
    assert sm32r.termination_type == cx.SUCCESS and np.abs(xm32r - x1).max() / scale < max(1e-8, 1e-2 * em), (em, np.abs(xm32r - x1).max() / scale)
    mev.close()
    mctx.close()
