"""Parity of the HIP path (libcxschur.so through its C ABI) against the oracle on
seeded BAL-shaped problems.  fp64 tolerances are stated per test; integer
structure (ordering, layout) is compared exactly."""
import os

import numpy as np
import pytest

from conftest import cx

pytestmark = pytest.mark.gpu

REL = 1e-12   # relative tolerance for single J passes (different summation order only)


@pytest.fixture(scope="module")
def ctx():
    c = cx.Context(0)
    yield c
    c.close()


def make(C, P, O, seed, values="eval", oracle=None):
    prob = cx.bal.make_bal_like(C, P, O, seed)
    bs, order = cx.bal.build_structure(prob)
    rng = np.random.default_rng(seed + 1000)
    if values == "random":
        vals = cx.bal.random_jacobian_values(O, seed)
        b = rng.standard_normal(2 * O)
    else:
        _, b, _, vals = oracle.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations,
                                            order, prob.state(), want_gradient=False)
    D = rng.uniform(0.5, 2.0, bs.num_cols) * (1.0 if values == "random" else 1e-2 * np.sqrt(np.abs(vals).mean()))
    return prob, bs, order, vals, b, D


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


PROBLEMS = [(6, 40, 130, 1), (16, 700, 2800, 2), (49, 7776, 31843, 49)]


@pytest.mark.parametrize("C,P,O,seed", PROBLEMS)
def test_products(ctx, oracle, C, P, O, seed):
    prob, bs, order, vals, b, D = make(C, P, O, seed, "random")
    A = cx.Matrix(ctx, bs, P)
    assert A.is_static_239 and A.num_rows == 2 * O and A.num_cols == 3 * P + 9 * C and A.num_nonzeros == 24 * O
    A.set_values(vals)
    assert np.array_equal(A.get_values(), vals)
    rng = np.random.default_rng(seed)
    x, y0 = rng.standard_normal(A.num_cols), rng.standard_normal(A.num_rows)
    assert relerr(A.right_multiply(x, y0), oracle.right_multiply(bs, vals, x, y0)) < REL
    z, c0 = rng.standard_normal(A.num_rows), rng.standard_normal(A.num_cols)
    assert relerr(A.left_multiply(z, c0), oracle.left_multiply(bs, vals, z, c0)) < REL
    assert relerr(A.squared_column_norm(), oracle.squared_column_norm(bs, vals)) < REL
    s = rng.uniform(0.5, 2.0, A.num_cols)
    A.scale_columns(s)
    np.testing.assert_array_equal(A.get_values(), oracle.scale_columns(bs, vals, s))   # one rounding each: bit exact
    # the camera-major copy must follow the new values
    assert relerr(A.left_multiply(z), oracle.left_multiply(bs, oracle.scale_columns(bs, vals, s), z)) < REL
    A.close()


def test_products_bit_exact_on_integers(ctx, oracle):
    """cuda_block_sparse_crs_view_test.cc:60-125 style: integer-valued matrix and unit
    vectors make every product exact, so device == CPU bit for bit."""
    prob, bs, order, vals, b, D = make(8, 60, 240, 7, "random")
    vals = np.arange(1, vals.size + 1, dtype=np.float64)
    A = cx.Matrix(ctx, bs, prob.num_points)
    A.set_values(vals)
    for j in (0, 5, 3 * 60, A.num_cols - 1):
        e = np.zeros(A.num_cols)
        e[j] = 1.0
        np.testing.assert_array_equal(A.right_multiply(e), oracle.right_multiply(bs, vals, e))
    for i in (0, 7, A.num_rows - 1):
        e = np.zeros(A.num_rows)
        e[i] = 1.0
        np.testing.assert_array_equal(A.left_multiply(e), oracle.left_multiply(bs, vals, e))
    A.close()


@pytest.mark.parametrize("C,P,O,seed", PROBLEMS[:2])
def test_implicit_schur_and_eliminator(ctx, oracle, C, P, O, seed):
    prob, bs, order, vals, b, D = make(C, P, O, seed, "random")
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    nf = 9 * C
    x = np.random.default_rng(3).standard_normal(nf)
    y_ref, rhs_ref = oracle.implicit_schur_multiply(bs, vals, D, b, P, x)
    y, rhs = cx.implicit_schur_multiply(ctx, A, D, b, x, nf)
    assert relerr(y, y_ref) < 1e-11 and relerr(rhs, rhs_ref) < 1e-11
    # explicit S (schur_eliminator_test.cc:137-177 bound: 1e-14 relative, looser here for longer sums)
    lhs_ref, r_ref = oracle.schur_eliminate_dense(bs, vals, b, D, P)
    lhs, r = cx.eliminate_dense(ctx, A, b, D, nf)
    assert relerr(np.triu(lhs), np.triu(lhs_ref)) < 1e-11 and relerr(r, r_ref) < 1e-11
    # block structure of lhs: strictly-lower blocks stay zero, like the reference's upper block triangle
    for i in range(C):
        assert not lhs[9 * i:9 * i + 9, :9 * i].any()
    z = np.random.default_rng(4).standard_normal(nf)
    xb_ref = oracle.schur_back_substitute(bs, vals, b, D, P, z)
    xb = cx.back_substitute(ctx, A, b, D, z)
    assert relerr(xb[:3 * P], xb_ref[:3 * P]) < 1e-10
    # the gather path sums every cell in a fixed order: bitwise reproducible
    lhs2, _ = cx.eliminate_dense(ctx, A, b, D, nf)
    assert np.array_equal(lhs, lhs2)
    A.close()
    # the scatter (atomics) path, used when the pair list would be too large
    os.environ["CX_ELIM_ATOMICS"] = "1"
    try:
        A2 = cx.Matrix(ctx, bs, P)
        A2.set_values(vals)
        lhs3, r3 = cx.eliminate_dense(ctx, A2, b, D, nf)
        A2.close()
    finally:
        del os.environ["CX_ELIM_ATOMICS"]
    assert relerr(np.triu(lhs3), np.triu(lhs_ref)) < 1e-11 and relerr(r3, r_ref) < 1e-11
    for i in range(C):
        assert not lhs3[9 * i:9 * i + 9, :9 * i].any()


@pytest.mark.parametrize("n", [9, 31, 64, 96, 97, 200, 441, 1000])
def test_dense_cholesky(ctx, oracle, n):
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 5))
    S = M @ M.T + n * np.eye(n)
    rhs = rng.standard_normal(n)
    lhs = np.triu(S) + np.tril(rng.standard_normal((n, n)), -1)   # lower triangle must be ignored
    x, s = cx.dense_cholesky_solve(ctx, lhs, rhs)
    assert s.termination_type == cx.SUCCESS
    ref = np.linalg.solve(S, rhs)
    assert relerr(x, ref) < 1e-10                                  # dense_cholesky_test.cc: 10 eps * cond
    xo, t = oracle.dense_cholesky_solve(lhs, rhs)
    assert relerr(x, xo) < 1e-10
    # not positive definite -> FAILURE
    bad = S.copy()
    bad[n // 2, n // 2] = -1.0
    x, s = cx.dense_cholesky_solve(ctx, bad, rhs)
    assert s.termination_type == cx.FAILURE


SOLVERS = [("ITERATIVE_SCHUR", "JACOBI"), ("ITERATIVE_SCHUR", "SCHUR_JACOBI"), ("ITERATIVE_SCHUR", "IDENTITY"),
           ("CGNR", "JACOBI"), ("CGNR", "IDENTITY"), ("DENSE_SCHUR", "IDENTITY")]


@pytest.mark.parametrize("stype,pre", SOLVERS)
@pytest.mark.parametrize("C,P,O,seed", PROBLEMS[:2])
def test_solvers_match_oracle(ctx, oracle, stype, pre, C, P, O, seed):
    """Same LM-style call the reference makes (levenberg_marquardt_strategy.cc:97-116):
    q_tolerance = eta = 0.1, r_tolerance = -1.  The iteration count is an integer
    outcome and must match; the solution matches to 1e-8 relative (CG amplifies
    summation-order differences by the condition number)."""
    prob, bs, order, vals, b, D = make(C, P, O, seed, "eval", oracle)
    nelim = 0 if stype == "CGNR" else P
    A = cx.Matrix(ctx, bs, nelim)
    A.set_values(vals)
    kw = dict(type=getattr(cx, stype), preconditioner_type=getattr(cx, pre), num_eliminate_blocks=nelim,
              max_num_iterations=200)
    S = cx.Solver(ctx, **kw)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    oo = oracle.make_options(type=getattr(oracle, stype), preconditioner_type=getattr(oracle, pre),
                             num_eliminate_blocks=P, max_num_iterations=200)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.1)
    assert s.termination_type == sr.termination_type, (s.message, sr.message)
    assert s.num_iterations == sr.num_iterations, (s.message, sr.message)
    assert np.all(np.isfinite(x))
    assert relerr(x, xr) < 1e-8
    S.close()
    A.close()


@pytest.mark.parametrize("C,P,O,seed", PROBLEMS[:2])
def test_explicit_schur_complement(ctx, oracle, C, P, O, seed):
    """ITERATIVE_SCHUR with use_explicit_schur_complement (solver.h:518-540): CG on the block-sparse S with
    the block Jacobi of S (schur_complement_solver.cc:337-420).  Cell set bit-exact with InitStorage's; same
    iteration count as the oracle; the step equals the implicit SCHUR_JACOBI solve's to CG accuracy."""
    prob, bs, order, vals, b, D = make(C, P, O, seed, "eval", oracle)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    r, c = cx.binding.schur_sparse_structure(A)
    r_ref, c_ref = oracle.schur_sparse_structure(bs, P)
    assert np.array_equal(r, r_ref) and np.array_equal(c, c_ref)        # integer structure: exact
    kw = dict(type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.SCHUR_JACOBI, num_eliminate_blocks=P, max_num_iterations=200)
    S = cx.Solver(ctx, use_explicit_schur_complement=1, **kw)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=oracle.SCHUR_JACOBI, num_eliminate_blocks=P,
                             max_num_iterations=200, use_explicit_schur_complement=1)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.1)
    assert s.termination_type == sr.termination_type, (s.message, sr.message)
    assert s.num_iterations == sr.num_iterations, (s.message, sr.message)
    assert relerr(x, xr) < 1e-8
    x2, s2 = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    assert np.array_equal(x, x2)                                          # fixed summation order
    Si = cx.Solver(ctx, **kw)
    xi, si = Si.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    assert si.num_iterations == s.num_iterations and relerr(x, xi) < 1e-7
    # option validation of the reference (solver.cc:277-290)
    with pytest.raises(cx.CxError):
        cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=P,
                  use_explicit_schur_complement=1)
    Si.close()
    S.close()
    A.close()


@pytest.mark.parametrize("stype,pre", [("ITERATIVE_SCHUR", "JACOBI"), ("CGNR", "JACOBI"), ("DENSE_SCHUR", "IDENTITY")])
def test_solvers_converged_solution(ctx, oracle, stype, pre):
    """Run to r_tolerance 1e-12 and compare with the dense normal-equation solution
    (schur_complement_solver_test.cc / iterative_schur_complement_solver_test.cc bound 1e-10)."""
    C, P, O = 5, 30, 100
    prob, bs, order, vals, b, D = make(C, P, O, 21, "random")
    J = bs.to_dense(vals)
    ref = np.linalg.solve(J.T @ J + np.diag(D ** 2), J.T @ b)
    nelim = 0 if stype == "CGNR" else P
    A = cx.Matrix(ctx, bs, nelim)
    A.set_values(vals)
    S = cx.Solver(ctx, type=getattr(cx, stype), preconditioner_type=getattr(cx, pre), num_eliminate_blocks=nelim,
                  max_num_iterations=A.num_cols * 2)
    x, s = S.solve(A, b, D, r_tolerance=1e-13, q_tolerance=0.0)
    assert s.termination_type == cx.SUCCESS, s.message
    assert np.linalg.norm(x - ref) / A.num_cols < 1e-10
    S.close()
    A.close()


def test_device_pointer_solve_matches_host_pointer_solve(ctx, oracle):
    prob, bs, order, vals, b, D = make(16, 700, 2800, 2, "eval", oracle)
    A = cx.Matrix(ctx, bs, prob.num_points)
    A.set_values(vals)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=prob.num_points)
    xh, sh = S.solve(A, b, D, q_tolerance=0.1)
    db, dD, dx = ctx.to_device(b), ctx.to_device(D), ctx.empty(A.num_cols)
    _, sd = S.solve(A, db, dD, q_tolerance=0.1, x=dx)
    np.testing.assert_array_equal(dx.to_host(), xh)          # deterministic kernels: bit identical
    assert sd.num_iterations == sh.num_iterations
    t = S.timing()
    assert t["total_ms"] > 0 and t["reduced_solve_ms"] > 0
    S.close()
    A.close()


@pytest.mark.parametrize("C,P,O,seed", PROBLEMS)
def test_evaluator(ctx, oracle, C, P, O, seed):
    prob = cx.bal.make_bal_like(C, P, O, seed)
    bs, order = cx.bal.build_structure(prob)
    ev = cx.Evaluator(ctx, prob)
    # integer layout: bit exact
    row_of_obs = ev.row_of_observation()
    assert np.array_equal(order[row_of_obs], np.arange(O))
    state = prob.state()
    cost, res, grad = ev.evaluate(state)
    cost_r, res_r, grad_r, vals_r = oracle.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index,
                                                        prob.observations, order, state)
    Jm = ev.jacobian(bs)
    vals = Jm.get_values()
    # device libm (sin, cos, sqrt) differs from glibc in the last ulps: 1e-11 relative to the largest entry
    assert relerr(res, res_r) < 1e-11 and relerr(vals, vals_r) < 1e-11
    assert abs(cost - cost_r) <= 1e-11 * cost_r
    assert relerr(grad, grad_r) < 1e-10
    # residual-only evaluation takes the plain-double path
    cost2, res2, _ = ev.evaluate(state, want_gradient=False, want_jacobian=False)
    assert relerr(res2, res_r) < 1e-11 and abs(cost2 - cost_r) <= 1e-11 * cost_r
    # a gradient WITHOUT a Jacobian (Evaluate(..., gradient, jacobian = nullptr)) at another state: g = J'r of that state,
    # while the matrix the caller holds -- column-scaled in the meantime -- keeps its values to the bit
    # (ProgramEvaluator computes such a gradient from scratch blocks, program_evaluator.h:186-258)
    Jm.scale_columns(1.0 / (1.0 + np.sqrt(Jm.squared_column_norm())))
    kept = Jm.get_values()
    state2 = state + 1e-3 * np.random.default_rng(seed).standard_normal(state.size)
    cost4, res4, grad4 = ev.evaluate(state2, want_jacobian=False)
    cost_r2, res_r2, grad_r2, _ = oracle.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations, order, state2)
    assert relerr(grad4, grad_r2) < 1e-10 and relerr(res4, res_r2) < 1e-11 and abs(cost4 - cost_r2) <= 1e-11 * cost_r2
    assert np.array_equal(Jm.get_values(), kept)
    y = np.random.default_rng(1).standard_normal(Jm.num_rows)
    assert relerr(Jm.left_multiply(y), oracle.left_multiply(bs, kept, y)) < 1e-12      # the camera-major copy is the scaled one still
    ev.close()


@pytest.mark.parametrize("loss", [None, (cx.binding.LOSS_HUBER, 1.0, 0.0)])
@pytest.mark.parametrize("quaternion", [False, True])
def test_evaluator_applies_a_registered_column_scale(ctx, oracle, loss, quaternion):
    """cx_evaluator_set_column_scale: TrustRegionMinimizer re-applies ONE scaling vector after every
    evaluation (trust_region_minimizer.cc:263-279); the evaluator that knows it writes J diag(s) in the evaluation kernel.
    Bit-identical to evaluate -> ScaleColumns (values and the camera-major copy behind the products), residuals and cost
    untouched, gradient still J'r of the unscaled problem; a gradient-only evaluation ignores the scale; clearing restores
    the plain evaluation.  More tiles than resident workgroups, so the persistent loop takes several rounds."""
    C, P, O = 40, 70000, 300000
    prob = cx.bal.make_bal_like(C, P, O, 21)
    ev = cx.Evaluator(ctx, prob)
    state = prob.state()
    if quaternion:
        ev.set_camera_model(cx.binding.CAMERA_QUATERNION_MANIFOLD)
        state = cx.bal.state_quaternion(prob)
    if loss:
        ev.set_loss(*loss)
    cost0, res0, grad0 = ev.evaluate(state)
    J = ev.jacobian()
    scale = 1.0 / (1.0 + np.sqrt(J.squared_column_norm()))
    J.scale_columns(scale)
    want = J.get_values()
    y = np.random.default_rng(2).standard_normal(J.num_rows)
    want_jty, want_sq = J.left_multiply(y), J.squared_column_norm()
    ev.set_column_scale(scale)
    cost1, res1, grad1 = ev.evaluate(state)
    assert np.array_equal(J.get_values(), want)
    assert np.array_equal(J.left_multiply(y), want_jty) and np.array_equal(J.squared_column_norm(), want_sq)
    assert cost1 == cost0 and np.array_equal(res1, res0)
    assert relerr(grad1, grad0) < 1e-14
    _, _, grad2 = ev.evaluate(state, want_jacobian=False)          # gradient-only: scratch J, unscaled
    assert relerr(grad2, grad0) < 1e-14 and np.array_equal(J.get_values(), want)
    ev.set_column_scale(None)
    ev.evaluate(state)
    J.scale_columns(scale)
    assert np.array_equal(J.get_values(), want)
    ev.close()


@pytest.mark.parametrize("env", [{}, {"CX_EVAL_VARIANT": "1"}, {"CX_EVAL_VARIANT": "4"}, {"CX_EVAL_PERSISTENT": "0"}],
                         ids=["default", "dual-numbers", "plain-gather", "one-tile-per-workgroup"])
def test_evaluator_variants(env):
    """k_bal_evaluate's variants (closed-form Jacobian / dual numbers, cooperative / plain camera gather, persistent /
    one tile per workgroup) against the oracle on a problem with more tiles than resident workgroups; each in its own
    process because the switches are read once (tests/evaluator_variant_check.py)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    child_env = dict(os.environ)
    for k in ("CX_EVAL_VARIANT", "CX_EVAL_PERSISTENT"):
        child_env.pop(k, None)
    child_env.update(env)
    out = subprocess.run([sys.executable, os.path.join(here, "evaluator_variant_check.py")], env=child_env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "EVALUATOR_VARIANT_OK" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("loss", [(cx.binding.LOSS_HUBER, 1.0, 0.0), (cx.binding.LOSS_HUBER, 0.25, 0.0),
                                  (cx.binding.LOSS_SOFT_L_ONE, 0.7, 0.0), (cx.binding.LOSS_CAUCHY, 1.3, 0.0),
                                  (cx.binding.LOSS_ARCTAN, 0.7, 0.0), (cx.binding.LOSS_TOLERANT, 0.7, 0.4),
                                  (cx.binding.LOSS_TUKEY, 1.3, 0.0)])
def test_evaluator_robust_loss(ctx, oracle, loss):
    """Corrector + LossFunction inside the Jacobian kernel (residual_block.cc:160-196): noisy
    observations put residual blocks on both sides of every loss's inlier/outlier boundary."""
    C, P, O = 12, 300, 2400
    prob = cx.bal.make_bal_like(C, P, O, 3)
    bs, order = cx.bal.build_structure(prob)
    ev = cx.Evaluator(ctx, prob)
    ev.set_loss(*loss)
    state = prob.state()
    cost, res, grad = ev.evaluate(state)
    cost_r, res_r, grad_r, vals_r = oracle.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index,
                                                        prob.observations, order, state, loss=loss)
    cost_plain = oracle.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations, order,
                                     state)[0]
    assert abs(cost_plain - cost_r) > 1e-3 * cost_plain  # the loss is active on this input
    vals = ev.jacobian(bs).get_values()
    assert relerr(res, res_r) < 1e-11 and relerr(vals, vals_r) < 1e-11
    assert abs(cost - cost_r) <= 1e-11 * abs(cost_r)
    assert relerr(grad, grad_r) < 1e-10
    cost2, res2, _ = ev.evaluate(state, want_gradient=False, want_jacobian=False)
    assert relerr(res2, res_r) < 1e-11 and abs(cost2 - cost_r) <= 1e-11 * abs(cost_r)
    ev.set_loss(cx.binding.LOSS_NONE)
    cost3, _, _ = ev.evaluate(state, want_gradient=False, want_jacobian=False)
    assert abs(cost3 - cost_plain) <= 1e-11 * cost_plain
    ev.close()


def test_big_chunk_paths(ctx, oracle):
    """A point seen by more cameras than a tile has rows exercises the long-chunk code paths."""
    C, P = 300, 40
    rng = np.random.default_rng(5)
    cam, pt = [], []
    for j in range(P):
        k = 290 if j in (3, 17) else int(rng.integers(2, 6))
        cams = np.sort(rng.choice(C, size=k, replace=False))
        cam.extend(cams.tolist())
        pt.extend([j] * k)
    cam, pt = np.array(cam, dtype=np.int32), np.array(pt, dtype=np.int32)
    O = cam.size
    prob = cx.bal.BalProblem(C, P, cam, pt, np.zeros((O, 2)), np.zeros((C, 9)), np.zeros((P, 3)))
    bs, order = cx.bal.build_structure(prob)
    vals = cx.bal.random_jacobian_values(O, 1)
    b = rng.standard_normal(2 * O)
    D = rng.uniform(0.5, 2.0, bs.num_cols)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    assert A.is_static_239
    z = rng.standard_normal(2 * O)
    assert relerr(A.left_multiply(z), oracle.left_multiply(bs, vals, z)) < REL
    x = rng.standard_normal(9 * C)
    y_ref, rhs_ref = oracle.implicit_schur_multiply(bs, vals, D, b, P, x)
    y, rhs = cx.implicit_schur_multiply(ctx, A, D, b, x, 9 * C)
    assert relerr(y, y_ref) < 1e-11 and relerr(rhs, rhs_ref) < 1e-11
    lhs_ref, r_ref = oracle.schur_eliminate_dense(bs, vals, b, D, P)
    lhs, r = cx.eliminate_dense(ctx, A, b, D, 9 * C)
    assert relerr(np.triu(lhs), np.triu(lhs_ref)) < 1e-11 and relerr(r, r_ref) < 1e-11
    xb = cx.back_substitute(ctx, A, b, D, x)
    assert relerr(xb[:3 * P], oracle.schur_back_substitute(bs, vals, b, D, P, x)[:3 * P]) < 1e-10
    A.close()


def test_generic_layout_products(ctx, oracle):
    """Matrices outside the static <2,3,9> layout (the reference fixtures) use the dynamic kernels."""
    from conftest import lls_problem
    for pid in (2, 4, 5, 6):
        bs, values, b, D, nelim, raw = lls_problem(pid)
        A = cx.Matrix(ctx, bs, nelim)
        assert not A.is_static_239
        A.set_values(values)
        rng = np.random.default_rng(pid)
        x, y = rng.standard_normal(bs.num_cols), rng.standard_normal(bs.num_rows)
        assert relerr(A.right_multiply(x), oracle.right_multiply(bs, values, x)) < 1e-14
        assert relerr(A.left_multiply(y), oracle.left_multiply(bs, values, y)) < 1e-14
        assert relerr(A.squared_column_norm(), oracle.squared_column_norm(bs, values)) < 1e-14
        A.close()


@pytest.mark.parametrize("wide", [20, 33, 64, 70, 300])
def test_generic_products_with_wide_column_blocks(ctx, oracle, wide):
    """the gather form of y += A'x assumed column blocks of at most 16 scalars.  A shared calibration
    block of 20 (or 33, 64) parameters seen by more than 128 rows -- several gather segments, partial sums `stride` apart --
    must give the oracle's product; blocks wider than 64 take the scatter kernel; a row block of 300 scalars keeps its size
    (it used to be packed into 8 bits)."""
    rng = np.random.default_rng(wide)
    rows, pos = [], 0
    num_e = 40
    col_sizes = [3] * num_e + [wide, 5]
    row_size = 300 if wide == 300 else 2
    for r in range(400):
        e = r % num_e
        cells = [(e, pos)]
        pos += row_size * 3
        cells.append((num_e, pos))          # every row sees the wide block: 400 cells > three segments of 128
        pos += row_size * wide
        if r % 3 == 0:
            cells.append((num_e + 1, pos))
            pos += row_size * 5
        rows.append((row_size, cells))
    bs = cx.BlockStructure.from_rows(col_sizes, rows)
    values = rng.standard_normal(pos)
    A = cx.Matrix(ctx, bs, 0)
    assert not A.is_static_239
    A.set_values(values)
    x, y = rng.standard_normal(bs.num_cols), rng.standard_normal(bs.num_rows)
    assert relerr(A.right_multiply(x), oracle.right_multiply(bs, values, x)) < 1e-13
    lt = A.left_multiply(y)
    assert relerr(lt, oracle.left_multiply(bs, values, y)) < 1e-13
    if wide <= 64:
        assert np.array_equal(lt, A.left_multiply(y))   # the gather form is repeatable to the bit
    assert relerr(A.squared_column_norm(), oracle.squared_column_norm(bs, values)) < 1e-13
    A.close()


def test_host_adapter_cpp():
    """The C++ adapter behind the mirrored LinearSolver interface (host/cx_linear_solver.h):
    every solver type against the normal equations, and the LM call sequence."""
    import os
    import subprocess
    from conftest import PKG_DIR
    host = os.path.join(PKG_DIR, "host")
    subprocess.check_call(["make", "-s", "-C", host, "test_host_adapter"])
    out = subprocess.run([os.path.join(host, "test_host_adapter")], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout


def test_rccl_single_rank_communicator(ctx):
    """librccl is resolved with dlopen, ncclCommInitRank forms a 1-rank communicator and cx_allreduce_sum really
    calls ncclAllReduce(sum, fp64) on the context's stream (the N > 1 data path; more ranks need more GPUs than
    the test box has -- the sharded logic itself runs with two ranks in tests/test_gpu_sharded.py)."""
    c2 = cx.Context(0)
    uid = cx.Context.unique_id()
    assert len(uid) == 128
    c2.set_comm(0, 1, uid)
    assert c2.rank == 0 and c2.num_ranks == 1
    v = np.arange(1000, dtype=np.float64)
    d = c2.to_device(v)
    c2.allreduce_sum(d)
    c2.allreduce_sum(d, offset=10, count=100)
    c2.synchronize()
    np.testing.assert_array_equal(d.to_host(), v)
    c2.close()


def test_shard_additivity_on_device(ctx, oracle):
    """What the multi-GPU path relies on: camera-space results of point shards add up to the
    result of the whole problem (S x, reduced rhs, explicit S), point-space results are local."""
    prob, bs, order, vals, b, D = make(12, 300, 1400, 11, "random")
    C, P, O = prob.num_cameras, prob.num_points, prob.num_observations
    nf = 9 * C
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    x = np.random.default_rng(1).standard_normal(nf)
    y_full, rhs_full = cx.implicit_schur_multiply(ctx, A, D, b, x, nf)
    lhs_full, r_full = cx.eliminate_dense(ctx, A, b, D, nf)
    z = np.random.default_rng(2).standard_normal(nf)
    xb_full = cx.back_substitute(ctx, A, b, D, z)
    Df2x = D[3 * P:] ** 2 * x
    bounds = cx.partition_points(bs, P, 3)
    y_sum, rhs_sum, lhs_sum, r_sum = np.zeros(nf), np.zeros(nf), np.zeros((nf, nf)), np.zeros(nf)
    rows_pt = bs.cells["block_id"][0::2]
    for k in range(3):
        lo, hi = int(bounds[k]), int(bounds[k + 1])
        sub = cx.bal.shard(prob, lo, hi)
        sbs, sorder = cx.bal.build_structure(sub)
        rows = np.flatnonzero((rows_pt >= lo) & (rows_pt < hi))
        svals = np.concatenate([vals[:6 * O].reshape(-1, 6)[rows].ravel(), vals[6 * O:].reshape(-1, 18)[rows].ravel()])
        sb = b.reshape(-1, 2)[rows].ravel()
        sD = np.concatenate([D[3 * lo:3 * hi], D[3 * P:]])
        SA = cx.Matrix(ctx, sbs, sub.num_points)
        SA.set_values(svals)
        y, rhs = cx.implicit_schur_multiply(ctx, SA, sD, sb, x, nf)
        y_sum += y - Df2x
        rhs_sum += rhs
        lhs, r = cx.eliminate_dense(ctx, SA, sb, sD, nf)
        lhs_sum += lhs - np.diag(D[3 * P:] ** 2)
        r_sum += r
        xb = cx.back_substitute(ctx, SA, sb, sD, z)
        assert relerr(xb[:3 * (hi - lo)], xb_full[3 * lo:3 * hi]) < 1e-12
        SA.close()
    assert relerr(y_sum + Df2x, y_full) < 1e-12 and relerr(rhs_sum, rhs_full) < 1e-12
    assert relerr(np.triu(lhs_sum + np.diag(D[3 * P:] ** 2)), np.triu(lhs_full)) < 1e-11 and relerr(r_sum, r_full) < 1e-12
    A.close()


def test_partitioned_products(ctx, oracle):
    """PartitionedMatrixView E / F products (partitioned_matrix_view_test.cc:103-204) on the static
    layout and on a fixture with dynamic blocks, against dense algebra."""
    from conftest import lls_problem
    prob, bs, order, vals, b, D = make(8, 120, 520, 31, "random")
    cases = [(bs, vals, prob.num_points)]
    for pid in (2, 4, 6):
        fbs, fvals, fb, fD, fnelim, raw = lls_problem(pid)
        cases.append((fbs, fvals, fnelim))
    rng = np.random.default_rng(5)
    for cbs, cvals, nelim in cases:
        J = cbs.to_dense(cvals)
        ne = int(cbs.col_blocks["size"][:nelim].sum())
        E, F = J[:, :ne], J[:, ne:]
        A = cx.Matrix(ctx, cbs, nelim)
        A.set_values(cvals)
        xe, xf = rng.standard_normal(ne), rng.standard_normal(J.shape[1] - ne)
        yr, y0 = rng.standard_normal(J.shape[0]), rng.standard_normal(J.shape[0])
        tol = 1e-12 * max(1.0, np.abs(J).max()) * J.shape[1]
        assert np.abs(A.partitioned_multiply("e", False, xe, y0) - (y0 + E @ xe)).max() < tol
        assert np.abs(A.partitioned_multiply("f", False, xf, y0) - (y0 + F @ xf)).max() < tol
        ce, cf = rng.standard_normal(ne), rng.standard_normal(J.shape[1] - ne)
        assert np.abs(A.partitioned_multiply("e", True, yr, ce) - (ce + E.T @ yr)).max() < tol * 10
        assert np.abs(A.partitioned_multiply("f", True, yr, cf) - (cf + F.T @ yr)).max() < tol * 10
        A.close()


@pytest.mark.parametrize("spse_init", [0, 1])
@pytest.mark.parametrize("pre", ["SCHUR_POWER_SERIES_EXPANSION", "JACOBI"])
def test_power_series_expansion(ctx, oracle, pre, spse_init):
    """SCHUR_POWER_SERIES_EXPANSION preconditioner and use_spse_initialization
    (power_series_expansion_preconditioner.cc:57-84, iterative_schur_complement_solver.cc:100-111)."""
    if pre == "JACOBI" and not spse_init:
        pytest.skip("covered by test_solvers_match_oracle")
    prob, bs, order, vals, b, D = make(16, 700, 2800, 2, "eval", oracle)
    P = prob.num_points
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P,
                  max_num_iterations=200, max_num_spse_iterations=4, use_spse_initialization=spse_init, spse_tolerance=0.1)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.05)
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                             max_num_iterations=200, max_num_spse_iterations=4, use_spse_initialization=spse_init,
                             spse_tolerance=0.1)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.05)
    assert s.termination_type == sr.termination_type and s.num_iterations == sr.num_iterations, (s.message, sr.message)
    assert relerr(x, xr) < 1e-8
    S.close()
    A.close()


def test_mixed_precision_cgnr(ctx, oracle):
    """fp32-storage CGNR (BASELINE config 5; not in the reference, whose mixed precision exists for
    Cholesky only).  Against the fp64 solve: the truncated LM-style solve agrees to fp32 level and
    takes the same number of iterations; with refinement steps and a tight tolerance the solution
    reaches the dense normal-equation solution like the fp64 path."""
    prob, bs, order, vals, b, D = make(16, 700, 2800, 2, "eval", oracle)
    A = cx.Matrix(ctx, bs, 0)
    A.set_values(vals)
    kw = dict(type=cx.CGNR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=0, max_num_iterations=300)
    S64 = cx.Solver(ctx, **kw)
    x64, s64 = S64.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    S32 = cx.Solver(ctx, use_mixed_precision_solves=1, **kw)
    x32, s32 = S32.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    assert s32.termination_type == cx.SUCCESS and s32.num_iterations == s64.num_iterations
    assert 0 < relerr(x32, x64) < 1e-5           # really took the fp32 path, and stayed close
    # converged: a few refinement rounds against the fp64 operator
    J = bs.to_dense(vals)
    ref = np.linalg.solve(J.T @ J + np.diag(D ** 2), J.T @ b)
    Sr = cx.Solver(ctx, use_mixed_precision_solves=1, max_num_refinement_iterations=3, **dict(kw, max_num_iterations=2000))
    xr, sr = Sr.solve(A, b, D, r_tolerance=1e-10, q_tolerance=0.0)
    S0 = cx.Solver(ctx, use_mixed_precision_solves=1, max_num_refinement_iterations=0, **dict(kw, max_num_iterations=2000))
    x0, s0 = S0.solve(A, b, D, r_tolerance=1e-10, q_tolerance=0.0)
    e0, er = relerr(x0, ref), relerr(xr, ref)
    assert er < 1e-9 and er < e0 / 10, (e0, er)
    for s in (S64, S32, Sr, S0):
        s.close()
    A.close()


def _custom_problem(C, cam_lists, seed):
    """BalProblem with explicit per-point camera lists (ascending, like BAL files)."""
    cam, pt = [], []
    for j, cams in enumerate(cam_lists):
        cam.extend(sorted(cams))
        pt.extend([j] * len(cams))
    cam, pt = np.array(cam, dtype=np.int32), np.array(pt, dtype=np.int32)
    O = cam.size
    return cx.bal.BalProblem(C, len(cam_lists), cam, pt, np.zeros((O, 2)), np.zeros((C, 9)), np.zeros((len(cam_lists), 3)))


@pytest.mark.parametrize("case", ["unseen_camera", "single_observation_points", "one_camera", "tile_boundaries"])
def test_edge_structures(ctx, oracle, case):
    """Edge cases of the structure: a camera nobody observes (its Schur block is D^2 only), points
    seen once, a single camera, and chunk sizes that land exactly on / next to tile boundaries
    (256 rows) -- all solver families against the oracle."""
    rng = np.random.default_rng({"unseen_camera": 1, "single_observation_points": 2, "one_camera": 3,
                                 "tile_boundaries": 4}[case])
    if case == "unseen_camera":
        C, lists = 6, [list(rng.choice([0, 1, 2, 4, 5], size=int(rng.integers(2, 5)), replace=False)) for _ in range(60)]
    elif case == "single_observation_points":
        C, lists = 5, [[int(rng.integers(0, 5))] if j % 3 == 0 else list(rng.choice(5, size=3, replace=False)) for j in range(90)]
    elif case == "one_camera":
        C, lists = 1, [[0] for _ in range(40)]
    else:
        C = 300
        sizes = [256, 1, 255, 2, 254, 3, 128, 128, 127, 129, 64] + [int(s) for s in rng.integers(2, 9, size=200)]
        lists = [list(rng.choice(C, size=s, replace=False)) for s in sizes]
    prob = _custom_problem(C, lists, 0)
    P, O = prob.num_points, prob.num_observations
    bs, order = cx.bal.build_structure(prob)
    vals = cx.bal.random_jacobian_values(O, 3)
    b = rng.standard_normal(2 * O)
    D = rng.uniform(0.5, 2.0, bs.num_cols)
    A = cx.Matrix(ctx, bs, P)
    assert A.is_static_239
    A.set_values(vals)
    x = rng.standard_normal(bs.num_cols)
    z = rng.standard_normal(bs.num_rows)
    assert relerr(A.right_multiply(x), oracle.right_multiply(bs, vals, x)) < REL
    assert relerr(A.left_multiply(z), oracle.left_multiply(bs, vals, z)) < REL
    for stype, pre in (("ITERATIVE_SCHUR", "JACOBI"), ("ITERATIVE_SCHUR", "SCHUR_JACOBI"), ("DENSE_SCHUR", "IDENTITY")):
        S = cx.Solver(ctx, type=getattr(cx, stype), preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P,
                      max_num_iterations=400)
        xs, s = S.solve(A, b, D, r_tolerance=1e-12, q_tolerance=0.0)
        oo = oracle.make_options(type=getattr(oracle, stype), preconditioner_type=getattr(oracle, pre),
                                 num_eliminate_blocks=P, max_num_iterations=400)
        xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=1e-12, q_tolerance=0.0)
        assert s.termination_type == cx.SUCCESS, (case, stype, pre, s.message)
        assert relerr(xs, xr) < 1e-7, (case, stype, pre)   # q_tolerance = 0 stops on rounding noise of Q
        S.close()
    A.close()
    # CGNR uses the unpartitioned matrix
    A0 = cx.Matrix(ctx, bs, 0)
    A0.set_values(vals)
    S = cx.Solver(ctx, type=cx.CGNR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=0, max_num_iterations=2000)
    xs, s = S.solve(A0, b, D, r_tolerance=1e-12, q_tolerance=0.0)
    J = bs.to_dense(vals) if bs.num_rows * bs.num_cols < 4e7 else None
    if J is not None:
        ref = np.linalg.solve(J.T @ J + np.diag(D ** 2), J.T @ b)
        assert relerr(xs, ref) < 1e-8
    S.close()
    A0.close()


@pytest.mark.parametrize("pre", ["JACOBI", "SCHUR_JACOBI"])
def test_mixed_precision_iterative_schur(ctx, oracle, pre):
    """use_mixed_precision_solves with ITERATIVE_SCHUR: S x inside CG streams fp32 copies of the cells (fp64
    accumulation and vectors), everything else stays fp64.  Not in the reference; compared with the fp64 path:
    the fp32 rounding of J (6e-8) shows up at the 1e-6 level after CG, iteration counts agree to one step."""
    C, P, O = 30, 4000, 20000
    prob, bs, order, vals, b, D = make(C, P, O, 9, "eval", oracle)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    kw = dict(type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P, max_num_iterations=200)
    S64 = cx.Solver(ctx, **kw)
    Smix = cx.Solver(ctx, use_mixed_precision_solves=1, **kw)
    for q in (0.1, 1e-3):
        x64, s64 = S64.solve(A, b, D, r_tolerance=-1.0, q_tolerance=q)
        xm, sm = Smix.solve(A, b, D, r_tolerance=-1.0, q_tolerance=q)
        assert sm.termination_type == cx.SUCCESS
        assert abs(sm.num_iterations - s64.num_iterations) <= 1, (sm.num_iterations, s64.num_iterations)
        assert relerr(xm, x64) < (1e-5 if sm.num_iterations == s64.num_iterations else 1e-2)
        assert relerr(xm, x64) > 0.0  # the fp32 copies really were used
    # fresh values invalidate the fp32 copies
    A.set_values(vals * 2.0)
    xm2, _ = Smix.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    x642, _ = S64.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    assert relerr(xm2, x642) < 1e-5
    S64.close()
    Smix.close()
    A.close()


@pytest.mark.parametrize("C,P,O,seed", [(12, 300, 1500, 1), (30, 800, 4000, 2), (100, 3000, 14000, 3), (400, 9000, 40000, 4)])
def test_sparse_schur_tile_cholesky(ctx, oracle, C, P, O, seed):
    """SPARSE_SCHUR through the tile-sparse Cholesky (forced with CX_SPARSE_CHOLESKY; by default it takes over
    from 2048 cameras): same step as the dense reduced solve and as the oracle (schur_complement_solver_test.cc
    bound 1e-10 on |dx| / n).  The camera ordering is this library's own (reverse Cuthill-McKee): parity
    unpinned for CHOLMOD's AMD / CAMD, pinned on the solution."""
    prob, bs, order, vals, b, D = make(C, P, O, seed, "eval", oracle)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    Sd = cx.Solver(ctx, type=cx.DENSE_SCHUR, num_eliminate_blocks=P)
    xd, sd = Sd.solve(A, b, D)
    os.environ["CX_SPARSE_CHOLESKY"] = "1"
    try:
        Ss = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P)
        xs, ss = Ss.solve(A, b, D)
        xs2, _ = Ss.solve(A, b, D)
    finally:
        del os.environ["CX_SPARSE_CHOLESKY"]
    assert ss.termination_type == cx.SUCCESS and ss.num_iterations == 1
    assert np.all(np.isfinite(xs))
    assert np.linalg.norm(xs - xd) / xs.size < 1e-10 and relerr(xs, xd) < 1e-8
    oo = oracle.make_options(type=oracle.SPARSE_SCHUR, num_eliminate_blocks=P)
    xr, sr = oracle.solve(bs, vals, b, D, oo)
    assert np.linalg.norm(xs - xr) / xs.size < 1e-10
    assert np.array_equal(xs, xs2)   # fixed order of every sum
    # not positive definite -> FAILURE, like the dense path
    Dneg = D.copy()
    vals_bad = vals.copy()
    vals_bad[6 * O:] = 0.0           # F = 0 and D_f = 0: S is singular
    Dneg[3 * P:] = 0.0
    A.set_values(vals_bad)
    os.environ["CX_SPARSE_CHOLESKY"] = "1"
    try:
        xbad, sbad = Ss.solve(A, b, Dneg)
    finally:
        del os.environ["CX_SPARSE_CHOLESKY"]
    assert sbad.termination_type == cx.FAILURE
    # host buffers of a failed solve come back as the zeros the reference writes first
    # (schur_complement_solver.cc:137), not as the staging copy's garbage or the factorisation's NaNs
    assert not xbad.any()
    xbad_d, sbad_d = Sd.solve(A, b, Dneg)
    assert sbad_d.termination_type == cx.FAILURE and not xbad_d.any()
    Sd.close()
    Ss.close()
    A.close()


@pytest.mark.parametrize("stype", ["DENSE_SCHUR", "SPARSE_SCHUR"])
@pytest.mark.parametrize("C,P,O,seed", [(12, 300, 1500, 1), (100, 3000, 14000, 3), (400, 9000, 40000, 4)])
def test_mixed_precision_and_refined_reduced_solves(ctx, oracle, stype, C, P, O, seed):
    """use_mixed_precision_solves / max_num_refinement_iterations on DENSE_SCHUR and SPARSE_SCHUR (solver.h:572-590;
    DenseCholesky::Create dense_cholesky.cc:84-136, SparseCholesky::Create sparse_cholesky.cc:45-118, iterative_refiner.cc): S is
    factored in SINGLE precision (tile pool in floats, v_mfma_f32_16x16x4_f32 updates) and / or the solution refined against
    the fp64 operator.  Checked as dense_cholesky_test.cc:70-117 checks the reference (mixed precision + 4 refinement steps
    reproduces the double precision answer), against the oracle's restatement of the same scheme, and for the float factor
    being what actually ran (without refinement the step carries single precision error, of the size the oracle's carries)."""
    prob, bs, order, vals, b, D = make(C, P, O, seed, "eval", oracle)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    kw = dict(type=getattr(cx, stype), num_eliminate_blocks=P)
    okw = dict(type=getattr(oracle, stype), num_eliminate_blocks=P)
    S64 = cx.Solver(ctx, **kw)
    x64, s64 = S64.solve(A, b, D)
    xr, sr = oracle.solve(bs, vals, b, D, oracle.make_options(**okw))
    assert s64.termination_type == sr.termination_type == cx.SUCCESS and relerr(x64, xr) < 1e-8
    # single precision factor, no refinement
    S32 = cx.Solver(ctx, use_mixed_precision_solves=1, **kw)
    x32, s32 = S32.solve(A, b, D)
    x32b, _ = S32.solve(A, b, D)
    xo32, so32 = oracle.solve(bs, vals, b, D, oracle.make_options(use_mixed_precision_solves=1, **okw))
    assert s32.termination_type == so32.termination_type == cx.SUCCESS and s32.num_iterations == 1
    assert b"single precision" in s32.message
    assert np.array_equal(x32, x32b)                                      # fixed order of every sum
    e_hip, e_orc = relerr(x32, x64), relerr(xo32, xr)
    assert 1e-10 < e_hip < 1e-3, e_hip                                    # a float factor, not the double one
    assert e_hip < 20 * e_orc + 1e-7, (e_hip, e_orc)                      # ... and no worse than the reference scheme's
    # + refinement (dense_cholesky_test.cc uses 4 steps): the double precision answer
    S32r = cx.Solver(ctx, use_mixed_precision_solves=1, max_num_refinement_iterations=4, **kw)
    x32r, s32r = S32r.solve(A, b, D)
    xo32r, _ = oracle.solve(bs, vals, b, D, oracle.make_options(use_mixed_precision_solves=1, max_num_refinement_iterations=4, **okw))
    assert s32r.termination_type == cx.SUCCESS and b"4 refinement steps" in s32r.message
    assert relerr(x32r, x64) < max(1e-8, 100 * relerr(xo32r, xr)), (relerr(x32r, x64), relerr(xo32r, xr))
    assert relerr(x32r, x64) < 1e-3 * e_hip                               # every step gained digits
    # refinement of the double precision factor (RefinedDenseCholesky / RefinedSparseCholesky wrap either): stays the answer
    S64r = cx.Solver(ctx, max_num_refinement_iterations=2, **kw)
    x64r, s64r = S64r.solve(A, b, D)
    assert s64r.termination_type == cx.SUCCESS and b"double precision" in s64r.message and relerr(x64r, xr) < 1e-8
    g = A.left_multiply(A.right_multiply(x64r) - b) + D * D * x64r
    g0 = A.left_multiply(A.right_multiply(x64) - b) + D * D * x64
    assert np.linalg.norm(g) <= 2 * np.linalg.norm(g0) + 1e-12 * np.linalg.norm(A.left_multiply(b))
    # not positive definite -> FAILURE, zeros, with the float factor too
    vals_bad = vals.copy()
    vals_bad[6 * O:] = 0.0
    Dneg = D.copy()
    Dneg[3 * P:] = 0.0
    A.set_values(vals_bad)
    xbad, sbad = S32r.solve(A, b, Dneg)
    assert sbad.termination_type == cx.FAILURE and not xbad.any()
    for S in (S64, S32, S32r, S64r):
        S.close()
    A.close()


def test_sparse_schur_with_hub_cameras(ctx, oracle):
    """A ring of cameras plus two hub cameras that see points everywhere: the group-minimum-degree ordering has to
    move the hubs' group to the end; whatever ordering wins, the step equals the dense solve's."""
    C, P = 330, 3000
    rng = np.random.default_rng(8)
    lists = []
    for j in range(P):
        c0 = int(j * (C - 2) / P)
        cams = {2 + (c0 + d) % (C - 2) for d in (0, 1, 2)}
        if j % 4 == 0:
            cams.add(0)
        if j % 7 == 0:
            cams.add(1)
        lists.append(sorted(cams))
    prob = _custom_problem(C, lists, 5)
    bs, order = cx.bal.build_structure(prob)
    O = prob.num_observations
    vals = cx.bal.random_jacobian_values(O, 3)
    b = rng.standard_normal(2 * O)
    D = rng.uniform(0.5, 2.0, bs.num_cols)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    Sd = cx.Solver(ctx, type=cx.DENSE_SCHUR, num_eliminate_blocks=P)
    xd, _ = Sd.solve(A, b, D)
    os.environ["CX_SPARSE_CHOLESKY"] = "1"
    try:
        Ss = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P)
        xs, ss = Ss.solve(A, b, D)
    finally:
        del os.environ["CX_SPARSE_CHOLESKY"]
    assert ss.termination_type == cx.SUCCESS
    assert np.linalg.norm(xs - xd) / xs.size < 1e-10 and relerr(xs, xd) < 1e-8
    Sd.close()
    Ss.close()
    A.close()


def test_sparse_schur_selected_by_fill(ctx, oracle):
    """From 512 cameras on SPARSE_SCHUR looks at the tile fill: a banded camera graph (every point seen by a few
    consecutive cameras) goes through the tile-sparse Cholesky without being forced, and must reproduce the dense
    reduced solve."""
    C, P = 640, 6000
    lists = [sorted({(int(j * C / P) + d) % C for d in (0, 1, 2, 3)}) for j in range(P)]
    prob = _custom_problem(C, lists, 9)
    bs, order = cx.bal.build_structure(prob)
    O = prob.num_observations
    rng = np.random.default_rng(10)
    vals = cx.bal.random_jacobian_values(O, 4)
    b = rng.standard_normal(2 * O)
    D = rng.uniform(0.5, 2.0, bs.num_cols)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    Sd = cx.Solver(ctx, type=cx.DENSE_SCHUR, num_eliminate_blocks=P)
    xd, _ = Sd.solve(A, b, D)
    Ss = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P)
    xs, ss = Ss.solve(A, b, D)
    assert ss.termination_type == cx.SUCCESS
    assert np.linalg.norm(xs - xd) / xs.size < 1e-10 and relerr(xs, xd) < 1e-8
    assert not np.array_equal(xs, xd)        # another elimination order: not the dense code path
    Sd.close()
    Ss.close()
    A.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_structures_all_solvers(ctx, oracle, seed):
    """Randomly drawn visibility structures (1..C observations per point, cameras nobody sees, single-camera
    problems, chunks of very different length) through every solver family, solved to convergence so that the
    comparison with the oracle does not depend on where a truncated CG stops: |dx| / n < 1e-9, same termination."""
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.integers(1, 40))
    P = int(rng.integers(1, 80))
    lists = []
    for j in range(P):
        k = int(rng.integers(1, min(C, 9) + 1))
        if rng.random() < 0.1:
            k = C                                   # a point every camera sees
        lists.append(sorted(rng.choice(C, size=k, replace=False).tolist()))
    prob = _custom_problem(C, lists, seed)
    bs, order = cx.bal.build_structure(prob)
    O = prob.num_observations
    vals = cx.bal.random_jacobian_values(O, seed + 1)
    b = rng.standard_normal(2 * O)
    D = rng.uniform(0.3, 2.0, bs.num_cols)           # regularised: every block is positive definite
    cases = [("DENSE_SCHUR", "IDENTITY", 0, False), ("SPARSE_SCHUR", "IDENTITY", 0, True),
             ("ITERATIVE_SCHUR", "JACOBI", 0, False), ("ITERATIVE_SCHUR", "SCHUR_JACOBI", 1, False),
             ("ITERATIVE_SCHUR", "SCHUR_POWER_SERIES_EXPANSION", 0, False), ("CGNR", "JACOBI", 0, False)]
    for stype, pre, explicit, force_sparse in cases:
        nelim = 0 if stype == "CGNR" else P
        A = cx.Matrix(ctx, bs, nelim)
        A.set_values(vals)
        if force_sparse:
            os.environ["CX_SPARSE_CHOLESKY"] = "1"
        try:
            S = cx.Solver(ctx, type=getattr(cx, stype), preconditioner_type=getattr(cx, pre), num_eliminate_blocks=nelim,
                          max_num_iterations=4 * bs.num_cols + 50, use_explicit_schur_complement=explicit)
            x, s = S.solve(A, b, D, r_tolerance=1e-13, q_tolerance=0.0)
        finally:
            os.environ.pop("CX_SPARSE_CHOLESKY", None)
        oo = oracle.make_options(type=getattr(oracle, stype), preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                                 max_num_iterations=4 * bs.num_cols + 50, use_explicit_schur_complement=explicit)
        xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=1e-13, q_tolerance=0.0)
        assert s.termination_type == sr.termination_type == cx.SUCCESS, (stype, pre, s.message, sr.message)
        assert np.all(np.isfinite(x))
        assert np.linalg.norm(x - xr) / x.size < 1e-9, (stype, pre, C, P, O)
        S.close()
        A.close()


@pytest.mark.parametrize("explicit", [0, 1])
def test_preconditioner_failure_is_reported(ctx, oracle, explicit):
    """A camera block that cannot be inverted (F = 0 and D_f = 0 for one camera): the device raises the flag during
    set-up, the CG prologue ends the run before the first iteration and the solve returns FAILURE, as
    IterativeSchurComplementSolver does when the preconditioner update fails (iterative_schur_complement_solver.cc:
    118-125) -- LM then shrinks the trust region.  No iteration is spent on the broken system."""
    C, P, O = 6, 80, 320
    prob, bs, order, vals, b, D = make(C, P, O, 5, "random")
    vals = vals.copy()
    D = D.copy()
    rows = np.nonzero(prob.camera_index[order] == 2)[0]
    for r in rows:
        vals[6 * O + 18 * r: 6 * O + 18 * r + 18] = 0.0
    D[3 * P + 18: 3 * P + 27] = 0.0
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.SCHUR_JACOBI, num_eliminate_blocks=P,
                  use_explicit_schur_complement=explicit)
    x, s = S.solve(A, b, D, q_tolerance=0.1)
    assert s.termination_type == cx.FAILURE and s.num_iterations == 0
    assert "Preconditioner update failed" in s.message.decode()
    assert not x.any()   # zeroed at the start of the solve, nothing else was written
    # the same solver object recovers on a healthy system
    prob2, bs2, order2, vals2, b2, D2 = make(C, P, O, 5, "random")
    A.set_values(vals2)
    x2, s2 = S.solve(A, b2, D2, q_tolerance=0.1)
    assert s2.termination_type == cx.SUCCESS and s2.num_iterations > 0
    S.close()
    A.close()



@pytest.mark.parametrize("stype,pre", SOLVERS + [("SPARSE_SCHUR", "IDENTITY")])
def test_points_and_a_camera_without_observations(ctx, oracle, stype, pre):
    """A structure whose column blocks are not all observed: two points in the middle, the LAST point and one camera have no
    residual block.  The static kernels write such entries themselves (the chunk-aligned tiles cover every point, the camera
    reduction every camera): J x, J'y and diag(J'J) against the oracle, and the LM-style solve of every solver type -- with
    D > 0 the unobserved blocks are still positive definite -- with the oracle's iteration count.  (Round 4 made the products of
    CGNR and of the evaluator's gradient write their results outright instead of accumulating into zeros.)"""
    base = cx.bal.make_bal_like(16, 700, 2800, 2)
    P, C = base.num_points, base.num_cameras
    drop_points = np.array([5, 6, P - 1])
    drop_camera = 7
    keep = ~np.isin(base.point_index, drop_points) & (base.camera_index != drop_camera)
    prob = cx.bal.BalProblem(C, P, base.camera_index[keep], base.point_index[keep], base.observations[keep], base.cameras, base.points)
    O = prob.num_observations
    bs, order = cx.bal.build_structure(prob)
    vals = cx.bal.random_jacobian_values(O, 11)
    rng = np.random.default_rng(12)
    b = rng.standard_normal(2 * O)
    D = rng.uniform(0.5, 2.0, bs.num_cols)
    nelim = 0 if stype == "CGNR" else P
    A = cx.Matrix(ctx, bs, nelim)
    assert A.is_static_239
    A.set_values(vals)
    x, y0 = rng.standard_normal(A.num_cols), rng.standard_normal(A.num_rows)
    assert relerr(A.right_multiply(x, y0), oracle.right_multiply(bs, vals, x, y0)) < REL
    z, c0 = rng.standard_normal(A.num_rows), rng.standard_normal(A.num_cols)
    assert relerr(A.left_multiply(z, c0), oracle.left_multiply(bs, vals, z, c0)) < REL
    sq = A.squared_column_norm()
    assert relerr(sq, oracle.squared_column_norm(bs, vals)) < REL
    assert np.all(sq[3 * 5:3 * 7] == 0.0) and np.all(sq[3 * (P - 1):3 * P] == 0.0) and np.all(sq[3 * P + 9 * drop_camera:3 * P + 9 * drop_camera + 9] == 0.0)
    kw = dict(type=getattr(cx, stype), preconditioner_type=getattr(cx, pre), num_eliminate_blocks=nelim, max_num_iterations=200)
    S = cx.Solver(ctx, **kw)
    xs, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    oo = oracle.make_options(type=getattr(oracle, stype), preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P, max_num_iterations=200)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.1)
    assert s.termination_type == sr.termination_type, (s.message, sr.message)
    assert s.num_iterations == sr.num_iterations, (s.message, sr.message)
    assert np.all(np.isfinite(xs)) and relerr(xs, xr) < 1e-8
    # the unobserved blocks: (D^2) x = 0
    assert np.all(xs[3 * 5:3 * 7] == 0.0) and np.all(xs[3 * P + 9 * drop_camera:3 * P + 9 * drop_camera + 9] == 0.0)
    S.close()
    A.close()


def test_evaluator_gradient_with_unobserved_blocks(ctx, oracle):
    """The evaluator's gradient J'r is written outright (no zeroing first): parameter blocks without residual blocks get
    their zeros from the product kernels themselves -- also when the buffer held something else before."""
    base = cx.bal.make_bal_like(16, 700, 2800, 2)
    P, C = base.num_points, base.num_cameras
    keep = ~np.isin(base.point_index, [5, 6, P - 1]) & (base.camera_index != 7)
    prob = cx.bal.BalProblem(C, P, base.camera_index[keep], base.point_index[keep], base.observations[keep], base.cameras, base.points)
    bs, order = cx.bal.build_structure(prob)
    ev = cx.Evaluator(ctx, prob)
    state = prob.state()
    for trial in range(2):   # (the second evaluation finds the first one's gradient in the library's scratch)
        st = state + trial * 1e-3 * np.random.default_rng(3).standard_normal(state.size)
        cost, res, grad = ev.evaluate(st)
        cost_r, res_r, grad_r, _ = oracle.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations, order, st)
        assert relerr(grad, grad_r) < 1e-10 and abs(cost - cost_r) <= 1e-11 * cost_r
        assert np.all(grad[3 * 5:3 * 7] == 0.0) and np.all(grad[3 * (P - 1):3 * P] == 0.0) and np.all(grad[3 * P + 9 * 7:3 * P + 9 * 8] == 0.0)
    ev.close()
