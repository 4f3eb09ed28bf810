"""Every configuration BASELINE.json names, by its preset (ceres-solver-ceres-solver_amd/bal.py PRESETS), through the
C ABI on the GPU:
  config 2  ladybug49     ITERATIVE_SCHUR (JACOBI, SCHUR_JACOBI) and CGNR against the oracle: equal iteration counts, 1e-8
  config 3  dubrovnik356  DENSE_SCHUR, SPARSE_SCHUR (the dense MFMA Cholesky) and the tile-sparse Cholesky forced onto the
                          same system, against the oracle's dense reduced solve (|dx| / n < 1e-10, the bound of
                          schur_complement_solver_test.cc:186-227); ITERATIVE_SCHUR iteration counts
  config 5  synthetic10M  fp32-stored Jacobian, fp64 accumulation: size-independent properties (the oracle does not finish
                          a 10 M residual-block solve in test time) -- adjointness of the fp32 operator pair, agreement with
                          the fp64 operator to fp32 rounding, normal equations after refinement
  config 1  ladybug16     SPARSE_SCHUR / DENSE_SCHUR exact steps against the oracle's SPARSE_SCHUR (the reference documents this
                          config as a CPU run; tests/test_oracle_sparse_schur.py runs the oracle alone on it)
(config 4, Final-13682: tests/test_gpu_full_size.py)"""
import os

import numpy as np
import pytest

from conftest import cx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = cx.Context(0)
    yield c
    c.close()


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def lm_system(oracle, preset):
    """J, r of the preset at its perturbed start point (oracle evaluation) and the first LM diagonal of
    LevenbergMarquardtStrategy (clamped diag(J'J) / radius, levenberg_marquardt_strategy.cc:81-98) after Jacobi scaling."""
    C, P, O, seed = cx.bal.PRESETS[preset]
    prob = cx.bal.make_preset(preset)
    bs, order = cx.bal.build_structure(prob)
    _, b, _, vals = oracle.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations, order,
                                        prob.state(), want_gradient=False)
    scale = 1.0 / (1.0 + np.sqrt(oracle.squared_column_norm(bs, vals)))
    vals = oracle.scale_columns(bs, vals, scale)
    D = np.sqrt(np.clip(oracle.squared_column_norm(bs, vals), 1e-6, 1e32) / 1e4)
    return prob, bs, vals, b, D


# ------------------------------------------------------------------------------------------------ config 1
@pytest.mark.parametrize("variant", ["SPARSE_SCHUR", "DENSE_SCHUR", "SPARSE_SCHUR tile-sparse"])
def test_config_1_ladybug16_schur(ctx, oracle, variant):
    """BASELINE config 1 (Ladybug-16: 16 cameras, 22 106 points, 83 718 residual blocks; the reference documents it as a
    CPU SPARSE_SCHUR run, installation.rst:198-216) on the device: DetectStructure reports <2,3,9> as the reference's log
    does (installation.rst:211-216, "Eliminate group 2,3,9"), and the exact step of SPARSE_SCHUR (144 x 144 reduced system:
    dense storage; the tile-sparse factorisation forced onto it as a third variant) and DENSE_SCHUR equals the oracle's
    SPARSE_SCHUR step to 1e-8 and |dx| / n < 1e-10 (schur_complement_solver_test.cc:186-227)."""
    prob, bs, vals, b, D = lm_system(oracle, "ladybug16")
    P = prob.num_points
    assert cx.binding.detect_structure(bs, P) == (2, 3, 9)
    xr, sr = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.SPARSE_SCHUR, num_eliminate_blocks=P))
    assert sr.termination_type == 0
    A = cx.Matrix(ctx, bs, P)
    assert A.is_static_239
    A.set_values(vals)
    forced = variant.endswith("tile-sparse")
    if forced:
        os.environ["CX_SPARSE_CHOLESKY"] = "1"
    try:
        S = cx.Solver(ctx, type=getattr(cx, variant.split()[0]), num_eliminate_blocks=P)
        x, s = S.solve(A, b, D)
    finally:
        if forced:
            del os.environ["CX_SPARSE_CHOLESKY"]
    assert s.termination_type == cx.SUCCESS and s.num_iterations == 1
    assert np.linalg.norm(x - xr) / x.size < 1e-10
    assert relerr(x, xr) < 1e-8
    S.close()
    A.close()


# ------------------------------------------------------------------------------------------------ config 2
@pytest.fixture(scope="module")
def ladybug49(oracle):
    return lm_system(oracle, "ladybug49")


@pytest.mark.parametrize("stype,pre", [("ITERATIVE_SCHUR", "JACOBI"), ("ITERATIVE_SCHUR", "SCHUR_JACOBI"), ("CGNR", "JACOBI")])
@pytest.mark.parametrize("eta", [0.1, 0.01])
def test_config_2_ladybug49_solvers(ctx, oracle, ladybug49, stype, pre, eta):
    """The LM call (q_tolerance = eta, r_tolerance = -1) at Solver::Options' eta = 0.1 and bundle_adjuster's 1e-2."""
    prob, bs, vals, b, D = ladybug49
    P = prob.num_points
    nelim = 0 if stype == "CGNR" else P
    A = cx.Matrix(ctx, bs, nelim)
    A.set_values(vals)
    S = cx.Solver(ctx, type=getattr(cx, stype), preconditioner_type=getattr(cx, pre), num_eliminate_blocks=nelim, max_num_iterations=500)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=eta)
    oo = oracle.make_options(type=getattr(oracle, stype), preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                             max_num_iterations=500)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=eta)
    assert s.termination_type == sr.termination_type == cx.SUCCESS, (s.message, sr.message)
    assert s.num_iterations == sr.num_iterations, (s.message, sr.message)
    assert relerr(x, xr) < 1e-8
    S.close()
    A.close()


# ------------------------------------------------------------------------------------------------ config 3
@pytest.fixture(scope="module")
def dubrovnik356(oracle):
    prob, bs, vals, b, D = lm_system(oracle, "dubrovnik356")
    oracle.set_num_threads(8)
    xr, sr = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=prob.num_points))
    oracle.set_num_threads(4)
    assert sr.termination_type == 0
    return prob, bs, vals, b, D, xr


@pytest.mark.parametrize("variant", ["DENSE_SCHUR", "SPARSE_SCHUR", "SPARSE_SCHUR tile-sparse"])
def test_config_3_dubrovnik356_schur(ctx, oracle, dubrovnik356, variant):
    """n = 9 * 356 = 3204: the reduced camera matrix is factored dense on fp64 MFMA (SPARSE_SCHUR maps there below 512
    cameras), and, forced, by the tile-sparse level-scheduled Cholesky; both against the oracle's dense reduced solve."""
    prob, bs, vals, b, D, xr = dubrovnik356
    P = prob.num_points
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    forced = variant.endswith("tile-sparse")
    if forced:
        os.environ["CX_SPARSE_CHOLESKY"] = "1"
    try:
        S = cx.Solver(ctx, type=getattr(cx, variant.split()[0]), num_eliminate_blocks=P)
        x, s = S.solve(A, b, D)
        x2, _ = S.solve(A, b, D)
    finally:
        if forced:
            del os.environ["CX_SPARSE_CHOLESKY"]
    assert s.termination_type == cx.SUCCESS and s.num_iterations == 1
    assert np.linalg.norm(x - xr) / x.size < 1e-10
    assert relerr(x, xr) < 1e-8
    assert np.array_equal(x, x2)                       # every sum has a fixed order
    # use_mixed_precision_solves + refinement (solver.h:572-590; dense_cholesky_test.cc runs the reference with 4 steps): the single
    # precision tile factorisation of this S and fp64 refinement through the implicit operator give the double precision step
    Sm = cx.Solver(ctx, type=getattr(cx, variant.split()[0]), num_eliminate_blocks=P, use_mixed_precision_solves=1,
                   max_num_refinement_iterations=4)
    xm, sm = Sm.solve(A, b, D)
    Sm0 = cx.Solver(ctx, type=getattr(cx, variant.split()[0]), num_eliminate_blocks=P, use_mixed_precision_solves=1)
    xm0, sm0 = Sm0.solve(A, b, D)
    assert sm.termination_type == sm0.termination_type == cx.SUCCESS and b"single precision" in sm.message
    assert 1e-10 < relerr(xm0, x) < 1e-2 and relerr(xm, x) < 1e-3 * relerr(xm0, x), (relerr(xm0, x), relerr(xm, x))
    Sm.close()
    Sm0.close()
    # the exact Newton step solves the normal equations (a code path independent of both Schur implementations)
    g = A.left_multiply(A.right_multiply(x) - b) + D * D * x
    assert np.linalg.norm(g) <= 1e-9 * np.linalg.norm(A.left_multiply(b))
    S.close()
    A.close()


def test_config_3_dubrovnik356_iterative_schur(ctx, oracle, dubrovnik356):
    prob, bs, vals, b, D, xr = dubrovnik356
    P = prob.num_points
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    for pre in ("JACOBI", "SCHUR_JACOBI"):
        S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P, max_num_iterations=500)
        x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
        oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                                 max_num_iterations=500)
        xo, so = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.1)
        assert s.termination_type == so.termination_type == cx.SUCCESS and s.num_iterations == so.num_iterations, (s.message, so.message)
        assert relerr(x, xo) < 1e-8
        S.close()
    A.close()


# ------------------------------------------------------------------------------------------------ config 5
@pytest.fixture(scope="module")
def synthetic10M():
    ctx = cx.Context(0)
    prob = cx.bal.make_preset("synthetic10M")
    ev = cx.Evaluator(ctx, prob)
    state = ctx.to_device(prob.state())
    res = ctx.empty(2 * prob.num_observations)
    ev.evaluate(state, residuals=res, gradient=None, want_jacobian=True)
    A = ev.jacobian()
    scale = 1.0 / (1.0 + np.sqrt(A.squared_column_norm()))
    A.scale_columns(scale)
    D = np.sqrt(np.clip(A.squared_column_norm(), 1e-6, 1e32) / 1e4)
    yield ctx, prob, ev, A, res.to_host(), D
    ev.close()
    ctx.close()


def test_config_5_synthetic10M_mixed_precision_cgnr(synthetic10M):
    """10 M residual blocks, 5 000 cameras, 1.5 M points: CGNR whose operator streams fp32 copies of the J values
    (fp64 accumulation and vectors).  Not in the reference (its mixed precision is for Cholesky, solver.h:572-590), so the
    checks are properties: the truncated solve follows the fp64 solve (same iteration count +- 1, solution to fp32
    rounding), refinement steps on the fp64 residual bring the normal equations down, and results are reproducible."""
    ctx, prob, ev, A, b, D = synthetic10M
    kw = dict(type=cx.CGNR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=0, max_num_iterations=500)
    S64 = cx.Solver(ctx, **kw)
    S32 = cx.Solver(ctx, use_mixed_precision_solves=1, **kw)
    x64, s64 = S64.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    x32, s32 = S32.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    assert s64.termination_type == cx.SUCCESS and s32.termination_type == cx.SUCCESS
    assert abs(s32.num_iterations - s64.num_iterations) <= 1
    assert relerr(x32, x64) < 1e-4
    x32b, _ = S32.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    assert np.array_equal(x32, x32b)
    # converged solves: fp64 normal-equation residual of the fp32-operator solution, without and with refinement
    jtb = A.left_multiply(b)

    def normal_residual(x):
        return np.linalg.norm(A.left_multiply(A.right_multiply(x) - b) + D * D * x) / np.linalg.norm(jtb)

    xc, sc = S32.solve(A, b, D, r_tolerance=1e-10, q_tolerance=0.0)
    Sref = cx.Solver(ctx, use_mixed_precision_solves=1, max_num_refinement_iterations=2, **kw)
    xr, sr = Sref.solve(A, b, D, r_tolerance=1e-10, q_tolerance=0.0)
    r0, r2 = normal_residual(xc), normal_residual(xr)
    assert r0 < 1e-5                 # fp32 storage of J: relative perturbation 6e-8, amplified by the conditioning
    assert r2 < 1e-9 and r2 < 1e-2 * r0
    for S in (S64, S32, Sref):
        S.close()


def test_config_5_synthetic10M_products(synthetic10M):
    """J x and J' y at 10 M residual blocks: adjointness and the column norms (the fp64 operator the refinement uses)."""
    ctx, prob, ev, A, b, D = synthetic10M
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(A.num_cols), rng.standard_normal(A.num_rows)
    jx, jty = A.right_multiply(x), A.left_multiply(y)
    assert abs(float(jx @ y) - float(x @ jty)) <= 1e-11 * np.linalg.norm(jx) * np.linalg.norm(y)
    sq = A.squared_column_norm()
    for j in (0, 3 * prob.num_points + 4, A.num_cols - 1):
        e = np.zeros(A.num_cols)
        e[j] = 1.0
        col = A.right_multiply(e)
        assert abs(float(col @ col) - sq[j]) <= 1e-12 * max(sq[j], 1e-300)


@pytest.mark.parametrize("shards", [2, 4, 8])
def test_config_5_synthetic10M_cgnr_on_shards(synthetic10M, shards):
    """BASELINE config 5 ("1 -> 8 GPU scaling sweep") on 2 / 4 / 8 logical shards of the one GPU: CGNR + JACOBI with fp64
    and with fp32-stored J equals the unsharded device solve -- same iteration count, x to 1e-8 (the point part of every
    CGNR dot product is a sum over the shards, so the last bits may differ)."""
    ctx, prob, ev, A, b, D = synthetic10M
    vals = A.get_values()                      # the Jacobi-scaled values of the fixture, handed over as a host matrix
    bs, _ = cx.bal.build_structure(prob)
    mctx = cx.Context(devices=[0] * shards)
    MA = cx.Matrix(mctx, bs, prob.num_points)
    MA.set_values(vals)
    for mixed in (0, 1):
        kw = dict(type=cx.CGNR, preconditioner_type=cx.JACOBI, max_num_iterations=500, use_mixed_precision_solves=mixed)
        S1 = cx.Solver(ctx, num_eliminate_blocks=0, **kw)
        MS = cx.Solver(mctx, num_eliminate_blocks=prob.num_points, **kw)
        x1, s1 = S1.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.01)
        xm, sm = MS.solve(MA, b, D, r_tolerance=-1.0, q_tolerance=0.01)
        assert sm.termination_type == s1.termination_type == cx.SUCCESS, (sm.message, s1.message)
        assert sm.num_iterations == s1.num_iterations, (shards, mixed, sm.num_iterations, s1.num_iterations)
        assert relerr(xm, x1) < 1e-8
        S1.close()
        MS.close()
    MA.close()
    mctx.close()
