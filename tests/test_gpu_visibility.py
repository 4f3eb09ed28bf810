"""CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL on the device (cx_visibility.cpp, cx_band_chol.hip) against the oracle:
the integer structure (clusters, cluster pairs, block pairs) exactly, the preconditioned CG path by iteration
count and solution (fp64, tolerances in the tests)."""
import numpy as np
import pytest

from conftest import cx, crafted_indefinite_tridiagonal

pytestmark = pytest.mark.gpu

PRE = ["CLUSTER_JACOBI", "CLUSTER_TRIDIAGONAL"]
CLUSTERING = ["CANONICAL_VIEWS", "SINGLE_LINKAGE"]


@pytest.fixture(scope="module")
def ctx():
    c = cx.Context(0)
    yield c
    c.close()


def make(oracle, C, P, O, seed):
    prob = cx.bal.make_bal_like(C, P, O, seed)
    bs, order = cx.bal.build_structure(prob)
    _, b, _, vals = oracle.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations, order,
                                        prob.state(), want_gradient=False)
    rng = np.random.default_rng(seed + 1000)
    D = rng.uniform(0.5, 2.0, bs.num_cols) * 1e-2 * np.sqrt(np.abs(vals).mean())
    return prob, bs, vals, b, D


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


PROBLEMS = [(6, 40, 130, 1), (16, 700, 2800, 2), (49, 7776, 31843, 49), (100, 3000, 14000, 3), (400, 9000, 40000, 4)]


@pytest.mark.parametrize("clustering", CLUSTERING)
@pytest.mark.parametrize("pre", PRE)
@pytest.mark.parametrize("C,P,O,seed", PROBLEMS)
def test_structure_is_bit_exact(ctx, oracle, C, P, O, seed, pre, clustering):
    """cluster_membership_, num_clusters_, cluster_pairs_, block_pairs_: integers, compared exactly."""
    prob = cx.bal.make_bal_like(C, P, O, seed)
    bs, _ = cx.bal.build_structure(prob)
    A = cx.Matrix(ctx, bs, P)
    m, k, cp, bp = cx.binding.visibility_structure(A, getattr(cx, pre), getattr(cx, clustering))
    mr, kr, cpr, bpr = oracle.visibility_structure(bs, P, getattr(oracle, pre), getattr(oracle, clustering))
    assert k == kr
    assert np.array_equal(m, mr)
    assert np.array_equal(cp, cpr)
    assert np.array_equal(bp, bpr)
    A.close()


@pytest.mark.parametrize("clustering", CLUSTERING)
@pytest.mark.parametrize("pre", PRE)
@pytest.mark.parametrize("C,P,O,seed", PROBLEMS[:4])
def test_solve_matches_oracle(ctx, oracle, C, P, O, seed, pre, clustering):
    """The LM call (q_tolerance = eta = 0.1, r_tolerance = -1): same iteration count, solution to 1e-8 relative
    (CG amplifies summation-order differences by the condition number)."""
    prob, bs, vals, b, D = make(oracle, C, P, O, seed)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P,
                  visibility_clustering_type=getattr(cx, clustering), max_num_iterations=200)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                             visibility_clustering_type=getattr(oracle, clustering), max_num_iterations=200)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.1)
    assert s.termination_type == sr.termination_type, (s.message, sr.message)
    assert s.num_iterations == sr.num_iterations, (s.message, sr.message)
    assert np.all(np.isfinite(x))
    assert relerr(x, xr) < 1e-8
    # a second solve reuses the plan and gives the same bits (fixed summation order everywhere)
    x2, s2 = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
    assert s2.num_iterations == s.num_iterations and np.array_equal(x, x2)
    S.close()
    A.close()


@pytest.mark.parametrize("clustering", CLUSTERING)
@pytest.mark.parametrize("pre", PRE)
@pytest.mark.parametrize("C,P,O,seed", PROBLEMS[1:])
def test_tile_sparse_factorisation_of_the_preconditioner(ctx, oracle, monkeypatch, C, P, O, seed, pre, clustering):
    """The preconditioner matrix factored by the level-scheduled tile-sparse Cholesky (cx_sparse_chol.hip) instead of the
    band -- the default for CLUSTER_TRIDIAGONAL when the cluster forest has a long path, forced here on small problems
    (CX_VISIBILITY_SPARSE=1): M^-1 r is the same operator, so iteration counts equal the oracle's and the band path's,
    solutions agree to 1e-8, and two solves give the same bits."""
    prob, bs, vals, b, D = make(oracle, C, P, O, seed)
    kw = dict(type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P,
              visibility_clustering_type=getattr(cx, clustering), max_num_iterations=300)
    eta = 0.1   # short CG runs: their outcome is stable against the different rounding of the two factorisations
    monkeypatch.setenv("CX_VISIBILITY_SPARSE", "1")
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    S = cx.Solver(ctx, **kw)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=eta)
    x2, s2 = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=eta)
    S.close()
    A.close()
    monkeypatch.setenv("CX_VISIBILITY_SPARSE", "0")
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    S = cx.Solver(ctx, **kw)
    xb, sb = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=eta)
    S.close()
    A.close()
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                             visibility_clustering_type=getattr(oracle, clustering), max_num_iterations=300)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=eta)
    assert s.termination_type == sr.termination_type == 0, (s.message, sr.message)
    assert s.num_iterations == sr.num_iterations == sb.num_iterations, (s.message, sr.message, sb.message)
    assert relerr(x, xr) < 1e-7 and relerr(x, xb) < 1e-8
    assert np.array_equal(x, x2)


def test_tile_sparse_retry_and_failure_paths(ctx, oracle, monkeypatch):
    """The CLUSTER_TRIDIAGONAL retry (indefinite until the off-diagonal cluster cells are halved) through the tile-sparse
    factorisation: same outcome as the band path and the oracle."""
    monkeypatch.setenv("CX_VISIBILITY_SPARSE", "1")
    prob, bs, vals, b, D, P = crafted_indefinite_tridiagonal(0)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.CLUSTER_TRIDIAGONAL, num_eliminate_blocks=P,
                  visibility_clustering_type=cx.SINGLE_LINKAGE, max_num_iterations=100)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.01)
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=oracle.CLUSTER_TRIDIAGONAL, num_eliminate_blocks=P,
                             visibility_clustering_type=oracle.SINGLE_LINKAGE, max_num_iterations=100)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.01)
    assert s.termination_type == sr.termination_type == 0, (s.message, sr.message)
    assert s.num_iterations == sr.num_iterations
    assert relerr(x, xr) < 1e-4
    S.close()
    A.close()


def test_global_memory_walk_agrees(ctx, oracle, monkeypatch):
    """k_band_solve (vector in global memory, used when the band does not fit the LDS window) and
    k_band_solve_lds do the same arithmetic (the backward dot products are grouped differently: 1e-12)."""
    C, P, O, seed = 100, 3000, 14000, 3
    prob, bs, vals, b, D = make(oracle, C, P, O, seed)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    out = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("CX_BAND_SOLVE_GLOBAL", "1")
        S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.CLUSTER_TRIDIAGONAL, num_eliminate_blocks=P, max_num_iterations=200)
        out.append(S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=1e-3))
        S.close()
    assert out[0][1].num_iterations == out[1][1].num_iterations
    assert relerr(out[0][0], out[1][0]) < 1e-12
    A.close()


@pytest.mark.parametrize("pre", PRE)
def test_tight_solve_and_fewer_iterations_than_schur_jacobi(ctx, oracle, pre):
    """eta = 1e-3 on 400 cameras (26 clusters, one forest path): iteration count equals the oracle's and is
    lower than SCHUR_JACOBI's -- the reason the preconditioner exists."""
    C, P, O, seed = 400, 9000, 40000, 4
    prob, bs, vals, b, D = make(oracle, C, P, O, seed)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P, max_num_iterations=500)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=1e-3)
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                             max_num_iterations=500)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=1e-3)
    assert s.termination_type == sr.termination_type == 0, (s.message, sr.message)
    assert s.num_iterations == sr.num_iterations, (s.message, sr.message)
    assert relerr(x, xr) < 1e-7
    J = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.SCHUR_JACOBI, num_eliminate_blocks=P, max_num_iterations=500)
    _, sj = J.solve(A, b, D, r_tolerance=-1.0, q_tolerance=1e-3)
    assert s.num_iterations < sj.num_iterations, (s.num_iterations, sj.num_iterations)
    J.close()
    S.close()
    A.close()


def test_tridiagonal_retry_with_halved_off_diagonal_cells(ctx, oracle):
    """The unscaled CLUSTER_TRIDIAGONAL matrix of this problem is indefinite (tests/test_oracle_visibility.py shows
    it): the device must notice, halve the cells between clusters, factor again and follow the oracle."""
    prob, bs, vals, b, D, P = crafted_indefinite_tridiagonal(0)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.CLUSTER_TRIDIAGONAL, num_eliminate_blocks=P,
                  visibility_clustering_type=cx.SINGLE_LINKAGE, max_num_iterations=100)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.01)
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=oracle.CLUSTER_TRIDIAGONAL, num_eliminate_blocks=P,
                             visibility_clustering_type=oracle.SINGLE_LINKAGE, max_num_iterations=100)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.01)
    assert s.termination_type == sr.termination_type == 0, (s.message, sr.message)
    assert s.num_iterations == sr.num_iterations, (s.message, sr.message)
    assert relerr(x, xr) < 1e-4   # ill conditioned on purpose (E'E + 1e-6 I): rounding differences are amplified
    # CLUSTER_JACOBI on the same problem needs no retry
    Sj = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.CLUSTER_JACOBI, num_eliminate_blocks=P,
                   visibility_clustering_type=cx.SINGLE_LINKAGE, max_num_iterations=100)
    xj, sj = Sj.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.01)
    oo.preconditioner_type = oracle.CLUSTER_JACOBI
    xjr, sjr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.01)
    assert sj.num_iterations == sjr.num_iterations and relerr(xj, xjr) < 1e-4
    Sj.close()
    S.close()
    A.close()


def test_setup_failure_survives_the_tridiagonal_retry(ctx):
    """The retry is the reference's answer to a failed Factorize() of the preconditioner only
    (visibility_based_preconditioner.cc:331-360).  A point block E'E + D_e^2 that cannot be inverted is a different
    failure with its own device flag: the retry must neither react to it nor erase it, and the solve must end with
    "Preconditioner update failed." instead of running CG on a broken (E'E)^-1."""
    prob, bs, vals, b, D, P = crafted_indefinite_tridiagonal(0)
    O = vals.size // 24
    vals, D = vals.copy(), D.copy()
    row_pt = bs.cells["block_id"][0::2]
    p_bad = P - 1                                   # a private point of camera 2: one row
    for r in np.nonzero(row_pt == p_bad)[0]:
        vals[6 * r: 6 * r + 6] = 0.0                # E = 0 ...
    D[3 * p_bad: 3 * p_bad + 3] = 0.0               # ... and D_e = 0: E'E + D_e^2 = 0
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.CLUSTER_TRIDIAGONAL, num_eliminate_blocks=P,
                  visibility_clustering_type=cx.SINGLE_LINKAGE, max_num_iterations=100)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.01)
    assert s.termination_type == cx.FAILURE and s.num_iterations == 0, s.message
    assert "Preconditioner update failed" in s.message.decode()
    assert not x.any()                              # x comes back zeroed, not uninitialised staging memory
    S.close()
    A.close()


def test_one_cluster_is_the_exact_inverse(ctx, oracle):
    """All cameras see all points: one cluster, the preconditioner is S, CG converges at once."""
    C, P = 5, 60
    cam = np.tile(np.arange(C, dtype=np.int32), P)
    pt = np.repeat(np.arange(P, dtype=np.int32), C)
    O = C * P
    prob = cx.bal.BalProblem(C, P, cam, pt, np.zeros((O, 2)), np.zeros((C, 9)), np.zeros((P, 3)))
    bs, _ = cx.bal.build_structure(prob)
    vals = cx.bal.random_jacobian_values(O, 3)
    b = np.random.default_rng(3).standard_normal(2 * O)
    D = np.full(bs.num_cols, 0.1)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    m, k, cp, bp = cx.binding.visibility_structure(A, cx.CLUSTER_JACOBI, cx.SINGLE_LINKAGE)
    assert k == 1 and len(bp) == C * (C + 1) // 2
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.CLUSTER_JACOBI, num_eliminate_blocks=P,
                  visibility_clustering_type=cx.SINGLE_LINKAGE, max_num_iterations=50)
    x, s = S.solve(A, b, D, r_tolerance=1e-10, q_tolerance=0.0)
    assert s.termination_type == 0 and s.num_iterations <= 2, s.message
    od = oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=P)
    xd, _ = oracle.solve(bs, vals, b, D, od)
    assert relerr(x, xd) < 1e-9
    S.close()
    A.close()


def test_failure_and_option_validation(ctx, oracle):
    """A camera whose S block is singular (F = 0, D_f = 0): "Preconditioner update failed." as in
    IterativeSchurComplementSolver::SolveImpl (:115-121); CLUSTER_* with CGNR or an explicit S is refused
    (solver.cc option validation, solver_test.cc:1082-1157)."""
    C, P, O, seed = 16, 700, 2800, 2
    prob, bs, vals, b, D = make(oracle, C, P, O, seed)
    row_cam = bs.cells["block_id"][1::2] - P
    vals = vals.copy()
    F = vals[6 * O:].reshape(O, 18)
    F[row_cam == 2] = 0.0
    D = D.copy()
    D[3 * P + 18:3 * P + 27] = 0.0
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    for pre in PRE:
        S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P, max_num_iterations=50)
        x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=0.1)
        assert s.termination_type == cx.FAILURE and s.num_iterations == 0
        assert b"Preconditioner update failed" in s.message
        S.close()
    with pytest.raises(cx.CxError):
        cx.Solver(ctx, type=cx.CGNR, preconditioner_type=cx.CLUSTER_JACOBI)
    with pytest.raises(cx.CxError):
        cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.CLUSTER_TRIDIAGONAL, num_eliminate_blocks=P,
                  use_explicit_schur_complement=1)
    with pytest.raises(cx.CxError):
        cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.CLUSTER_JACOBI, num_eliminate_blocks=P,
                  visibility_clustering_type=7)
    A.close()


def test_lm_loop_with_cluster_jacobi(ctx, oracle):
    """cx_minimize with CLUSTER_JACOBI inside: the iterations follow the oracle's TrustRegionMinimizer."""
    C, P, O = 12, 300, 2400
    prob = cx.bal.make_bal_like(C, P, O, 3)
    ev = cx.Evaluator(ctx, prob)
    solver = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.CLUSTER_JACOBI, num_eliminate_blocks=P,
                       max_num_iterations=500)
    x, summ, its = cx.binding.minimize(ev, solver, prob.state(), cx.binding.minimizer_options(max_num_iterations=8))
    so = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=oracle.CLUSTER_JACOBI, num_eliminate_blocks=P,
                             max_num_iterations=500)
    x_r, summ_r, its_r = oracle.minimize_bal(C, P, prob.camera_index, prob.point_index, prob.observations, prob.state(), so,
                                             oracle.minimizer_options(max_num_iterations=8))
    assert summ["termination_type"] == summ_r["termination_type"] and len(its) == len(its_r)
    for a, b in zip(its, its_r):
        assert a["step_is_successful"] == b["step_is_successful"]
        assert abs(a["cost"] - b["cost"]) <= 1e-5 * abs(b["cost"])
    solver.close()
    ev.close()


def _custom_problem(C, cam_lists):
    cam, pt = [], []
    for j, cams in enumerate(cam_lists):
        cam.extend(sorted(int(c) for c in cams))
        pt.extend([j] * len(cams))
    cam, pt = np.array(cam, dtype=np.int32), np.array(pt, dtype=np.int32)
    O = cam.size
    return cx.bal.BalProblem(C, len(cam_lists), cam, pt, np.zeros((O, 2)), np.zeros((C, 9)), np.zeros((len(cam_lists), 3)))


@pytest.mark.parametrize("clustering", CLUSTERING)
@pytest.mark.parametrize("pre", PRE)
@pytest.mark.parametrize("case", ["unseen_camera", "one_camera", "two_components", "identical_views", "big_cluster"])
def test_edge_structures(ctx, oracle, case, pre, clustering):
    """A camera nobody observes (no edges, its block is D^2), a single camera, two groups of cameras that share
    no point (forest with two paths), groups of cameras with identical visibility (similarity exactly 1: exact
    ties between candidate views), and one cluster of 120 cameras (band 1080+: more than two 256-column passes of
    the solve kernel's window).  Structure exact, solve against the oracle."""
    rng = np.random.default_rng({"unseen_camera": 1, "one_camera": 2, "two_components": 3, "identical_views": 4, "big_cluster": 5}[case])
    if case == "unseen_camera":
        C, lists = 6, [list(rng.choice([0, 1, 2, 4, 5], size=int(rng.integers(2, 5)), replace=False)) for _ in range(60)]
    elif case == "one_camera":
        C, lists = 1, [[0] for _ in range(40)]
    elif case == "two_components":
        C = 24
        lists = [list(rng.choice(12, size=4, replace=False) + (12 if j % 2 else 0)) for j in range(300)]
    elif case == "identical_views":
        C = 12
        groups = [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]]
        lists = [groups[j % 4] + ([groups[(j + 1) % 4][0]] if j % 5 == 0 else []) for j in range(120)]
    else:
        C = 120
        lists = [list(range(C)) for _ in range(40)] + [list(rng.choice(C, size=3, replace=False)) for _ in range(200)]
    prob = _custom_problem(C, lists)
    P, O = prob.num_points, prob.num_observations
    bs, _ = cx.bal.build_structure(prob)
    vals = cx.bal.random_jacobian_values(O, 3)
    b = rng.standard_normal(2 * O)
    D = rng.uniform(0.5, 2.0, bs.num_cols)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    got = cx.binding.visibility_structure(A, getattr(cx, pre), getattr(cx, clustering))
    ref = oracle.visibility_structure(bs, P, getattr(oracle, pre), getattr(oracle, clustering))
    assert got[1] == ref[1]
    for a, r in zip((got[0], got[2], got[3]), (ref[0], ref[2], ref[3])):
        assert np.array_equal(a, r)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P,
                  visibility_clustering_type=getattr(cx, clustering), max_num_iterations=300)
    x, s = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=1e-4)
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                             visibility_clustering_type=getattr(oracle, clustering), max_num_iterations=300)
    xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=1e-4)
    assert s.termination_type == sr.termination_type == 0, (s.message, sr.message)
    assert s.num_iterations == sr.num_iterations, (s.message, sr.message)
    assert relerr(x, xr) < 1e-8
    S.close()
    A.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_structures(ctx, oracle, seed):
    """Randomly drawn visibility structures (as test_gpu_parity.test_random_structures_all_solvers: 1..C observations per
    point, cameras nobody sees, single-camera problems) through both CLUSTER_* types and both clusterings: the
    structure exactly, the solve to convergence (|dx| / n < 1e-9, same termination as the oracle)."""
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.integers(1, 40))
    P = int(rng.integers(1, 80))
    lists = []
    for j in range(P):
        k = int(rng.integers(1, min(C, 9) + 1))
        if rng.random() < 0.1:
            k = C
        lists.append(sorted(rng.choice(C, size=k, replace=False).tolist()))
    prob = _custom_problem(C, lists)
    bs, _ = cx.bal.build_structure(prob)
    O = prob.num_observations
    vals = cx.bal.random_jacobian_values(O, seed + 1)
    b = rng.standard_normal(2 * O)
    D = rng.uniform(0.3, 2.0, bs.num_cols)
    A = cx.Matrix(ctx, bs, P)
    A.set_values(vals)
    for pre in PRE:
        for clustering in CLUSTERING:
            got = cx.binding.visibility_structure(A, getattr(cx, pre), getattr(cx, clustering))
            ref = oracle.visibility_structure(bs, P, getattr(oracle, pre), getattr(oracle, clustering))
            assert got[1] == ref[1] and all(np.array_equal(a, r) for a, r in zip((got[0], got[2], got[3]), (ref[0], ref[2], ref[3])))
            S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P,
                          visibility_clustering_type=getattr(cx, clustering), max_num_iterations=4 * bs.num_cols + 50)
            x, s = S.solve(A, b, D, r_tolerance=1e-13, q_tolerance=0.0)
            oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                                     visibility_clustering_type=getattr(oracle, clustering), max_num_iterations=4 * bs.num_cols + 50)
            xr, sr = oracle.solve(bs, vals, b, D, oo, r_tolerance=1e-13, q_tolerance=0.0)
            assert s.termination_type == sr.termination_type == cx.SUCCESS, (pre, clustering, s.message, sr.message)
            assert np.linalg.norm(x - xr) / x.size < 1e-9, (pre, clustering, C, P, O)
            S.close()
    A.close()
