"""The oracle's SPARSE_SCHUR: SparseSchurComplementSolver with a sparse direct reduced solve
(schur_complement_solver.cc:101-159, 224-335).  The reference gets the factorisation from SuiteSparse (third party,
absent); the oracle's stand-in (oracle/orc_sparse_chol.*) is pinned the way the reference pins its own wrapper:
against a dense Cholesky of the same matrix (sparse_cholesky_test.cc:160-169) and against the reference's fixture
problems (schur_complement_solver_test.cc:186-227: |x - x_reference| / n < 1e-10).  The fill-reducing ordering is
"parity unpinned" (CHOLMOD's AMD / nested dissection are not reproduced); only its validity is checked."""
import numpy as np
import pytest

from conftest import cx, lls_problem


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def make(oracle, C, P, O, seed):
    prob = cx.bal.make_bal_like(C, P, O, seed)
    bs, order = cx.bal.build_structure(prob)
    _, b, _, vals = oracle.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations, order,
                                        prob.state(), want_gradient=False)
    rng = np.random.default_rng(seed + 1000)
    D = rng.uniform(0.5, 2.0, bs.num_cols) * 1e-2 * np.sqrt(np.abs(vals).mean())
    return prob, bs, vals, b, D


@pytest.mark.parametrize("C,P,O,seed", [(6, 40, 130, 1), (49, 7776, 31843, 49), (100, 3000, 14000, 3), (400, 9000, 40000, 4)])
def test_sparse_schur_equals_dense_schur(oracle, C, P, O, seed):
    prob, bs, vals, b, D = make(oracle, C, P, O, seed)
    xs, ss = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.SPARSE_SCHUR, num_eliminate_blocks=P))
    xd, sd = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=P))
    assert ss.termination_type == sd.termination_type == 0 and ss.num_iterations == 1
    assert np.linalg.norm(xs - xd) / xs.size < 1e-10 and relerr(xs, xd) < 1e-9
    st = oracle.sparse_schur_stats()
    assert st["s_cells"] >= C and st["factor_blocks"] >= st["s_cells"]      # L holds at least the cells of S


def test_config_1_ladybug16_shape(oracle):
    """BASELINE config 1 (Ladybug-16 SPARSE_SCHUR, CPU plumbing): the oracle runs at that shape -- 16 cameras,
    22 106 points, 83 718 observations -- and its exact Newton step solves the normal equations."""
    C, P, O, seed = cx.bal.PRESETS["ladybug16"]
    prob, bs, vals, b, D = make(oracle, C, P, O, seed)
    x, s = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.SPARSE_SCHUR, num_eliminate_blocks=P))
    assert s.termination_type == 0
    Jx = oracle.right_multiply(bs, vals, x)
    g = oracle.left_multiply(bs, vals, Jx - b) + D * D * x
    rhs = oracle.left_multiply(bs, vals, b)
    assert np.linalg.norm(g) < 1e-9 * np.linalg.norm(rhs)
    # the truncated-Newton solvers of config 2 at the same shape agree with it to their tolerance
    xi, si = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=oracle.JACOBI,
                                                             num_eliminate_blocks=P, max_num_iterations=500),
                          r_tolerance=1e-12, q_tolerance=0.0)
    assert si.termination_type == 0 and np.linalg.norm(xi - x) < 1e-4 * np.linalg.norm(x)   # CG until Q stops decreasing


@pytest.mark.parametrize("pid", [2, 4, 5, 6])
def test_fixture_problems(oracle, pid):
    """linear_least_squares_problems.cc problems with e-blocks: block sizes 1-3, rows without e-blocks (problem 6)."""
    bs, vals, b, D, nelim, raw = lls_problem(pid)
    xs, ss = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.SPARSE_SCHUR, num_eliminate_blocks=nelim))
    xd, sd = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=nelim))
    assert ss.termination_type == 0
    assert np.linalg.norm(xs - xd) / xs.size < 1e-10
    J = bs.to_dense(vals)
    ref = np.linalg.lstsq(np.vstack([J, np.diag(D)]), np.concatenate([b, np.zeros(bs.num_cols)]), rcond=None)[0]
    assert np.linalg.norm(xs - ref) / xs.size < 1e-10                        # schur_complement_solver_test.cc:186-227


def test_hub_cameras_and_two_components(oracle):
    """Two disconnected camera groups plus a hub camera seeing points of both halves of one group: the ordering has
    to handle components, and a vertex adjacent to everything; the step still equals the dense solve's."""
    rng = np.random.default_rng(8)
    C, P = 120, 1500
    cam, pt = [], []
    for j in range(P):
        half = 0 if j < P // 2 else 1
        base = half * 60
        c0 = base + rng.integers(0, 55)
        cams = {int(c0 + k) for k in range(rng.integers(2, 5))}
        if half == 0 and j % 7 == 0:
            cams.add(59)                                                     # hub of the first component
        for c in sorted(cams):
            cam.append(c)
            pt.append(j)
    cam, pt = np.array(cam, dtype=np.int32), np.array(pt, dtype=np.int32)
    O = cam.size
    prob = cx.bal.BalProblem(C, P, cam, pt, np.zeros((O, 2)), np.zeros((C, 9)), np.zeros((P, 3)))
    bs, _ = cx.bal.build_structure(prob)
    vals = cx.bal.random_jacobian_values(O, 5)
    b = rng.standard_normal(2 * O)
    D = rng.uniform(0.5, 2.0, bs.num_cols)
    xs, ss = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.SPARSE_SCHUR, num_eliminate_blocks=P))
    xd, sd = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=P))
    assert ss.termination_type == 0 and np.linalg.norm(xs - xd) / xs.size < 1e-10
    st = oracle.sparse_schur_stats()
    dense_blocks = C * (C + 1) // 2
    assert st["factor_blocks"] < 0.6 * dense_blocks                         # the two components never mix


def test_not_positive_definite_is_a_failure(oracle):
    prob, bs, vals, b, D = make(oracle, 16, 700, 2800, 2)
    O = vals.size // 24
    vals = vals.copy()
    D = D.copy()
    vals[6 * O:] = 0.0
    D[3 * 700:] = 0.0
    x, s = oracle.solve(bs, vals, b, D, oracle.make_options(type=oracle.SPARSE_SCHUR, num_eliminate_blocks=700))
    assert s.termination_type == oracle.FAILURE
