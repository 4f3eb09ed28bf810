"""The oracle (oracle/liborc.so) against the reference's own fixtures and against
dense algebra in numpy -- the procedures of schur_eliminator_test.cc:82-177,
implicit_schur_complement_test.cc:70-213, partitioned_matrix_view_test.cc,
block_sparse_matrix_test.cc:488-631, conjugate_gradients_solver_test.cc:46-153,
schur_complement_solver_test.cc and iterative_schur_complement_solver_test.cc.
CPU only."""
import numpy as np
import pytest

from conftest import load_golden, lls_problem, cx


def dense_reference(bs, values, b, D, nelim):
    """Dense H = J'J + D'D block elimination (schur_eliminator_test.cc:82-135)."""
    J = bs.to_dense(values)
    ne = int(bs.col_blocks["size"][:nelim].sum())
    H = J.T @ J + (np.diag(D ** 2) if D is not None else 0)
    g = J.T @ b
    P, Q, R = H[:ne, :ne], H[:ne, ne:], H[ne:, ne:]
    # block-diagonal P: invert block by block like the reference test
    Pinv = np.zeros_like(P)
    for blk in bs.col_blocks[:nelim]:
        p, s = int(blk["position"]), int(blk["size"])
        Pinv[p:p + s, p:p + s] = np.linalg.inv(P[p:p + s, p:p + s])
    S = R - Q.T @ Pinv @ Q
    rhs = g[ne:] - Q.T @ Pinv @ g[:ne]
    sol = np.linalg.solve(H, g)
    return J, H, g, S, rhs, sol


def upper_block_triangle(S, bs, nelim):
    """The reference only writes blocks (i, j) with i <= j of the reduced matrix."""
    out = np.zeros_like(S)
    cols = bs.col_blocks[nelim:]
    p0 = int(cols[0]["position"])
    for i, bi in enumerate(cols):
        for bj in cols[i:]:
            r, rs = int(bi["position"]) - p0, int(bi["size"])
            c, cs = int(bj["position"]) - p0, int(bj["size"])
            out[r:r + rs, c:c + cs] = S[r:r + rs, c:c + cs]
    return out


@pytest.mark.parametrize("m", range(3))
def test_bsm_dense_and_crs_goldens(oracle, m):
    g = load_golden("block_sparse_matrix_goldens.json")["matrices"][m]
    rows = [(rs, [tuple(c) for c in cells]) for rs, cells in g["rows"]]
    bs = cx.BlockStructure.from_rows(g["col_sizes"], rows)
    values = np.arange(1, g["nnz"] + 1, dtype=np.float64)
    assert bs.num_nonzeros == g["nnz"]
    assert np.array_equal(bs.to_dense(values), np.array(g["dense"], dtype=np.float64))
    r, c, v = oracle.to_crs(bs, values, transpose=False)
    assert r.tolist() == g["crs"]["rows"] and c.tolist() == g["crs"]["cols"] and v.tolist() == g["crs"]["values"]
    r, c, v = oracle.to_crs(bs, values, transpose=True)
    assert r.tolist() == g["crs_t"]["rows"] and c.tolist() == g["crs_t"]["cols"] and v.tolist() == g["crs_t"]["values"]


@pytest.mark.parametrize("pid", [0, 2, 3, 4, 5, 6])
def test_bsm_products_match_dense(oracle, pid):
    bs, values, b, D, nelim, raw = lls_problem(pid)
    J = bs.to_dense(values)
    if "dense" in raw:
        assert np.array_equal(J, np.array(raw["dense"], dtype=np.float64))
    rng = np.random.default_rng(pid)
    x = rng.standard_normal(bs.num_cols)
    y = rng.standard_normal(bs.num_rows)
    np.testing.assert_allclose(oracle.right_multiply(bs, values, x), J @ x, rtol=0, atol=1e-13)
    np.testing.assert_allclose(oracle.left_multiply(bs, values, y), J.T @ y, rtol=0, atol=1e-12)
    np.testing.assert_allclose(oracle.squared_column_norm(bs, values), (J * J).sum(axis=0), rtol=1e-15)
    s = rng.uniform(0.5, 2.0, bs.num_cols)
    np.testing.assert_allclose(bs.to_dense(oracle.scale_columns(bs, values, s)), J * s, rtol=1e-15)


def test_problem2_comment_goldens(oracle):
    """linear_least_squares_problems.cc:138-187: A'A, A'b, S, r, S\\r, A\\b (4 decimals, D = 0)."""
    bs, values, b, D, nelim, raw = lls_problem(2)
    J = bs.to_dense(values)
    np.testing.assert_array_equal(J.T @ J, np.array(raw["AtA"], dtype=float))
    np.testing.assert_array_equal(J.T @ b, np.array(raw["Atb"], dtype=float))
    lhs, rhs = oracle.schur_eliminate_dense(bs, values, b, None, nelim)
    S = np.triu(lhs) + np.triu(lhs, 1).T
    np.testing.assert_allclose(S, np.array(raw["S"]), atol=5e-5)
    np.testing.assert_allclose(rhs, np.array(raw["r"]), atol=5e-5)
    z, t = oracle.dense_cholesky_solve(lhs, rhs)
    assert t == oracle.SUCCESS
    np.testing.assert_allclose(z, np.array(raw["S_solve_r"]), atol=5e-5)
    x = oracle.schur_back_substitute(bs, values, b, None, nelim, z)
    x[bs.num_cols - len(z):] = z
    np.testing.assert_allclose(x, np.array(raw["x"]), atol=5e-5)


def test_problem0_known_solutions(oracle):
    """linear_least_squares_problems.cc:72-136: x and x_D through CGNR run to convergence."""
    bs, values, b, D, nelim, raw = lls_problem(0)
    o = oracle.make_options(type=oracle.CGNR, preconditioner_type=oracle.JACOBI, max_num_iterations=50)
    x, s = oracle.solve(bs, values, b, None, o, r_tolerance=1e-14)
    assert s.termination_type == oracle.SUCCESS
    np.testing.assert_allclose(x, raw["x"], atol=1e-10)
    x, s = oracle.solve(bs, values, b, D, o, r_tolerance=1e-14)
    np.testing.assert_allclose(x, raw["x_D"], atol=5e-9)


@pytest.mark.parametrize("pid", [2, 4, 5, 6])
@pytest.mark.parametrize("use_D", [False, True])
def test_schur_eliminator_vs_dense(oracle, pid, use_D):
    """schur_eliminator_test.cc:137-177 (1e-14 relative)."""
    bs, values, b, D, nelim, raw = lls_problem(pid)
    if not use_D and pid in (4, 6):
        pytest.skip("rank deficient without regularisation (fixture note)")
    Duse = D if use_D else None
    J, H, g, S, rhs_ref, sol = dense_reference(bs, values, b, Duse, nelim)
    lhs, rhs = oracle.schur_eliminate_dense(bs, values, b, Duse, nelim)
    ref_upper = upper_block_triangle(S, bs, nelim)
    scale = np.abs(S).max()
    assert np.abs(lhs - ref_upper).max() <= 1e-14 * scale * 10
    assert np.abs(rhs - rhs_ref).max() <= 1e-13 * max(1.0, np.abs(rhs_ref).max())
    ne = bs.num_cols - lhs.shape[0]
    x = oracle.schur_back_substitute(bs, values, b, Duse, nelim, sol[ne:])
    np.testing.assert_allclose(x[:ne], sol[:ne], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("pid", [2, 3, 4, 5, 6])
def test_schur_solver_vs_dense(oracle, pid):
    """schur_complement_solver_test.cc:52-118: DENSE_SCHUR solution vs dense solve, 1e-10."""
    bs, values, b, D, nelim, raw = lls_problem(pid)
    J, H, g, S, rhs_ref, sol = dense_reference(bs, values, b, D, nelim)
    o = oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=nelim)
    x, s = oracle.solve(bs, values, b, D, o)
    assert s.termination_type == oracle.SUCCESS
    assert np.linalg.norm(x - sol) / bs.num_cols < 1e-10


@pytest.mark.parametrize("pid", [2, 4, 5, 6])
def test_implicit_schur_vs_explicit(oracle, pid):
    """implicit_schur_complement_test.cc:112-213: every column of implicit S, rhs, back substitution."""
    bs, values, b, D, nelim, raw = lls_problem(pid)
    J, H, g, S, rhs_ref, sol = dense_reference(bs, values, b, D, nelim)
    n = S.shape[0]
    for i in range(n):
        e = np.zeros(n)
        e[i] = 1.0
        y, rhs = oracle.implicit_schur_multiply(bs, values, D, b, nelim, e)
        assert np.abs(y - S[:, i]).max() <= 1e-14 * np.abs(S).max() * 10
    assert np.abs(rhs - rhs_ref).max() <= 1e-13 * max(1.0, np.abs(rhs_ref).max())


@pytest.mark.parametrize("pid", [2, 3, 5])
@pytest.mark.parametrize("pre", ["SCHUR_JACOBI", "JACOBI", "IDENTITY", "SCHUR_POWER_SERIES_EXPANSION"])
def test_iterative_schur_vs_dense(oracle, pid, pre):
    """iterative_schur_complement_solver_test.cc:59-149: max_iter = num_cols, r_tol 1e-12."""
    bs, values, b, D, nelim, raw = lls_problem(pid)
    J, H, g, S, rhs_ref, sol = dense_reference(bs, values, b, D, nelim)
    o = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=getattr(oracle, pre),
                            num_eliminate_blocks=nelim, max_num_iterations=bs.num_cols,
                            max_num_spse_iterations=100)
    x, s = oracle.solve(bs, values, b, D, o, r_tolerance=1e-12)
    assert s.termination_type == oracle.SUCCESS, s.message
    assert np.linalg.norm(x - sol) < 1e-11


@pytest.mark.parametrize("pid", [2, 3, 4, 5, 6])
def test_explicit_schur_complement_cg_vs_dense(oracle, pid):
    """use_explicit_schur_complement (ITERATIVE_SCHUR on the block-sparse S, SCHUR_JACOBI): the procedure of
    iterative_schur_complement_solver_test.cc:59-149 (max_iter = num_cols, r_tol 1e-12, dense gold) applied to
    SparseSchurComplementSolver::SolveReducedLinearSystemUsingConjugateGradients (schur_complement_solver.cc:337-420)."""
    bs, values, b, D, nelim, raw = lls_problem(pid)
    J, H, g, S, rhs_ref, sol = dense_reference(bs, values, b, D, nelim)
    o = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=oracle.SCHUR_JACOBI,
                            num_eliminate_blocks=nelim, max_num_iterations=bs.num_cols, use_explicit_schur_complement=1)
    x, s = oracle.solve(bs, values, b, D, o, r_tolerance=1e-12)
    assert s.termination_type == oracle.SUCCESS, s.message
    assert np.linalg.norm(x - sol) < 1e-11


@pytest.mark.parametrize("pid", [2, 3, 4, 5, 6])
def test_sparse_schur_cell_structure(oracle, pid):
    """SparseSchurComplementSolver::InitStorage (schur_complement_solver.cc:224-290): the cell set is exactly the
    upper block triangle of the non-zero pattern of S (plus every diagonal cell), in lexicographic order."""
    bs, values, b, D, nelim, raw = lls_problem(pid)
    r, c = oracle.schur_sparse_structure(bs, nelim)
    cells = list(zip(r.tolist(), c.tolist()))
    assert cells == sorted(set(cells)) and all(i <= j for i, j in cells)
    nf = bs.num_col_blocks - nelim
    assert all((i, i) in cells for i in range(nf))
    # structural pattern from the Jacobian: f-blocks sharing an e-block, or a row without e-block
    pat = set((i, i) for i in range(nf))
    rows = {}
    for rb in range(bs.num_row_blocks):
        ids = [int(bs.cells["block_id"][k]) for k in range(bs.row_cell_begin[rb], bs.row_cell_begin[rb + 1])]
        if ids[0] < nelim:
            rows.setdefault(ids[0], set()).update(i - nelim for i in ids[1:])
        else:
            for i in ids:
                for j in ids:
                    if i <= j:
                        pat.add((i - nelim, j - nelim))
    for fs in rows.values():
        for i in fs:
            for j in fs:
                if i < j:
                    pat.add((i, j))
    assert set(cells) == pat


@pytest.mark.parametrize("pid", [0, 2, 5])
@pytest.mark.parametrize("pre", ["JACOBI", "IDENTITY"])
def test_cgnr_vs_dense(oracle, pid, pre):
    bs, values, b, D, nelim, raw = lls_problem(pid)
    J = bs.to_dense(values)
    sol = np.linalg.solve(J.T @ J + np.diag(D ** 2), J.T @ b)
    o = oracle.make_options(type=oracle.CGNR, preconditioner_type=getattr(oracle, pre), max_num_iterations=100)
    x, s = oracle.solve(bs, values, b, D, o, r_tolerance=1e-13)
    assert s.termination_type == oracle.SUCCESS, s.message
    np.testing.assert_allclose(x, sol, atol=1e-10)


def test_cg_goldens(oracle):
    for case in load_golden("conjugate_gradients_goldens.json")["cases"]:
        x, s = oracle.cg_dense(np.array(case["A"], dtype=float), case["b"], case["x0"],
                               case["max_num_iterations"], r_tolerance=case["r_tolerance"],
                               q_tolerance=case["q_tolerance"], min_num_iterations=case["min_num_iterations"],
                               residual_reset_period=case["residual_reset_period"])
        assert s.termination_type == case["termination"]
        if "num_iterations" in case:
            assert s.num_iterations == case["num_iterations"]
        np.testing.assert_allclose(x, case["x"], rtol=0, atol=4 * np.finfo(float).eps * 4)


def test_block_diagonal_inverses(oracle):
    """partitioned_matrix_view_test.cc:206-262 + AddDiagonalAndInvert."""
    for pid in (2, 4, 6):
        bs, values, b, D, nelim, raw = lls_problem(pid)
        J = bs.to_dense(values)
        H = J.T @ J + np.diag(D ** 2)
        ete, ftf = oracle.block_diagonal_inverses(bs, values, D, nelim)
        o = 0
        for blk in bs.col_blocks[:nelim]:
            p, s = int(blk["position"]), int(blk["size"])
            np.testing.assert_allclose(ete[o:o + s * s].reshape(s, s), np.linalg.inv(H[p:p + s, p:p + s]), rtol=1e-12)
            o += s * s
        o = 0
        for blk in bs.col_blocks[nelim:]:
            p, s = int(blk["position"]), int(blk["size"])
            np.testing.assert_allclose(ftf[o:o + s * s].reshape(s, s), np.linalg.inv(H[p:p + s, p:p + s]), rtol=1e-12)
            o += s * s


def test_detect_structure(oracle):
    """detect_structure_test.cc semantics on the fixtures."""
    bs, values, b, D, nelim, raw = lls_problem(2)
    assert oracle.detect_structure(bs, nelim) == (1, 1, 1)
    bs, values, b, D, nelim, raw = lls_problem(4)
    assert oracle.detect_structure(bs, nelim) == (2, 2, 2)
    bs, values, b, D, nelim, raw = lls_problem(6)
    assert oracle.detect_structure(bs, nelim) == (2, 2, 2)


def test_one_f_block_2_3_6(oracle):
    """schur_eliminator_test.cc:221-372 (SchurEliminatorForOneFBlock, MatchesSchurEliminator): lhs, rhs and the
    back-substituted e-part on the <2,3,6> single-f-block structure, here against dense algebra."""
    from conftest import one_f_block_problem
    bs, values, b, D, nelim = one_f_block_problem()
    J, H, g, S, rhs_ref, sol = dense_reference(bs, values, b, D, nelim)
    lhs, rhs = oracle.schur_eliminate_dense(bs, values, b, D, nelim)
    assert lhs.shape == (6, 6)
    assert np.abs(np.triu(lhs) - np.triu(S)).max() < 1e-13 * np.abs(S).max()
    assert np.abs(rhs - rhs_ref).max() < 1e-13 * np.abs(rhs_ref).max()
    f_sol = np.random.default_rng(1).uniform(-1, 1, 6)
    x = oracle.schur_back_substitute(bs, values, b, D, nelim, f_sol)
    ne = 3 * nelim
    e_ref = np.linalg.solve(H[:ne, :ne], g[:ne] - H[:ne, ne:] @ f_sol)
    assert np.abs(x[:ne] - e_ref).max() < 1e-12 * np.abs(e_ref).max()


def test_dense_cholesky_mixed_precision_and_refinement(oracle):
    """dense_cholesky_test.cc:70-117 (DenseCholeskyTest.FactorAndSolve, kMixedPrecision with 4 refinement steps and kFullPrecision):
    lhs = a'a + I, rhs = lhs x, 1..9 columns, 10 trials each: |x - actual| / |x| within 10 eps -- restated by
    orc_dense_cholesky_solve_refined; iterative_refiner_test.cc:152-190: 30 refinement steps with an exact and with a single
    precision factor of m m' (5 columns) bring |lhs x - rhs| to 10 eps."""
    rng = np.random.default_rng(12)
    eps = np.finfo(float).eps
    for n in range(1, 10):
        for trial in range(10):
            a = rng.uniform(-1, 1, (n, n))
            lhs = a.T @ a + np.eye(n)
            x = rng.uniform(-1, 1, n)
            rhs = lhs @ x
            for use_float, refinements in ((1, 4), (0, 0)):
                actual, t = oracle.dense_cholesky_solve_refined(np.triu(lhs), rhs, use_float, refinements)
                assert t == oracle.SUCCESS
                assert np.linalg.norm(x - actual) / np.linalg.norm(x) <= 10 * eps * max(1.0, np.linalg.cond(lhs)), (n, trial, use_float)
            # single precision alone is visibly single precision (the factor really is a float one)
            x32, _ = oracle.dense_cholesky_solve_refined(np.triu(lhs), rhs, 1, 0)
            if n > 1:
                assert 1e-9 < np.linalg.norm(x - x32) / np.linalg.norm(x) < 1e-4
    m = rng.uniform(-1, 1, (5, 5))
    lhs = m @ m.T + 1e-3 * np.eye(5)
    sol = rng.uniform(-1, 1, 5)
    rhs = lhs @ sol
    for use_float in (0, 1):
        refined, t = oracle.dense_cholesky_solve_refined(np.triu(lhs), rhs, use_float, 30)
        assert t == oracle.SUCCESS and np.linalg.norm(lhs @ refined - rhs) <= 10 * eps * max(1.0, np.linalg.norm(rhs))
    # not positive definite -> FAILURE with either factor
    bad = np.triu(lhs).copy()
    bad[2, 2] = -1.0
    for use_float in (0, 1):
        assert oracle.dense_cholesky_solve_refined(bad, rhs, use_float, 2)[1] == oracle.FAILURE
