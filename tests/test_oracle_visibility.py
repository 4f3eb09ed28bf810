"""The oracle's visibility clustering and CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL preconditioners against the
reference's own known answers (re-typed as data):
  canonical_views_clustering_test.cc:41-139, single_linkage_clustering_test.cc:40-122,
  graph_algorithms_test.cc:105-169 (Degree2MaximumSpanningForest), visibility_test.cc:45-197,
and the procedures of the (disabled) visibility_based_preconditioner_test.cc:110-160: sparsity structure valid,
preconditioner values equal the matching cells of S.  CPU only."""
import numpy as np
import pytest

from conftest import cx


# ---------------------------------------------------------------- canonical_views_clustering_test.cc
# V0 -0.8- V1 -0.9- V2 -0.3- V3, vertex weights 0, 2, 2, -1, self edges 1.0
CV_EDGES = [(0, 0, 1.0), (1, 1, 1.0), (2, 2, 1.0), (3, 3, 1.0), (0, 1, 0.8), (1, 2, 0.9), (2, 3, 0.3)]
CV_WEIGHTS = [0.0, 2.0, 2.0, -1.0]


def test_canonical_views(oracle):
    centers, membership = oracle.canonical_views(4, CV_WEIGHTS, CV_EDGES, min_views=0, size_penalty_weight=0.5,
                                                 similarity_penalty_weight=0.0, view_score_weight=0.0)
    assert centers == [1, 3]
    assert membership == [0, 0, 0, 1]


def test_canonical_views_size_penalty(oracle):
    centers, _ = oracle.canonical_views(4, CV_WEIGHTS, CV_EDGES, min_views=0, size_penalty_weight=2.0,
                                        similarity_penalty_weight=0.0, view_score_weight=0.0)
    assert centers == [1]


def test_canonical_views_view_score(oracle):
    centers, _ = oracle.canonical_views(4, CV_WEIGHTS, CV_EDGES, min_views=0, size_penalty_weight=0.5,
                                        similarity_penalty_weight=0.0, view_score_weight=1.0)
    assert centers == [1, 2]


def test_canonical_views_similarity_penalty(oracle):
    centers, _ = oracle.canonical_views(4, CV_WEIGHTS, CV_EDGES, min_views=0, size_penalty_weight=0.5,
                                        similarity_penalty_weight=3.0, view_score_weight=1.0)
    assert centers == [1]


# ---------------------------------------------------------------- single_linkage_clustering_test.cc
def test_single_linkage_two_components(oracle):
    k, m = oracle.single_linkage(6, [(0, 1, 1.0), (1, 2, 1.0), (2, 3, 1.0), (4, 5, 1.0)])
    assert m[1] == m[0] and m[2] == m[0] and m[3] == m[0]
    assert m[4] != m[0] and m[5] != m[0] and m[4] == m[5]
    assert k == 2


def test_single_linkage_weak_link(oracle):
    k, m = oracle.single_linkage(6, [(0, 1, 1.0), (1, 2, 1.0), (2, 3, 1.0), (4, 5, 0.5)])
    assert m[1] == m[0] and m[2] == m[0] and m[3] == m[0]
    assert m[4] != m[0] and m[5] != m[0] and m[4] != m[5]
    assert k == 3


def test_single_linkage_weak_and_strong_link(oracle):
    k, m = oracle.single_linkage(6, [(0, 1, 1.0), (1, 2, 1.0), (2, 3, 0.5), (0, 3, 1.0), (4, 5, 1.0)])
    assert m[1] == m[0] and m[2] == m[0] and m[3] == m[0] and m[4] == m[5]
    assert k == 2


# ---------------------------------------------------------------- graph_algorithms_test.cc
def test_degree2_forest_preserves_edge(oracle):
    assert oracle.degree2_forest(2, [(0, 1, 0.5)]) == [(0, 1)]


def test_degree2_forest_star_graph(oracle):
    forest = oracle.degree2_forest(5, [(0, 1, 1.0), (0, 2, 2.0), (0, 3, 3.0), (0, 4, 4.0)])
    assert sorted(forest) == [(0, 3), (0, 4)]   # the hub keeps its two heaviest edges; 1 and 2 stay isolated


def test_degree2_forest_no_cycles(oracle):
    # triangle + pendant: the lightest triangle edge would close a cycle
    forest = oracle.degree2_forest(4, [(0, 1, 3.0), (1, 2, 2.0), (0, 2, 1.0), (2, 3, 0.5)])
    assert sorted(forest) == [(0, 1), (1, 2), (2, 3)]


# ---------------------------------------------------------------- visibility_test.cc
def test_schur_complement_graph_simple_matrix(oracle):
    # A = [1 0 0 0 0 1; 1 0 0 1 0 0; 0 1 1 0 0 0; 0 1 0 0 1 0], 2 e-blocks
    rows = [(2, [(0, 0), (5, 0)]), (2, [(0, 1), (3, 1)]), (2, [(1, 2), (2, 2)]), (2, [(1, 3), (4, 3)])]
    bs = cx.BlockStructure.from_rows([1] * 6, rows)
    w = np.zeros((4, 4))
    for u, v, weight in oracle.schur_complement_graph(bs, 2):
        w[u, v] = weight
    expect = np.eye(4)
    expect[1, 3] = 1.0
    expect[0, 2] = 1.0
    assert np.array_equal(w, expect)


def test_schur_complement_graph_no_e_blocks_seen(oracle):
    rows = [(2, [(0, 0)]), (2, [(0, 1)]), (2, [(1, 2)]), (2, [(1, 3)])]
    bs = cx.BlockStructure.from_rows([1] * 6, rows)
    edges = oracle.schur_complement_graph(bs, 2)
    assert sorted(edges) == [(i, i, 1.0) for i in range(4)]


# ---------------------------------------------------------------- preconditioner structure and values
def _problem(oracle, C, P, O, seed):
    prob = cx.bal.make_bal_like(C, P, O, seed)
    bs, order = cx.bal.build_structure(prob)
    _, b, _, vals = oracle.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations, order,
                                        prob.state(), want_gradient=False)
    rng = np.random.default_rng(seed)
    D = rng.uniform(0.5, 2.0, bs.num_cols) * 1e-2 * np.sqrt(np.abs(vals).mean())
    return prob, bs, vals, b, D


@pytest.mark.parametrize("clustering", [0, 1])
@pytest.mark.parametrize("pre", ["CLUSTER_JACOBI", "CLUSTER_TRIDIAGONAL"])
def test_preconditioner_structure_is_valid(oracle, pre, clustering):
    """IsSparsityStructureValid (visibility_based_preconditioner_test.cc:110-133): a block pair is in the
    preconditioner iff its cluster pair is -- and, beyond the reference's check, iff the cameras are co-visible."""
    C, P = 40, 900
    prob, bs, vals, b, D = _problem(oracle, C, P, 4200, 5)
    membership, k, cluster_pairs, block_pairs = oracle.visibility_structure(bs, P, getattr(oracle, pre), clustering)
    assert membership.min() == 0 and membership.max() == k - 1 and len(np.unique(membership)) == k
    cps = {tuple(p) for p in cluster_pairs.tolist()}
    assert all((i, i) in cps for i in range(k))
    if pre == "CLUSTER_JACOBI":
        assert len(cps) == k
    else:
        # a degree-2 forest: no cluster has more than two partners, no cycles
        partners = np.zeros(k, dtype=int)
        parent = list(range(k))

        def find(v):
            while parent[v] != v:
                v = parent[v]
            return v
        for a, c in cps:
            if a == c:
                continue
            partners[a] += 1
            partners[c] += 1
            ra, rc = find(a), find(c)
            assert ra != rc
            parent[ra] = rc
        assert partners.max() <= 2
    r_all, c_all = oracle.schur_sparse_structure(bs, P)
    covisible = set(zip(r_all.tolist(), c_all.tolist()))
    bps = {tuple(p) for p in block_pairs.tolist()}
    for i in range(C):
        for j in range(i, C):
            cp = (min(membership[i], membership[j]), max(membership[i], membership[j]))
            assert ((i, j) in bps) == (cp in cps and (i, j) in covisible)


@pytest.mark.parametrize("pre", ["CLUSTER_JACOBI", "CLUSTER_TRIDIAGONAL"])
def test_cluster_preconditioned_solve(oracle, pre):
    """PreconditionerValuesMatch + the solve: with the preconditioner built from the matching cells of S in numpy,
    PCG must take the oracle's path -- same iteration count, same solution; and the converged step equals
    DENSE_SCHUR's."""
    C, P = 30, 700
    prob, bs, vals, b, D = _problem(oracle, C, P, 3300, 11)
    membership, k, cluster_pairs, block_pairs = oracle.visibility_structure(bs, P, getattr(oracle, pre), 0)
    S, rhs = oracle.schur_eliminate_dense(bs, vals, b, D, P)
    S = np.triu(S) + np.triu(S, 1).T
    M = np.zeros_like(S)
    for i, j in block_pairs.tolist():
        M[9 * i:9 * i + 9, 9 * j:9 * j + 9] = S[9 * i:9 * i + 9, 9 * j:9 * j + 9]
        M[9 * j:9 * j + 9, 9 * i:9 * i + 9] = S[9 * j:9 * j + 9, 9 * i:9 * i + 9]
    try:
        np.linalg.cholesky(M)
    except np.linalg.LinAlgError:
        for i, j in block_pairs.tolist():
            if membership[i] != membership[j]:
                M[9 * i:9 * i + 9, 9 * j:9 * j + 9] *= 0.5
                M[9 * j:9 * j + 9, 9 * i:9 * i + 9] *= 0.5
    Minv = np.linalg.inv(M)
    # plain PCG with the reference's stopping rule (conjugate_gradients_solver.h:107-305), q_tolerance = 0.1
    x = np.zeros(9 * C)
    r = rhs.copy()
    Q0 = 0.0
    rho = 1.0
    iters = 0
    for it in range(1, 201):
        z = Minv @ r
        last_rho, rho = rho, r @ z
        p = z.copy() if it == 1 else z + (rho / last_rho) * p
        q = S @ p
        alpha = rho / (p @ q)
        x += alpha * p
        r = rhs - S @ x if it % 10 == 0 else r - alpha * q
        Q1 = -x @ (rhs + r)
        zeta = it * (Q1 - Q0) / Q1
        iters = it
        if zeta < 0.1:
            break
        Q0 = Q1
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                             max_num_iterations=200)
    xo, so = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.1)
    assert so.termination_type == 0, so.message
    assert so.num_iterations == iters
    assert np.abs(xo[3 * P:] - x).max() < 1e-8 * np.abs(x).max()
    # converged: the step of the direct solver
    xc, sc = oracle.solve(bs, vals, b, D, oo, r_tolerance=1e-13, q_tolerance=0.0)
    od = oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=P)
    xd, _ = oracle.solve(bs, vals, b, D, od)
    assert np.abs(S @ xc[3 * P:] - rhs).max() < 1e-7 * np.abs(rhs).max()
    assert np.abs(xc - xd).max() < 1e-3 * np.abs(xd).max()   # cond(S) ~ 1e8 on this problem
    # a sharper preconditioner than the block Jacobi of S: no more iterations than SCHUR_JACOBI to converge
    oj = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=oracle.SCHUR_JACOBI, num_eliminate_blocks=P,
                             max_num_iterations=500)
    _, sj = oracle.solve(bs, vals, b, D, oj, r_tolerance=1e-13, q_tolerance=0.0)
    assert sc.num_iterations <= sj.num_iterations


def test_one_cluster_is_the_exact_inverse(oracle):
    """Cameras that all see the same points fall into one cluster under SINGLE_LINKAGE (similarity 1): the
    preconditioner is S itself and PCG needs a single step."""
    C, P = 5, 60
    cam = np.tile(np.arange(C, dtype=np.int32), P)
    pt = np.repeat(np.arange(P, dtype=np.int32), C)
    O = C * P
    prob = cx.bal.BalProblem(C, P, cam, pt, np.zeros((O, 2)), np.zeros((C, 9)), np.zeros((P, 3)))
    bs, order = cx.bal.build_structure(prob)
    vals = cx.bal.random_jacobian_values(O, 3)
    b = np.random.default_rng(3).standard_normal(2 * O)
    D = np.full(bs.num_cols, 0.1)
    membership, k, _, block_pairs = oracle.visibility_structure(bs, P, oracle.CLUSTER_JACOBI, oracle.SINGLE_LINKAGE)
    assert k == 1 and len(block_pairs) == C * (C + 1) // 2
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=oracle.CLUSTER_JACOBI, num_eliminate_blocks=P,
                             visibility_clustering_type=oracle.SINGLE_LINKAGE, max_num_iterations=50)
    xo, so = oracle.solve(bs, vals, b, D, oo, r_tolerance=1e-10, q_tolerance=0.0)
    assert so.termination_type == 0 and so.num_iterations <= 2, so.message


def test_tridiagonal_retry_with_halved_off_diagonal_cells(oracle):
    """ScaleOffDiagonalCells (visibility_based_preconditioner.cc:331-393): the unscaled matrix is indefinite, the
    solve still succeeds and takes the path of PCG with the halved matrix."""
    from conftest import crafted_indefinite_tridiagonal
    prob, bs, vals, b, D, P = crafted_indefinite_tridiagonal(0)
    membership, k, _, block_pairs = oracle.visibility_structure(bs, P, oracle.CLUSTER_TRIDIAGONAL, oracle.SINGLE_LINKAGE)
    assert k == 3 and len(block_pairs) == 5
    S, rhs = oracle.schur_eliminate_dense(bs, vals, b, D, P)
    S = np.triu(S) + np.triu(S, 1).T
    M = np.zeros_like(S)
    for i, j in block_pairs.tolist():
        M[9 * i:9 * i + 9, 9 * j:9 * j + 9] = S[9 * i:9 * i + 9, 9 * j:9 * j + 9]
        M[9 * j:9 * j + 9, 9 * i:9 * i + 9] = S[9 * j:9 * j + 9, 9 * i:9 * i + 9]
    assert np.linalg.eigvalsh(M).min() < -1.0
    for i, j in block_pairs.tolist():
        if membership[i] != membership[j]:
            M[9 * i:9 * i + 9, 9 * j:9 * j + 9] *= 0.5
            M[9 * j:9 * j + 9, 9 * i:9 * i + 9] *= 0.5
    assert np.linalg.eigvalsh(M).min() > 0.1
    oo = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=oracle.CLUSTER_TRIDIAGONAL, num_eliminate_blocks=P,
                             visibility_clustering_type=oracle.SINGLE_LINKAGE, max_num_iterations=100)
    x, s = oracle.solve(bs, vals, b, D, oo, r_tolerance=1e-12, q_tolerance=0.0)
    assert s.termination_type == 0, s.message
    assert np.abs(S @ x[3 * P:] - rhs).max() < 1e-5 * np.abs(rhs).max()   # (E'E + 1e-6 I)^-1 of the private points
    # same iterates as PCG preconditioned with the halved matrix (loose solve: the problem is ill conditioned on
    # purpose, E'E + 1e-6 I, so a long CG run amplifies the rounding differences of the threaded elimination)
    xl, sl = oracle.solve(bs, vals, b, D, oo, r_tolerance=-1.0, q_tolerance=0.01)
    assert sl.termination_type == 0 and 2 <= sl.num_iterations <= 30
    Minv = np.linalg.inv(M)
    xs = np.zeros(27)
    r = rhs.copy()
    rho = 1.0
    for it in range(1, sl.num_iterations + 1):
        z = Minv @ r
        last_rho, rho = rho, r @ z
        p = z.copy() if it == 1 else z + (rho / last_rho) * p
        q = S @ p
        alpha = rho / (p @ q)
        xs += alpha * p
        r = rhs - S @ xs if it % 10 == 0 else r - alpha * q
    assert np.abs(xs - xl[3 * P:]).max() < 1e-3 * np.abs(xs).max()
