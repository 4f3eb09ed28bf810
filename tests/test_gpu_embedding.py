"""The reference's other small Schur structures on the static kernels (csrc/cx_embed.hip).

SchurEliminator / PartitionedMatrixView are instantiated for 22 (row, e, f) triples (schur_eliminator.cc:55-143) and rows
without an e-block ride along (NoEBlockRowsUpdate, schur_eliminator_impl.h:567-659).  Structures with 2-row e-rows,
e <= 3 and f <= 9 -- <2,3,6>, <2,3,3>, <2,3,4>, <2,2,2>, <2,2,3>, <2,2,4>, and <2,3,9> in a cell layout other than
BuildJacobianLayout's -- get an embedded <2,3,9> image, with camera priors (rows holding one f cell) as chunks of dummy
points.  Every solver through the C ABI against the oracle's dynamic-size restatement of the reference on the SAME block
structure: iteration counts equal, solutions 1e-8; products and column norms 1e-12."""
import numpy as np
import pytest

from conftest import cx

pytestmark = pytest.mark.gpu


def make_structure(e, f, C, P, seed, priors=(), interleaved=False):
    """Random bundle-adjustment-shaped structure: P e-blocks of size e, C f-blocks of size f, every point seen by 2..6
    cameras of a window; `priors` = row sizes of trailing rows that hold one f cell (camera c = index mod C).
    interleaved: cell values laid out row by row ([E | F] per row) instead of all E cells then all F cells."""
    rng = np.random.default_rng(seed)
    rows_main = []
    for p in range(P):
        k = int(rng.integers(2, 7))
        home = int(rng.integers(0, C))
        cams = sorted({(home + i * int(rng.integers(1, 3))) % C for i in range(k)})
        for c in cams[::-1]:                       # (any order inside a chunk)
            rows_main.append((p, c))
    O = len(rows_main)
    rows, pos = [], 0
    if interleaved:
        for p, c in rows_main:
            rows.append((2, [(p, pos), (P + c, pos + 2 * e)]))
            pos += 2 * e + 2 * f
    else:
        for r, (p, c) in enumerate(rows_main):
            rows.append((2, [(p, 2 * e * r), (P + c, 2 * e * O + 2 * f * r)]))
        pos = (2 * e + 2 * f) * O
    for i, s in enumerate(priors):
        rows.append((s, [(P + (7 * i) % C, pos)]))
        pos += s * f
    bs = cx.BlockStructure.from_rows([e] * P + [f] * C, rows)
    values = rng.standard_normal(pos)
    b = rng.standard_normal(bs.num_rows)
    D = rng.uniform(0.5, 2.0, bs.num_cols) * 0.3
    return bs, values, b, D


CASES = [
    pytest.param(3, 6, (), False, id="<2,3,6>"),
    pytest.param(3, 3, (), False, id="<2,3,3>"),
    pytest.param(3, 4, (), True, id="<2,3,4> interleaved cells"),
    pytest.param(2, 2, (), False, id="<2,2,2>"),
    pytest.param(2, 3, (), False, id="<2,2,3>"),
    pytest.param(2, 4, (), True, id="<2,2,4> interleaved cells"),
    pytest.param(1, 5, (), False, id="<2,1,5>"),
    pytest.param(3, 9, (), True, id="<2,3,9> interleaved cells"),
    pytest.param(3, 9, (9, 9, 3, 1, 2), False, id="<2,3,9> + camera priors"),
    pytest.param(3, 6, (6, 6, 6, 1, 5), True, id="<2,3,6> + camera priors, interleaved"),
]
SOLVERS = [("ITERATIVE_SCHUR", "JACOBI", 0), ("ITERATIVE_SCHUR", "SCHUR_JACOBI", 0), ("ITERATIVE_SCHUR", "IDENTITY", 0),
           ("ITERATIVE_SCHUR", "SCHUR_POWER_SERIES_EXPANSION", 0), ("ITERATIVE_SCHUR", "SCHUR_JACOBI", 1),
           ("CGNR", "JACOBI", 0), ("DENSE_SCHUR", "IDENTITY", 0), ("SPARSE_SCHUR", "IDENTITY", 0)]


@pytest.fixture(scope="module")
def ctx():
    c = cx.Context(0)
    yield c
    c.close()


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("e,f,priors,interleaved", CASES)
def test_embedded_structures_match_the_oracle(ctx, oracle, e, f, priors, interleaved):
    C, P = 23, 400
    bs, values, b, D = make_structure(e, f, C, P, seed=10 * e + f, priors=priors, interleaved=interleaved)
    if not priors:
        assert cx.binding.detect_structure(bs, P) == (2, e, f)
    A = cx.Matrix(ctx, bs, P)
    assert A.static_path == 2 and not A.is_static_239
    A.set_values(values)
    assert np.array_equal(A.get_values(), values)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(bs.num_cols), rng.standard_normal(bs.num_rows)
    y0, x0 = rng.standard_normal(bs.num_rows), rng.standard_normal(bs.num_cols)
    jx, jty = oracle.right_multiply(bs, values, x), oracle.left_multiply(bs, values, y)
    assert relerr(A.right_multiply(x, y0), y0 + jx) < 1e-12
    assert relerr(A.left_multiply(y, x0), x0 + jty) < 1e-12
    assert relerr(A.squared_column_norm(), oracle.squared_column_norm(bs, values)) < 1e-12
    for stype, pre, explicit in SOLVERS:
        for use_D in (True, False):
            if not use_D and (stype, pre) != ("DENSE_SCHUR", "IDENTITY"):
                continue                                   # D = nullptr (the reference's tests): one direct solve is enough
            Dv = D if use_D else None
            o = oracle.make_options(type=getattr(oracle, stype), preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P,
                                    max_num_iterations=300, use_explicit_schur_complement=explicit)
            x_r, s_r = oracle.solve(bs, values, b, Dv if use_D else np.zeros(bs.num_cols), o, r_tolerance=-1.0, q_tolerance=1e-3)
            S = cx.Solver(ctx, type=getattr(cx, stype), preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P,
                          max_num_iterations=300, use_explicit_schur_complement=explicit)
            x_d, s_d = S.solve(A, b, Dv, r_tolerance=-1.0, q_tolerance=1e-3)
            assert s_d.termination_type == s_r.termination_type, (stype, pre, s_d.message, s_r.message)
            assert s_d.num_iterations == s_r.num_iterations, (stype, pre, s_d.num_iterations, s_r.num_iterations)
            assert np.all(np.isfinite(x_d)) and relerr(x_d, x_r) < 1e-8, (stype, pre, relerr(x_d, x_r))
            S.close()
    # ScaleColumns acts on the caller's values; the static image follows
    scale = rng.uniform(0.5, 1.5, bs.num_cols)
    A.scale_columns(scale)
    scaled = oracle.scale_columns(bs, values, scale)
    assert relerr(A.get_values(), scaled) < 1e-15
    assert relerr(A.right_multiply(x), oracle.right_multiply(bs, scaled, x)) < 1e-12
    A.close()


@pytest.mark.parametrize("pre", ["CLUSTER_JACOBI", "CLUSTER_TRIDIAGONAL"])
def test_visibility_preconditioners_on_an_embedded_structure(ctx, oracle, pre):
    """<2,3,6> with camera priors under the visibility based preconditioners: the clustering sees the real points only
    (rows without an e-block see no point, visibility.cc:50-85), so structure and iteration counts equal the oracle's."""
    C, P = 40, 900
    bs, values, b, D = make_structure(3, 6, C, P, seed=5, priors=(6, 6, 2), interleaved=False)
    A = cx.Matrix(ctx, bs, P)
    assert A.static_path == 2
    A.set_values(values)
    o = oracle.make_options(type=oracle.ITERATIVE_SCHUR, preconditioner_type=getattr(oracle, pre), num_eliminate_blocks=P, max_num_iterations=300)
    x_r, s_r = oracle.solve(bs, values, b, D, o, r_tolerance=-1.0, q_tolerance=1e-3)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=getattr(cx, pre), num_eliminate_blocks=P, max_num_iterations=300)
    x_d, s_d = S.solve(A, b, D, r_tolerance=-1.0, q_tolerance=1e-3)
    assert s_d.termination_type == s_r.termination_type and s_d.num_iterations == s_r.num_iterations, (s_d.message, s_r.message)
    assert relerr(x_d, x_r) < 1e-8
    S.close()
    A.close()


def test_what_does_not_embed_keeps_the_dynamic_size_path(ctx, oracle):
    """e = 4 (homogeneous points), f = 10, a row with two f cells: the dynamic-size kernels, results still the oracle's."""
    for e, f in ((4, 9), (3, 10)):
        bs, values, b, D = make_structure(e, f, 9, 60, seed=e + f)
        A = cx.Matrix(ctx, bs, 60)
        assert A.static_path == 0
        A.set_values(values)
        o = oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=60)
        x_r, _ = oracle.solve(bs, values, b, D, o)
        S = cx.Solver(ctx, type=cx.DENSE_SCHUR, num_eliminate_blocks=60)
        x_d, _ = S.solve(A, b, D)
        assert relerr(x_d, x_r) < 1e-9
        S.close()
        A.close()


def test_embedded_236_runs_at_static_speed(ctx):
    """500 k observations of a <2,3,6> problem: y += J x and S x (through ITERATIVE_SCHUR) run on the static kernels --
    within 1.5 x of the <2,3,9> time on the same visibility (the embedding streams the padded cells: 192 instead of 144
    bytes per row block, 1.33 x) -- where the dynamic-size path took two orders of magnitude longer."""
    prob = cx.bal.make_bal_like(400, 80000, 500000, seed=3)
    bs9, _ = cx.bal.build_structure(prob)
    O, P, C = prob.num_observations, prob.num_points, prob.num_cameras
    rows = [(2, [(int(bs9.cells["block_id"][2 * r]), 6 * r), (int(bs9.cells["block_id"][2 * r + 1]), 6 * O + 12 * r)]) for r in range(O)]
    bs6 = cx.BlockStructure.from_rows([3] * P + [6] * C, rows)
    rng = np.random.default_rng(1)
    times = {}
    for name, bs, nnz in (("239", bs9, 24 * O), ("236", bs6, 18 * O)):
        A = cx.Matrix(ctx, bs, P)
        assert A.static_path == (1 if name == "239" else 2)
        A.set_values(rng.standard_normal(nnz))
        x = rng.standard_normal(bs.num_cols)
        ms = []
        for _ in range(6):
            A.right_multiply(x)
            ms.append(A.last_kernel_ms)
        times[name] = float(np.median(ms[1:]))
        A.close()
    assert times["236"] < 2.0 * times["239"] + 0.2, times          # includes the widening of x and the narrowing of y
