"""Ceres' own linear-system dumps (Solver::Options::trust_region_minimizer_iterations_to_dump, TEXTFILE format of
linear_least_squares_problems.cc:966-1022) through the loader of ceres-solver-ceres-solver_amd/dumps.py: the committed
fixture tests/golden/ceres_dump/ (written by tests/golden/make_ceres_dump.py -- the reference itself cannot run here, so
its x is the oracle's) and a round trip.  The GPU half solves the loaded system through the C ABI."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, cx

BASE = os.path.join(GOLDEN, "ceres_dump", "ceres_solver_iteration_000")


def test_dump_loader_rebuilds_the_block_structure(oracle, tmp_path):
    d = cx.dumps.read_dump(BASE)
    bs = d["bs"]
    assert (d["num_rows"], d["num_cols"]) == (340, 174) and d["values"].size == 4080
    # the structure of the problem the fixture was written from, bit for bit (integers)
    prob = cx.bal.make_bal_like(6, 40, 170, seed=77)
    ref, _ = cx.bal.build_structure(prob)
    for name in ("row_blocks", "col_blocks", "row_cell_begin"):
        assert np.array_equal(getattr(bs, name), getattr(ref, name)), name
    assert np.array_equal(bs.cells["block_id"], ref.cells["block_id"])
    assert np.array_equal(bs.cells["position"], np.arange(len(bs.cells)) // 2 * 24 + (np.arange(len(bs.cells)) % 2) * 6)  # file order
    assert cx.dumps.leading_eliminate_blocks(bs) == prob.num_points
    # ... and in BuildJacobianLayout order ([E cells | F cells]) it IS the structure the evaluator's Jacobian has
    jbs, jvalues = cx.dumps.to_jacobian_layout(bs, d["values"], prob.num_points)
    assert np.array_equal(jbs.cells, ref.cells)
    assert np.array_equal(jbs.to_dense(jvalues), bs.to_dense(d["values"]))
    # the dumped x solves the dumped system: normal-equation residual of the oracle's own solve of the LOADED values
    o = oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=prob.num_points)
    x, s = oracle.solve(bs, d["values"], d["b"], d["D"], o)
    assert s.termination_type == oracle.SUCCESS
    # %17f keeps six decimals of A, b, D and x: the solution of the ROUNDED system is the dumped x only as far as the rounding
    # of a Jacobi-scaled J (entries below 1) lets it be -- 1e-4 of its size here; the structure and the values are exact
    assert np.abs(x - d["x"]).max() <= 5e-4 * np.abs(d["x"]).max()
    # round trip: what is written is what is read (to the six decimals of the format), structure exact
    base2 = str(tmp_path / "again")
    cx.dumps.write_dump(base2, bs, d["values"], b=d["b"], D=d["D"], x=d["x"])
    d2 = cx.dumps.read_dump(base2)
    assert np.array_equal(d2["bs"].cells, bs.cells) and np.array_equal(d2["values"], d["values"]) and np.array_equal(d2["x"], d["x"])


def test_dump_loader_on_other_block_shapes(tmp_path):
    # <2,3,6> rows, a trailing row block with one f cell, an untouched column block in the middle
    rows, pos = [], 0
    for i in range(5):
        rows.append((2, [(i, pos), (6, pos + 6)]))
        pos += 6 + 12
    rows.append((3, [(7, pos)]))
    pos += 12
    bs = cx.BlockStructure.from_rows([3] * 5 + [2, 6, 4], rows)
    values = np.round(np.random.default_rng(0).standard_normal(pos), 6)
    base = str(tmp_path / "shapes")
    cx.dumps.write_dump(base, bs, values, b=np.arange(bs.num_rows, dtype=float))
    d = cx.dumps.read_dump(base)
    assert np.array_equal(d["bs"].col_blocks, bs.col_blocks) and np.array_equal(d["bs"].row_blocks, bs.row_blocks)
    assert np.array_equal(d["bs"].cells, bs.cells) and np.array_equal(d["values"], values)
    assert d["D"] is None and d["x"] is None and np.array_equal(d["b"], np.arange(bs.num_rows))
    assert cx.dumps.leading_eliminate_blocks(d["bs"]) == 5


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["DENSE_SCHUR", "SPARSE_SCHUR", "ITERATIVE_SCHUR", "CGNR"])
def test_dumped_system_through_the_c_abi(solver, oracle):
    d = cx.dumps.read_dump(BASE)
    nelim = cx.dumps.leading_eliminate_blocks(d["bs"])
    bs, values = cx.dumps.to_jacobian_layout(d["bs"], d["values"], nelim)
    ctx = cx.Context(0)
    A = cx.Matrix(ctx, bs, nelim if solver != "CGNR" else 0)
    assert A.is_static_239 or solver == "CGNR"
    A.set_values(values)
    S = cx.Solver(ctx, type=getattr(cx, solver), preconditioner_type=cx.JACOBI, num_eliminate_blocks=nelim if solver != "CGNR" else 0,
                  max_num_iterations=2000)
    x, s = S.solve(A, d["b"], d["D"], r_tolerance=1e-12, q_tolerance=0.0)
    assert s.termination_type == cx.SUCCESS, s.message
    # parity: the oracle's exact solve of the same loaded values; the dumped x (six decimals of a system whose values had
    # more) only as far as the format's rounding allows
    x_r, _ = oracle.solve(bs, values, d["b"], d["D"], oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=nelim))
    # (direct solves to rounding; the CG solvers as far as 2000 iterations at r_tolerance 1e-12 take them -- CGNR works on
    # the squared condition number)
    tol = {"DENSE_SCHUR": 1e-10, "SPARSE_SCHUR": 1e-10, "ITERATIVE_SCHUR": 1e-6, "CGNR": 2e-5}[solver]
    assert np.abs(x - x_r).max() <= tol * np.abs(x_r).max(), solver
    assert np.abs(x - d["x"]).max() <= 5e-4 * np.abs(d["x"]).max(), solver
    S.close()
    A.close()
    ctx.close()
