"""The drop-in boundary with HOST vectors: the LM call sequence of TrustRegionMinimizer against the
C ABI exactly as the host adapters issue it (ceres-solver-ceres-solver_amd/boundary.py is the ctypes twin of
host/test_host_adapter.cpp's loop).  What is checked here is that the transfer machinery -- caller arrays registered with
the HIP runtime, the model-cost product into a zeroed target, slices of the caller's arrays copied straight to the shards
of a multi-shard front -- never changes a bit of the loop, and that the traffic counters say what crossed PCIe."""
import ctypes

import numpy as np
import pytest

from conftest import cx

pytestmark = pytest.mark.gpu

boundary = cx.boundary


@pytest.fixture(scope="module")
def problem():
    return cx.bal.make_bal_like(40, 6000, 30000, seed=21)


def _solver_kw(prob):
    return dict(type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=prob.num_points,
                max_num_iterations=500, min_num_iterations=0, residual_reset_period=10)


def _run(prob, devices=None, policy=1, iterations=5, **opt_ins):
    ctx = cx.Context(devices=devices) if devices else cx.Context(0)
    loop = boundary.BoundaryLoop(ctx, prob, _solver_kw(prob), eta=1e-2, register_arrays=policy, **opt_ins)
    report = loop.run(iterations)
    loop.close()     # releases the registrations while the loop's vectors are still alive
    ctx.close()
    return report


def test_registration_and_opt_ins_change_no_bit(problem):
    plain = _run(problem, policy=0, fuse_scaling=False, alias_residuals=False, zeroed_target=False)
    assert len(plain["costs"]) >= 4 and plain["costs"][-1] < 0.5 * plain["costs"][0]
    assert plain["registered_fraction"] == 0.0
    n, m = 3 * problem.num_points + 9 * problem.num_cameras, 2 * problem.num_observations
    for policy in (0, 1, 2):
        full = _run(problem, policy=policy)
        assert full["costs"] == plain["costs"], "policy %d" % policy
        assert full["linear_iterations"] == plain["linear_iterations"]
        if policy == 1:
            assert full["registered_fraction"] > 0.95
        # per iteration with all opt-ins: H2D = state, D, step, candidate state (4 column vectors; a rejected step has
        # no second state), D2H = residuals, gradient, column norms, step, model residuals
        assert full["h2d_bytes"] <= 8 * (4 * n) + 4096
        assert full["d2h_bytes"] <= 8 * (3 * n + 2 * m) + 4096
    # without the opt-ins the same iteration also uploads b and the zeros of the product target, and scales J
    assert plain["h2d_bytes"] >= 8 * (3 * n + 2 * m)


@pytest.mark.parametrize("shards", [2, 4])
def test_front_copies_slices_and_keeps_the_bits(problem, shards):
    a = _run(problem, devices=[0] * shards, policy=0)
    b = _run(problem, devices=[0] * shards, policy=1)
    assert a["costs"] == b["costs"] and a["linear_iterations"] == b["linear_iterations"]
    one = _run(problem, policy=1)
    assert len(a["costs"]) == len(one["costs"])
    assert np.allclose(a["costs"], one["costs"], rtol=1e-9, atol=0.0)
    n, m = 3 * problem.num_points + 9 * problem.num_cameras, 2 * problem.num_observations
    nf = 9 * problem.num_cameras
    # the shards read their own slices: the camera part travels once per shard, the point part once in total
    assert b["h2d_bytes"] <= 8 * (4 * (n + (shards - 1) * nf)) + 4096
    assert b["d2h_bytes"] <= 8 * (3 * n + 2 * m) + 4096
    assert b["registered_fraction"] > 0.95


def test_overwrite_product_equals_accumulating_into_zeros(problem):
    ctx = cx.Context(0)
    ev = cx.Evaluator(ctx, problem)
    ev.evaluate(problem.state())
    A = ev.jacobian()
    x = np.random.default_rng(5).standard_normal(A.num_cols)
    y0 = A.right_multiply(x)
    y1 = np.full(A.num_rows, 7.0)   # whatever the target held is ignored
    cx.binding._check(ctx.lib.cx_matrix_right_multiply_overwrite(A._h, x.ctypes.data_as(ctypes.c_void_p),
                                                                 y1.ctypes.data_as(ctypes.c_void_p), cx.HOST))
    assert np.array_equal(y0, y1)
    ev.close()
    ctx.close()


def test_registry_is_off_by_default_and_explicit_registration_is_released():
    """Nothing is registered unless the caller asks (a registered array that is freed and whose addresses come back
    faults the GPU: the owner of the lifetime decides); an explicitly registered array is copied as registered memory and
    is gone from the registry after the release."""
    ctx = cx.Context(0)
    a = np.random.default_rng(0).standard_normal(1 << 20)
    ctx.transfer_stats(reset=True)
    d = ctx.to_device(a)
    assert np.array_equal(d.to_host(), a)
    stats = ctx.transfer_stats(reset=True)
    assert stats["num_registered"] == 0 and stats["h2d_registered_bytes"] == 0 and stats["h2d_bytes"] == a.nbytes
    cx.host_register(a)
    d2 = ctx.to_device(a)
    back = d2.to_host()
    assert np.array_equal(back, a)
    stats = ctx.transfer_stats(reset=True)
    assert stats["num_registered"] == 1 and stats["h2d_registered_bytes"] == a.nbytes
    cx.host_registrations_release()
    assert ctx.transfer_stats()["num_registered"] == 0
    del a
    ctx.close()
