"""cx_minimize (device-resident TrustRegionMinimizer + LevenbergMarquardtStrategy) against the
oracle's restatement of trust_region_minimizer.cc on the same bundle-adjustment inputs."""
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
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, float(np.max(np.abs(b)))))


CASES = [
    ("ITERATIVE_SCHUR", "JACOBI", None),
    ("ITERATIVE_SCHUR", "SCHUR_JACOBI", None),
    ("DENSE_SCHUR", "IDENTITY", None),
    ("CGNR", "JACOBI", None),
    ("DENSE_SCHUR", "IDENTITY", (cx.binding.LOSS_HUBER, 1.0, 0.0)),
    ("ITERATIVE_SCHUR", "JACOBI", (cx.binding.LOSS_CAUCHY, 2.0, 0.0)),
]


@pytest.mark.parametrize("stype,ptype,loss", CASES)
def test_minimize_matches_oracle(ctx, oracle, stype, ptype, loss):
    C, P, O = 12, 300, 2400
    prob = cx.bal.make_bal_like(C, P, O, 3)
    nelim = 0 if stype == "CGNR" else P
    max_lin = 500 if stype != "CGNR" else 2000
    mo = cx.binding.minimizer_options(max_num_iterations=8)
    ev = cx.Evaluator(ctx, prob)
    if loss:
        ev.set_loss(*loss)
    solver = cx.Solver(ctx, type=getattr(cx.binding, stype), preconditioner_type=getattr(cx.binding, ptype),
                       num_eliminate_blocks=nelim, max_num_iterations=max_lin)
    x, summ, its = cx.binding.minimize(ev, solver, prob.state(), mo)
    so = oracle.make_options(type=getattr(oracle, stype), preconditioner_type=getattr(oracle, ptype),
                             num_eliminate_blocks=nelim, max_num_iterations=max_lin)
    omo = oracle.minimizer_options(max_num_iterations=8)
    x_r, summ_r, its_r = oracle.minimize_bal(C, P, prob.camera_index, prob.point_index, prob.observations, prob.state(),
                                             so, omo, loss=loss)
    assert summ["termination_type"] == summ_r["termination_type"], (summ, summ_r)
    assert len(its) == len(its_r)
    assert summ["num_successful_steps"] == summ_r["num_successful_steps"]
    assert summ["num_unsuccessful_steps"] == summ_r["num_unsuccessful_steps"]
    # fp64 tolerance: an inexact (CG) step differs between device and host at the CG tolerance, so the
    # iterates agree to ~1e-6 relative while the flags / iteration counts (integers) agree exactly
    tol = 1e-9 if stype == "DENSE_SCHUR" else 1e-5
    for a, b in zip(its, its_r):
        assert a["iteration"] == b["iteration"]
        assert a["step_is_valid"] == b["step_is_valid"] and a["step_is_successful"] == b["step_is_successful"]
        assert abs(a["cost"] - b["cost"]) <= tol * abs(b["cost"])
        assert abs(a["trust_region_radius"] - b["trust_region_radius"]) <= 1e-4 * b["trust_region_radius"]
        assert abs(a["gradient_max_norm"] - b["gradient_max_norm"]) <= 1e-3 * its_r[0]["gradient_max_norm"]
        if stype == "DENSE_SCHUR":
            assert abs(a["step_norm"] - b["step_norm"]) <= 1e-7 * max(1.0, b["step_norm"])
            assert abs(a["relative_decrease"] - b["relative_decrease"]) <= 1e-6
    assert abs(summ["final_cost"] - summ_r["final_cost"]) <= tol * summ_r["final_cost"]
    assert summ["final_cost"] < 0.05 * summ["initial_cost"]
    assert relerr(x, x_r) < (1e-8 if stype == "DENSE_SCHUR" else 1e-4)
    # the returned state is the minimum-cost iterate
    cost_at_x = ev.evaluate(x, want_gradient=False, want_jacobian=False)[0]
    assert abs(cost_at_x - summ["final_cost"]) <= 1e-12 * summ["final_cost"]
    solver.close()
    ev.close()


@pytest.mark.parametrize("stype", ["DENSE_SCHUR", "SPARSE_SCHUR"])
def test_minimize_with_mixed_precision_solves(ctx, oracle, stype):
    """bundle_adjuster --mixed_precision_solves [--max_num_refinement_iterations N] (bundle_adjuster.cc:141-142, 170-173): the LM
    loop on single precision reduced solves.  An LM step only has to be a descent step of the right size: with the float factor
    the iterations accept / reject as the oracle's mixed run does and land on the same cost (to single precision of the step);
    with refinement steps the trajectory is the double precision one."""
    C, P, O = 12, 300, 2400
    prob = cx.bal.make_bal_like(C, P, O, 3)
    mo = cx.binding.minimizer_options(max_num_iterations=8)
    omo = oracle.minimizer_options(max_num_iterations=8)
    ev = cx.Evaluator(ctx, prob)
    kw = dict(type=getattr(cx.binding, stype), num_eliminate_blocks=P)
    okw = dict(type=getattr(oracle, stype), num_eliminate_blocks=P)
    run = lambda **extra: cx.binding.minimize(ev, cx.Solver(ctx, **kw, **extra), prob.state(), mo)
    orun = lambda **extra: oracle.minimize_bal(C, P, prob.camera_index, prob.point_index, prob.observations, prob.state(),
                                               oracle.make_options(**okw, **extra), omo)
    x64, s64, it64 = run()
    x32, s32, it32 = run(use_mixed_precision_solves=1)
    x32r, s32r, it32r = run(use_mixed_precision_solves=1, max_num_refinement_iterations=3)
    xo32, so32, ito32 = orun(use_mixed_precision_solves=1)
    assert s32["termination_type"] == so32["termination_type"] and len(it32) == len(ito32)
    assert s32["num_successful_steps"] == so32["num_successful_steps"]
    for a, b in zip(it32, ito32):
        assert a["step_is_successful"] == b["step_is_successful"] and abs(a["cost"] - b["cost"]) <= 1e-4 * abs(b["cost"])
    assert abs(s32["final_cost"] - s64["final_cost"]) <= 1e-4 * s64["final_cost"]
    assert x32.tobytes() != x64.tobytes()                                  # the float factor really ran
    assert len(it32r) == len(it64)
    for a, b in zip(it32r, it64):
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * abs(b["cost"])
    assert relerr(x32r, x64) < 1e-7
    ev.close()


def test_minimize_rejected_steps_and_nonmonotonic_option(ctx, oracle):
    """Bundle adjustment from a perturbed start is so close to Gauss-Newton-ideal that no step is ever rejected
    (and starts that do get rejections are singular: costs ~1e30).  Rejections are therefore forced through
    min_relative_decrease = 0.99: the strategy's radius halving / quartering and reuse of the LM diagonal
    (levenberg_marquardt_strategy.cc:153-169) and the step evaluator with and without the non-monotonic option
    (trust_region_step_evaluator.cc:40-117) must walk the oracle's path.  DENSE_SCHUR: every step exact."""
    C, P, O = 10, 200, 1500
    prob = cx.bal.perturb(cx.bal.make_bal_like(C, P, O, 5), 1, 0, 0.15, seed=2)
    for nonmono in (0, 1):
        kw = dict(max_num_iterations=12, min_relative_decrease=0.99, use_nonmonotonic_steps=nonmono,
                  max_consecutive_nonmonotonic_steps=3, function_tolerance=1e-12)
        ev = cx.Evaluator(ctx, prob)
        solver = cx.Solver(ctx, type=cx.binding.DENSE_SCHUR, num_eliminate_blocks=P)
        x, summ, its = cx.binding.minimize(ev, solver, prob.state(), cx.binding.minimizer_options(**kw))
        so = oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=P)
        x_r, summ_r, its_r = oracle.minimize_bal(C, P, prob.camera_index, prob.point_index, prob.observations,
                                                 prob.state(), so, oracle.minimizer_options(**kw))
        assert len(its) == len(its_r) and summ["termination_type"] == summ_r["termination_type"]
        flags = [i["step_is_successful"] for i in its]
        assert flags == [i["step_is_successful"] for i in its_r]
        assert flags.count(0) >= 4 and flags.count(1) >= 4
        assert summ["num_unsuccessful_steps"] == summ_r["num_unsuccessful_steps"]
        for a, b in zip(its, its_r):
            assert abs(a["cost"] - b["cost"]) <= 1e-8 * abs(b["cost"])
            assert abs(a["trust_region_radius"] - b["trust_region_radius"]) <= 1e-5 * b["trust_region_radius"]
            assert abs(a["relative_decrease"] - b["relative_decrease"]) <= 1e-5
        assert relerr(x, x_r) < 1e-7
        solver.close()
        ev.close()


def test_minimize_device_state_and_limits(ctx, oracle):
    """DEVICE memspace updates the caller's array in place; max_num_iterations = 0 evaluates only."""
    C, P, O = 8, 120, 900
    prob = cx.bal.make_bal_like(C, P, O, 11)
    ev = cx.Evaluator(ctx, prob)
    solver = cx.Solver(ctx, type=cx.binding.DENSE_SCHUR, num_eliminate_blocks=P)
    d_state = ctx.to_device(prob.state())
    _, summ0, its0 = cx.binding.minimize(ev, solver, d_state, cx.binding.minimizer_options(max_num_iterations=0))
    assert summ0["termination_type"] == cx.binding.MIN_NO_CONVERGENCE and len(its0) == 1
    assert np.array_equal(d_state.to_host(), prob.state())
    _, summ, its = cx.binding.minimize(ev, solver, d_state, cx.binding.minimizer_options(max_num_iterations=20))
    assert summ["termination_type"] == cx.binding.CONVERGENCE
    x_host, summ_h, _ = cx.binding.minimize(ev, solver, prob.state(), cx.binding.minimizer_options(max_num_iterations=20))
    # same result through either memspace: every sum has a fixed order (S is gathered, not scattered), so bitwise
    assert np.array_equal(d_state.to_host(), x_host)
    assert abs(summ["final_cost"] - summ_h["final_cost"]) <= 1e-12 * summ_h["final_cost"]
    solver.close()
    ev.close()


def test_quaternion_cameras_on_manifold(ctx, oracle):
    """bundle_adjuster --use_quaternions --use_manifolds: 10-parameter cameras on ProductManifold<QuaternionManifold,
    EuclideanManifold<6>>; the kernel projects the ambient Jacobian with the PlusJacobian so J keeps the <2,3,9>
    layout (residual_block.cc:136-159).  Evaluation, Plus and the whole trust-region loop against the oracle."""
    C, P, O = 12, 300, 2400
    prob = cx.bal.make_bal_like(C, P, O, 3)
    bs, order = cx.bal.build_structure(prob)
    state = cx.bal.state_quaternion(prob)
    ev = cx.Evaluator(ctx, prob)
    ev.set_camera_model(cx.binding.CAMERA_QUATERNION_MANIFOLD)
    assert ev.num_parameters == 3 * P + 10 * C and ev.num_effective_parameters == 3 * P + 9 * C
    cost, res, grad = ev.evaluate(state)
    cost_r, res_r, grad_r, vals_r = oracle.bal_evaluate_model(bs, C, P, prob.camera_index, prob.point_index,
                                                              prob.observations, order, state, oracle.QUATERNION_MANIFOLD)
    vals = ev.jacobian(bs).get_values()
    assert relerr(res, res_r) < 1e-11 and relerr(vals, vals_r) < 1e-11 and relerr(grad, grad_r) < 1e-10
    assert abs(cost - cost_r) <= 1e-11 * cost_r
    cost2, res2, _ = ev.evaluate(state, want_gradient=False, want_jacobian=False)       # value-only kernel
    assert relerr(res2, res_r) < 1e-11
    # same residuals as the angle-axis parameterisation of the same cameras
    ev_aa = cx.Evaluator(ctx, prob)
    _, res_aa, _ = ev_aa.evaluate(prob.state(), want_gradient=False, want_jacobian=False)
    assert relerr(res, res_aa) < 1e-10
    # Evaluator::Plus
    delta = np.random.default_rng(1).standard_normal(3 * P + 9 * C) * 0.05
    xp = ev.plus(state, delta)
    xp_r = oracle.bal_plus(C, P, oracle.QUATERNION_MANIFOLD, state, delta)
    assert relerr(xp, xp_r) < 1e-14
    assert np.array_equal(ev.plus(state, np.zeros_like(delta)), state)
    # the trust-region loop
    mo = cx.binding.minimizer_options(max_num_iterations=8)
    solver = cx.Solver(ctx, type=cx.binding.DENSE_SCHUR, num_eliminate_blocks=P)
    x, summ, its = cx.binding.minimize(ev, solver, state, mo)
    so = oracle.make_options(type=oracle.DENSE_SCHUR, num_eliminate_blocks=P)
    x_r, summ_r, its_r = oracle.minimize_bal(C, P, prob.camera_index, prob.point_index, prob.observations, state, so,
                                             oracle.minimizer_options(max_num_iterations=8),
                                             camera_model=oracle.QUATERNION_MANIFOLD)
    assert len(its) == len(its_r) and summ["termination_type"] == summ_r["termination_type"]
    for a, b in zip(its, its_r):
        assert a["step_is_successful"] == b["step_is_successful"]
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * abs(b["cost"])
        assert abs(a["gradient_max_norm"] - b["gradient_max_norm"]) <= 1e-6 * its_r[0]["gradient_max_norm"]
        assert abs(a["step_norm"] - b["step_norm"]) <= 1e-7 * max(1.0, b["step_norm"])
    assert relerr(x, x_r) < 1e-8
    q = x[3 * P:].reshape(C, 10)[:, 0:4]
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-12)
    # and the same optimum as the angle-axis run
    x_aa, summ_aa, _ = cx.binding.minimize(ev_aa, solver, prob.state(), mo)
    assert abs(summ["final_cost"] - summ_aa["final_cost"]) <= 1e-6 * summ_aa["final_cost"]
    solver.close()
    ev_aa.close()
    ev.close()
