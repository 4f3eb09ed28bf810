"""Oracle pins for LossFunction / Corrector, following the reference's own test procedures:
loss_function_test.cc:44-175 (finite-difference validity at s = 0.357, 1.792, ..., values at
s = 0) and corrector_test.cc:56-266 (scalar known answers, Gauss-Newton approximation)."""
import numpy as np
import pytest

import orc

L = dict(HUBER=1, SOFT_L_ONE=2, CAUCHY=3, ARCTAN=4, TOLERANT=5, TUKEY=6)


def assert_loss_valid(t, a, b, s):
    # AssertLossFunctionIsValid, loss_function_test.cc:48-70
    rho = orc.loss_evaluate(t, a, b, s)
    h = 1e-4
    fwd, bwd = orc.loss_evaluate(t, a, b, s + h), orc.loss_evaluate(t, a, b, s - h)
    assert abs((fwd[0] - bwd[0]) / (2 * h) - rho[1]) < 1e-6
    assert abs((fwd[0] - 2 * rho[0] + bwd[0]) / (h * h) - rho[2]) < 1e-6


@pytest.mark.parametrize("name", ["HUBER", "SOFT_L_ONE", "CAUCHY", "ARCTAN", "TUKEY"])
def test_loss_derivatives(name):
    for a in (0.7, 1.3):
        for s in (0.357, 1.792):
            assert_loss_valid(L[name], a, 0.0, s)


def test_trivial_loss():
    for s in (0.357, 1.792, 0.0):
        assert np.allclose(orc.loss_evaluate(0, 0, 0, s), [s, 1.0, 0.0])


def test_loss_at_zero():
    # loss_function_test.cc:97-101,110-114,123-127,136-140,169-173
    assert np.allclose(orc.loss_evaluate(L["HUBER"], 0.7, 0, 0.0), [0, 1, 0], atol=1e-6)
    assert np.allclose(orc.loss_evaluate(L["SOFT_L_ONE"], 0.7, 0, 0.0), [0, 1, -0.5 / 0.49], atol=1e-6)
    assert np.allclose(orc.loss_evaluate(L["CAUCHY"], 0.7, 0, 0.0), [0, 1, -1.0 / 0.49], atol=1e-6)
    assert np.allclose(orc.loss_evaluate(L["ARCTAN"], 0.7, 0, 0.0), [0, 1, 0], atol=1e-6)
    assert np.allclose(orc.loss_evaluate(L["TUKEY"], 0.7, 0, 0.0), [0, 1, -2.0 / 0.49], atol=1e-6)


def test_tolerant_loss():
    # loss_function_test.cc:143-160
    for a, b in ((0.7, 0.4), (1.3, 0.1)):
        for s in (0.357, 1.792, 55.5):
            assert_loss_valid(L["TOLERANT"], a, b, s)
    assert abs(orc.loss_evaluate(L["TOLERANT"], 0.7, 0.4, 0.0)[0]) < 1e-6
    for s in (20.0 + 36.6, 20.0 + 36.7, 20.0 + 36.8, 20.0 + 1000.0):
        assert_loss_valid(L["TOLERANT"], 20.0, 1.0, s)


@pytest.mark.parametrize("res,rho", [(np.sqrt(3.0), [3.0, 0.1, -0.01]), (0.0, [0.0, 0.1, -0.01]),
                                     (np.sqrt(3.0), [3.0, 0.1, -0.1])])
def test_corrector_scalar(res, rho):
    # corrector_test.cc:56-136: rho'' < 0 or zero residual -> alpha = 0
    r, j = orc.corrector_apply(res * res, rho, [res], [[10.0]])
    assert abs(r[0] - res * np.sqrt(rho[1])) < 1e-6
    assert abs(j[0, 0] - np.sqrt(rho[1]) * 10.0) < 1e-6


def test_corrector_gauss_newton_approximation():
    # corrector_test.cc:140-205 (procedure; numpy's generator instead of std::mt19937)
    rng = np.random.default_rng(5)
    for _ in range(2000):
        jac = rng.uniform(0, 1, (3, 2))
        res = rng.uniform(0, 1, 3)
        sq = float(res @ res)
        rho = [sq, rng.uniform(0, 1), rng.uniform(-1, 1)]
        kD = 1 + 2 * rho[2] / rho[1] * sq
        alpha = 1 - np.sqrt(kD) if rho[2] > 0 else 0.0
        g_res = np.sqrt(rho[1]) / (1 - alpha) * res
        g_jac = np.sqrt(rho[1]) * (jac - alpha / sq * np.outer(res, res) @ jac)
        g_grad = rho[1] * jac.T @ res
        g_hess = rho[1] * jac.T @ jac + 2 * rho[2] * jac.T @ np.outer(res, res) @ jac
        c_res, c_jac = orc.corrector_apply(sq, rho, res, jac)
        assert np.linalg.norm(g_res - c_res) < 1e-10
        assert np.linalg.norm(g_jac - c_jac) < 1e-10
        assert np.linalg.norm(g_grad - c_jac.T @ c_res) < 1e-10
        if rho[2] > 0:  # corrector_test.cc:207-266 checks the Hessian where the correction is exact
            assert np.linalg.norm(g_hess - c_jac.T @ c_jac) < 1e-9
