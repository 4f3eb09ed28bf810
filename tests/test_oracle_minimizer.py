"""Pins the oracle's TrustRegionMinimizer / LevenbergMarquardtStrategy restatement with the
reference's own known-answer test: Powell's singular function, every subset of active columns the
reference runs (trust_region_minimizer_test.cc:215-290), must reach the origin to 1e-3."""
import itertools

import numpy as np
import pytest

import orc

# the 14 column activations of PowellsSingularFunctionUsingLevenbergMarquardt
# (trust_region_minimizer_test.cc:269-290; <1,1,0,1> is excluded there: local minimum)
ACTIVE = [c for c in itertools.product([True, False], repeat=4) if any(c) and c != (True, True, False, True)]


def powell(active):
    cols = [i for i in range(4) if active[i]]

    def fun(xa):
        # PowellEvaluator2::Evaluate (trust_region_minimizer_test.cc:91-190); inactive columns stay 0
        x = np.zeros(4)
        x[cols] = xa
        x1, x2, x3, x4 = x
        f = np.array([x1 + 10.0 * x2, np.sqrt(5.0) * (x3 - x4), (x2 - 2.0 * x3) ** 2, np.sqrt(10.0) * (x1 - x4) ** 2])
        J = np.array([[1.0, 10.0, 0.0, 0.0],
                      [0.0, 0.0, np.sqrt(5.0), -np.sqrt(5.0)],
                      [0.0, 2.0 * (x2 - 2.0 * x3), -4.0 * (x2 - 2.0 * x3), 0.0],
                      [np.sqrt(10.0) * 2.0 * (x1 - x4), 0.0, 0.0, -np.sqrt(10.0) * 2.0 * (x1 - x4)]])
        return f, J[:, cols]

    return fun, np.array([3.0, -1.0, 0.0, 1.0])[cols]


@pytest.mark.parametrize("active", ACTIVE)
def test_powell_levenberg_marquardt(active):
    fun, x0 = powell(active)
    opts = orc.minimizer_options(gradient_tolerance=1e-26, function_tolerance=1e-26, parameter_tolerance=1e-26,
                                 max_trust_region_radius=1e20)
    x, summary, its = orc.minimize_dense(fun, x0, opts)
    assert np.all(np.abs(x) < 1e-3), (x, summary)
    # bookkeeping of Minimize(): one summary per iteration, steps add up, cost decreases monotonically
    assert summary["num_iterations"] == len(its) == its[-1]["iteration"] + 1
    assert summary["num_successful_steps"] + summary["num_unsuccessful_steps"] == len(its)
    costs = [i["cost"] for i in its if i["step_is_successful"]]
    assert all(b <= a for a, b in zip(costs, costs[1:]))
    assert summary["final_cost"] <= summary["initial_cost"]


def test_lm_radius_update_rule():
    # LevenbergMarquardtStrategy::StepAccepted / StepRejected (levenberg_marquardt_strategy.cc:153-169):
    # on a linear least squares problem the model is exact (rho = 1) so the radius triples each step
    A = np.array([[1.0, 2.0], [3.0, 4.0], [5.0, 7.0]])
    b = np.array([1.0, -2.0, 0.5])
    x, summary, its = orc.minimize_dense(lambda x: (A @ x - b, A), np.zeros(2),
                                         orc.minimizer_options(jacobi_scaling=0, max_num_iterations=3,
                                                               function_tolerance=0.0, gradient_tolerance=0.0,
                                                               parameter_tolerance=0.0))
    radii = [i["trust_region_radius"] for i in its]
    assert radii[0] == 1e4 and np.allclose(radii[1:], [3e4, 9e4, 27e4])
    assert np.allclose(x, np.linalg.lstsq(A, b, rcond=None)[0], atol=1e-4)
    assert abs(its[1]["relative_decrease"] - 1.0) < 1e-6
