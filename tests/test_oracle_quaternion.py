"""Oracle pins for the quaternion camera parameterisation (bundle_adjuster --use_quaternions --use_manifolds):
rotation.h conversions against rotation_test.cc:205-330 known answers, QuaternionManifold against
manifold_test.cc:500-605 procedures, and the projected Jacobian against finite differences through Plus."""
import numpy as np
import pytest

from conftest import cx, orc

bal = cx.bal


def test_angle_axis_quaternion_known_answers():
    # rotation_test.cc:205-247
    assert np.allclose(orc.angle_axis_to_quaternion([0, 0, 0]), [1, 0, 0, 0], atol=1e-15)
    th = 1e-2
    assert np.allclose(orc.angle_axis_to_quaternion([th, 0, 0]), [np.cos(th / 2), np.sin(th / 2), 0, 0], atol=1e-15)
    tiny = np.finfo(float).tiny ** 0.75
    assert np.allclose(orc.angle_axis_to_quaternion([tiny, 0, 0]), [np.cos(tiny / 2), np.sin(tiny / 2), 0, 0], atol=1e-300)
    h = np.sqrt(0.5)
    assert np.allclose(orc.angle_axis_to_quaternion([np.pi / 2, 0, 0]), [h, h, 0, 0], atol=1e-15)
    # rotation_test.cc:249-275
    assert np.allclose(orc.quaternion_to_angle_axis([1, 0, 0, 0]), [0, 0, 0], atol=1e-15)
    assert np.allclose(orc.quaternion_to_angle_axis([0, 0, 1, 0]), [0, np.pi, 0], atol=1e-14)
    assert np.allclose(orc.quaternion_to_angle_axis([h, 0, 0, h]), [0, 0, np.pi / 2], atol=1e-14)
    # rotation_test.cc:298-330: angle < pi, round trips
    rng = np.random.default_rng(0)
    for _ in range(200):
        aa = rng.standard_normal(3)
        aa *= rng.uniform(0, np.pi - 1e-3) / np.linalg.norm(aa)
        q = orc.angle_axis_to_quaternion(aa)
        assert abs(np.linalg.norm(q) - 1.0) < 1e-14
        assert np.allclose(orc.quaternion_to_angle_axis(q), aa, atol=1e-12)
    # the vectorised numpy versions of bal.py agree with the scalar restatement
    aas = rng.standard_normal((50, 3))
    aas[0] = 0.0
    qs = bal.angle_axis_to_quaternion(aas)
    assert np.allclose(qs, [orc.angle_axis_to_quaternion(a) for a in aas], atol=1e-15)
    assert np.allclose(bal.quaternion_to_angle_axis(qs), [orc.quaternion_to_angle_axis(q) for q in qs], atol=1e-14)


def _quat_product(z, w):  # rotation.h:763-773
    return np.array([z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3],
                     z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2],
                     z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1],
                     z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0]])


@pytest.mark.parametrize("scale", ["generic", 1e-6, np.pi - 1e-6])
def test_quaternion_manifold(scale):
    # manifold_test.cc:500-605: Plus == QuaternionProduct(AngleAxisToQuaternion(2 delta), x); PlusJacobian is
    # the derivative of Plus at delta = 0; Plus keeps the norm
    rng = np.random.default_rng(1)
    for _ in range(100):
        x = rng.standard_normal(4)
        x /= np.linalg.norm(x)
        delta = rng.uniform(-1, 1, 3)
        if scale != "generic":
            delta *= scale / np.linalg.norm(delta)
        expected = _quat_product(orc.angle_axis_to_quaternion(2 * delta), x)
        actual = orc.quaternion_plus(x, delta)
        assert np.linalg.norm(actual - expected) / np.linalg.norm(expected) < 1e-9
        assert abs(np.linalg.norm(actual) - 1.0) < 1e-12
        J = orc.quaternion_plus_jacobian(x)
        h = 1e-6
        fd = np.stack([(orc.quaternion_plus(x, h * e) - orc.quaternion_plus(x, -h * e)) / (2 * h) for e in np.eye(3)], axis=1)
        assert np.abs(J - fd).max() < 1e-9
    # PlusPiBy2 (manifold_test.cc:500-531)
    for i in range(3):
        d = np.zeros(3)
        d[i] = np.pi / 2
        out = orc.quaternion_plus([1, 0, 0, 0], d)
        for j in range(4):
            assert abs(abs(out[j]) - (1.0 if j == i + 1 else 0.0)) < 1e-15
    assert np.array_equal(orc.quaternion_plus([0.5, 0.5, 0.5, 0.5], [0, 0, 0]), [0.5, 0.5, 0.5, 0.5])


def test_quaternion_snavely_matches_angle_axis_and_finite_differences():
    prob = bal.make_bal_like(6, 40, 200, seed=3)
    cams10 = bal.quaternion_cameras(prob)
    rng = np.random.default_rng(2)
    for k in rng.choice(prob.num_observations, 25, replace=False):
        c, p = prob.camera_index[k], prob.point_index[k]
        r_aa = np.zeros(2)
        orc.lib().orc_snavely(orc._p(prob.cameras[c].copy()), orc._p(prob.points[p].copy()),
                              orc._p(prob.observations[k].copy()), orc._p(r_aa), None, None)
        r_q, jc, jp = orc.snavely_quaternion(cams10[c], prob.points[p], prob.observations[k])
        assert np.allclose(r_q, r_aa, rtol=1e-11, atol=1e-9)          # same rotation, same residual
        # tangent-space Jacobian: d residual(Plus(camera, delta)) / d delta at 0
        h = 1e-6

        def res_at(delta):
            cam = cams10[c].copy()
            cam[0:4] = orc.quaternion_plus(cam[0:4], delta[0:3])
            cam[4:10] += delta[3:9]
            return orc.snavely_quaternion(cam, prob.points[p], prob.observations[k], want_jacobian=False)[0]
        fd = np.stack([(res_at(h * e) - res_at(-h * e)) / (2 * h) for e in np.eye(9)], axis=1)
        assert np.abs(jc - fd).max() <= 1e-6 * max(1.0, np.abs(fd).max())
        # a non-unit quaternion gives the same residual (QuaternionRotatePoint normalises)
        cam = cams10[c].copy()
        cam[0:4] *= 1.7
        assert np.allclose(orc.snavely_quaternion(cam, prob.points[p], prob.observations[k], want_jacobian=False)[0], r_q,
                           rtol=1e-12, atol=1e-9)


def test_quaternion_program_minimizes_like_angle_axis():
    C, P, O = 8, 120, 900
    prob = bal.make_bal_like(C, P, O, 11)
    so = orc.make_options(type=orc.DENSE_SCHUR, num_eliminate_blocks=P)
    mo = orc.minimizer_options(max_num_iterations=10)
    x_aa, s_aa, its_aa = orc.minimize_bal(C, P, prob.camera_index, prob.point_index, prob.observations, prob.state(), so, mo)
    x_q, s_q, its_q = orc.minimize_bal(C, P, prob.camera_index, prob.point_index, prob.observations,
                                       bal.state_quaternion(prob), so, mo, camera_model=orc.QUATERNION_MANIFOLD)
    assert s_q["initial_cost"] == pytest.approx(s_aa["initial_cost"], rel=1e-12)
    assert s_q["termination_type"] == orc.CONVERGENCE
    assert s_q["final_cost"] == pytest.approx(s_aa["final_cost"], rel=1e-6)   # same minimum, other chart
    q = x_q[3 * P:].reshape(C, 10)[:, 0:4]
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-12)            # Plus keeps the quaternions on the sphere
