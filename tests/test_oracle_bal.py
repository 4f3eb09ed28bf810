"""Bundle-adjustment half of the oracle: Snavely residual/Jacobian, rotation,
residual ordering, Jacobian layout, Schur ordering.  The reference has no numeric
fixture for the Snavely arithmetic (its BAL data file is a stripped blob), so this
part of the oracle is PARITY UNPINNED against the reference and is pinned here by
independent means: scipy's Rodrigues rotation and central differences."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import cx


def test_angle_axis_rotate_point_vs_scipy(oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        aa = rng.standard_normal(3) * rng.uniform(1e-3, 3.0)
        pt = rng.standard_normal(3) * 10
        np.testing.assert_allclose(oracle.angle_axis_rotate_point(aa, pt),
                                   Rotation.from_rotvec(aa).apply(pt), rtol=1e-12, atol=1e-12)
    # theta == 0 exactly: first-order Taylor branch (rotation.h:836-856)
    pt = np.array([1.0, -2.0, 3.0])
    np.testing.assert_array_equal(oracle.angle_axis_rotate_point(np.zeros(3), pt), pt)


def test_snavely_jacobian_central_differences(oracle):
    prob = cx.bal.make_bal_like(6, 40, 120, seed=3)
    rng = np.random.default_rng(0)
    for i in rng.integers(0, prob.num_observations, 25):
        cam = prob.cameras[prob.camera_index[i]].copy()
        pt = prob.points[prob.point_index[i]].copy()
        obs = prob.observations[i]
        r, jc, jp = oracle.snavely(cam, pt, obs)
        r0, _, _ = oracle.snavely(cam, pt, obs, jacobians=False)
        # the dual-number path divides by multiplying with 1/g (jet.h:367-379), so the
        # residual differs from the plain-double path in the last bits, as in the reference
        np.testing.assert_allclose(r, r0, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(r, cx.bal.project(cam[None], pt[None], [0], [0])[0] - obs, rtol=1e-11, atol=1e-9)
        for k in range(9):
            h = 1e-6 * max(1.0, abs(cam[k]))
            cp, cm = cam.copy(), cam.copy()
            cp[k] += h
            cm[k] -= h
            fd = (oracle.snavely(cp, pt, obs, False)[0] - oracle.snavely(cm, pt, obs, False)[0]) / (2 * h)
            np.testing.assert_allclose(jc[:, k], fd, rtol=2e-5, atol=1e-5 * np.abs(jc).max())
        for k in range(3):
            h = 1e-6 * max(1.0, abs(pt[k]))
            pp, pm = pt.copy(), pt.copy()
            pp[k] += h
            pm[k] -= h
            fd = (oracle.snavely(cam, pp, obs, False)[0] - oracle.snavely(cam, pm, obs, False)[0]) / (2 * h)
            np.testing.assert_allclose(jp[:, k], fd, rtol=2e-5, atol=1e-5 * np.abs(jp).max())


def test_residual_order_reverses_inside_chunks(oracle):
    """LexicographicallyOrderResidualBlocks: buckets by point, each filled back to front."""
    prob = cx.bal.make_bal_like(5, 30, 100, seed=5)
    # shuffle the input so the bucket order is not trivially sorted
    perm = np.random.default_rng(2).permutation(prob.num_observations)
    pt = prob.point_index[perm]
    order = oracle.bal_residual_order(prob.num_points, pt)
    assert np.array_equal(order, cx.bal.residual_order(pt, prob.num_points))
    assert np.all(np.diff(pt[order]) >= 0)
    for j in range(prob.num_points):
        idx = order[pt[order] == j]
        assert np.all(np.diff(idx) < 0)            # reverse input order inside the chunk


def test_bal_structure_layout(oracle):
    prob = cx.bal.make_bal_like(7, 50, 160, seed=9)
    bs, order = cx.bal.build_structure(prob)
    rb, cb, rcb, cells = oracle.bal_structure_arrays(prob.num_cameras, prob.num_points, prob.camera_index,
                                                    prob.point_index, order)
    assert np.array_equal(rb, bs.row_blocks) and np.array_equal(cb, bs.col_blocks)
    assert np.array_equal(rcb, bs.row_cell_begin) and np.array_equal(cells, bs.cells)
    assert oracle.detect_structure(bs, prob.num_points) == (2, 3, 9)
    O = prob.num_observations
    assert bs.num_nonzeros == 24 * O and bs.num_rows == 2 * O and bs.num_cols == 3 * 50 + 9 * 7


def test_bal_evaluate_consistency(oracle):
    prob = cx.bal.make_bal_like(6, 60, 200, seed=11)
    bs, order = cx.bal.build_structure(prob)
    state = prob.state()
    cost, res, grad, vals = oracle.bal_evaluate(bs, prob.num_cameras, prob.num_points, prob.camera_index,
                                                prob.point_index, prob.observations, order, state)
    assert abs(cost - 0.5 * res @ res) <= 1e-12 * cost
    J = bs.to_dense(vals)
    np.testing.assert_allclose(grad, J.T @ res, rtol=1e-12, atol=1e-9)
    # directional derivative of the residual vector
    rng = np.random.default_rng(0)
    dx = rng.standard_normal(state.shape[0]) * 1e-6
    _, rp, _, _ = oracle.bal_evaluate(bs, prob.num_cameras, prob.num_points, prob.camera_index, prob.point_index,
                                      prob.observations, order, state + dx, want_gradient=False, want_jacobian=False)
    _, rm, _, _ = oracle.bal_evaluate(bs, prob.num_cameras, prob.num_points, prob.camera_index, prob.point_index,
                                      prob.observations, order, state - dx, want_gradient=False, want_jacobian=False)
    np.testing.assert_allclose((rp - rm) / 2, J @ dx, rtol=1e-4, atol=1e-7)


def test_stable_schur_ordering_bal(oracle):
    """ComputeStableSchurOrdering: on BAL graphs where every point has smaller degree
    than every camera the independent set is exactly the points, in index order."""
    prob = cx.bal.make_bal_like(8, 200, 900, seed=13)
    C, P = prob.num_cameras, prob.num_points
    ordering, k = oracle.stable_schur_ordering(C, P, prob.camera_index, prob.point_index)
    deg_pt = np.bincount(prob.point_index, minlength=P)
    deg_cam = np.bincount(prob.camera_index, minlength=C)
    assert deg_pt.max() < deg_cam.min()
    assert k == P
    # stable sort by degree keeps index order among equal degrees
    expect = C + np.argsort(deg_pt, kind="stable")
    assert np.array_equal(ordering[:P], expect)
    assert sorted(ordering[P:].tolist()) == list(range(C))


def test_stable_schur_ordering_small_camera_first(oracle):
    """A camera that sees fewer points than its points see cameras enters the
    independent set first and turns its points grey (graph_algorithms.h:165-227)."""
    cam = np.array([0, 1, 2, 1, 2, 1, 2], dtype=np.int32)
    pt = np.array([0, 0, 0, 1, 1, 2, 2], dtype=np.int32)
    ordering, k = oracle.stable_schur_ordering(3, 3, cam, pt)
    # degrees: cam0=1, cam1=3, cam2=3, pt0=3, pt1=2, pt2=2 ; queue = cam0, pt1, pt2, cam1, cam2, pt0
    assert ordering.tolist() == [0, 4, 5, 1, 2, 3] and k == 3
