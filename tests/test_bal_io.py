"""BAL file reader / writer, Normalize and Perturb (examples/bal_problem.cc:73-333) -- host I/O."""
import numpy as np

from conftest import cx

bal = cx.bal


def test_write_read_round_trip(tmp_path):
    prob = bal.make_bal_like(7, 60, 300, seed=4)
    path = tmp_path / "problem-7-60-pre.txt"
    bal.write_bal(path, prob)
    back = bal.read_bal(path)
    assert (back.num_cameras, back.num_points, back.num_observations) == (7, 60, 300)
    assert np.array_equal(back.camera_index, prob.camera_index) and np.array_equal(back.point_index, prob.point_index)
    # parameters are written with %.16g; 17 significant digits would be exact, 16 is ~1 ulp
    assert np.allclose(back.cameras, prob.cameras, rtol=1e-15, atol=0) and np.allclose(back.points, prob.points, rtol=1e-15)
    # observations go through %g (6 significant digits), bal_problem.cc:149-151
    assert np.allclose(back.observations, prob.observations, rtol=1e-5, atol=1e-5)
    text = open(path).read().split("\n")
    assert text[0] == "7 60 300" and len(text[1].split()) == 4


def test_reader_accepts_free_form_whitespace(tmp_path):
    # fscanf("%d") / fscanf("%lf") skip any whitespace: one token per line is as valid as four
    path = tmp_path / "p.txt"
    cams = np.arange(18, dtype=float) * 0.5
    pts = np.array([1.0, 2.0, 3.0])
    tokens = ["2", "1", "2", "0", "0", "-1.5", "2e0", "1", "0", "3.25", "-4"] + [repr(float(v)) for v in cams] + [repr(float(v)) for v in pts]
    path.write_text("\n".join(tokens) + "\n")
    p = bal.read_bal(path)
    assert p.num_cameras == 2 and p.num_points == 1 and p.num_observations == 2
    assert p.camera_index.tolist() == [0, 1] and p.point_index.tolist() == [0, 0]
    assert np.array_equal(p.observations, [[-1.5, 2.0], [3.25, -4.0]])
    assert np.array_equal(p.cameras.ravel(), cams) and np.array_equal(p.points.ravel(), pts)


def test_normalize():
    prob = bal.make_bal_like(9, 401, 2000, seed=8)
    # move the scene away from the origin first
    shifted = bal.dataclasses.replace(prob, points=prob.points * 3.0 + np.array([5.0, -7.0, 11.0]))
    norm = bal.normalize(shifted)
    med = np.array([np.partition(norm.points[:, i], 200)[200] for i in range(3)])
    assert np.allclose(med, 0.0, atol=1e-9)
    mad = np.partition(np.abs(norm.points).sum(axis=1), 200)[200]
    assert abs(mad - 100.0) < 1e-9
    # a similarity transform of points and camera centres leaves every projection unchanged
    cams = shifted.cameras.copy()
    c0 = bal.camera_centers(cams)
    proj0 = bal.project(cams, shifted.points, shifted.camera_index, shifted.point_index)
    proj1 = bal.project(norm.cameras, norm.points, norm.camera_index, norm.point_index)
    assert np.allclose(proj0, proj1, rtol=1e-9, atol=1e-9)
    assert np.allclose(norm.cameras[:, [0, 1, 2, 6, 7, 8]], cams[:, [0, 1, 2, 6, 7, 8]])
    assert not np.allclose(bal.camera_centers(norm.cameras), c0)


def test_perturb():
    prob = bal.make_bal_like(9, 100, 500, seed=8)
    same = bal.perturb(prob, 0.0, 0.0, 0.0)
    assert np.allclose(same.cameras, prob.cameras, atol=1e-12) and np.array_equal(same.points, prob.points)
    p = bal.perturb(prob, 0.1, 0.5, 0.2, seed=1)
    assert 0.1 < np.std(p.points - prob.points) < 0.3
    assert not np.allclose(p.cameras[:, 0:6], prob.cameras[:, 0:6])
    assert np.array_equal(p.cameras[:, 6:9], prob.cameras[:, 6:9])
