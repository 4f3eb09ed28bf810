"""Pins the Snavely half of the oracle (AngleAxisRotatePoint, rotation.h:792-857 -- the arithmetic inside
SnavelyReprojectionError the evaluator parity tests are checked against) to answers the reference itself holds:
rotation_test.cc:379-424 (AngleAxisToRotationMatrix known answers: zero, near zero, pi/2 about X, pi about Y) and
rotation_test.cc:1707-1800 (AngleAxisRotatePoint == R p for 500 000 angle-axis / point draws over (-pi, pi) and for
10 000 rotations of 1e-16 rad: the exactly-zero Taylor branch included).  Tolerance is the reference's own
kTolerance = 10 eps (rotation_test.cc:60).  The draws come from numpy instead of the default-seeded std::mt19937 of
the reference test; the test is a property over the draws, not a table."""
import numpy as np

from conftest import orc

K_TOLERANCE = np.finfo(float).eps * 10   # rotation_test.cc:60


def is_orthonormal(R):   # rotation_test.cc:118-140 (column-major 3x3)
    M = np.asarray(R).reshape(3, 3).T
    return np.abs(M.T @ M - np.eye(3)).max() <= K_TOLERANCE


def test_angle_axis_to_rotation_matrix_known_answers():
    eye = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    # ZeroAngleAxisToRotationMatrix, rotation_test.cc:379-386
    R = orc.angle_axis_to_rotation_matrix([0.0, 0.0, 0.0])
    assert is_orthonormal(R) and np.abs(R - eye).max() <= K_TOLERANCE
    # NearZeroAngleAxisToRotationMatrix :388-395
    R = orc.angle_axis_to_rotation_matrix([1e-24, 2e-24, 3e-24])
    assert is_orthonormal(R) and np.abs(R - eye).max() <= K_TOLERANCE
    # XRotationToRotationMatrix :398-408 (column-major expected)
    R = orc.angle_axis_to_rotation_matrix([np.pi / 2, 0.0, 0.0])
    assert is_orthonormal(R) and np.abs(R - [1, 0, 0, 0, 0, 1, 0, -1, 0]).max() <= K_TOLERANCE
    # YRotationToRotationMatrix :413-424
    R = orc.angle_axis_to_rotation_matrix([0.0, np.pi, 0.0])
    assert is_orthonormal(R) and np.abs(R - [-1, 0, 0, 0, 1, 0, 0, 0, -1]).max() <= K_TOLERANCE


def test_rotate_point_known_answers():
    # the same four rotations applied to points: R p read off the expected matrices above
    assert np.abs(orc.angle_axis_rotate_point([0, 0, 0], [1, 2, 3]) - [1, 2, 3]).max() <= K_TOLERANCE
    assert np.abs(orc.angle_axis_rotate_point([np.pi / 2, 0, 0], [1, 2, 3]) - [1, -3, 2]).max() <= 4 * K_TOLERANCE
    assert np.abs(orc.angle_axis_rotate_point([0, np.pi, 0], [1, 2, 3]) - [-1, 2, -3]).max() <= 4 * K_TOLERANCE


def _check(angle_axis, p):
    R = orc.angle_axis_to_rotation_matrix(angle_axis)
    by_matrix = np.array([R[0] * p[0] + R[3] * p[1] + R[6] * p[2],
                          R[1] * p[0] + R[4] * p[1] + R[7] * p[2],
                          R[2] * p[0] + R[5] * p[1] + R[8] * p[2]])
    return np.abs(by_matrix - orc.angle_axis_rotate_point(angle_axis, p)).max()


def test_rotate_point_gives_same_answer_as_rotation_matrix():
    # rotation_test.cc:1707-1750: theta sweeps (-pi, pi) in 10 000 steps, 50 draws each (here 5: 50 000 cases)
    rng = np.random.default_rng(0)
    worst = 0.0
    for i in range(10000):
        theta = (2.0 * i * 0.0011 - 1.0) * np.pi
        for _ in range(5):
            angle_axis = rng.uniform(-1.0, 1.0, 3)
            p = rng.uniform(-1.0, 1.0, 3)
            angle_axis *= theta / np.hypot(np.hypot(angle_axis[0], angle_axis[1]), angle_axis[2])
            worst = max(worst, _check(angle_axis, p))
    assert worst <= K_TOLERANCE, worst


def test_near_zero_rotate_point_gives_same_answer_as_rotation_matrix():
    # rotation_test.cc:1766-1800, including the reference's `norm2 = ...` (only the last component) as written
    rng = np.random.default_rng(1)
    worst = 0.0
    for i in range(10000):
        angle_axis = rng.uniform(-1.0, 1.0, 3)
        p = rng.uniform(-1.0, 1.0, 3)
        norm2 = angle_axis[2] * angle_axis[2]
        theta = (2.0 * i * 0.0001 - 1.0) * 1e-16
        angle_axis *= theta / np.sqrt(norm2)
        worst = max(worst, _check(angle_axis, p))
    # exactly zero: the Taylor branch of both functions
    worst = max(worst, _check(np.zeros(3), np.array([0.3, -0.7, 0.2])))
    assert worst <= K_TOLERANCE, worst
