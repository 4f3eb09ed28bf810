"""Synthetic BAL-shaped bundle-adjustment problems (host side, numpy).

No BAL file exists in the build image (the reference's
data/problem-16-22106-pre.txt is a stripped blob), so benchmarks and parity tests
run on deterministic problems with the public BAL header sizes: C cameras with 9
parameters (angle-axis, translation, focal, k1, k2 -- the camera model of
examples/snavely_reprojection_error.h:53-104), P points, O observations listed
point-major like BAL files (examples/bal_problem.cc:73-130).
"""
import dataclasses
from dataclasses import dataclass

import numpy as np

from .structure import BLOCK_DTYPE, CELL_DTYPE, BlockStructure

# name -> (cameras, points, observations, seed); sizes are the public BAL headers
PRESETS = {
    "ladybug16": (16, 22106, 83718, 16),
    "ladybug49": (49, 7776, 31843, 49),
    "dubrovnik356": (356, 226730, 1255268, 356),
    "final13682": (13682, 4456117, 28987644, 13682),
    "synthetic10M": (5000, 1500000, 10000000, 10),
    # the Final-13682 sizes on a second kind of scene: the trajectory passes every place twice (5 % of the points are also
    # seen from the cameras half a ring away), so the camera graph is a ring with chords -- loop closures -- instead of a band
    "final13682_revisit": (13682, 4456117, 28987644, 13682),
    "dubrovnik356_revisit": (356, 226730, 1255268, 356),
}
# generator options of a preset beyond (cameras, points, observations, seed)
PRESET_OPTIONS = {"final13682_revisit": {"revisit_fraction": 0.05}, "dubrovnik356_revisit": {"revisit_fraction": 0.05}}


@dataclass
class BalProblem:
    num_cameras: int
    num_points: int
    camera_index: np.ndarray   # [O] int32, input (file) order
    point_index: np.ndarray    # [O] int32
    observations: np.ndarray   # [O, 2] float64
    cameras: np.ndarray        # [C, 9]
    points: np.ndarray         # [P, 3]

    @property
    def num_observations(self):
        return int(self.camera_index.shape[0])

    def state(self):
        """Parameter vector in column order: points then cameras (the order
        ApplyOrdering gives bundle_adjuster's user ordering, reorder_program.cc:216-254)."""
        return np.concatenate([self.points.ravel(), self.cameras.ravel()])


def _rodrigues(aa, X):
    theta = np.linalg.norm(aa, axis=1, keepdims=True)
    theta = np.where(theta == 0.0, 1.0, theta)
    w = aa / theta
    c, s = np.cos(theta), np.sin(theta)
    return X * c + np.cross(w, X) * s + w * (np.sum(w * X, axis=1, keepdims=True) * (1.0 - c))


def project(cameras, points, camera_index, point_index):
    """Snavely projection of the listed (camera, point) pairs; [O, 2]."""
    cam = cameras[camera_index]
    p = _rodrigues(cam[:, 0:3], points[point_index]) + cam[:, 3:6]
    xp = -p[:, 0] / p[:, 2]
    yp = -p[:, 1] / p[:, 2]
    r2 = xp * xp + yp * yp
    d = 1.0 + r2 * (cam[:, 7] + cam[:, 8] * r2)
    return np.stack([cam[:, 6] * d * xp, cam[:, 6] * d * yp], axis=1)


def _track_lengths(rng, C, P, O):
    """k_j >= 2 observations per point, sum = O, k_j <= C."""
    assert 2 * P <= O <= C * P, "need 2P <= O <= C*P"
    extra = O - 2 * P
    mean_extra = extra / P
    kmax = C - 2
    if mean_extra <= 0:
        k = np.zeros(P, dtype=np.int64)
    else:
        p = 1.0 / (1.0 + mean_extra)
        k = np.minimum(rng.geometric(p, size=P) - 1, kmax).astype(np.int64)
    diff = int(extra - k.sum())
    # fix the total deterministically by single increments / decrements
    while diff != 0:
        if diff > 0:
            cand = np.flatnonzero(k < kmax)
            take = cand[rng.permutation(cand.size)[:min(diff, cand.size)]]
            k[take] += 1
            diff -= take.size
        else:
            cand = np.flatnonzero(k > 0)
            take = cand[rng.permutation(cand.size)[:min(-diff, cand.size)]]
            k[take] -= 1
            diff += take.size
    return k + 2


def make_bal_like(C, P, O, seed, noise_px=0.5, perturb=True, revisit_fraction=0.0):
    """Cameras on a noisy ring looking at a point cloud; each point is seen from a
    window of neighbouring cameras, so the reduced camera matrix is banded like
    real SfM data.  Deterministic in (C, P, O, seed).
    revisit_fraction > 0: that share of the points (those with at least four observations) has its last two
    observations taken by the cameras half a ring away instead -- loop closures: the co-visibility graph of the
    cameras becomes a ring with chords, the structure nested dissection and the visibility clusterings of real
    reconstructions have to cope with.  The observation count stays O."""
    rng = np.random.default_rng(seed)
    # ---- scene
    ang = 2.0 * np.pi * np.arange(C) / C
    radius = 300.0 * (1.0 + 0.05 * rng.standard_normal(C))
    centers = np.stack([radius * np.cos(ang), radius * np.sin(ang), 20.0 * rng.standard_normal(C)], axis=1)
    z = centers / np.linalg.norm(centers, axis=1, keepdims=True)      # camera looks down -z at the origin
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(up, z)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=1)                                   # world -> camera
    from scipy.spatial.transform import Rotation
    aa = Rotation.from_matrix(R).as_rotvec()
    t = -np.einsum("cij,cj->ci", R, centers)
    cameras = np.zeros((C, 9))
    cameras[:, 0:3] = aa
    cameras[:, 3:6] = t
    cameras[:, 6] = 1000.0 * (1.0 + 0.1 * rng.standard_normal(C))
    cameras[:, 7] = -0.05 + 0.01 * rng.standard_normal(C)
    cameras[:, 8] = 0.01 + 0.002 * rng.standard_normal(C)
    points = 30.0 * rng.standard_normal((P, 3))
    # ---- visibility: point j is seen by k_j distinct cameras around a home camera
    k = _track_lengths(rng, C, P, O)
    home = (np.arange(P, dtype=np.int64) * C) // max(P, 1)
    home = (home + rng.integers(0, max(1, C // 8) + 1, size=P)) % C
    max_stride = np.maximum(1, np.minimum(3, C // np.maximum(k, 1)))
    stride = 1 + (rng.integers(0, 3, size=P) % max_stride)
    point_index = np.repeat(np.arange(P, dtype=np.int64), k)
    first = np.concatenate([[0], np.cumsum(k)[:-1]])
    within = np.arange(O, dtype=np.int64) - np.repeat(first, k)
    camera_index = (np.repeat(home, k) + within * np.repeat(stride, k)) % C
    if revisit_fraction > 0.0:
        # a generator of its own: the presets without revisits keep their random stream (and their published numbers)
        revisit = (np.random.default_rng(seed + 7919).random(P) < revisit_fraction) & (k >= 4) & (3 * k < C // 2)
        far = np.repeat(revisit, k) & (within >= np.repeat(k, k) - 2)
        camera_index = np.where(far, (camera_index + C // 2) % C, camera_index)
    # BAL lists observations point-major with ascending camera index
    key = point_index * C + camera_index
    order = np.argsort(key, kind="stable")
    camera_index = camera_index[order].astype(np.int32)
    point_index = point_index[order].astype(np.int32)
    obs = project(cameras, points, camera_index, point_index)
    obs += noise_px * rng.standard_normal(obs.shape)
    if perturb:  # BALProblem::Perturb (bal_problem.cc:294-333): iid noise on every block ...
        cameras[:, 0:3] += 1e-4 * rng.standard_normal((C, 3))
        cameras[:, 3:6] += 1e-2 * rng.standard_normal((C, 3))
        points += 5e-2 * rng.standard_normal((P, 3))
        # ... plus a smooth drift along the camera ring (the low-frequency error real SfM
        # reconstructions accumulate); without it CG on the reduced system converges in two
        # iterations, which no real BAL problem does
        s = np.arange(C) / C
        for m in (1, 2, 3, 5, 8):
            cameras[:, 3:6] += (2.0 / m) * np.sin(2 * np.pi * m * s)[:, None] * rng.standard_normal(3)
            cameras[:, 0:3] += (2e-3 / m) * np.cos(2 * np.pi * m * s)[:, None] * rng.standard_normal(3)
    return BalProblem(C, P, camera_index, point_index, obs, cameras, points)


def make_preset(name, **kw):
    C, P, O, seed = PRESETS[name]
    return make_bal_like(C, P, O, seed, **{**PRESET_OPTIONS.get(name, {}), **kw})


def residual_order(point_index, num_points):
    """order[k] = input observation at row block k: residual blocks bucketed by
    point, each bucket filled back to front, i.e. REVERSE input order inside a
    chunk (LexicographicallyOrderResidualBlocks, reorder_program.cc:256-338)."""
    O = point_index.shape[0]
    rev = point_index[::-1]
    return (O - 1 - np.argsort(rev, kind="stable")).astype(np.int64)


def build_structure(problem, order=None):
    """Block structure of J in the BuildJacobianLayout layout
    (block_jacobian_writer.cc:68-167): E cells (2x3) packed first in row order,
    then F cells (2x9); column blocks = points then cameras.  Returns
    (BlockStructure, order)."""
    C, P, O = problem.num_cameras, problem.num_points, problem.num_observations
    assert 24 * O < 2 ** 31, "cell positions are int32 (block_jacobian_writer.cc:95-99)"
    if order is None:
        order = residual_order(problem.point_index, P)
    cols = np.zeros(P + C, dtype=BLOCK_DTYPE)
    cols["size"][:P] = 3
    cols["size"][P:] = 9
    cols["position"][:P] = 3 * np.arange(P)
    cols["position"][P:] = 3 * P + 9 * np.arange(C)
    rows = np.zeros(O, dtype=BLOCK_DTYPE)
    rows["size"] = 2
    rows["position"] = 2 * np.arange(O)
    rcb = (2 * np.arange(O + 1)).astype(np.int32)
    cells = np.zeros(2 * O, dtype=CELL_DTYPE)
    cells["block_id"][0::2] = problem.point_index[order]
    cells["position"][0::2] = 6 * np.arange(O)
    cells["block_id"][1::2] = P + problem.camera_index[order]
    cells["position"][1::2] = 6 * O + 18 * np.arange(O)
    return BlockStructure(rows, cols, rcb, cells), order


def random_jacobian_values(num_observations, seed):
    """J values ~ N(0,1) on the BAL structure, as the reference's kernel
    benchmarks do (evaluation_benchmark.cc:147-153)."""
    return np.random.default_rng(seed).standard_normal(24 * num_observations)


def partition_points(problem_or_counts, nranks, num_points=None):
    """Contiguous point ranges with about equal observation counts (host mirror
    of cx_partition_points); bounds[nranks + 1]."""
    if isinstance(problem_or_counts, BalProblem):
        counts = np.bincount(problem_or_counts.point_index, minlength=problem_or_counts.num_points)
    else:
        counts = np.asarray(problem_or_counts)
    cum = np.concatenate([[0], np.cumsum(counts)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, nranks):
        bounds.append(int(np.searchsorted(cum, total * r / nranks, side="left")))
    bounds.append(len(counts))
    return np.maximum.accumulate(np.array(bounds, dtype=np.int64))


def shard(problem, lo, hi):
    """The sub-problem holding points [lo, hi) and every camera (cameras stay
    replicated; point indices are renumbered from 0)."""
    sel = (problem.point_index >= lo) & (problem.point_index < hi)
    return BalProblem(problem.num_cameras, hi - lo, problem.camera_index[sel],
                      (problem.point_index[sel] - lo).astype(np.int32), problem.observations[sel],
                      problem.cameras, problem.points[lo:hi])


# ----------------------------------------------------------------------------------------------
# BAL text files (examples/bal_problem.cc:73-130 reader, :137-176 writer), Normalize (:249-292)
# and Perturb (:294-333).  Host-side I/O only; nothing here touches the device.
# ----------------------------------------------------------------------------------------------
def read_bal(path):
    """BALProblem::BALProblem: '<C> <P> <O>', O lines 'cam pt x y', then 9C camera and 3P point
    parameters, one per line.  Every token is whitespace separated, as fscanf reads it."""
    with open(path, "r") as f:
        header = []
        while len(header) < 3:
            header += f.readline().split()
        C, P, O = (int(v) for v in header[:3])
        rest = header[3:]
        data = np.fromfile(f, dtype=np.float64, sep=" ")
    if rest:
        data = np.concatenate([np.array([float(v) for v in rest]), data])
    need = 4 * O + 9 * C + 3 * P
    if data.size < need:
        raise ValueError("%s: expected %d numbers after the header, found %d" % (path, need, data.size))
    obs = data[:4 * O].reshape(O, 4)
    cam_idx = obs[:, 0].astype(np.int32)
    pt_idx = obs[:, 1].astype(np.int32)
    if O and (cam_idx.min() < 0 or cam_idx.max() >= C or pt_idx.min() < 0 or pt_idx.max() >= P):
        raise ValueError("%s: observation index out of range" % path)
    cams = data[4 * O:4 * O + 9 * C].reshape(C, 9).copy()
    pts = data[4 * O + 9 * C:need].reshape(P, 3).copy()
    return BalProblem(C, P, cam_idx, pt_idx, np.ascontiguousarray(obs[:, 2:4]), cams, pts)


def write_bal(path, prob):
    """BALProblem::WriteToFile: observations with %g, parameters with %.16g."""
    with open(path, "w") as f:
        f.write("%d %d %d\n" % (prob.num_cameras, prob.num_points, prob.num_observations))
        for c, q, (x, y) in zip(prob.camera_index, prob.point_index, prob.observations):
            f.write("%d %d %g %g\n" % (c, q, x, y))
        for v in prob.cameras.ravel():
            f.write("%.16g\n" % v)
        for v in prob.points.ravel():
            f.write("%.16g\n" % v)


def _median_nth(values):
    """Median() of bal_problem.cc:66-70: the element at index size/2 after nth_element."""
    v = np.asarray(values)
    k = v.size // 2
    return float(np.partition(v, k)[k])


def camera_centers(cameras):
    """CameraToAngleAxisAndCenter: c = -R' t."""
    return -_rodrigues(-cameras[:, 0:3], cameras[:, 3:6])


def normalize(prob):
    """BALProblem::Normalize: marginal median to the origin, median absolute deviation (l1) to 100."""
    pts = prob.points
    median = np.array([_median_nth(pts[:, i]) for i in range(3)])
    mad = _median_nth(np.abs(pts - median).sum(axis=1))
    scale = 100.0 / mad
    new_pts = scale * (pts - median)
    cams = prob.cameras.copy()
    center = scale * (camera_centers(cams) - median)
    cams[:, 3:6] = -_rodrigues(cams[:, 0:3], center)  # t = -R c
    return dataclasses.replace(prob, cameras=cams, points=new_pts)


def perturb(prob, rotation_sigma, translation_sigma, point_sigma, seed=0):
    """BALProblem::Perturb.  The reference draws from a default-seeded std::mt19937 through
    std::normal_distribution, whose stream numpy cannot reproduce: same distributions, other
    draws.  (As in the reference, the rotation noise uses point_sigma, bal_problem.cc:310-311.)"""
    assert rotation_sigma >= 0 and translation_sigma >= 0 and point_sigma >= 0
    rng = np.random.default_rng(seed)
    pts = prob.points.copy()
    if point_sigma > 0:
        pts += rng.normal(0.0, point_sigma, pts.shape)
    cams = prob.cameras.copy()
    center = camera_centers(cams)
    if rotation_sigma > 0:
        cams[:, 0:3] += rng.normal(0.0, point_sigma, (prob.num_cameras, 3))
    cams[:, 3:6] = -_rodrigues(cams[:, 0:3], center)
    if translation_sigma > 0:
        cams[:, 3:6] += rng.normal(0.0, translation_sigma, (prob.num_cameras, 3))
    return dataclasses.replace(prob, cameras=cams, points=pts)


# ----------------------------------------------------------------------------------------------
# Quaternion cameras (BALProblem(filename, use_quaternions = true), bal_problem.cc:111-130):
# [w x y z | translation 3 | focal k1 k2], conversions of include/ceres/rotation.h:315-388.
# ----------------------------------------------------------------------------------------------
def angle_axis_to_quaternion(aa):
    """AngleAxisToQuaternion for an [n, 3] array -> [n, 4], w first."""
    aa = np.asarray(aa, dtype=np.float64)
    theta = np.sqrt((aa * aa).sum(axis=1))
    nz = theta != 0.0
    safe = np.where(nz, theta, 1.0)
    k = np.where(nz, np.sin(0.5 * safe) / safe, 0.5)
    w = np.where(nz, np.cos(0.5 * safe), 1.0)
    return np.concatenate([w[:, None], aa * k[:, None]], axis=1)


def quaternion_to_angle_axis(q):
    """QuaternionToAngleAxis for an [n, 4] array -> [n, 3]."""
    q = np.asarray(q, dtype=np.float64)
    s = np.sqrt((q[:, 1:] ** 2).sum(axis=1))
    nz = s != 0.0
    c = q[:, 0]
    two_theta = 2.0 * np.where(c < 0.0, np.arctan2(-s, -c), np.arctan2(s, c))
    k = np.where(nz, two_theta / np.where(nz, s, 1.0), 2.0)
    return q[:, 1:] * k[:, None]


def quaternion_cameras(prob):
    """[C, 10] camera array of the quaternion parameterisation."""
    return np.concatenate([angle_axis_to_quaternion(prob.cameras[:, 0:3]), prob.cameras[:, 3:9]], axis=1)


def state_quaternion(prob):
    """Ambient parameter vector [points | 10-parameter cameras] for CX_CAMERA_QUATERNION_MANIFOLD."""
    return np.concatenate([prob.points.ravel(), quaternion_cameras(prob).ravel()])


def cameras_from_quaternion_state(prob, state):
    """Back to a BalProblem with angle-axis cameras (BALProblem::WriteToFile converts the same way)."""
    P, C = prob.num_points, prob.num_cameras
    cams10 = np.asarray(state[3 * P:]).reshape(C, 10)
    cams = np.concatenate([quaternion_to_angle_axis(cams10[:, 0:4]), cams10[:, 4:10]], axis=1)
    return dataclasses.replace(prob, points=np.asarray(state[:3 * P]).reshape(P, 3).copy(), cameras=cams)
