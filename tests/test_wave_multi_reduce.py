"""The transpose-reduce of cx_kernels.h (WaveMulti: N sums over the 64 lanes of a wavefront in ~2N shuffles) restated on
the host: after the six steps every value index in [0, N) is the finished total in EXACTLY one lane, and that lane is the
one WaveMulti<N, 32>::index names.  CPU only -- the device code is exercised by the parity tests of k_cam_init / k_cam_diag /
k_cam_ft (test_gpu_parity.py); this pins the algorithm they rely on, for every N the kernels use and all others up to 64."""
import numpy as np
import pytest


def run(values):
    """values[lane][j]: what WaveMulti<N, 32>::run leaves in every lane (one number each)."""
    lanes, n = values.shape
    cur = [list(values[l]) for l in range(lanes)]
    mask = 32
    while mask >= 1:
        h = (n + 1) // 2
        nxt = []
        for l in range(lanes):
            upper = (l & mask) != 0
            partner = l ^ mask
            row = []
            for j in range(h):
                a = cur[l][j]
                b = cur[l][j + h] if j + h < n else 0.0
                pa = cur[partner][j]
                pb = cur[partner][j + h] if j + h < n else 0.0
                keep = b if upper else a
                recv = pb if upper else pa          # the partner hands over the half it does not keep
                row.append(keep + recv)
            nxt.append(row)
        cur, n, mask = nxt, h, mask // 2
    assert n == 1
    return np.array([c[0] for c in cur])


def index(lane, n, mask=32):
    h = (n + 1) // 2
    pos = 0
    if mask > 1:
        pos = index(lane, h, mask // 2)
        if pos < 0:
            return -1
    src = pos + (h if lane & mask else 0)
    return src if src < n else -1


@pytest.mark.parametrize("n", [1, 2, 3, 9, 45, 54, 63, 64] + list(range(4, 64, 7)))
def test_every_value_ends_in_exactly_one_lane(n):
    idx = [index(l, n) for l in range(64)]
    owners = [i for i in idx if i >= 0]
    assert sorted(owners) == list(range(n))


@pytest.mark.parametrize("n", [9, 45, 54])
def test_totals(n):
    rng = np.random.default_rng(n)
    v = rng.integers(-1000, 1000, size=(64, n)).astype(np.float64)   # integers: the sums are exact in any order
    out = run(v)
    for lane in range(64):
        i = index(lane, n)
        if i >= 0:
            assert out[lane] == v[:, i].sum()
