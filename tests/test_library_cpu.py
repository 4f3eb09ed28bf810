"""CPU-side checks of the C-ABI library: it loads, exports every symbol
include/cxschur.h declares, fails loudly without a GPU, and the host-side helpers
(no device needed) agree with the oracle."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, cx, lls_problem


def test_header_symbols_are_exported():
    lib = cx.load_library()
    header = open(os.path.join(ROOT, "include", "cxschur.h")).read()
    declared = set(re.findall(r"\b(cx_[a-z0-9_]+)\s*\(", header))
    declared -= {"cx_block", "cx_cell"}
    assert declared, "no declarations found"
    assert declared == set(cx.EXPORTED_SYMBOLS), declared ^ set(cx.EXPORTED_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(lib, name), "libcxschur.so does not export %s" % name


def test_no_cpu_fallback():
    """Without a gfx950 device context creation must fail with an error, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(cx.CxError):
        cx.Context(0)


def test_detect_structure_host(oracle):
    for pid in (2, 4, 6):
        bs, values, b, D, nelim, raw = lls_problem(pid)
        assert cx.detect_structure(bs, nelim) == oracle.detect_structure(bs, nelim)
    prob = cx.bal.make_bal_like(7, 50, 160, seed=9)
    bs, order = cx.bal.build_structure(prob)
    assert cx.detect_structure(bs, prob.num_points) == (2, 3, 9)


def test_partition_points_balanced():
    prob = cx.bal.make_bal_like(12, 500, 2600, seed=4)
    bs, order = cx.bal.build_structure(prob)
    for nranks in (1, 2, 3, 8):
        bounds = cx.partition_points(bs, prob.num_points, nranks)
        assert bounds[0] == 0 and bounds[-1] == prob.num_points and np.all(np.diff(bounds) >= 0)
        counts = np.bincount(prob.point_index, minlength=prob.num_points)
        per = [counts[bounds[i]:bounds[i + 1]].sum() for i in range(nranks)]
        assert sum(per) == prob.num_observations
        assert max(per) - min(per) <= counts.max() * 2 + 1
        assert np.array_equal(bounds, cx.bal.partition_points(prob, nranks))


def test_stable_schur_ordering_matches_oracle(oracle):
    """Integer outcome, must be identical: ComputeStableSchurOrdering on BAL graphs, including a
    camera with fewer points than its points have cameras and a repeated observation."""
    rng = np.random.default_rng(0)
    cases = []
    prob = cx.bal.make_bal_like(9, 300, 1300, seed=21)
    cases.append((prob.num_cameras, prob.num_points, prob.camera_index, prob.point_index))
    cam = np.array([0, 1, 2, 1, 2, 1, 2, 2], dtype=np.int32)       # duplicate (2, 2)
    pt = np.array([0, 0, 0, 1, 1, 2, 2, 2], dtype=np.int32)
    cases.append((3, 3, cam, pt))
    C, P = 20, 60
    cam = rng.integers(0, C, 400).astype(np.int32)
    pt = rng.integers(0, P, 400).astype(np.int32)
    cases.append((C, P, cam, pt))
    for C, P, cam, pt in cases:
        o1, k1 = cx.stable_schur_ordering(C, P, cam, pt)
        o2, k2 = oracle.stable_schur_ordering(C, P, cam, pt)
        assert k1 == k2 and np.array_equal(o1, o2)


@pytest.mark.parametrize("clustering", ["CANONICAL_VIEWS", "SINGLE_LINKAGE"])
@pytest.mark.parametrize("pre", ["CLUSTER_JACOBI", "CLUSTER_TRIDIAGONAL"])
@pytest.mark.parametrize("C,P,O,seed", [(6, 40, 130, 1), (16, 700, 2800, 2), (49, 7776, 31843, 49), (100, 3000, 14000, 3),
                                        (400, 9000, 40000, 4)])
def test_visibility_clusters_host_matches_oracle(oracle, C, P, O, seed, pre, clustering):
    """The library's host-side visibility clustering (flat arrays, cached quality differences, per-point cluster
    counting) against the oracle's restatement with ordered sets and maps: integers, compared exactly.  No device."""
    prob = cx.bal.make_bal_like(C, P, O, seed)
    bs, _ = cx.bal.build_structure(prob)
    m, k, cp = cx.binding.visibility_clusters_host(bs, P, getattr(cx, pre), getattr(cx, clustering))
    mr, kr, cpr, _ = oracle.visibility_structure(bs, P, getattr(oracle, pre), getattr(oracle, clustering))
    assert k == kr
    assert np.array_equal(m, mr)
    assert np.array_equal(cp, cpr)


def test_visibility_clusters_host_larger_problem_and_layout_check():
    """2 000 cameras: the clustering alone (no oracle at this size, it is quadratic in the clusters): every camera
    gets a cluster, clusters are numbered by first appearance, the forest has at most K - 1 edges of degree <= 2;
    structures outside the static layout are refused."""
    prob = cx.bal.make_bal_like(2000, 60000, 300000, 11)
    bs, _ = cx.bal.build_structure(prob)
    m, k, cp = cx.binding.visibility_clusters_host(bs, prob.num_points, cx.CLUSTER_TRIDIAGONAL, cx.CANONICAL_VIEWS)
    assert m.min() == 0 and m.max() == k - 1
    first = {}
    for c, cl in enumerate(m.tolist()):
        first.setdefault(cl, c)
    assert [first[i] for i in range(k)] == sorted(first.values())
    off = cp[cp[:, 0] != cp[:, 1]]
    assert len(off) <= k - 1 and np.bincount(off.ravel(), minlength=k).max() <= 2
    bs6, values, b, D, nelim, raw = lls_problem(6)
    with pytest.raises(cx.CxError):
        cx.binding.visibility_clusters_host(bs6, nelim, cx.CLUSTER_JACOBI)


@pytest.mark.parametrize("seed", range(40))
def test_visibility_clusters_host_fuzz(oracle, seed):
    """Random small structures -- few cameras, repeated visibility patterns, cameras nobody sees, so exact ties between
    candidate views and equal forest weights are common: the library's clustering and the oracle's must still agree
    entry for entry (both break ties by ascending id)."""
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.integers(2, 14))
    P = int(rng.integers(5, 60))
    patterns = [sorted(rng.choice(C, size=int(rng.integers(1, min(C, 5) + 1)), replace=False).tolist())
                for _ in range(int(rng.integers(1, 6)))]
    lists = [patterns[int(rng.integers(0, len(patterns)))] for _ in range(P)]
    cam, pt = [], []
    for j, cams in enumerate(lists):
        cam.extend(cams)
        pt.extend([j] * len(cams))
    cam, pt = np.array(cam, dtype=np.int32), np.array(pt, dtype=np.int32)
    O = cam.size
    prob = cx.bal.BalProblem(C, P, cam, pt, np.zeros((O, 2)), np.zeros((C, 9)), np.zeros((P, 3)))
    bs, _ = cx.bal.build_structure(prob)
    for pre in ("CLUSTER_JACOBI", "CLUSTER_TRIDIAGONAL"):
        for clustering in ("CANONICAL_VIEWS", "SINGLE_LINKAGE"):
            m, k, cp = cx.binding.visibility_clusters_host(bs, P, getattr(cx, pre), getattr(cx, clustering))
            mr, kr, cpr, _ = oracle.visibility_structure(bs, P, getattr(oracle, pre), getattr(oracle, clustering))
            assert k == kr and np.array_equal(m, mr) and np.array_equal(cp, cpr), (seed, pre, clustering)


def _plan_checks(oracle, C, P, O, seed):
    """Host half of the tile-sparse Cholesky plan (no device) on the InitStorage cell list of a synthetic scene."""
    prob = cx.bal.make_bal_like(C, P, O, seed)
    bs, _ = cx.bal.build_structure(prob)
    r, c = oracle.schur_sparse_structure(bs, P)
    plan = cx.binding.sparse_cholesky_plan_host(C, r, c)
    T, first = plan["num_tile_rows"], plan["camera_first_row"]
    # layout: every camera owns 9 rows of its own inside the padded range
    rows = (first[:, None] + np.arange(9)[None, :]).ravel()
    assert rows.min() >= 0 and rows.max() < 64 * T and np.unique(rows).size == 9 * C
    # symbolic fill, recomputed with dense booleans at tile level
    nz = np.zeros((T, T), dtype=bool)
    for c1, c2 in zip(r, c):
        a, b = sorted((first[c1], first[c2]))
        nz[np.ix_(np.arange(a >> 6, ((a + 8) >> 6) + 1), np.arange(b >> 6, ((b + 8) >> 6) + 1))] = True
    nz = np.triu(nz | nz.T) | np.eye(T, dtype=bool)
    for k in range(T):
        later = np.nonzero(nz[k, k + 1:])[0] + k + 1
        nz[np.ix_(later, later)] |= later[:, None] <= later[None, :]
    start, cols, level = plan["tile_row_start"], plan["tile_cols"], plan["tile_row_level"]
    assert start[-1] == plan["num_tiles"] == cols.size
    pairs = 0
    for i in range(T):
        lst = cols[start[i]:start[i + 1]]
        assert lst[-1] == T                                     # the right-hand-side tile closes every row
        assert np.array_equal(lst[:-1], np.nonzero(nz[i])[0])   # diagonal first, then the filled row, ascending
        m = lst.size - 1                                        # real tiles incl. the diagonal
        pairs += (m - 1) * m // 2 + (m - 1)                      # (Ja <= Jb) right of the diagonal + the right-hand side column
        # a tile row is factored after every row that updates it
        for j in lst[1:-1]:
            assert level[j] > level[i]
    assert pairs == plan["num_tile_pair_updates"]
    assert plan["num_levels"] == level.max() + 1
    again = cx.binding.sparse_cholesky_plan_host(C, r, c)
    assert np.array_equal(again["camera_first_row"], first) and np.array_equal(again["tile_cols"], cols)
    return plan


@pytest.mark.parametrize("C,P,O,seed", [(6, 40, 130, 1), (49, 7776, 31843, 49), (400, 9000, 40000, 4)])
def test_sparse_cholesky_plan_host(oracle, C, P, O, seed):
    _plan_checks(oracle, C, P, O, seed)


def test_sparse_cholesky_plan_host_dissects_a_long_scene(oracle):
    """3 000 cameras on the generator's ring with a narrow visibility window: the dissection must find independent subtrees
    (levels well below the number of tile rows) without blowing up the fill."""
    plan = _plan_checks(oracle, 3000, 60000, 260000, 5)
    T = plan["num_tile_rows"]
    assert T >= 3000 * 9 // 64
    assert plan["num_levels"] < T // 2, (plan["num_levels"], T)


def test_loss_function_recovery_on_the_host():
    """host/cx_loss_probe.h: the constructor arguments of every built-in LossFunction (private in the reference,
    loss_function.h:174-292) are recovered through LossFunction::Evaluate, so that CxBalEvaluator::TryCreate can accept
    bundle_adjuster --robustify programs; ScaledLoss and other non-built-in losses are declined.  Pure host code."""
    import subprocess
    host = os.path.join(ROOT, "ceres-solver-ceres-solver_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host, "test_loss_probe"])
    out = subprocess.run([os.path.join(host, "test_loss_probe")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("C,P,O,seed", [(6, 40, 150, 1), (14, 900, 4200, 6), (60, 1500, 7000, 12), (3, 4, 9, 1)])
def test_pair_lists_host_match_the_oracle_cell_set(oracle, C, P, O, seed):
    """cx_schur_pair_lists_host (the structure half of the gather assembly of S, no device): the cell list is
    SparseSchurComplementSolver::InitStorage's (schur_complement_solver.cc:224-290) -- bit-exact against the oracle -- and
    the pairs of a cell are exactly the row pairs (row of c1, row of c2) of every chunk that sees both cameras, chunks in
    order: integers, compared exactly with a brute-force numpy enumeration."""
    prob = cx.bal.make_bal_like(C, P, O, seed=seed)
    bs, _ = cx.bal.build_structure(prob)
    r, c, num_pairs, num_items, pairs = cx.binding.schur_pair_lists_host(bs, P, want_pairs=True)
    r_ref, c_ref = oracle.schur_sparse_structure(bs, P)
    assert np.array_equal(r, r_ref) and np.array_equal(c, c_ref)
    row_pt = bs.cells["block_id"][0::2].astype(np.int64)
    row_cam = bs.cells["block_id"][1::2].astype(np.int64) - P
    want = {}
    start = np.searchsorted(row_pt, np.arange(P + 1))
    for p in range(P):
        rows = range(start[p], start[p + 1])
        for i in rows:
            for j in rows:
                if j < i:
                    continue
                ci, cj = row_cam[i], row_cam[j]
                key = (min(ci, cj), max(ci, cj))
                want.setdefault(key, []).append((i, j) if ci <= cj else (j, i))
    assert num_pairs == sum(len(v) for v in want.values())
    pairs = pairs.reshape(-1, 2)
    pos = 0
    for c1, c2 in zip(r, c):
        lst = want.get((c1, c2), [])
        got = [tuple(q) for q in pairs[pos:pos + len(lst)]]
        assert got == lst, (c1, c2)
        pos += len(lst)
    assert pos == num_pairs and num_items >= len(r)


@pytest.mark.parametrize("C,P,O,seed", [(700, 5000, 26000, 8), (2000, 20000, 110000, 3)])
def test_distribution_of_the_tile_sparse_factorisation(C, P, O, seed):
    """cx_sparse_cholesky_distribution_host (the host half of the distributed SPARSE_SCHUR factorisation): every tile row is
    either owned by exactly one rank or part of the replicated top; an owned row's whole subtree of the elimination tree
    belongs to the same rank (so a rank's phase needs nothing from the others), the parent chain of a replicated row is
    replicated (so the top needs nothing after the one exchange), and the ranks' shares of the tile-pair updates are
    balanced.  Integers, checked exactly."""
    prob = cx.bal.make_bal_like(C, P, O, seed=seed)
    bs, _ = cx.bal.build_structure(prob)
    r, c, _, _ = cx.binding.schur_pair_lists_host(bs, P)
    plan = cx.binding.sparse_cholesky_plan_host(C, r, c)
    T, start, cols = plan["num_tile_rows"], plan["tile_row_start"], plan["tile_cols"]
    parent = np.array([cols[start[i] + 1] if cols[start[i] + 1] < T else -1 for i in range(T)])
    one = cx.binding.sparse_cholesky_distribution_host(C, r, c, 1)
    assert one["updates_per_rank"][0] == plan["num_tile_pair_updates"] and one["tiles_replicated"] == 0
    for nranks in (2, 4, 8):
        d = cx.binding.sparse_cholesky_distribution_host(C, r, c, nranks)
        owner = d["owner"]
        assert owner.size == T and owner.min() >= -1 and owner.max() < nranks
        for i in range(T):
            p = parent[i]
            if p < 0:
                continue
            if owner[i] < 0:
                assert owner[p] < 0                          # above a replicated row everything is replicated
            else:
                assert owner[p] in (-1, owner[i])            # a subtree does not change hands
        per = d["updates_per_rank"]
        assert per.sum() > 0 and d["updates_replicated"] >= 0
        if T >= 16 * nranks:
            assert per.max() <= 1.5 * per.mean() + 1, (nranks, per)      # largest-first onto the least loaded rank
        replicated_rows = int((owner < 0).sum())
        assert d["tiles_replicated"] == sum(start[i + 1] - start[i] for i in range(T) if owner[i] < 0)
        assert replicated_rows < T


@pytest.mark.parametrize("C,P,O,seed", [(700, 5000, 26000, 8), (2000, 20000, 110000, 3)])
def test_update_windows_of_the_tile_sparse_factorisation(C, P, O, seed, monkeypatch):
    """cx_sparse_cholesky_schedule_host: the schedule of the tile-pair updates under an update window (a target's contributions of
    `window` levels subtracted as one chain), checked from its definition on the host: every product of the symbolic factor is
    scheduled exactly once, after its source row is factored and before its target's row is, ascending source rows inside a chain;
    on a distributed plan a rank's own products all run before the exchange.  Integers, checked exactly."""
    prob = cx.bal.make_bal_like(C, P, O, seed=seed)
    bs, _ = cx.bal.build_structure(prob)
    r, c, _, _ = cx.binding.schur_pair_lists_host(bs, P)
    plan = cx.binding.sparse_cholesky_plan_host(C, r, c)
    # (windows open only at levels wide enough to fill the chip: these small structures have none, so the rule is switched off)
    wide = cx.binding.sparse_cholesky_schedule_host(C, r, c, window=4)
    monkeypatch.setenv("CX_SPARSE_WINDOW_MIN_PRODUCTS", "0")
    one = cx.binding.sparse_cholesky_schedule_host(C, r, c, window=1)
    assert wide["violations"] == 0 and wide["num_chains"] == one["num_chains"]
    assert one["violations"] == 0 and one["num_tile_rows"] == plan["num_tile_rows"]
    assert one["num_products"] == plan["num_tile_pair_updates"]
    chains = [one["num_chains"]]
    for window in (2, 4, 16, 1000):
        w = cx.binding.sparse_cholesky_schedule_host(C, r, c, window=window)
        assert w["violations"] == 0 and w["num_products"] == one["num_products"]
        assert w["num_chains"] <= chains[-1] and w["longest_chain"] >= one["longest_chain"]   # same work in fewer, longer chains
        chains.append(w["num_chains"])
    assert chains[-1] < chains[0]
    for nranks in (2, 4):
        total = 0
        for rank in range(nranks):
            w = cx.binding.sparse_cholesky_schedule_host(C, r, c, window=4, nranks=nranks, rank=rank)
            assert w["violations"] == 0
            total += w["num_products"]
        d = cx.binding.sparse_cholesky_distribution_host(C, r, c, nranks)
        m = np.diff(plan["tile_row_start"]) - 1                         # tiles right of the diagonal, right-hand side included
        replicated = int((m * (m + 1) // 2 - 1)[d["owner"] < 0].sum())
        assert total == one["num_products"] + (nranks - 1) * replicated  # the top is factored by every rank


def test_ticket_reductions_keep_their_instruction_order():
    """The fused reductions (cx_solver.hip: dot2_finish, cx_eval.hip: k_sum_partials) finish in the workgroup that draws the
    last ticket.  Their ordering rests on instructions, not on a fence: the partial sums are stored with
    agent-scope (sc1) stores that are COMPLETE (s_waitcnt vmcnt(0)) before the ticket atomic is issued, and the last workgroup
    reads them with agent-scope (sc1) loads issued only after its own ticket atomic has returned (s_waitcnt vmcnt(0) behind
    the returning atomic).  A compiler that reorders or weakens any of it would break the reductions silently on some
    launches, so the gfx950 code objects of the build are disassembled and checked here, kernel by kernel.  No GPU."""
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    csrc = os.path.join(ROOT, "ceres-solver-ceres-solver_amd", "csrc")
    checked = 0
    for unit in ("cx_solver.o", "cx_eval.o"):
        obj = os.path.join(csrc, unit)
        if not os.path.exists(obj):      # (a checkout that was never built: compile now, as __graft_entry__.build() does)
            subprocess.check_call([sys.executable, os.path.join(ROOT, "ceres-solver-ceres-solver_amd", "build.py")], stdout=subprocess.DEVNULL)
        subprocess.run([objdump, "--offloading", obj], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        code = [f for f in os.listdir(csrc) if f.startswith(unit + ".") and "gfx950" in f]
        try:
            assert code, "no gfx950 code object in " + unit
            asm = subprocess.run([objdump, "-d", os.path.join(csrc, code[0])], capture_output=True, text=True, check=True).stdout
        finally:
            for f in os.listdir(csrc):
                if f.startswith(unit + "."):
                    os.remove(os.path.join(csrc, f))
        for fn in re.split(r"\n(?=[0-9a-f]{16} <)", asm):
            name = fn.split(">:")[0]
            ins = [l.split("//")[0].strip() for l in fn.splitlines()[1:] if l.strip()]
            tickets = [i for i, l in enumerate(ins) if re.match(r"global_atomic_add(_u32)? v\d+,", l)]   # the returning ticket atomic
            for t in tickets:
                stores = [i for i in range(max(0, t - 60), t) if ins[i].startswith("global_store_dwordx2")]
                if not stores:
                    continue            # (an atomic add that is not a reduction's ticket)
                assert all("sc1" in ins[i] for i in stores[-2:]), (name, [ins[i] for i in stores[-2:]])
                assert any(ins[i].startswith("s_waitcnt vmcnt(0)") for i in range(stores[-1], t)), (name, "stores not complete before the ticket")
                loads = [i for i in range(t + 1, min(len(ins), t + 120)) if ins[i].startswith("global_load_dwordx2") and "sc1" in ins[i]]
                assert loads, (name, "the partial sums are not read with agent-scope loads")
                assert any(ins[i].startswith("s_waitcnt vmcnt(0)") for i in range(t, loads[0])), (name, "loads issued before the ticket returned")
                checked += 1
    assert checked >= 5, checked     # k_dot2_fused, k_blockdiag9_dot, k_cam_reduce9_dot, k_update_xr_dot, k_sum_partials, ...
