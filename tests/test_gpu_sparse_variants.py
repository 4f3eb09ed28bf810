"""The update kernels of the tile-sparse factorisation that stage their operands through LDS claim the bits of the kernels
they replaced (cx_sparse_chol.hip: same products in the same order).  The kernel switches are read once per process, so each
variant runs in a process of its own (tools/sparse_bits.py prints a SHA-1 of the SPARSE_SCHUR step of a 2 000-camera problem)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(env, *flags):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sparse_bits.py"), *flags], env=e, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]
    assert " 0 " in line, line          # termination_type SUCCESS
    return line.split()[-1]


def test_lds_staged_update_kernels_keep_the_bits():
    assert _digest({"CX_SPARSE_F64_LDS": "1"}) == _digest({"CX_SPARSE_F64_LDS": "0"})                      # k_sp_update_f64_lds == k_sp_update_slices
    assert _digest({"CX_SPARSE_F32_LDS": "1"}, "--mixed") == _digest({"CX_SPARSE_F32_LDS": "0"}, "--mixed")  # k_sp_update_f32_lds == k_sp_update_f32


def test_single_workgroup_cg_tail_keeps_the_bits():
    """problems whose camera vectors have at most 4 096 entries close a CG iteration (and open the next)
    in ONE single-workgroup launch (k_cg_small_tail) instead of four ticket-finished ones -- with the same sums in the same
    order: steps and iteration counts of tools/small_cg_bits.py are those of the general path, to the last bit."""
    def run(env):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "small_cg_bits.py")], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stdout.strip().splitlines()[-2:]
    fused, general = run({}), run({"CX_NO_SMALL_CG": "1"})
    assert fused == general, (fused, general)
    # ... and the set-up of the smallest ones (k_cg_small_setup: ten launches and copies in one) keeps them too
    assert run({"CX_NO_SMALL_SETUP": "1"}) == fused
    assert "(0, " in fused[0]      # SUCCESS somewhere, and more than one residual reset period was crossed
    assert any(int(t.split(")")[0]) > 20 for t in fused[0].split(", ")[1::2])


def test_cgnr_operator_without_memsets_keeps_the_bits():
    """CGNR's operator writes J x and J'(J x) + D^2 x outright (no zeroing, the LM diagonal inside the kernels that write y)
    and its Jacobi preconditioner sums r.z in the launch that forms z: steps and iteration counts of tools/cgnr_bits.py
    (fp64 and fp32 products, with and without D, JACOBI / IDENTITY) are those of round 3's separate launches."""
    def run(env):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cgnr_bits.py")], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stdout.strip().splitlines()[-2:]
    assert run({}) == run({"CX_CGNR_PLAIN": "1"})


def test_assembly_that_also_delivers_the_right_hand_side(tmp_path):
    """The explicit Schur assembly takes the reduced right-hand side out of its own two set-up passes (k_chunk_init<cofactor>
    writes t' next to the inverses, k_cam_init sums F't' next to the F'F blocks): DENSE_SCHUR, SPARSE_SCHUR (fp64 and float
    factor, tile-sparse forced) and explicit-S ITERATIVE_SCHUR steps of tools/schur_bits.py against the four separate passes
    of round 3 -- same termination and iteration counts, steps equal to rounding (1e-11 of |x| through a double precision
    factor; 1e-7 where a single precision factor or a truncated CG run amplifies the last bits of the right-hand side)."""
    import numpy as np

    def run(env, name):
        e = dict(os.environ)
        e.update(env)
        e["CX_SPARSE_CHOLESKY"] = "1"
        path = str(tmp_path / name)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "schur_bits.py"), "--dump", path], env=e, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        z = np.load(path)
        return out.stdout.strip().splitlines()[-2], [z[k] for k in z.files]
    counts_a, xa = run({}, "fused.npz")
    counts_b, xb = run({"CX_ELIMINATE_RHS_SEPARATE": "1"}, "separate.npz")
    assert counts_a == counts_b
    assert len(xa) == len(xb) == 16
    for k, (a, b) in enumerate(zip(xa, xb)):
        exact = (k % 8) in (0, 1, 2, 3)            # DENSE_SCHUR and fp64 SPARSE_SCHUR, with and without D
        tol = 1e-11 if exact else 1e-7
        assert np.linalg.norm(a - b) <= tol * np.linalg.norm(b), (k, np.linalg.norm(a - b) / np.linalg.norm(b))


def test_launch_bound_solver_samples_its_timings_on_request(cxschur):
    """A launch-bound solver takes phase times and kernel samples on its first solve and every 16th after it; in between
    cx_solve_timing.sampled is 0 and cx_solver_sample_next() asks for the next one."""
    cx = cxschur
    if os.environ.get("CX_DIAG_PERIOD"):
        pytest.skip("CX_DIAG_PERIOD overrides the period this test states")
    ctx = cx.Context(0)
    prob = cx.bal.make_preset("ladybug49")
    ev = cx.Evaluator(ctx, prob)
    _, res, _ = ev.evaluate(prob.state())
    A = ev.jacobian()
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=prob.num_points)
    flags = []
    for i in range(4):
        S.solve(A, res, None, r_tolerance=-1.0, q_tolerance=0.1)
        flags.append(S.timing()["sampled"])
    assert flags[0] == 1.0 and flags[1:] == [0.0, 0.0, 0.0]
    assert S.timing()["total_ms"] > 0.0 and S.timing()["reduced_solve_ms"] > 0.0    # the phases of solve 0 stay
    S.sample_next()
    S.solve(A, res, None, r_tolerance=-1.0, q_tolerance=0.1)
    assert S.timing()["sampled"] == 1.0 and len(S.kernel_stats()) > 0
    S.close()
    ev.close()
