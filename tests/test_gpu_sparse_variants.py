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
