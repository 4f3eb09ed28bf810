"""Host-side sanitizer runs of the threaded structure planners (tools/sanitize): the host-only entry points of the
library -- pair lists of S on 16 threads, nested dissection + symbolic fill + level schedule, visibility clustering,
stable Schur ordering, point partition -- built with AddressSanitizer + UBSan and with ThreadSanitizer (host side only;
the reference wires the same sanitizers into its build, CMakeLists.txt:118-120, cmake/EnableSanitizer.cmake:38-98) and run
on the CPU on a ring scene, a ring with 5 % long-range observations, and a tiny scene.  Never on the GPU box: GPU
sanitizer runs are not available on the pool."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SAN = os.path.join(ROOT, "tools", "sanitize")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm clang with its sanitizer runtimes")
def test_host_planners_are_clean_under_asan_ubsan_and_tsan():
    out = subprocess.run(["make", "-s", "-j8", "-C", SAN, "check"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    for name in ("asan", "tsan"):
        log = open(os.path.join(SAN, "out", name + ".log")).read()
        assert "ALL OK" in log and log.rstrip().endswith("exit 0"), log[-3000:]
        for bad in ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "FAILED"):
            assert bad not in log, log[-3000:]
        shutil.copy(os.path.join(SAN, "out", name + ".log"), os.path.join(ROOT, "profiles", "r04_sanitizer_%s.log" % name))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm clang with its sanitizer runtimes")
def test_shard_group_threading_is_clean_under_asan_ubsan_and_tsan():
    """the multi-shard front's threading (csrc/cx_shard_group.h -- job dispatch, the rendezvous of the
    in-process exchange step, the abort path) driven with dummy jobs: clean runs, one shard failing before / between /
    after the exchange steps, a failing combine step, a length mismatch, a late failure while the others wait, runs after
    failed runs, destruction while idle.  Every scenario must return, naming the shard that failed first."""
    out = subprocess.run(["make", "-s", "-C", SAN, "check-shard-group"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    for name in ("asan", "tsan"):
        log = open(os.path.join(SAN, "out", "shard_group_%s.log" % name)).read()
        assert "ALL OK" in log and log.rstrip().endswith("exit 0"), log[-3000:]
        for bad in ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "FAILED"):
            assert bad not in log, log[-3000:]
        shutil.copy(os.path.join(SAN, "out", "shard_group_%s.log" % name), os.path.join(ROOT, "profiles", "r04_sanitizer_shard_group_%s.log" % name))
