"""Build csrc/libcxschur.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libcxschur.so")
SOURCES = ["cx_context.cpp", "cx_transfer.cpp", "cx_ordering.cpp", "cx_matrix.hip", "cx_schur.hip", "cx_solver.hip", "cx_cholesky.hip",
           "cx_eval.hip", "cx_generic.hip", "cx_minimizer.hip", "cx_sparse_chol.hip", "cx_visibility.cpp", "cx_band_chol.hip", "cx_multi.hip", "cx_embed.hip"]
HEADERS = ["cx_internal.h", "cx_kernels.h", "cx_schur.h", "cx_solver_internal.h", "cx_chol_blocks.h", "cx_visibility.h", os.path.join("..", "..", "include", "cxschur.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"] + \
    os.environ.get("CX_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DCX_POTRF_SCALAR for an A/B build


# cx_eval.hip: the persistent evaluator loop must not have the literal constants of its sin / cos / exp expansions
# hoisted into (and spilled from) vector registers
PER_FILE_FLAGS = {"cx_eval.hip": ["-mllvm", "-disable-machine-licm"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every translation unit to an object and link the shared library."""
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(src, []) + ["-x", "hip", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("==== %s\n%s\n" % (src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-ldl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
