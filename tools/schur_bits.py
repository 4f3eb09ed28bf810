#!/usr/bin/env python3
"""DENSE_SCHUR, SPARSE_SCHUR (tile-sparse factorisation forced) and explicit-S ITERATIVE_SCHUR steps of two problems, as a
SHA-1 and (--dump file.npz) as arrays: the explicit assembly that delivers the reduced right-hand side from its own two set-up
passes against the separate passes of round 3 (CX_ELIMINATE_RHS_SEPARATE=1) -- equal to rounding, not to the bit: t' comes out
of k_chunk_init with E'b summed in two interleaved halves, out of k_chunk_pass<1> in row order.  One process per variant: the
switch is read once."""
import hashlib
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = os.path.join(ROOT, "ceres-solver-ceres-solver_amd")
spec = importlib.util.spec_from_file_location("cxschur", os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
cx = importlib.util.module_from_spec(spec)
sys.modules["cxschur"] = cx
spec.loader.exec_module(cx)

h = hashlib.sha1()
out = []
steps = []
ctx = cx.Context(0)
for prob in (cx.bal.make_preset("ladybug49"), cx.bal.make_bal_like(300, 40000, 250000, seed=6)):
    ev = cx.Evaluator(ctx, prob)
    _, res, _ = ev.evaluate(prob.state())
    A = ev.jacobian()
    D = np.sqrt(np.clip(A.squared_column_norm(), 1e-6, 1e32) / 1e4)
    for kw in (dict(type=cx.DENSE_SCHUR), dict(type=cx.SPARSE_SCHUR), dict(type=cx.SPARSE_SCHUR, use_mixed_precision_solves=1),
               dict(type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.SCHUR_JACOBI, use_explicit_schur_complement=1, max_num_iterations=50)):
        S = cx.Solver(ctx, num_eliminate_blocks=prob.num_points, **kw)
        for Dv in (D, None):
            x, s = S.solve(A, res, Dv, r_tolerance=-1.0, q_tolerance=1e-3)
            h.update(x.tobytes())
            steps.append(x.copy())
            out.append((int(s.termination_type), int(s.num_iterations)))
        S.close()
    ev.close()
if "--dump" in sys.argv:
    np.savez(sys.argv[sys.argv.index("--dump") + 1], *steps)
print(out)
print(h.hexdigest())
