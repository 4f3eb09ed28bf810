#!/usr/bin/env python3
"""SHA-1 of a few CGNR steps (ladybug49 preset, two synthetic problems; JACOBI and IDENTITY; fp64 and fp32 products) with
their iteration counts: the operator that writes J x and J'(J x) + D^2 x outright and the preconditioner kernel that also
sums r.z must give the bits of the separate launches (CX_CGNR_PLAIN=1).  One process per variant: the switch is read once."""
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
its = []
ctx = cx.Context(0)
for prob in (cx.bal.make_preset("ladybug49"), cx.bal.make_bal_like(7, 300, 1400, seed=3), cx.bal.make_bal_like(300, 40000, 250000, seed=6)):
    ev = cx.Evaluator(ctx, prob)
    _, res, _ = ev.evaluate(prob.state())
    A = ev.jacobian()
    D = np.sqrt(np.clip(A.squared_column_norm(), 1e-6, 1e32) / 1e4)
    for pre in (cx.JACOBI, cx.IDENTITY):
        for mixed in (0, 1):
            for eta in (0.1, 1e-3):
                S = cx.Solver(ctx, type=cx.CGNR, preconditioner_type=pre, max_num_iterations=60, use_mixed_precision_solves=mixed,
                              max_num_refinement_iterations=1 if mixed else 0)
                x, s = S.solve(A, res, D, r_tolerance=-1.0, q_tolerance=eta)
                h.update(x.tobytes())
                its.append((int(s.termination_type), int(s.num_iterations)))
                S.close()
    S = cx.Solver(ctx, type=cx.CGNR, preconditioner_type=cx.JACOBI, max_num_iterations=60)
    x, s = S.solve(A, res, None, r_tolerance=-1.0, q_tolerance=0.1)   # no LM diagonal
    h.update(x.tobytes())
    its.append((int(s.termination_type), int(s.num_iterations)))
    S.close()
    ev.close()
print(its)
print(h.hexdigest())
