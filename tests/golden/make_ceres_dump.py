#!/usr/bin/env python3
"""Writes tests/golden/ceres_dump/iteration_000_{A,b,D,x}.txt + .m: a tiny bundle-adjustment linear system in the TEXTFILE
format of Ceres' DumpLinearLeastSquaresProblem (linear_least_squares_problems.cc:966-1022; what
Solver::Options::trust_region_minimizer_iterations_to_dump produces).  The system is the first LM iteration's
(Jacobi-scaled J, residuals, LM diagonal at radius 1e4) of a 6-camera synthetic problem evaluated by the ORACLE, and x is
the oracle's DENSE_SCHUR solution of it -- standing in for the reference's x, which cannot be produced here (the reference
is unbuildable in this image).  Run from the repo root: python tests/golden/make_ceres_dump.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import cx, orc  # noqa: E402

orc.lib()
orc.set_num_threads(1)
prob = cx.bal.make_bal_like(6, 40, 170, seed=77)
bs, order = cx.bal.build_structure(prob)
_, res, _, vals = orc.bal_evaluate(bs, prob.num_cameras, prob.num_points, prob.camera_index, prob.point_index, prob.observations,
                                   order, prob.state())
scale = 1.0 / (1.0 + np.sqrt(orc.squared_column_norm(bs, vals)))
vals = orc.scale_columns(bs, vals, scale)
D = np.sqrt(np.clip(orc.squared_column_norm(bs, vals), 1e-6, 1e32) / 1e4)
x, s = orc.solve(bs, vals, res, D, orc.make_options(type=orc.DENSE_SCHUR, num_eliminate_blocks=prob.num_points))
assert s.termination_type == orc.SUCCESS
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ceres_dump"))   # Ceres writes its dumps with the base it was given
base = "ceres_solver_iteration_000"   # the name TrustRegionMinimizer gives them (trust_region_minimizer.cc:  "ceres_solver_iteration_%03d")
cx.dumps.write_dump(base, bs, vals, b=res, D=D, x=x)
print("wrote", base, bs.num_rows, "x", bs.num_cols, vals.size, "entries")
