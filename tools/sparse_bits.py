"""SHA-1 of the SPARSE_SCHUR step of a 2 000-camera problem (tile-sparse factorisation) -- run under different kernel switches
(CX_SPARSE_F64_LDS, CX_SPARSE_F32_LDS, ...) to check the 'bitwise the same result' statements of cx_sparse_chol.hip:
  for v in 0 1 324; do CX_SPARSE_F64_LDS=$v python tools/sparse_bits.py; done;  for v in 0 1 644; do CX_SPARSE_F32_LDS=$v python tools/sparse_bits.py --mixed; done"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import conftest, numpy as np
cx = conftest.cx
ctx = cx.Context(0)
prob = cx.bal.make_bal_like(2000, 20000, 110000, seed=3)
bs, _ = cx.bal.build_structure(prob)
O, P = prob.num_observations, prob.num_points
rng = np.random.default_rng(5)
vals = cx.bal.random_jacobian_values(O, 4)
b = rng.standard_normal(2 * O)
D = rng.uniform(0.5, 2.0, bs.num_cols)
A = cx.Matrix(ctx, bs, P)
A.set_values(vals)
S = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=P, use_mixed_precision_solves=1 if "--mixed" in sys.argv else 0)
x, s = S.solve(A, b, D)
print({k: v for k, v in os.environ.items() if k.startswith("CX_SPARSE")}, "--mixed" in sys.argv, s.termination_type, hashlib.sha1(x.tobytes()).hexdigest())
