"""Products and a 20-iteration ITERATIVE_SCHUR solve of a <2,3,6> problem (400 cameras, 500 k observations) through the embedded static
image, beside the native <2,3,9> problem on the same visibility; CX_NO_EMBEDDING=1: the dynamic-size path."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import conftest, numpy as np
cx=conftest.cx
ctx=cx.Context(0)
prob = cx.bal.make_bal_like(400, 80000, 500000, seed=3)
bs9, _ = cx.bal.build_structure(prob)
O, P, C = prob.num_observations, prob.num_points, prob.num_cameras
rows = [(2, [(int(bs9.cells["block_id"][2 * r]), 6 * r), (int(bs9.cells["block_id"][2 * r + 1]), 6 * O + 12 * r)]) for r in range(O)]
bs6 = cx.BlockStructure.from_rows([3] * P + [6] * C, rows)
rng = np.random.default_rng(1)
for name, bs, nnz in (("239", bs9, 24 * O), ("236", bs6, 18 * O)):
    A = cx.Matrix(ctx, bs, P); A.set_values(rng.standard_normal(nnz))
    x = rng.standard_normal(bs.num_cols); y = rng.standard_normal(bs.num_rows)
    r=[];l=[]
    for _ in range(6):
        A.right_multiply(x); r.append(A.last_kernel_ms); A.left_multiply(y); l.append(A.last_kernel_ms)
    b = rng.standard_normal(bs.num_rows); D = np.full(bs.num_cols, 0.1)
    S = cx.Solver(ctx, type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=P, max_num_iterations=20, min_num_iterations=20)
    S.solve(A,b,D); S.solve(A,b,D); tm=S.timing()
    print(name, "static_path", A.static_path, "Jx %.4f ms  J'y %.4f ms  solve(20 CG) device %.3f ms  cg/it %.4f" % (np.median(r[1:]), np.median(l[1:]), tm["total_ms"], tm["reduced_solve_ms"]/20))
    S.close(); A.close()
