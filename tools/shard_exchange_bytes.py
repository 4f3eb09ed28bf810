import sys, os, importlib.util
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
pkg = os.path.join(ROOT, "ceres-solver-ceres-solver_amd")
spec = importlib.util.spec_from_file_location("cxschur", os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
cx = importlib.util.module_from_spec(spec); sys.modules["cxschur"] = cx; spec.loader.exec_module(cx)
prob = cx.bal.make_preset("final13682")
for shards in (2, 4, 8):
    ctx = cx.Context(devices=[0] * shards)
    ev = cx.Evaluator(ctx, prob)
    _, res, _ = ev.evaluate(prob.state(), want_gradient=False)
    A = ev.jacobian()
    D = np.sqrt(np.clip(A.squared_column_norm(), 1e-6, 1e32) / 1e4)
    S = cx.Solver(ctx, type=cx.SPARSE_SCHUR, num_eliminate_blocks=prob.num_points)
    x, s = S.solve(A, res, D)
    x2, s2 = S.solve(A, res, D)
    tm = S.timing()
    print(shards, s2.termination_type, "calls", tm["allreduce_calls"], "GB", tm["allreduce_bytes"] / 1e9, "total_ms", tm["total_ms"], bool(np.array_equal(x, x2)), flush=True)
    S.close(); ev.close(); ctx.close()
