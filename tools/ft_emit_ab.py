#!/usr/bin/env python3
"""A/B of the kernels that keep the camera-major copy of F current (k_scale_239, k_bal_evaluate) on the
Final-13682 shape.  Variants are chosen by environment variables read once per process, so run it once per
variant:  CX_NO_FT_EMIT=1 (separate k_permute_ft), CX_NO_XCD_TILES=1 (plain blockIdx -> tile map)."""
import importlib.util, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (shares its HIP runtime)
import bench
cx = bench.load_cx()
ctx = cx.Context(0)
prob = cx.bal.make_preset(sys.argv[1] if len(sys.argv) > 1 else "final13682")
ev = cx.Evaluator(ctx, prob)
A = ev.jacobian()
state = ctx.to_device(prob.state())
res = ctx.empty(2 * prob.num_observations)
scale = ctx.to_device(np.full(A.num_cols, 1.0))
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("CX_")}}
for emit in (1, 0):
    ev.set_emit_camera_major(bool(emit))
    t = []
    for _ in range(6):
        ev.evaluate(state, residuals=res, gradient=None, want_jacobian=True)
        t.append(ev.last_kernel_ms)
    out["evaluate_emit%d_ms" % emit] = float(np.median(t[1:]))
t = []
for _ in range(6):
    ctx.synchronize(); t0 = time.perf_counter(); A.scale_columns(scale); ctx.synchronize()
    t.append((time.perf_counter() - t0) * 1e3)
out["scale_columns_ms"] = float(np.median(t[1:]))
# the cost of finding the copy stale: one product with F' after invalidating it
x = ctx.to_device(np.ones(A.num_rows)); y = ctx.zeros(A.num_cols)
t = []
for _ in range(4):
    A.values_changed(); ctx.synchronize(); t0 = time.perf_counter(); A.left_multiply(x, y); ctx.synchronize()
    t.append((time.perf_counter() - t0) * 1e3)
out["left_multiply_with_stale_copy_ms"] = float(np.median(t[1:]))
t = []
for _ in range(4):
    ctx.synchronize(); t0 = time.perf_counter(); A.left_multiply(x, y); ctx.synchronize()
    t.append((time.perf_counter() - t0) * 1e3)
out["left_multiply_ms"] = float(np.median(t[1:]))
print(json.dumps(out))
