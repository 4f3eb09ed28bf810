#!/usr/bin/env python3
"""Where k_bal_evaluate's time goes on the Final-13682 shape: the Jacobian variant (Jet arithmetic, 2 330 vector
instructions, 208 B written per residual block), the value-only variant with and without its 16-byte residual
store (1 351 vector instructions), per process one setting of CX_EVAL_WG_PER_CU (unset = 138 VGPRs, 3 workgroups
per CU; 4 = 128 VGPRs with 10 spilled)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (shares its HIP runtime)
import bench
cx = bench.load_cx()
ctx = cx.Context(0)
prob = cx.bal.make_preset(sys.argv[1] if len(sys.argv) > 1 else "final13682")
ev = cx.Evaluator(ctx, prob)
state = ctx.to_device(prob.state())
res = ctx.empty(2 * prob.num_observations)
ev.set_emit_camera_major(False)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("CX_")}, "residual_blocks": int(prob.num_observations)}


def timed(**kw):
    t = []
    for _ in range(6):
        ev.evaluate(state, **kw)
        t.append(ev.last_kernel_ms)
    return float(np.median(t[1:]))


out["jacobian_ms"] = timed(residuals=res, gradient=None, want_jacobian=True)
ev.set_emit_camera_major(True)
out["jacobian_and_camera_major_copy_ms"] = timed(residuals=res, gradient=None, want_jacobian=True)
ev.set_emit_camera_major(False)
out["value_and_residuals_ms"] = timed(residuals=res, gradient=None, want_jacobian=False)
out["value_only_ms"] = timed(residuals=None, gradient=None, want_jacobian=False)
# ScaleColumns (k_scale_239), which also writes the camera-major copy
import time
A = ev.jacobian()
scale = ctx.to_device(np.full(A.num_cols, 1.0))
t = []
for _ in range(6):
    ctx.synchronize(); t0 = time.perf_counter(); A.scale_columns(scale); ctx.synchronize()
    t.append((time.perf_counter() - t0) * 1e3)
out["scale_columns_ms"] = float(np.median(t[1:]))
print(json.dumps(out))
