"""What an LM iteration spends on producing the scaled Jacobian (Final-13682 shape): evaluation + ScaleColumns (round 2)
against the evaluation that applies the registered column scale itself (round 3).  Prints one JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
cx = conftest.cx
ctx = cx.Context(0)
prob = cx.bal.make_preset(sys.argv[1] if len(sys.argv) > 1 else "final13682")
ev = cx.Evaluator(ctx, prob)
A = ev.jacobian()
state = ctx.to_device(prob.state())
res = ctx.empty(2 * prob.num_observations)
ev.evaluate(state, residuals=res, gradient=None, want_jacobian=True)
sq = ctx.empty(A.num_cols)
A.squared_column_norm(sq)
ctx.synchronize()
scale = ctx.to_device(1.0 / (1.0 + np.sqrt(sq.to_host())))

def timed_scale():
    ctx.synchronize(); t0 = time.perf_counter(); A.scale_columns(scale); ctx.synchronize(); return (time.perf_counter() - t0) * 1e3

out = {}
for name, emit, fused in (("evaluate(no copy)+ScaleColumns(copy)", False, False), ("evaluate(copy)+ScaleColumns(copy)", True, False),
                          ("scaled evaluation, copy scattered by the kernel", True, True), ("scaled evaluation + gather pass for the copy", False, True)):
    ev.set_emit_camera_major(emit)
    ev.set_column_scale(scale if fused else None)
    ev_ms, sc_ms = [], []
    for _ in range(6):
        ev.evaluate(state, residuals=res, gradient=None, want_jacobian=True)
        ev_ms.append(ev.last_kernel_ms)
        sc_ms.append(0.0 if fused else timed_scale())
    out[name] = {"evaluate_ms": float(np.median(ev_ms[1:])), "scale_columns_ms": float(np.median(sc_ms[1:]))}
    out[name]["total_ms"] = out[name]["evaluate_ms"] + out[name]["scale_columns_ms"]
print(json.dumps(out))
