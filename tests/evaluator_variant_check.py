"""Helper of test_gpu_parity.py::test_evaluator_variants (run as a child process: the library reads its A/B switches
CX_EVAL_VARIANT / CX_EVAL_PERSISTENT once per process).  Evaluates a problem with more tiles than the chip holds
workgroups -- so that the persistent loop of k_bal_evaluate takes several tiles per workgroup, with and without a loss,
with and without the camera-major copy -- and compares residuals, cost, gradient and Jacobian with the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: E402  (loads torch first)

cx = conftest._load_package()
orc = conftest._load_oracle()


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, float(np.max(np.abs(b)))))


def main():
    C, P, O = 60, 40000, 330000      # 1290 tiles of 256 rows: more than 3 x 256 resident workgroups
    prob = cx.bal.make_bal_like(C, P, O, 11)
    bs, order = cx.bal.build_structure(prob)
    ctx = cx.Context(0)
    ev = cx.Evaluator(ctx, prob)
    state = prob.state()
    for loss in (None, (cx.binding.LOSS_HUBER, 1.0, 0.0)):
        if loss:
            ev.set_loss(*loss)
        ref = orc.bal_evaluate(bs, C, P, prob.camera_index, prob.point_index, prob.observations, order, state,
                               **({"loss": loss} if loss else {}))
        cost_r, res_r, grad_r, vals_r = ref
        for emit in (True, False):
            ev.set_emit_camera_major(emit)
            cost, res, grad = ev.evaluate(state)
            vals = ev.jacobian(bs).get_values()
            assert relerr(res, res_r) < 1e-11 and relerr(vals, vals_r) < 1e-11, (loss, emit)
            assert abs(cost - cost_r) <= 1e-11 * abs(cost_r) and relerr(grad, grad_r) < 1e-10, (loss, emit)
            if emit:   # the camera-major copy the kernel wrote gives the same J' x as the row-major cells
                A = ev.jacobian()
                x = np.random.default_rng(3).standard_normal(A.num_rows)
                y1 = A.left_multiply(x)
                A.values_changed()
                y2 = A.left_multiply(x)
                assert np.array_equal(y1, y2), (loss, "camera-major copy")
        cost2, res2, _ = ev.evaluate(state, want_gradient=False, want_jacobian=False)
        assert relerr(res2, res_r) < 1e-11 and abs(cost2 - cost_r) <= 1e-11 * abs(cost_r)
    ev.close()
    print("EVALUATOR_VARIANT_OK")


if __name__ == "__main__":
    main()
