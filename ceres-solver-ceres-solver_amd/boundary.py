"""The trust-region loop of the reference reduced to the calls that cross the drop-in boundary, with HOST vectors.

This is the ctypes twin of host/test_host_adapter.cpp's RunTrustRegionLoop: the C-ABI calls below are exactly the
ones CxBalEvaluator / CxDeviceJacobian / CxLinearSolver issue when TrustRegionMinimizer drives them
(trust_region_minimizer.cc:246-313 EvaluateGradientAndJacobian, :381-463 ComputeTrustRegionStep, :720-748
ComputeCandidatePointAndEvaluateCost; levenberg_marquardt_strategy.cc:69-156 ComputeStep), every vector a
long-lived numpy array in host memory as the minimizer's Eigen vectors are.  bench.py uses it for the `boundary`
block (what one LM iteration costs THROUGH the interfaces, and what crosses PCIe for it), tests use it to check that
the transfer machinery (registered arrays, the zero-target product, direct slices on a multi-shard front) leaves
every cost bit for bit where it was.

The arithmetic between the calls (negating the step, the LM diagonal, the model-cost dot product, x + delta) is the
caller's own work in the reference (Eigen expressions in TrustRegionMinimizer); here it is numpy and is timed
separately ("caller_ms") -- it is not part of the boundary.
"""
import ctypes
import time

import numpy as np

from . import binding as B


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class BoundaryLoop:
    def __init__(self, ctx, problem, solver_kw, eta=0.1, fuse_scaling=True, alias_residuals=True, zeroed_target=True,
                 register_arrays=0):
        """fuse_scaling / alias_residuals / zeroed_target are the adapters' three opt-ins:
        CxBalEvaluator::set_fuse_jacobi_scaling, CxLinearSolver::set_alias_evaluator_residuals and
        CxDeviceJacobian::set_assume_zeroed_product_target.  register_arrays: cx_host_registration_policy's `sightings`
        for the life of this object (its vectors live as long as it does; close() releases the registrations)."""
        self.ctx, self.lib, self.problem = ctx, ctx.lib, problem
        self.register_arrays = int(register_arrays)
        B.host_registration_policy(self.register_arrays, 4096, 16 << 30)
        self.ev = B.Evaluator(ctx, problem)
        self.J = self.ev.jacobian()
        self.S = B.Solver(ctx, **solver_kw)
        self.eta = eta
        self.fuse_scaling, self.alias_residuals, self.zeroed_target = fuse_scaling, alias_residuals, zeroed_target
        n, m = self.ev.num_effective_parameters, self.ev.num_rows
        self.n, self.m = n, m
        # TrustRegionMinimizer::Init (trust_region_minimizer.cc:181-203) and LevenbergMarquardtStrategy (:77-99)
        self.x = np.array(problem.state(), dtype=np.float64)
        self.candidate = np.zeros_like(self.x)
        self.residuals = np.zeros(m)
        self.model_residuals = np.zeros(m)
        self.gradient = np.zeros(n)
        self.scaling = np.zeros(n)
        self.diagonal = np.zeros(n)
        self.lm_diagonal = np.zeros(n)
        self.step = np.zeros(n)
        self.delta = np.zeros(n)
        self.tmp_rows = np.zeros(m)
        self.cost = 0.0
        self.radius, self.decrease_factor = 1e4, 2.0  # solver.h:270-290
        self.reuse_diagonal = False
        self.scale_registered = False
        self.values_carry_scale = False
        self.t = {}
        self.costs, self.linear_iterations = [], []
        if fuse_scaling:
            self.ev.set_emit_camera_major(False)

    def close(self):
        self.S.close()
        self.ev.close()
        B.host_registrations_release()   # before the vectors can be freed
        B.host_registration_policy(0)

    # ---- timing
    def _timed(self, key, fn):
        t0 = time.perf_counter()
        r = fn()
        self.t[key] = self.t.get(key, 0.0) + (time.perf_counter() - t0) * 1e3
        return r

    # ---- the boundary calls
    def _evaluate(self, state, with_jacobian):
        cost = ctypes.c_double()
        if with_jacobian:
            rc = self.lib.cx_evaluator_evaluate(self.ev._h, _p(state), ctypes.byref(cost), _p(self.residuals), _p(self.gradient), 1, B.HOST)
        else:
            rc = self.lib.cx_evaluator_evaluate(self.ev._h, _p(state), ctypes.byref(cost), None, None, 0, B.HOST)
        B._check(rc)
        return cost.value

    def _scale_columns(self):
        # CxDeviceJacobian::ScaleColumns (host/cx_device_jacobian.h)
        if self.fuse_scaling:
            if self.scale_registered and self.values_carry_scale:
                self.values_carry_scale = False
                return
            if not self.scale_registered:
                B._check(self.lib.cx_matrix_scale_columns(self.J._h, _p(self.scaling), B.HOST))
                B._check(self.lib.cx_evaluator_set_column_scale(self.ev._h, _p(self.scaling), B.HOST))
                self.scale_registered = True
                return
        B._check(self.lib.cx_matrix_scale_columns(self.J._h, _p(self.scaling), B.HOST))

    def _evaluate_gradient_and_jacobian(self, first):
        self.cost = self._timed("evaluate_jacobian_ms", lambda: self._evaluate(self.x, True))
        self.values_carry_scale = self.fuse_scaling and self.scale_registered
        if first:
            self._timed("squared_column_norm_ms", lambda: B._check(self.lib.cx_matrix_squared_column_norm(self.J._h, _p(self.scaling), B.HOST)))
            t0 = time.perf_counter()
            np.sqrt(self.scaling, out=self.scaling)
            self.scaling += 1.0
            np.reciprocal(self.scaling, out=self.scaling)
            self.t["caller_ms"] = self.t.get("caller_ms", 0.0) + (time.perf_counter() - t0) * 1e3
        self._timed("scale_columns_ms", self._scale_columns)

    def start(self):
        self._evaluate_gradient_and_jacobian(True)
        self.costs.append(self.cost)

    def iterate(self):
        """One iteration of the loop (successful or not).  Returns True when the step was accepted."""
        lib = self.lib
        caller0 = time.perf_counter()
        caller = 0.0

        def boundary(key, fn):
            nonlocal caller, caller0
            caller += time.perf_counter() - caller0
            r = self._timed(key, fn)
            caller0 = time.perf_counter()
            return r

        # LevenbergMarquardtStrategy::ComputeStep
        if not self.reuse_diagonal:
            boundary("squared_column_norm_ms", lambda: B._check(lib.cx_matrix_squared_column_norm(self.J._h, _p(self.diagonal), B.HOST)))
            np.clip(self.diagonal, 1e-6, 1e32, out=self.diagonal)
        np.divide(self.diagonal, self.radius, out=self.lm_diagonal)
        np.sqrt(self.lm_diagonal, out=self.lm_diagonal)
        self.step.fill(np.nan)  # InvalidateArray
        ps = B.cx_per_solve_options()
        ps.D = _p(self.lm_diagonal).value
        ps.r_tolerance, ps.q_tolerance, ps.memspace = -1.0, self.eta, B.HOST
        b = _p(self.residuals)
        if self.alias_residuals:
            token = lib.cx_evaluator_device_residuals(self.ev._h)
            if token:
                b = ctypes.c_void_p(token)
                ps.b_on_device = 1
        s = B.cx_summary()
        boundary("solve_ms", lambda: B._check(lib.cx_solver_solve(self.S._h, self.J._h, b, ctypes.byref(ps), _p(self.step), ctypes.byref(s))))
        accepted = False
        if s.termination_type not in (B.FAILURE, B.FATAL_ERROR) and np.isfinite(self.step).all():
            np.negative(self.step, out=self.step)
            self.reuse_diagonal = True
            self.linear_iterations.append(int(s.num_iterations))
            # ComputeTrustRegionStep: model_residuals = J step
            self.model_residuals.fill(0.0)
            if self.zeroed_target:
                boundary("model_cost_product_ms", lambda: B._check(lib.cx_matrix_right_multiply_overwrite(self.J._h, _p(self.step), _p(self.model_residuals), B.HOST)))
            else:
                boundary("model_cost_product_ms", lambda: B._check(lib.cx_matrix_right_multiply(self.J._h, _p(self.step), _p(self.model_residuals), B.HOST)))
            np.multiply(self.model_residuals, 0.5, out=self.tmp_rows)
            self.tmp_rows += self.residuals
            # (no BLAS here on purpose: the worker threads a threaded ddot leaves spinning steal the cores the HIP runtime's
            # copy threads need, and the evaluation that follows was measured anywhere between 14 and 87 ms because of it)
            np.multiply(self.tmp_rows, self.model_residuals, out=self.tmp_rows)
            model_cost_change = -float(self.tmp_rows.sum())
            if model_cost_change > 0.0:
                np.multiply(self.step, self.scaling, out=self.delta)
                np.add(self.x, self.delta, out=self.candidate)  # Evaluator::Plus, Euclidean blocks (host, as CxBalEvaluator::Plus)
                candidate_cost = boundary("evaluate_cost_ms", lambda: self._evaluate(self.candidate, False))
                relative_decrease = (self.cost - candidate_cost) / model_cost_change
                if relative_decrease > 1e-3:
                    self.x[:] = self.candidate
                    self.radius = min(1e16, self.radius / max(1.0 / 3.0, 1.0 - (2.0 * relative_decrease - 1.0) ** 3))
                    self.decrease_factor = 2.0
                    self.reuse_diagonal = False
                    caller += time.perf_counter() - caller0
                    self._evaluate_gradient_and_jacobian(False)
                    caller0 = time.perf_counter()
                    self.costs.append(self.cost)
                    accepted = True
        if not accepted:
            self.radius /= self.decrease_factor
            self.decrease_factor *= 2.0
            self.reuse_diagonal = True
        caller += time.perf_counter() - caller0
        self.t["caller_ms"] = self.t.get("caller_ms", 0.0) + caller * 1e3
        return accepted

    def reset(self):
        """Back to the start point with every array, registration and device-side cache kept: what a second
        Solver::Solve on the same problem object would find."""
        self.x[:] = self.problem.state()
        self.radius, self.decrease_factor = 1e4, 2.0
        self.reuse_diagonal = False
        self.costs, self.linear_iterations = [], []
        if self.fuse_scaling and self.scale_registered:
            # iteration 0 of the reference computes the scaling from the unscaled J: the evaluator goes back to plain values
            B._check(self.lib.cx_evaluator_set_column_scale(self.ev._h, None, B.HOST))
            self.scale_registered = False
        self.values_carry_scale = False

    def run(self, iterations, warm_pass=True):
        """Runs `iterations` iterations from the start point and reports each of them: wall ms inside the boundary
        calls (by call), the caller's own ms between them, bytes across PCIe.  warm_pass: the same iterations are run
        once before, untimed (first-sight registrations, structure caches, solver set-up), then the loop is reset."""
        if warm_pass:
            self.start()
            for _ in range(iterations):
                self.iterate()
            self.reset()
        self.ctx.transfer_stats(reset=True)
        self.t = {}
        self.start()
        start_ms = dict(self.t)
        per_iteration = []
        for _ in range(iterations):
            self.t = {}
            self.ctx.transfer_stats(reset=True)
            t0 = time.perf_counter()
            accepted = self.iterate()
            wall = (time.perf_counter() - t0) * 1e3
            stats = self.ctx.transfer_stats(reset=False)
            calls = {key: v for key, v in self.t.items() if key != "caller_ms"}
            per_iteration.append({
                "through_interfaces_ms": sum(calls.values()), "calls_ms": calls, "caller_ms": self.t.get("caller_ms", 0.0),
                "wall_ms": wall, "accepted": bool(accepted),
                "cg_iterations": self.linear_iterations[-1] if self.linear_iterations else 0,
                "h2d_bytes": stats["h2d_bytes"], "d2h_bytes": stats["d2h_bytes"], "h2d_ms": stats["h2d_ms"], "d2h_ms": stats["d2h_ms"],
                "registered_bytes_moved": stats["h2d_registered_bytes"] + stats["d2h_registered_bytes"],
            })
        stats = self.ctx.transfer_stats(reset=False)
        k = float(max(1, iterations))
        mean = lambda key: sum(r[key] for r in per_iteration) / k  # noqa: E731
        moved = sum(r["h2d_bytes"] + r["d2h_bytes"] for r in per_iteration)
        return {
            "iterations": iterations, "accepted": sum(r["accepted"] for r in per_iteration),
            "lm_iteration_through_interfaces_ms": mean("through_interfaces_ms"),
            "calls_ms": {key: sum(r["calls_ms"].get(key, 0.0) for r in per_iteration) / k
                         for key in sorted({c for r in per_iteration for c in r["calls_ms"]})},
            "caller_ms": mean("caller_ms"), "wall_ms_per_iteration": mean("wall_ms"),
            "h2d_bytes": mean("h2d_bytes"), "d2h_bytes": mean("d2h_bytes"),
            "h2d_ms": mean("h2d_ms"), "d2h_ms": mean("d2h_ms"), "transfer_ms": mean("h2d_ms") + mean("d2h_ms"),
            "h2d_GBps": sum(r["h2d_bytes"] for r in per_iteration) / max(sum(r["h2d_ms"] for r in per_iteration), 1e-9) / 1e6,
            "d2h_GBps": sum(r["d2h_bytes"] for r in per_iteration) / max(sum(r["d2h_ms"] for r in per_iteration), 1e-9) / 1e6,
            "registered_fraction": sum(r["registered_bytes_moved"] for r in per_iteration) / max(1.0, float(moved)),
            "registered_arrays": stats["num_registered"], "registered_bytes": stats["registered_bytes"],
            "iteration_zero_ms": start_ms,
            "per_iteration": per_iteration,
            "costs": list(self.costs), "linear_iterations": list(self.linear_iterations),
        }
