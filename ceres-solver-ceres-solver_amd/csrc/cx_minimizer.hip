// Device-resident trust-region loop for the bundle-adjustment evaluator:
// TrustRegionMinimizer::Minimize (trust_region_minimizer.cc:68-840),
// LevenbergMarquardtStrategy (levenberg_marquardt_strategy.cc:50-175, .h:63-69) and
// TrustRegionStepEvaluator (trust_region_step_evaluator.cc:40-117).
//
// The reference keeps x, residuals, gradient, step, the LM diagonal and the Jacobi scaling in
// host Eigen vectors and J in host memory; here they all live in HBM next to J and only the
// scalars the control flow branches on (costs, the model cost change, norms) cross to the
// host, a few doubles per iteration.  The control flow itself is host code, as in the reference.
#include <chrono>
#include <limits>

#include "cx_internal.h"
#include "cx_kernels.h"
#include "cx_solver_internal.h"

namespace {

constexpr int kMinRedBlocks = 512;

enum RedMode {
  RED_GRADIENT = 0,    // a = x, b = Plus(x, -gradient): v0 = sum d^2, v1 = max |d|, d = a - b
  RED_MODEL_COST = 1,  // a = model residuals, b = residuals: v0 = sum m (r + m / 2)
  RED_STEP = 2         // a = x, b = candidate: v0 = sum x^2, v1 = sum (x - c)^2
};

template <int MODE>
__device__ __forceinline__ void red_term(double a, double b, double& v0, double& v1) {
  if (MODE == RED_GRADIENT) {
    const double d = a - b;  // x - Plus(x, -gradient), trust_region_minimizer.cc:283-298
    v0 += d * d;
    v1 = fmax(v1, fabs(d));
  } else if (MODE == RED_MODEL_COST) {
    v0 += a * (b + a / 2.0);            // trust_region_minimizer.cc:430-434
  } else {
    const double d = a - b;             // trust_region_minimizer.cc:700-705
    v0 += a * a;
    v1 += d * d;
  }
}

__device__ __forceinline__ double block_max(double v, double* scratch) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmax(fmax(scratch[0], scratch[1]), fmax(scratch[2], scratch[3]));
  __syncthreads();
  return v;
}

// grid-stride partial reduction; partial[2 * block + {0, 1}]
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_reduce2(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                    double* __restrict__ partial) {
  __shared__ double red[8];
  double v0 = 0.0, v1 = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock)
    red_term<MODE>(a[i], b[i], v0, v1);
  double s[1] = {v0};
  block_sum<1>(s, red);
  double t;
  if (MODE == RED_GRADIENT) {
    t = block_max(v1, red);
  } else {
    double u[1] = {v1};
    block_sum<1>(u, red);
    t = u[0];
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = s[0];
    partial[2 * blockIdx.x + 1] = t;
  }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_reduce2_final(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
  __shared__ double red[8];
  double v0 = 0.0, v1 = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += kBlock) {
    v0 += partial[2 * i];
    if (MODE == RED_GRADIENT) v1 = fmax(v1, partial[2 * i + 1]);
    else v1 += partial[2 * i + 1];
  }
  double s[1] = {v0};
  block_sum<1>(s, red);
  double t;
  if (MODE == RED_GRADIENT) {
    t = block_max(v1, red);
  } else {
    double u[1] = {v1};
    block_sum<1>(u, red);
    t = u[0];
  }
  if (threadIdx.x == 0) {
    out[0] = s[0];
    out[1] = t;
  }
}

// sharded exchange buffer: buf[0] = local sum, buf[1 + r] = rank r's second value
__global__ void k_pack_slots(const double* __restrict__ local, int rank, int nranks, double* __restrict__ buf) {
  const int t = threadIdx.x;
  if (t == 0) buf[0] = local[0];
  if (t < nranks) buf[1 + t] = (t == rank) ? local[1] : 0.0;
}

// jacobian_scaling = 1 / (1 + sqrt(diag J'J))  (trust_region_minimizer.cc:252-261)
__global__ __launch_bounds__(kBlock) void k_jacobi_scaling(double* __restrict__ s, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) s[i] = 1.0 / (1.0 + sqrt(s[i]));
}

// diagonal = clamp(diag J'J), lm_diagonal = sqrt(diagonal / radius)  (levenberg_marquardt_strategy.cc:81-98)
__global__ __launch_bounds__(kBlock) void k_lm_diagonal(double* __restrict__ diagonal, double* __restrict__ lm, int64_t n,
                                                        double min_diagonal, double max_diagonal, double radius, int clamp) {
  const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  double d = diagonal[i];
  if (clamp) {
    d = fmin(fmax(d, min_diagonal), max_diagonal);
    diagonal[i] = d;
  }
  lm[i] = sqrt(d / radius);
}

// step = -step (levenberg_marquardt_strategy.cc:126-128); delta = step .* scaling (trust_region_minimizer.cc:443-448)
__global__ __launch_bounds__(kBlock) void k_negate_and_unscale(double* __restrict__ step, const double* __restrict__ scaling,
                                                               double* __restrict__ delta, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const double s = -step[i];
  step[i] = s;
  delta[i] = s * scaling[i];
}

__global__ __launch_bounds__(kBlock) void k_fill(double* __restrict__ p, int64_t n, double v) {
  const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) p[i] = v;
}

inline int grid_for(int64_t n) { return int((n + kBlock - 1) / kBlock); }

using Clock = std::chrono::steady_clock;
double MsSince(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

// levenberg_marquardt_strategy.cc:50-175 (host scalars only)
struct LmStrategy {
  double radius, max_radius, min_diagonal, max_diagonal;
  double decrease_factor = 2.0;
  bool reuse_diagonal = false;
  void StepAccepted(double step_quality) {
    radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * step_quality - 1.0, 3));
    radius = std::min(max_radius, radius);
    decrease_factor = 2.0;
    reuse_diagonal = false;
  }
  void StepRejected() {
    radius = radius / decrease_factor;
    decrease_factor *= 2.0;
    reuse_diagonal = true;
  }
};

// trust_region_step_evaluator.cc:40-117
struct StepEvaluator {
  int max_consecutive_nonmonotonic_steps;
  double minimum_cost, current_cost, reference_cost, candidate_cost;
  double accumulated_reference_model_cost_change = 0.0, accumulated_candidate_model_cost_change = 0.0;
  int num_consecutive_nonmonotonic_steps = 0;
  StepEvaluator(double initial_cost, int max_nonmonotonic)
      : max_consecutive_nonmonotonic_steps(max_nonmonotonic), minimum_cost(initial_cost), current_cost(initial_cost),
        reference_cost(initial_cost), candidate_cost(initial_cost) {}
  double StepQuality(double cost, double model_cost_change) const {
    if (cost >= std::numeric_limits<double>::max()) return std::numeric_limits<double>::lowest();
    const double relative_decrease = (current_cost - cost) / model_cost_change;
    const double historical_relative_decrease =
        (reference_cost - cost) / (accumulated_reference_model_cost_change + model_cost_change);
    return std::max(relative_decrease, historical_relative_decrease);
  }
  void StepAccepted(double cost, double model_cost_change) {
    current_cost = cost;
    accumulated_candidate_model_cost_change += model_cost_change;
    accumulated_reference_model_cost_change += model_cost_change;
    if (current_cost < minimum_cost) {
      minimum_cost = current_cost;
      num_consecutive_nonmonotonic_steps = 0;
      candidate_cost = current_cost;
      accumulated_candidate_model_cost_change = 0.0;
    } else {
      ++num_consecutive_nonmonotonic_steps;
      if (current_cost > candidate_cost) {
        candidate_cost = current_cost;
        accumulated_candidate_model_cost_change = 0.0;
      }
    }
    if (num_consecutive_nonmonotonic_steps == max_consecutive_nonmonotonic_steps) {
      reference_cost = candidate_cost;
      accumulated_reference_model_cost_change = accumulated_candidate_model_cost_change;
    }
  }
};

struct Minimizer {
  cx_evaluator* e;
  cx_solver* solver;
  cx_context* ctx;
  cx_matrix* J;
  cx_minimizer_options o;
  cx_minimizer_summary* out;
  cx_iteration_summary* iterations;
  int capacity;
  // n: tangent size (gradient, step, J columns); n_amb: ambient size of the state (> n for cameras on a
  // manifold); n_local: leading entries owned by this rank (the points); the rest is replicated
  int64_t n, n_amb, m, n_local;
  DevBuf<double> x, candidate_x, residuals, gradient, step, delta, scaling, diagonal, lm_diagonal, model_residuals;
  DevBuf<double> partial, red_out, slots;
  double* parameters = nullptr;  // device: the minimum-cost iterate
  double x_cost = std::numeric_limits<double>::max(), minimum_cost = x_cost, candidate_cost = 0.0;
  double model_cost_change = 0.0;
  int num_consecutive_invalid_steps = 0;
  cx_iteration_summary it{}, last{};
  int num_written = 0;

  // v0 always a sum; v1 a max (RED_GRADIENT) or a sum.  Entries [0, n_local) of the vectors are
  // this rank's own, [n_local, len) are replicated on every rank.
  template <int MODE>
  int Reduce(const double* a, const double* b, int64_t len, int64_t len_local, double& v0, double& v1) {
    hipStream_t st = ctx->stream;
    auto pass = [&](const double* pa, const double* pb, int64_t cnt, double* dst) -> int {
      const int blocks = int(std::max<int64_t>(1, std::min<int64_t>(kMinRedBlocks, (cnt + kBlock - 1) / kBlock)));
      hipLaunchKernelGGL(k_reduce2<MODE>, dim3(blocks), dim3(kBlock), 0, st, pa, pb, cnt, partial.p);
      hipLaunchKernelGGL(k_reduce2_final<MODE>, dim3(1), dim3(kBlock), 0, st, (const double*)partial.p, blocks, dst);
      CX_HIP(hipGetLastError());
      return CX_OK;
    };
    double h[4] = {0, 0, 0, 0};
    if (ctx->nranks <= 1) {
      CX_TRY(pass(a, b, len, red_out.p));
      CX_TRY(cx_read_back(ctx, h, red_out.p, 2 * sizeof(double), st));
      CX_TRY(cx_stream_sync(ctx, st));
      v0 = h[0];
      v1 = h[1];
      return CX_OK;
    }
    // own part: summed / maxed over ranks through one sum-all-reduce of [sum | one slot per rank]
    CX_TRY(pass(a, b, len_local, red_out.p));
    hipLaunchKernelGGL(k_pack_slots, dim3(1), dim3(64), 0, st, (const double*)red_out.p, ctx->rank, ctx->nranks, slots.p);
    CX_TRY(cx_allreduce_device(ctx, slots.p, 1 + ctx->nranks));
    if (len > len_local) CX_TRY(pass(a + len_local, b + len_local, len - len_local, red_out.p + 2));
    else CX_HIP(hipMemsetAsync(red_out.p + 2, 0, 2 * sizeof(double), st));
    std::vector<double> hs(size_t(1 + ctx->nranks));
    CX_TRY(cx_read_back(ctx, hs.data(), slots.p, hs.size() * sizeof(double), st));
    CX_TRY(cx_read_back(ctx, h, red_out.p + 2, 2 * sizeof(double), st));
    CX_TRY(cx_stream_sync(ctx, st));
    v0 = hs[0] + h[0];
    v1 = h[1];
    for (int r = 0; r < ctx->nranks; ++r) v1 = (MODE == RED_GRADIENT) ? std::max(v1, hs[1 + r]) : v1 + hs[1 + r];
    return CX_OK;
  }

  void Message(const char* fmt, double a, double b) { std::snprintf(out->message, sizeof(out->message), fmt, a, b); }

  int SquaredColumnNorm(double* dst) {
    CX_TRY(cx_matrix_squared_column_norm(J, dst, CX_DEVICE));
    if (ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, dst + n_local, n - n_local));
    return CX_OK;
  }

  // trust_region_minimizer.cc:228-299; *ok = false on an evaluation failure
  int EvaluateGradientAndJacobian(bool* ok) {
    hipStream_t st = ctx->stream;
    *ok = false;
    // with Jacobi scaling the ScaleColumns below rewrites F and its camera-major copy: no point in the
    // evaluation kernel writing that copy as well
    static const bool fuse_scaling = std::getenv("CX_NO_FUSED_SCALING") == nullptr;  // A/B switch
    const bool emit_saved = e->emit_ft;
    // with Jacobi scaling either a ScaleColumns follows (iteration 0: it writes the camera-major copy) or the evaluator
    // applies the scale itself and the copy is rebuilt by the gather pass at its first use, which is cheaper than
    // scattering it from the evaluation kernel (cx_eval.hip)
    if (o.jacobi_scaling) e->emit_ft = false;
    const int eval_rc = cx_evaluator_evaluate(e, x.p, &x_cost, residuals.p, gradient.p, 1, CX_DEVICE);
    e->emit_ft = emit_saved;
    CX_TRY(eval_rc);
    it.jacobian_ms = cx_evaluator_last_kernel_ms(e);
    if (!std::isfinite(x_cost)) {
      std::snprintf(out->message, sizeof(out->message), "Residual and Jacobian evaluation failed.");
      out->termination_type = CX_MIN_FAILURE;
      return CX_OK;
    }
    it.cost = x_cost;
    if (o.jacobi_scaling && it.iteration == 0) {
      // jacobian_scaling_ is computed once (trust_region_minimizer.cc:263-279).  This first evaluation is scaled by a
      // pass of its own; from then on the evaluator applies the same vector while it writes J (cx_evaluator_set_column_scale),
      // which is what ScaleColumns after every later evaluation amounts to, without the second pass over J.
      CX_TRY(SquaredColumnNorm(scaling.p));
      hipLaunchKernelGGL(k_jacobi_scaling, dim3(grid_for(n)), dim3(kBlock), 0, st, scaling.p, n);
      CX_TRY(cx_matrix_scale_columns(J, scaling.p, CX_DEVICE));
      if (fuse_scaling) CX_TRY(cx_evaluator_set_column_scale(e, scaling.p, CX_DEVICE));
    } else if (o.jacobi_scaling && !fuse_scaling) {
      CX_TRY(cx_matrix_scale_columns(J, scaling.p, CX_DEVICE));
    }
    // |x - Plus(x, -gradient)| in the ambient space; candidate_x is free at this point and serves as scratch
    CX_TRY(cxe_plus(e, x.p, gradient.p, -1.0, candidate_x.p));
    double sq = 0.0, mx = 0.0;
    CX_TRY(Reduce<RED_GRADIENT>(x.p, candidate_x.p, n_amb, n_local, sq, mx));
    it.gradient_max_norm = mx;
    it.gradient_norm = std::sqrt(sq);
    *ok = true;
    return CX_OK;
  }

  // trust_region_minimizer.cc:312-361
  int Finalize(LmStrategy& strategy, bool* can_continue) {
    if (it.step_is_successful) {
      ++out->num_successful_steps;
      if (x_cost < minimum_cost) {
        minimum_cost = x_cost;
        CX_HIP(hipMemcpyAsync(parameters, x.p, size_t(n_amb) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        it.step_is_nonmonotonic = 0;
      } else {
        it.step_is_nonmonotonic = 1;
      }
    } else {
      ++out->num_unsuccessful_steps;
    }
    it.trust_region_radius = strategy.radius;
    if (iterations && num_written < capacity) iterations[num_written] = it;
    ++num_written;
    last = it;
    *can_continue = false;
    if (it.iteration >= o.max_num_iterations) {
      std::snprintf(out->message, sizeof(out->message), "Maximum number of iterations reached. Number of iterations: %d.",
                    it.iteration);
      out->termination_type = CX_MIN_NO_CONVERGENCE;
      return CX_OK;
    }
    if (it.step_is_successful && it.gradient_max_norm <= o.gradient_tolerance) {
      Message("Gradient tolerance reached. Gradient max norm: %e <= %e", it.gradient_max_norm, o.gradient_tolerance);
      out->termination_type = CX_CONVERGENCE;
      return CX_OK;
    }
    if (it.trust_region_radius <= o.min_trust_region_radius) {
      Message("Minimum trust region radius reached. Trust region radius: %e <= %e", it.trust_region_radius,
              o.min_trust_region_radius);
      out->termination_type = CX_CONVERGENCE;
      return CX_OK;
    }
    *can_continue = true;
    return CX_OK;
  }

  // LevenbergMarquardtStrategy::ComputeStep + TrustRegionMinimizer::ComputeTrustRegionStep
  // (levenberg_marquardt_strategy.cc:69-151, trust_region_minimizer.cc:381-463); *fatal on FATAL_ERROR
  int ComputeTrustRegionStep(LmStrategy& strategy, bool* fatal) {
    hipStream_t st = ctx->stream;
    auto t0 = Clock::now();
    *fatal = false;
    it.step_is_valid = 0;
    if (!strategy.reuse_diagonal) CX_TRY(SquaredColumnNorm(diagonal.p));
    hipLaunchKernelGGL(k_lm_diagonal, dim3(grid_for(n)), dim3(kBlock), 0, st, diagonal.p, lm_diagonal.p, n, strategy.min_diagonal,
                       strategy.max_diagonal, strategy.radius, strategy.reuse_diagonal ? 0 : 1);
    // InvalidateArray (levenberg_marquardt_strategy.cc:110)
    hipLaunchKernelGGL(k_fill, dim3(grid_for(n)), dim3(kBlock), 0, st, step.p, n, std::numeric_limits<double>::quiet_NaN());
    CX_HIP(hipGetLastError());
    cx_per_solve_options ps{};
    ps.D = lm_diagonal.p;
    ps.q_tolerance = o.eta;
    ps.r_tolerance = -1.0;
    ps.memspace = CX_DEVICE;
    cx_summary ls{};
    CX_TRY(cx_solver_solve(solver, J, residuals.p, &ps, step.p, &ls));
    strategy.reuse_diagonal = true;
    if (ls.termination_type == CX_FATAL_ERROR) {
      std::snprintf(out->message, sizeof(out->message),
                    "Linear solver failed due to unrecoverable non-numeric causes: %.150s", ls.message);
      out->termination_type = CX_MIN_FAILURE;
      *fatal = true;
      return CX_OK;
    }
    it.linear_solver_iterations = ls.num_iterations;
    if (ls.termination_type == CX_FAILURE) {
      it.linear_solver_ms = MsSince(t0);
      return CX_OK;
    }
    hipLaunchKernelGGL(k_negate_and_unscale, dim3(grid_for(n)), dim3(kBlock), 0, st, step.p, (const double*)scaling.p, delta.p, n);
    it.linear_solver_ms = MsSince(t0);
    // model_cost_change = -(J step)' (f + J step / 2); a non-finite step makes it NaN, hence invalid,
    // which is what IsArrayValid -> FAILURE leads to in the reference
    CX_TRY(cx_matrix_right_multiply_overwrite(J, step.p, model_residuals.p, CX_DEVICE));  // (written outright: no zeroing, no read of the old value)
    double dot = 0.0, unused = 0.0;
    CX_TRY(Reduce<RED_MODEL_COST>(model_residuals.p, residuals.p, m, m, dot, unused));
    model_cost_change = -dot;
    it.step_is_valid = (model_cost_change > 0.0) ? 1 : 0;
    if (it.step_is_valid) num_consecutive_invalid_steps = 0;
    return CX_OK;
  }

  int Run(double* state, int32_t memspace) {
    hipStream_t st = ctx->stream;
    auto start = Clock::now();
    J = e->J;
    n = cx_evaluator_num_effective_parameters(e);
    n_amb = cx_evaluator_num_parameters(e);
    m = 2 * e->O;
    n_local = 3 * int64_t(e->P);
    for (DevBuf<double>* b : {&gradient, &step, &delta, &scaling, &diagonal, &lm_diagonal}) CX_TRY(b->alloc(size_t(n)));
    for (DevBuf<double>* b : {&x, &candidate_x}) CX_TRY(b->alloc(size_t(n_amb)));
    CX_TRY(residuals.alloc(size_t(m)));
    CX_TRY(model_residuals.alloc(size_t(m)));
    CX_TRY(partial.alloc(2 * kMinRedBlocks));
    CX_TRY(red_out.alloc(4));
    CX_TRY(slots.alloc(size_t(1 + ctx->nranks)));
    DevBuf<double> best;
    if (cx_is_host_space(memspace)) {
      CX_TRY(best.alloc(size_t(n_amb)));
      parameters = best.p;
      CX_TRY(cx_vector_in(ctx, x.p, state, size_t(n_amb), memspace));
      CX_HIP(hipMemcpyAsync(best.p, x.p, size_t(n_amb) * sizeof(double), hipMemcpyDeviceToDevice, st));
    } else {
      parameters = state;
      CX_HIP(hipMemcpyAsync(x.p, state, size_t(n_amb) * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(k_fill, dim3(grid_for(n)), dim3(kBlock), 0, st, scaling.p, n, 1.0);
    out->termination_type = CX_MIN_NO_CONVERGENCE;
    out->num_successful_steps = out->num_unsuccessful_steps = 0;
    out->message[0] = 0;
    // IterationZero (trust_region_minimizer.cc:170-214)
    it = cx_iteration_summary{};
    it.eta = o.eta;
    auto iteration_start = Clock::now();
    bool ok = false;
    CX_TRY(EvaluateGradientAndJacobian(&ok));
    if (ok) {
      out->initial_cost = x_cost;
      it.step_is_valid = 1;
      it.step_is_successful = 1;
      LmStrategy strategy;
      strategy.radius = o.initial_trust_region_radius;
      strategy.max_radius = o.max_trust_region_radius;
      strategy.min_diagonal = o.min_lm_diagonal;
      strategy.max_diagonal = o.max_lm_diagonal;
      StepEvaluator step_evaluator(x_cost, o.use_nonmonotonic_steps ? o.max_consecutive_nonmonotonic_steps : 0);
      bool atleast_one_successful_step = false;
      for (;;) {
        it.iteration_ms = MsSince(iteration_start);
        bool can_continue = false;
        CX_TRY(Finalize(strategy, &can_continue));
        if (!can_continue) break;
        iteration_start = Clock::now();
        const double previous_gradient_norm = it.gradient_norm;
        const double previous_gradient_max_norm = it.gradient_max_norm;
        const int next = last.iteration + 1;
        it = cx_iteration_summary{};
        it.iteration = next;
        bool fatal = false;
        CX_TRY(ComputeTrustRegionStep(strategy, &fatal));
        if (fatal) break;
        if (!it.step_is_valid) {
          // HandleInvalidStep (trust_region_minimizer.cc:468-499)
          if (++num_consecutive_invalid_steps >= o.max_num_consecutive_invalid_steps) {
            std::snprintf(out->message, sizeof(out->message),
                          "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d",
                          o.max_num_consecutive_invalid_steps);
            out->termination_type = CX_MIN_FAILURE;
            break;
          }
          strategy.StepRejected();  // StepIsInvalid
          it.cost = x_cost;
          it.cost_change = 0.0;
          it.gradient_max_norm = last.gradient_max_norm;
          it.gradient_norm = last.gradient_norm;
          it.step_norm = 0.0;
          it.relative_decrease = 0.0;
          it.eta = o.eta;
          continue;
        }
        // ComputeCandidatePointAndEvaluateCost (trust_region_minimizer.cc:753-774)
        CX_TRY(cxe_plus(e, x.p, delta.p, 1.0, candidate_x.p));
        CX_TRY(cx_evaluator_evaluate(e, candidate_x.p, &candidate_cost, nullptr, nullptr, 0, CX_DEVICE));
        it.residual_ms = cx_evaluator_last_kernel_ms(e);
        if (!std::isfinite(candidate_cost)) candidate_cost = std::numeric_limits<double>::max();
        if (atleast_one_successful_step) {
          // ParameterToleranceReached (trust_region_minimizer.cc:700-723)
          double x_sq = 0.0, d_sq = 0.0;
          CX_TRY(Reduce<RED_STEP>(x.p, candidate_x.p, n_amb, n_local, x_sq, d_sq));
          const double x_norm = std::sqrt(x_sq);
          it.step_norm = std::sqrt(d_sq);
          if (it.step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
            Message("Parameter tolerance reached. Relative step_norm: %e <= %e.",
                    it.step_norm / (x_norm + o.parameter_tolerance), o.parameter_tolerance);
            out->termination_type = CX_CONVERGENCE;
            break;
          }
        }
        // FunctionToleranceReached (trust_region_minimizer.cc:728-748)
        it.cost_change = x_cost - candidate_cost;
        if (std::fabs(it.cost_change) <= o.function_tolerance * x_cost) {
          Message("Function tolerance reached. |cost_change|/cost: %e <= %e", std::fabs(it.cost_change) / x_cost,
                  o.function_tolerance);
          out->termination_type = CX_CONVERGENCE;
          break;
        }
        // IsStepSuccessful (trust_region_minimizer.cc:777-820)
        it.relative_decrease = step_evaluator.StepQuality(candidate_cost, model_cost_change);
        if (it.relative_decrease > o.min_relative_decrease) {
          atleast_one_successful_step = true;
          // HandleSuccessfulStep (trust_region_minimizer.cc:825-840)
          std::swap(x.p, candidate_x.p);
          const double residual_ms = it.residual_ms;
          bool evaluated = false;
          CX_TRY(EvaluateGradientAndJacobian(&evaluated));
          it.residual_ms = residual_ms;
          if (!evaluated) break;
          it.step_is_successful = 1;
          strategy.StepAccepted(it.relative_decrease);
          step_evaluator.StepAccepted(candidate_cost, model_cost_change);
        } else {
          it.step_is_successful = 0;
          it.cost = candidate_cost;
          it.gradient_norm = previous_gradient_norm;
          it.gradient_max_norm = previous_gradient_max_norm;
          strategy.StepRejected();
        }
      }
    }
    out->num_iterations = num_written;
    out->final_cost = minimum_cost;
    if (cx_is_host_space(memspace)) CX_TRY(cx_vector_out(ctx, state, parameters, size_t(n_amb), memspace));
    CX_TRY(cx_stream_sync(ctx, st));
    out->total_ms = MsSince(start);
    return CX_OK;
  }
};

}  // namespace

extern "C" {

void cx_minimizer_default_options(cx_minimizer_options* o) {
  if (!o) return;
  *o = cx_minimizer_options{};
  o->max_num_iterations = 50;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->use_nonmonotonic_steps = 0;
  o->max_consecutive_nonmonotonic_steps = 5;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->eta = 1e-1;
}

int cx_minimize(cx_evaluator* e, cx_solver* s, const cx_minimizer_options* options, double* state, int32_t memspace,
                cx_minimizer_summary* summary, cx_iteration_summary* iterations, int32_t capacity) {
  CX_CHECK_ARG(e && s && options && state && summary);
  if (!e->parts.empty() || cxm_is_front(s->ctx)) {
    CX_CHECK_ARG(capacity >= 0 && (iterations != nullptr || capacity == 0));
    return cxm_minimize(e, s, options, state, memspace, summary, iterations, capacity);
  }
  CX_CHECK_ARG(s->ctx == e->ctx);
  CX_CHECK_ARG(memspace == CX_HOST || memspace == CX_DEVICE || memspace == CX_HOST_SLICES);
  CX_CHECK_ARG(capacity >= 0 && (iterations != nullptr || capacity == 0));
  // LevenbergMarquardtStrategy's constructor checks (levenberg_marquardt_strategy.cc:62-65)
  CX_CHECK_ARG(options->min_lm_diagonal > 0.0 && options->min_lm_diagonal <= options->max_lm_diagonal);
  CX_CHECK_ARG(options->max_trust_region_radius > 0.0 && options->initial_trust_region_radius > 0.0);
  *summary = cx_minimizer_summary{};
  Minimizer mz;
  mz.e = e;
  mz.solver = s;
  mz.ctx = e->ctx;
  mz.o = *options;
  mz.out = summary;
  mz.iterations = iterations;
  mz.capacity = capacity;
  (void)cx_evaluator_set_column_scale(e, nullptr, CX_HOST);
  const int rc = mz.Run(state, memspace);
  // the Jacobi scaling the loop registered with the evaluator ends with the minimisation
  (void)cx_evaluator_set_column_scale(e, nullptr, CX_HOST);
  return rc;
}

}  // extern "C"
