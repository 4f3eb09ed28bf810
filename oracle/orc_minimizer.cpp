// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// TrustRegionMinimizer::Minimize (trust_region_minimizer.cc:68-840) with
// LevenbergMarquardtStrategy (levenberg_marquardt_strategy.cc:50-175, .h:63-69) and
// TrustRegionStepEvaluator (trust_region_step_evaluator.cc:40-117), restated for an
// unconstrained program on Euclidean manifolds (Plus(x, d) = x + d), without inner
// iterations, line search, callbacks or a time limit.
//
// Pinned by the reference's own known-answer test for this loop: Powell's singular function
// from (3, -1, 0, 1) must reach the origin to 1e-3 for every subset of active columns
// (trust_region_minimizer_test.cc:215-290) -- tests/test_oracle_minimizer.py.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <limits>

#include "orc_api.h"
#include "orc_common.h"

namespace orc {
namespace {

// levenberg_marquardt_strategy.cc:50-175
struct LmStrategy {
  const orc_min_problem* p;
  int n = 0;  // tangent size (jacobian->num_cols())
  double radius, max_radius, min_diagonal, max_diagonal;
  double decrease_factor = 2.0;
  bool reuse_diagonal = false;
  std::vector<double> diagonal, lm_diagonal;

  // returns cx_termination
  int ComputeStep(double eta, const double* residuals, double* step, int* num_iterations) {
    if (!reuse_diagonal) {
      diagonal.assign(n, 0.0);
      p->squared_column_norm(p->user, diagonal.data());
      for (double& d : diagonal) d = std::min(std::max(d, min_diagonal), max_diagonal);
    }
    lm_diagonal.resize(n);
    for (int i = 0; i < n; ++i) lm_diagonal[i] = std::sqrt(diagonal[i] / radius);
    for (int i = 0; i < n; ++i) step[i] = std::numeric_limits<double>::quiet_NaN();  // InvalidateArray
    int type = p->solve(p->user, residuals, lm_diagonal.data(), eta, step, num_iterations);
    if (type != CX_FATAL_ERROR && type != CX_FAILURE) {
      bool valid = true;
      for (int i = 0; i < n; ++i) valid = valid && std::isfinite(step[i]);
      if (!valid) type = CX_FAILURE;
      else for (int i = 0; i < n; ++i) step[i] = -step[i];
    }
    reuse_diagonal = true;
    return type;
  }
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
      : max_consecutive_nonmonotonic_steps(max_nonmonotonic), minimum_cost(initial_cost),
        current_cost(initial_cost), reference_cost(initial_cost), candidate_cost(initial_cost) {}
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

using Clock = std::chrono::steady_clock;
double MsSince(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

struct Minimizer {
  const orc_min_problem* p;
  cx_minimizer_options o;
  cx_minimizer_summary* out;
  cx_iteration_summary* iterations;
  int capacity;
  int n, m, n_amb;  // n: tangent size (gradient, step), n_amb: ambient size (x)
  std::vector<double> x, candidate_x, residuals, gradient, step, delta, scaling, model_residuals;
  double x_cost = std::numeric_limits<double>::max(), minimum_cost = x_cost, candidate_cost = 0.0;
  double model_cost_change = 0.0;
  int num_consecutive_invalid_steps = 0;
  double* parameters;
  cx_iteration_summary it{};
  cx_iteration_summary last{};
  int num_written = 0;

  void Message(const char* fmt, double a, double b) { std::snprintf(out->message, sizeof(out->message), fmt, a, b); }

  void Plus(const double* xx, const double* dd, double* out) const {
    if (p->plus) p->plus(p->user, xx, dd, out);
    else for (int i = 0; i < n; ++i) out[i] = xx[i] + dd[i];
  }

  // trust_region_minimizer.cc:228-299
  bool EvaluateGradientAndJacobian() {
    if (!p->evaluate(p->user, x.data(), &x_cost, residuals.data(), gradient.data(), 1)) {
      std::snprintf(out->message, sizeof(out->message), "Residual and Jacobian evaluation failed.");
      out->termination_type = CX_MIN_FAILURE;
      return false;
    }
    it.cost = x_cost;
    if (o.jacobi_scaling) {
      if (it.iteration == 0) {
        p->squared_column_norm(p->user, scaling.data());
        for (double& s : scaling) s = 1.0 / (1.0 + std::sqrt(s));
      }
      p->scale_columns(p->user, scaling.data());
    }
    // |Plus(x, -gradient) - x| in the ambient space
    std::vector<double> negative_gradient(n), projected(n_amb);
    for (int i = 0; i < n; ++i) negative_gradient[i] = -gradient[i];
    Plus(x.data(), negative_gradient.data(), projected.data());
    double max_norm = 0.0, sq = 0.0;
    for (int i = 0; i < n_amb; ++i) {
      const double d = x[i] - projected[i];
      max_norm = std::max(max_norm, std::fabs(d));
      sq += d * d;
    }
    it.gradient_max_norm = max_norm;
    it.gradient_norm = std::sqrt(sq);
    return true;
  }

  // trust_region_minimizer.cc:312-361
  bool FinalizeIterationAndCheckIfMinimizerCanContinue(LmStrategy& strategy) {
    if (it.step_is_successful) {
      ++out->num_successful_steps;
      if (x_cost < minimum_cost) {
        minimum_cost = x_cost;
        std::copy(x.begin(), x.end(), parameters);
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
    if (it.iteration >= o.max_num_iterations) {
      std::snprintf(out->message, sizeof(out->message), "Maximum number of iterations reached. Number of iterations: %d.",
                    it.iteration);
      out->termination_type = CX_MIN_NO_CONVERGENCE;
      return false;
    }
    if (it.step_is_successful && it.gradient_max_norm <= o.gradient_tolerance) {
      Message("Gradient tolerance reached. Gradient max norm: %e <= %e", it.gradient_max_norm, o.gradient_tolerance);
      out->termination_type = CX_CONVERGENCE;
      return false;
    }
    if (it.trust_region_radius <= o.min_trust_region_radius) {
      Message("Minimum trust region radius reached. Trust region radius: %e <= %e", it.trust_region_radius,
              o.min_trust_region_radius);
      out->termination_type = CX_CONVERGENCE;
      return false;
    }
    return true;
  }

  // trust_region_minimizer.cc:381-463
  bool ComputeTrustRegionStep(LmStrategy& strategy) {
    auto t0 = Clock::now();
    it.step_is_valid = 0;
    int num_iterations = 0;
    const int type = strategy.ComputeStep(o.eta, residuals.data(), step.data(), &num_iterations);
    if (type == CX_FATAL_ERROR) {
      std::snprintf(out->message, sizeof(out->message),
                    "Linear solver failed due to unrecoverable non-numeric causes. Please see the error log for clues. ");
      out->termination_type = CX_MIN_FAILURE;
      return false;
    }
    it.linear_solver_ms = MsSince(t0);
    it.linear_solver_iterations = num_iterations;
    if (type == CX_FAILURE) return true;
    std::fill(model_residuals.begin(), model_residuals.end(), 0.0);
    p->right_multiply(p->user, step.data(), model_residuals.data());
    double dot = 0.0;
    for (int i = 0; i < m; ++i) dot += model_residuals[i] * (residuals[i] + model_residuals[i] / 2.0);
    model_cost_change = -dot;
    it.step_is_valid = (model_cost_change > 0.0);
    if (it.step_is_valid) {
      for (int i = 0; i < n; ++i) delta[i] = step[i] * scaling[i];
      num_consecutive_invalid_steps = 0;
    }
    return true;
  }

  // trust_region_minimizer.cc:468-499
  bool HandleInvalidStep(LmStrategy& strategy) {
    if (++num_consecutive_invalid_steps >= o.max_num_consecutive_invalid_steps) {
      std::snprintf(out->message, sizeof(out->message),
                    "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d",
                    o.max_num_consecutive_invalid_steps);
      out->termination_type = CX_MIN_FAILURE;
      return false;
    }
    strategy.StepRejected();  // StepIsInvalid, levenberg_marquardt_strategy.h:63-69
    it.cost = x_cost;
    it.cost_change = 0.0;
    it.gradient_max_norm = last.gradient_max_norm;
    it.gradient_norm = last.gradient_norm;
    it.step_norm = 0.0;
    it.relative_decrease = 0.0;
    it.eta = o.eta;
    return true;
  }

  void Run() {
    auto start = Clock::now();
    n_amb = p->num_parameters;
    n = (p->plus && p->num_effective_parameters > 0) ? p->num_effective_parameters : n_amb;
    m = p->num_residuals;
    x.assign(parameters, parameters + n_amb);
    candidate_x.resize(n_amb); residuals.resize(m); gradient.resize(n); step.resize(n); delta.resize(n);
    model_residuals.resize(m);
    scaling.assign(n, 1.0);
    out->termination_type = CX_MIN_NO_CONVERGENCE;
    out->num_successful_steps = out->num_unsuccessful_steps = 0;
    out->message[0] = 0;
    // IterationZero (trust_region_minimizer.cc:170-214)
    it = cx_iteration_summary{};
    it.eta = o.eta;
    auto iteration_start = Clock::now();
    bool ok = EvaluateGradientAndJacobian();
    if (ok) {
      out->initial_cost = x_cost;
      it.step_is_valid = 1;
      it.step_is_successful = 1;
      LmStrategy strategy;
      strategy.p = p;
      strategy.n = n;
      strategy.radius = o.initial_trust_region_radius;
      strategy.max_radius = o.max_trust_region_radius;
      strategy.min_diagonal = o.min_lm_diagonal;
      strategy.max_diagonal = o.max_lm_diagonal;
      StepEvaluator step_evaluator(x_cost, o.use_nonmonotonic_steps ? o.max_consecutive_nonmonotonic_steps : 0);
      bool atleast_one_successful_step = false;
      for (;;) {
        it.iteration_ms = MsSince(iteration_start);
        if (!FinalizeIterationAndCheckIfMinimizerCanContinue(strategy)) break;
        iteration_start = Clock::now();
        const double previous_gradient_norm = it.gradient_norm;
        const double previous_gradient_max_norm = it.gradient_max_norm;
        const int next = last.iteration + 1;
        it = cx_iteration_summary{};
        it.iteration = next;
        if (!ComputeTrustRegionStep(strategy)) break;
        if (!it.step_is_valid) {
          if (!HandleInvalidStep(strategy)) break;
          continue;
        }
        // ComputeCandidatePointAndEvaluateCost (trust_region_minimizer.cc:753-774)
        Plus(x.data(), delta.data(), candidate_x.data());
        if (!p->evaluate(p->user, candidate_x.data(), &candidate_cost, nullptr, nullptr, 0))
          candidate_cost = std::numeric_limits<double>::max();
        if (atleast_one_successful_step) {
          // ParameterToleranceReached (trust_region_minimizer.cc:700-723)
          double x_sq = 0.0, d_sq = 0.0;
          for (int i = 0; i < n_amb; ++i) {
            x_sq += x[i] * x[i];
            const double d = x[i] - candidate_x[i];
            d_sq += d * d;
          }
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
          x = candidate_x;
          if (!EvaluateGradientAndJacobian()) break;
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
    out->total_ms = MsSince(start);
  }
};

// ------------------------------------------------------------------ BAL program
struct BalProgram {
  int camera_model = CX_CAMERA_ANGLE_AXIS;
  int C, P;
  int64_t O;
  const int32_t *cam, *pt;
  const double* obs;
  int loss_type;
  double loss_a, loss_b;
  cx_solver_options solver_options;
  std::vector<int64_t> order;
  std::vector<cx_block> rows, cols;
  std::vector<int32_t> rcb;
  std::vector<cx_cell> cells;
  cx_block_structure bs;
  std::vector<double> values;
};
int BalEvaluate(void* u, const double* x, double* cost, double* residuals, double* gradient, int want_jacobian) {
  auto* b = static_cast<BalProgram*>(u);
  orc_bal_evaluate_model(&b->bs, b->C, b->P, b->O, b->cam, b->pt, b->obs, b->order.data(), x, b->camera_model, b->loss_type,
                         b->loss_a, b->loss_b, cost, residuals, gradient, want_jacobian ? b->values.data() : nullptr);
  return std::isfinite(*cost) ? 1 : 0;
}
void BalSqNorm(void* u, double* out) {
  auto* b = static_cast<BalProgram*>(u);
  orc_squared_column_norm(&b->bs, b->values.data(), out);
}
void BalScale(void* u, const double* scale) {
  auto* b = static_cast<BalProgram*>(u);
  orc_scale_columns(&b->bs, b->values.data(), scale);
}
void BalRight(void* u, const double* x, double* y) {
  auto* b = static_cast<BalProgram*>(u);
  orc_right_multiply(&b->bs, b->values.data(), x, y);
}
void BalPlus(void* u, const double* x, const double* delta, double* out) {
  auto* b = static_cast<BalProgram*>(u);
  orc_bal_plus(b->C, b->P, b->camera_model, x, delta, out);
}
int BalSolve(void* u, const double* rhs, const double* D, double q_tolerance, double* x, int* num_iterations) {
  auto* b = static_cast<BalProgram*>(u);
  cx_summary s{};
  orc_solve(&b->bs, b->values.data(), rhs, D, &b->solver_options, -1.0, q_tolerance, x, &s);
  *num_iterations = s.num_iterations;
  return s.termination_type;
}

}  // namespace
}  // namespace orc

extern "C" {

int orc_minimize(const orc_min_problem* problem, const cx_minimizer_options* options, double* parameters,
                 cx_minimizer_summary* summary, cx_iteration_summary* iterations, int capacity) {
  orc::Minimizer mz;
  mz.p = problem;
  mz.o = *options;
  mz.out = summary;
  mz.iterations = iterations;
  mz.capacity = capacity;
  mz.parameters = parameters;
  mz.Run();
  return 0;
}

int orc_minimize_bal(int C, int P, int64_t O, const int32_t* cam, const int32_t* pt, const double* obs, int camera_model,
                     int loss_type,
                     double loss_a, double loss_b, const cx_solver_options* solver_options,
                     const cx_minimizer_options* options, double* state, cx_minimizer_summary* summary,
                     cx_iteration_summary* iterations, int capacity) {
  orc::BalProgram b;
  b.C = C; b.P = P; b.O = O; b.cam = cam; b.pt = pt; b.obs = obs;
  b.camera_model = camera_model;
  b.loss_type = loss_type; b.loss_a = loss_a; b.loss_b = loss_b;
  b.solver_options = *solver_options;
  b.order.resize(O);
  orc_bal_residual_order(P, O, pt, b.order.data());
  b.rows.resize(O); b.cols.resize(size_t(P) + C); b.rcb.resize(O + 1); b.cells.resize(2 * O);
  orc_bal_structure(C, P, O, cam, pt, b.order.data(), b.rows.data(), b.cols.data(), b.rcb.data(), b.cells.data());
  b.bs = cx_block_structure{int32_t(O), P + C, b.rows.data(), b.cols.data(), b.rcb.data(), b.cells.data()};
  b.values.assign(size_t(24) * O, 0.0);
  const int cam_size = camera_model == CX_CAMERA_ANGLE_AXIS ? 9 : 10;
  orc_min_problem p{3 * P + cam_size * C, int32_t(2 * O), &b, orc::BalEvaluate, orc::BalSqNorm, orc::BalScale, orc::BalRight,
                    orc::BalSolve, 3 * P + 9 * C, orc::BalPlus};
  return orc_minimize(&p, options, state, summary, iterations, capacity);
}

}  // extern "C"
