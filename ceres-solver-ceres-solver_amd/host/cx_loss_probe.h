// Which built-in LossFunction is this, and with which constructor arguments?
//
// The device evaluator applies a robust loss by (type, a, b) -- cx_evaluator_set_loss -- but the reference's loss objects
// keep their parameters private (include/ceres/loss_function.h:174-292: HuberLoss::a_, SoftLOneLoss::b_, ...), and a
// program built by the user (examples/bundle_adjuster.cc:327-328, `new HuberLoss(1.0)` per residual block) carries only
// `const LossFunction*`.  What IS public is Evaluate(s, rho[3]) (loss_function.h:86-88), so the parameters are recovered
// from it: every candidate type has a closed form for its argument(s) in terms of rho'(s) at one or two abscissae; the
// recovered value (rounded to a short decimal first, which is what a constructor argument usually is) must then make the
// candidate reproduce the object's three outputs to a few ulps on a grid of abscissae that covers inlier and outlier
// regions.  A loss that matches no candidate
// (ScaledLoss, ComposedLoss, a user's own class) is declined and the factory falls back to ProgramEvaluator.
// The formulas are those of loss_function.cc:46-144, which the device kernel repeats (cx_eval.hip: loss_evaluate).
#ifndef CX_LOSS_PROBE_H_
#define CX_LOSS_PROBE_H_

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "../../include/cxschur.h"
#ifdef CX_USE_CERES_HEADERS
#include "ceres/loss_function.h"
#else
#include "ceres_mirror.h"
#endif

namespace ceres::internal {

// rho(s), rho'(s), rho''(s) of a built-in loss by (type, a, b): loss_function.cc:46-144
inline void CxLossEvaluate(int32_t type, double a, double b, double s, double rho[3]) {
  const double kMin = std::numeric_limits<double>::min();
  switch (type) {
    case CX_LOSS_HUBER: {
      const double bb = a * a;
      if (s > bb) {
        const double r = std::sqrt(s);
        rho[0] = 2.0 * a * r - bb;
        rho[1] = std::max(kMin, a / r);
        rho[2] = -rho[1] / (2.0 * s);
      } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
      }
      return;
    }
    case CX_LOSS_SOFT_L_ONE: {
      const double bb = a * a, c = 1 / bb;
      const double sum = 1.0 + s * c, tmp = std::sqrt(sum);
      rho[0] = 2.0 * bb * (tmp - 1.0);
      rho[1] = std::max(kMin, 1.0 / tmp);
      rho[2] = -(c * rho[1]) / (2.0 * sum);
      return;
    }
    case CX_LOSS_CAUCHY: {
      const double bb = a * a, c = 1 / bb;
      const double sum = 1.0 + s * c, inv = 1.0 / sum;
      rho[0] = bb * std::log(sum);
      rho[1] = std::max(kMin, inv);
      rho[2] = -c * (inv * inv);
      return;
    }
    case CX_LOSS_ARCTAN: {
      const double bb = 1 / (a * a);
      const double sum = 1 + s * s * bb, inv = 1 / sum;
      rho[0] = a * std::atan2(s, a);
      rho[1] = std::max(kMin, inv);
      rho[2] = -2.0 * s * bb * (inv * inv);
      return;
    }
    case CX_LOSS_TOLERANT: {
      const double c = b * std::log(1.0 + std::exp(-a / b));
      const double x = (s - a) / b;
      if (x > 36.7) {
        rho[0] = s - a - c; rho[1] = 1.0; rho[2] = 0.0;
      } else {
        const double e_x = std::exp(x);
        rho[0] = b * std::log(1.0 + e_x) - c;
        rho[1] = std::max(kMin, e_x / (1.0 + e_x));
        rho[2] = 0.5 / (b * (1.0 + std::cosh(x)));
      }
      return;
    }
    case CX_LOSS_TUKEY: {
      const double a2 = a * a;
      if (s <= a2) {
        const double value = 1.0 - s / a2, value_sq = value * value;
        rho[0] = a2 / 3.0 * (1.0 - value_sq * value);
        rho[1] = value_sq;
        rho[2] = -2.0 / a2 * value;
      } else {
        rho[0] = a2 / 3.0; rho[1] = 0.0; rho[2] = 0.0;
      }
      return;
    }
    default:
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

namespace cx_loss_probe_detail {

// equal up to a few units in the last place (the reference library and this header may be compiled with different
// floating-point contraction, so bit equality would be too strict; a different loss type or argument differs by far more)
inline bool Close(const double x[3], const double y[3]) {
  for (int i = 0; i < 3; ++i) {
    const double tol = 8.0 * std::numeric_limits<double>::epsilon() * std::max(std::abs(x[i]), std::abs(y[i]));
    if (!(std::abs(x[i] - y[i]) <= tol)) return false;
  }
  return true;
}

// does (type, a, b) reproduce `loss` on abscissae spread around its own scale (inlier and outlier regions)?
inline bool Reproduces(const LossFunction& loss, int32_t type, double a, double b) {
  const double scale = type == CX_LOSS_NONE ? 1.0 : (type == CX_LOSS_ARCTAN ? a : (type == CX_LOSS_TOLERANT ? std::max(a, b) : a * a));
  static const double kFactors[] = {0.0, 1e-9, 1e-3, 0.25, 0.5, 0.9, 1.1, 2.0, 7.5, 1e2, 1e5, 1e12};
  for (double f : kFactors) {
    for (double s : {f * scale, f}) {
      double want[3], got[3];
      loss.Evaluate(s, want);
      CxLossEvaluate(type, a, b, s, got);
      if (!Close(want, got)) return false;
    }
  }
  return true;
}

// a0 (and b0) are the closed-form estimates of the constructor arguments: exact up to rounding, so a short decimal
// (what a constructor argument usually is) is tried first, then the estimate itself
inline bool Search(const LossFunction& loss, int32_t type, double a0, double b0, bool two, double* a, double* b) {
  if (!(a0 > 0.0 || (two && a0 >= 0.0)) || !std::isfinite(a0) || (two && !(b0 > 0.0 && std::isfinite(b0)))) return false;
  for (double digits : {1e3, 1e6, 1e9}) {
    const double nice_a = std::round(a0 * digits) / digits, nice_b = two ? std::round(b0 * digits) / digits : 0.0;
    if ((nice_a > 0.0 || (two && nice_a >= 0.0)) && (!two || nice_b > 0.0) && Reproduces(loss, type, nice_a, nice_b)) {
      *a = nice_a; *b = nice_b;
      return true;
    }
  }
  if (Reproduces(loss, type, a0, b0)) { *a = a0; *b = two ? b0 : 0.0; return true; }
  return false;
}

}  // namespace cx_loss_probe_detail

// true and (type, a, b) when `loss` (nullptr: no loss) behaves exactly like a built-in loss function.
inline bool CxIdentifyLoss(const LossFunction* loss, int32_t* type, double* a, double* b) {
  using namespace cx_loss_probe_detail;
  *type = CX_LOSS_NONE; *a = 0.0; *b = 0.0;
  if (loss == nullptr) return true;
  if (Reproduces(*loss, CX_LOSS_NONE, 0.0, 0.0)) return true;  // TrivialLoss
  double r[3], r2[3];
  // Huber: rho' = a / sqrt(s) in the outlier region
  loss->Evaluate(1e100, r);
  if (Search(*loss, CX_LOSS_HUBER, r[1] * 1e50, 0.0, false, a, b)) { *type = CX_LOSS_HUBER; return true; }
  // SoftLOne: rho'(s) = 1 / sqrt(1 + s / a^2)   Cauchy: rho'(s) = 1 / (1 + s / a^2)   Arctan: rho'(s) = 1 / (1 + s^2 / a^2)
  for (double s : {1.0, 1e-6, 1e6}) {
    loss->Evaluate(s, r);
    if (!(r[1] > 0.0 && r[1] < 1.0)) continue;
    if (Search(*loss, CX_LOSS_SOFT_L_ONE, std::sqrt(s / (1.0 / (r[1] * r[1]) - 1.0)), 0.0, false, a, b)) { *type = CX_LOSS_SOFT_L_ONE; return true; }
    if (Search(*loss, CX_LOSS_CAUCHY, std::sqrt(s / (1.0 / r[1] - 1.0)), 0.0, false, a, b)) { *type = CX_LOSS_CAUCHY; return true; }
    if (Search(*loss, CX_LOSS_ARCTAN, s / std::sqrt(1.0 / r[1] - 1.0), 0.0, false, a, b)) { *type = CX_LOSS_ARCTAN; return true; }
  }
  // Tukey: rho = a^2 / 3 in the outlier region
  loss->Evaluate(1e300, r);
  if (r[1] == 0.0 && r[2] == 0.0 && Search(*loss, CX_LOSS_TUKEY, std::sqrt(3.0 * r[0]), 0.0, false, a, b)) { *type = CX_LOSS_TUKEY; return true; }
  // Tolerant: logit(rho'(s)) = (s - a) / b, two abscissae where rho' is well inside (0, 1)
  for (double s1 : {0.0, 1.0, 1e2, 1e4}) {
    const double s2 = s1 + 1.0;
    loss->Evaluate(s1, r);
    loss->Evaluate(s2, r2);
    if (!(r[1] > 1e-6 && r[1] < 1.0 - 1e-6 && r2[1] > 1e-6 && r2[1] < 1.0 - 1e-6)) continue;
    const double x1 = std::log(r[1] / (1.0 - r[1])), x2 = std::log(r2[1] / (1.0 - r2[1]));
    const double b0 = (s2 - s1) / (x2 - x1), a0 = s1 - b0 * x1;
    if (Search(*loss, CX_LOSS_TOLERANT, std::max(a0, 0.0), b0, true, a, b)) { *type = CX_LOSS_TOLERANT; return true; }
  }
  return false;
}

}  // namespace ceres::internal
#endif
