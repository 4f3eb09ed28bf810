// CxIdentifyLoss (cx_loss_probe.h) on the host alone: every built-in loss of include/ceres/loss_function.h:131-292 is
// recognised with its constructor arguments from LossFunction::Evaluate, what is not a built-in loss is declined.
// No device is touched: part of the CPU test suite (tests/test_library_cpu.py).
#include <cstdio>
#include <memory>

#include "cx_loss_probe.h"

using namespace ceres;
using namespace ceres::internal;

static int failures = 0;
static void Check(const char* name, const LossFunction* loss, int32_t want_type, double want_a, double want_b) {
  int32_t type = -1;
  double a = -1.0, b = -1.0;
  const bool ok = CxIdentifyLoss(loss, &type, &a, &b);
  const bool good = ok && type == want_type && std::abs(a - want_a) <= 4e-16 * std::abs(want_a) && std::abs(b - want_b) <= 4e-16 * std::abs(want_b);
  if (!good) ++failures;
  std::printf("%-34s %s  type %d a %.17g b %.17g\n", name, good ? "ok    " : "FAILED", type, a, b);
}
static void CheckDeclined(const char* name, const LossFunction* loss) {
  int32_t type = -1;
  double a, b;
  const bool declined = !CxIdentifyLoss(loss, &type, &a, &b);
  if (!declined) ++failures;
  std::printf("%-34s %s\n", name, declined ? "ok     (declined)" : "FAILED (accepted)");
}

int main() {
  Check("nullptr", nullptr, CX_LOSS_NONE, 0.0, 0.0);
  { TrivialLoss l; Check("TrivialLoss", &l, CX_LOSS_NONE, 0.0, 0.0); }
  for (double a : {1.0, 0.5, 2.25, 0.037, 1e-3, 1e3, 1.2345678901234567, 3.141592653589793}) {
    char name[64];
    { HuberLoss l(a); std::snprintf(name, sizeof(name), "HuberLoss(%.17g)", a); Check(name, &l, CX_LOSS_HUBER, a, 0.0); }
    { SoftLOneLoss l(a); std::snprintf(name, sizeof(name), "SoftLOneLoss(%.17g)", a); Check(name, &l, CX_LOSS_SOFT_L_ONE, a, 0.0); }
    { CauchyLoss l(a); std::snprintf(name, sizeof(name), "CauchyLoss(%.17g)", a); Check(name, &l, CX_LOSS_CAUCHY, a, 0.0); }
    { ArctanLoss l(a); std::snprintf(name, sizeof(name), "ArctanLoss(%.17g)", a); Check(name, &l, CX_LOSS_ARCTAN, a, 0.0); }
    { TukeyLoss l(a); std::snprintf(name, sizeof(name), "TukeyLoss(%.17g)", a); Check(name, &l, CX_LOSS_TUKEY, a, 0.0); }
  }
  for (auto ab : {std::pair<double, double>{0.3, 0.1}, {1.0, 0.5}, {0.0, 1.0}, {5.0, 2.0}}) {
    char name[64];
    TolerantLoss l(ab.first, ab.second);
    std::snprintf(name, sizeof(name), "TolerantLoss(%g, %g)", ab.first, ab.second);
    Check(name, &l, CX_LOSS_TOLERANT, ab.first, ab.second);
  }
  { ScaledLoss l(new HuberLoss(1.0), 2.0); CheckDeclined("ScaledLoss(HuberLoss(1), 2)", &l); }
  { ScaledLoss l(nullptr, 0.5); CheckDeclined("ScaledLoss(nullptr, 0.5)", &l); }
  std::printf("%s\n", failures ? "FAILED" : "ALL OK");
  return failures ? 1 : 0;
}
