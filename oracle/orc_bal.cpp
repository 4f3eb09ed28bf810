// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// Bundle-adjustment evaluation and ordering: SnavelyReprojectionError through
// forward-mode duals (include/ceres/jet.h, internal/autodiff.h),
// AngleAxisRotatePoint (include/ceres/rotation.h:792-857),
// LexicographicallyOrderResidualBlocks (reorder_program.cc:256-338),
// ComputeStableSchurOrdering (parameter_block_ordering.cc:50-83,
// graph_algorithms.h:165-227), BuildJacobianLayout
// (block_jacobian_writer.cc:68-167) and ProgramEvaluator::Evaluate
// (program_evaluator.h:137-304) for the BAL program.
//
// PARITY UNPINNED for the Snavely arithmetic: the reference holds no numeric
// fixture for it (data/problem-16-22106-pre.txt is a stripped blob); tests pin
// it by central differences and scipy's rotation instead.
#include <omp.h>

#include <algorithm>
#include <limits>
#include <numeric>

#include "orc_api.h"
#include "orc_common.h"

namespace orc {

// include/ceres/jet.h restated for T = double.
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  explicit Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};
template <int N> inline Jet<N> operator-(const Jet<N>& f) { Jet<N> r; r.a = -f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }
template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a + g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a - g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i]; return r; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> r = f; r.a = f.a + s; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> r = f; r.a = f.a - s; return r; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> r; r.a = s - f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }
// jet.h:349-352
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a * g.a; for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) { Jet<N> r; r.a = f.a * s; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s; return r; }
// jet.h:367-379
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  const double g_a_inverse = 1.0 / g.a;
  const double f_a_by_g_a = f.a * g_a_inverse;
  Jet<N> r; r.a = f_a_by_g_a;
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
  return r;
}
// jet.h:382-386
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) {
  const double m = -s / (g.a * g.a);
  Jet<N> r; r.a = s / g.a; for (int i = 0; i < N; ++i) r.v[i] = g.v[i] * m; return r;
}
// jet.h:614-629
template <int N> inline Jet<N> cos(const Jet<N>& f) { Jet<N> r; r.a = std::cos(f.a); const double m = -std::sin(f.a); for (int i = 0; i < N; ++i) r.v[i] = m * f.v[i]; return r; }
template <int N> inline Jet<N> sin(const Jet<N>& f) { Jet<N> r; r.a = std::sin(f.a); const double m = std::cos(f.a); for (int i = 0; i < N; ++i) r.v[i] = m * f.v[i]; return r; }
// jet.h:733-748
template <int N> inline Jet<N> hypot3(const Jet<N>& x, const Jet<N>& y, const Jet<N>& z) {
  const double tmp = std::hypot(x.a, y.a, z.a);
  Jet<N> r; r.a = tmp;
  const double cx = x.a / tmp, cy = y.a / tmp, cz = z.a / tmp;
  for (int i = 0; i < N; ++i) r.v[i] = cx * x.v[i] + cy * y.v[i] + cz * z.v[i];
  return r;
}
inline double hypot3(double x, double y, double z) { return std::hypot(x, y, z); }
// jet.h:483-487
template <int N> inline Jet<N> sqrt(const Jet<N>& f) {
  const double tmp = std::sqrt(f.a);
  const double two_a_inverse = 1.0 / (2.0 * tmp);
  Jet<N> r; r.a = tmp;
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * two_a_inverse;
  return r;
}
inline double value_of(double x) { return x; }
template <int N> inline double value_of(const Jet<N>& x) { return x.a; }
inline double make(double, double s) { return s; }
template <int N> inline Jet<N> make(const Jet<N>&, double s) { return Jet<N>(s); }

// rotation.h:792-857
template <typename T>
inline void AngleAxisRotatePoint(const T aa[3], const T pt[3], T result[3]) {
  using std::cos;
  using std::sin;
  const T theta = hypot3(aa[0], aa[1], aa[2]);
  if (std::fpclassify(value_of(theta)) != FP_ZERO) {
    const T costheta = cos(theta);
    const T sintheta = sin(theta);
    const T theta_inverse = 1.0 / theta;
    const T w[3] = {aa[0] * theta_inverse, aa[1] * theta_inverse, aa[2] * theta_inverse};
    const T w_cross_pt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - costheta);
    result[0] = pt[0] * costheta + w_cross_pt[0] * sintheta + w[0] * tmp;
    result[1] = pt[1] * costheta + w_cross_pt[1] * sintheta + w[1] * tmp;
    result[2] = pt[2] * costheta + w_cross_pt[2] * sintheta + w[2] * tmp;
  } else {
    const T w_cross_pt[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0]};
    result[0] = pt[0] + w_cross_pt[0];
    result[1] = pt[1] + w_cross_pt[1];
    result[2] = pt[2] + w_cross_pt[2];
  }
}

// examples/snavely_reprojection_error.h:57-92
template <typename T>
inline void Snavely(const T* camera, const T* point, double ox, double oy, T* residuals) {
  T p[3];
  AngleAxisRotatePoint(camera, point, p);
  p[0] = p[0] + camera[3];
  p[1] = p[1] + camera[4];
  p[2] = p[2] + camera[5];
  const T xp = -p[0] / p[2];
  const T yp = -p[1] / p[2];
  const T& l1 = camera[7];
  const T& l2 = camera[8];
  const T r2 = xp * xp + yp * yp;
  const T distortion = 1.0 + r2 * (l1 + l2 * r2);
  const T& focal = camera[6];
  const T predicted_x = focal * distortion * xp;
  const T predicted_y = focal * distortion * yp;
  residuals[0] = predicted_x - ox;
  residuals[1] = predicted_y - oy;
}

// AutoDiffCostFunction<Snavely,2,9,3>::Evaluate (internal/autodiff.h:296-360):
// one pass with Jet<double,12>, camera parameters in dual slots 0..8, point in 9..11.
static void SnavelyAutoDiff(const double* cam, const double* pt, const double* obs, double* res,
                            double* jc, double* jp) {
  if (!jc && !jp) {
    Snavely<double>(cam, pt, obs[0], obs[1], res);
    return;
  }
  using J = Jet<12>;
  J c[9], p[3], r[2];
  for (int i = 0; i < 9; ++i) c[i] = J(cam[i], i);
  for (int i = 0; i < 3; ++i) p[i] = J(pt[i], 9 + i);
  Snavely<J>(c, p, obs[0], obs[1], r);
  for (int k = 0; k < 2; ++k) {
    res[k] = r[k].a;
    if (jc) for (int i = 0; i < 9; ++i) jc[k * 9 + i] = r[k].v[i];
    if (jp) for (int i = 0; i < 3; ++i) jp[k * 3 + i] = r[k].v[9 + i];
  }
}

// rotation.h:722-760
template <typename T>
inline void UnitQuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
  T uv0 = q[2] * pt[2] - q[3] * pt[1];
  T uv1 = q[3] * pt[0] - q[1] * pt[2];
  T uv2 = q[1] * pt[1] - q[2] * pt[0];
  uv0 = uv0 + uv0;
  uv1 = uv1 + uv1;
  uv2 = uv2 + uv2;
  result[0] = pt[0] + q[0] * uv0;
  result[1] = pt[1] + q[0] * uv1;
  result[2] = pt[2] + q[0] * uv2;
  result[0] = result[0] + (q[2] * uv2 - q[3] * uv1);
  result[1] = result[1] + (q[3] * uv0 - q[1] * uv2);
  result[2] = result[2] + (q[1] * uv1 - q[2] * uv0);
}
template <typename T>
inline void QuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
  using std::sqrt;
  const T scale = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const T unit[4] = {scale * q[0], scale * q[1], scale * q[2], scale * q[3]};
  UnitQuaternionRotatePoint(unit, pt, result);
}
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }

// examples/snavely_reprojection_error.h:111-170 (camera: quaternion 4, translation 3, focal, k1, k2)
template <typename T>
inline void SnavelyQuaternion(const T* camera, const T* point, double ox, double oy, T* residuals) {
  T p[3];
  QuaternionRotatePoint(camera, point, p);
  p[0] = p[0] + camera[4];
  p[1] = p[1] + camera[5];
  p[2] = p[2] + camera[6];
  const T xp = -p[0] / p[2];
  const T yp = -p[1] / p[2];
  const T& l1 = camera[8];
  const T& l2 = camera[9];
  const T r2 = xp * xp + yp * yp;
  const T distortion = 1.0 + r2 * (l1 + l2 * r2);
  const T& focal = camera[7];
  const T predicted_x = focal * distortion * xp;
  const T predicted_y = focal * distortion * yp;
  residuals[0] = predicted_x - ox;
  residuals[1] = predicted_y - oy;
}

// QuaternionManifold (manifold.cc:27-78), w first
static void QuaternionPlus(const double* x, const double* delta, double* x_plus_delta) {
  const double norm_delta = std::hypot(delta[0], delta[1], delta[2]);
  if (std::fpclassify(norm_delta) == FP_ZERO) {
    std::copy(x, x + 4, x_plus_delta);
    return;
  }
  const double sin_delta_by_delta = std::sin(norm_delta) / norm_delta;
  const double q[4] = {std::cos(norm_delta), sin_delta_by_delta * delta[0], sin_delta_by_delta * delta[1],
                       sin_delta_by_delta * delta[2]};
  x_plus_delta[0] = q[0] * x[0] - q[1] * x[1] - q[2] * x[2] - q[3] * x[3];
  x_plus_delta[1] = q[0] * x[1] + q[1] * x[0] + q[2] * x[3] - q[3] * x[2];
  x_plus_delta[2] = q[0] * x[2] - q[1] * x[3] + q[2] * x[0] + q[3] * x[1];
  x_plus_delta[3] = q[0] * x[3] + q[1] * x[2] - q[2] * x[1] + q[3] * x[0];
}
static void QuaternionPlusJacobian(const double* x, double* j /* 4x3 row-major */) {
  j[0] = -x[1]; j[1] = -x[2]; j[2] = -x[3];
  j[3] = x[0];  j[4] = x[3];  j[5] = -x[2];
  j[6] = -x[3]; j[7] = x[0];  j[8] = x[1];
  j[9] = x[2];  j[10] = -x[1]; j[11] = x[0];
}

// AutoDiffCostFunction<SnavelyReprojectionErrorWithQuaternions, 2, 10, 3> followed by the projection with
// the PlusJacobian of ProductManifold<QuaternionManifold, EuclideanManifold<6>> (residual_block.cc:136-159):
// jc is the 2x9 tangent-space block
static void SnavelyQuaternionAutoDiff(const double* cam, const double* pt, const double* obs, double* res, double* jc,
                                      double* jp) {
  if (!jc && !jp) {
    SnavelyQuaternion<double>(cam, pt, obs[0], obs[1], res);
    return;
  }
  using J = Jet<13>;
  J c[10], p[3], r[2];
  for (int i = 0; i < 10; ++i) c[i] = J(cam[i], i);
  for (int i = 0; i < 3; ++i) p[i] = J(pt[i], 10 + i);
  SnavelyQuaternion<J>(c, p, obs[0], obs[1], r);
  double pj[12];
  QuaternionPlusJacobian(cam, pj);
  for (int k = 0; k < 2; ++k) {
    res[k] = r[k].a;
    if (jc) {
      for (int t = 0; t < 3; ++t) {
        double sum = 0.0;
        for (int a = 0; a < 4; ++a) sum += r[k].v[a] * pj[a * 3 + t];
        jc[k * 9 + t] = sum;
      }
      for (int i = 0; i < 6; ++i) jc[k * 9 + 3 + i] = r[k].v[4 + i];
    }
    if (jp) for (int i = 0; i < 3; ++i) jp[k * 3 + i] = r[k].v[10 + i];
  }
}

}  // namespace orc

using namespace orc;

extern "C" {

void orc_snavely(const double* cam, const double* pt, const double* obs, double* res, double* jc, double* jp) {
  SnavelyAutoDiff(cam, pt, obs, res, jc, jp);
}

void orc_angle_axis_rotate_point(const double* aa, const double* pt, double* out) {
  AngleAxisRotatePoint<double>(aa, pt, out);
}

// AngleAxisToRotationMatrix (rotation.h:452-494), COLUMN-major R like the reference's default
// (ColumnMajorAdapter3x3, rotation.h:441-444).  Not on the solve path: it is the independent formula
// rotation_test.cc:1707-1800 checks AngleAxisRotatePoint against, restated for the same purpose.
void orc_angle_axis_to_rotation_matrix(const double* aa, double* R) {
  const double theta = std::hypot(aa[0], aa[1], aa[2]);
  auto at = [&](int r, int c) -> double& { return R[c * 3 + r]; };
  if (std::fpclassify(theta) != FP_ZERO) {
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double costheta = std::cos(theta), sintheta = std::sin(theta);
    at(0, 0) = costheta + wx * wx * (1.0 - costheta);
    at(1, 0) = wz * sintheta + wx * wy * (1.0 - costheta);
    at(2, 0) = -wy * sintheta + wx * wz * (1.0 - costheta);
    at(0, 1) = wx * wy * (1.0 - costheta) - wz * sintheta;
    at(1, 1) = costheta + wy * wy * (1.0 - costheta);
    at(2, 1) = wx * sintheta + wy * wz * (1.0 - costheta);
    at(0, 2) = wy * sintheta + wx * wz * (1.0 - costheta);
    at(1, 2) = -wx * sintheta + wy * wz * (1.0 - costheta);
    at(2, 2) = costheta + wz * wz * (1.0 - costheta);
  } else {  // first order Taylor expansion at zero
    at(0, 0) = 1.0; at(1, 0) = aa[2]; at(2, 0) = -aa[1];
    at(0, 1) = -aa[2]; at(1, 1) = 1.0; at(2, 1) = aa[0];
    at(0, 2) = aa[1]; at(1, 2) = -aa[0]; at(2, 2) = 1.0;
  }
}

// reorder_program.cc:256-338 with MinParameterBlock == point index (the only
// parameter block of a BAL residual that lies in the first elimination group)
void orc_bal_residual_order(int num_points, int64_t num_obs, const int32_t* point_index, int64_t* order) {
  std::vector<int64_t> offsets(num_points + 1, 0);
  for (int64_t i = 0; i < num_obs; ++i) offsets[point_index[i]]++;
  std::partial_sum(offsets.begin(), offsets.end(), offsets.begin());
  // each bucket is filled from its back to its front
  for (int64_t i = 0; i < num_obs; ++i) {
    const int bucket = point_index[i];
    offsets[bucket]--;
    order[offsets[bucket]] = i;
  }
}

// parameter_block_ordering.cc:50-83 + graph_algorithms.h:165-227.
// Hessian graph (parameter_block_ordering.cc:120-160): vertices = parameter
// blocks, an edge between every pair of blocks sharing a residual.  For BAL that
// is the bipartite camera-point graph.  Degree = number of distinct neighbours.
int orc_stable_schur_ordering(int C, int P, int64_t O, const int32_t* cam, const int32_t* pt, int32_t* ordering) {
  const int n = C + P;
  std::vector<std::vector<int32_t>> nbr(n);
  for (int64_t i = 0; i < O; ++i) {
    nbr[cam[i]].push_back(C + pt[i]);
    nbr[C + pt[i]].push_back(cam[i]);
  }
  for (auto& v : nbr) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
  // CreateHessianGraph adds every non-constant parameter block as a vertex
  // (parameter_block_ordering.cc:126-135), so isolated blocks are kept; the
  // queue starts in program order (cameras, then points).
  std::vector<int32_t> queue(n);
  std::iota(queue.begin(), queue.end(), 0);
  std::stable_sort(queue.begin(), queue.end(), [&](int32_t a, int32_t b) { return nbr[a].size() < nbr[b].size(); });
  std::vector<char> color(n, 0);  // 0 white, 1 grey, 2 black
  int k = 0;
  for (int32_t v : queue) {
    if (color[v] != 0) continue;
    ordering[k++] = v;
    color[v] = 2;
    for (int32_t u : nbr[v]) color[u] = 1;
  }
  const int independent_set_size = k;
  for (int32_t v : queue) if (color[v] != 2) ordering[k++] = v;
  return independent_set_size;
}

// block_jacobian_writer.cc:68-263 for residual blocks {point (e), camera (f)}:
// E cells packed first in row order, then F cells.
void orc_bal_structure(int C, int P, int64_t O, const int32_t* cam, const int32_t* pt,
                       const int64_t* order, cx_block* row_blocks, cx_block* col_blocks,
                       int32_t* row_cell_begin, cx_cell* cells) {
  for (int j = 0; j < P; ++j) { col_blocks[j].size = 3; col_blocks[j].position = 3 * j; }
  for (int i = 0; i < C; ++i) { col_blocks[P + i].size = 9; col_blocks[P + i].position = 3 * P + 9 * i; }
  const int64_t f_block_pos = 6 * O;
  for (int64_t k = 0; k < O; ++k) {
    const int64_t i = order[k];
    row_blocks[k].size = 2;
    row_blocks[k].position = int32_t(2 * k);
    row_cell_begin[k] = int32_t(2 * k);
    cells[2 * k].block_id = pt[i];
    cells[2 * k].position = int32_t(6 * k);
    cells[2 * k + 1].block_id = P + cam[i];
    cells[2 * k + 1].position = int32_t(f_block_pos + 18 * k);
  }
  row_cell_begin[O] = int32_t(2 * O);
}

// program_evaluator.h:137-304 (no loss function, no manifolds: bundle_adjuster
// defaults, bundle_adjuster.cc:112,327-328)
// LossFunction::Evaluate for the built-in losses (loss_function.cc:46-144, constructors
// include/ceres/loss_function.h:176-292); type codes are cx_loss_type of include/cxschur.h.
void orc_loss_evaluate(int type, double a, double b, double s, double* rho) {
  const double kMin = std::numeric_limits<double>::min();
  switch (type) {
    case CX_LOSS_HUBER: {
      const double b_ = a * a;
      if (s > b_) {
        const double r = std::sqrt(s);
        rho[0] = 2.0 * a * r - b_;
        rho[1] = std::max(kMin, a / r);
        rho[2] = -rho[1] / (2.0 * s);
      } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
      }
      return;
    }
    case CX_LOSS_SOFT_L_ONE: {
      const double b_ = a * a, c_ = 1 / b_;
      const double sum = 1.0 + s * c_, tmp = std::sqrt(sum);
      rho[0] = 2.0 * b_ * (tmp - 1.0);
      rho[1] = std::max(kMin, 1.0 / tmp);
      rho[2] = -(c_ * rho[1]) / (2.0 * sum);
      return;
    }
    case CX_LOSS_CAUCHY: {
      const double b_ = a * a, c_ = 1 / b_;
      const double sum = 1.0 + s * c_, inv = 1.0 / sum;
      rho[0] = b_ * std::log(sum);
      rho[1] = std::max(kMin, inv);
      rho[2] = -c_ * (inv * inv);
      return;
    }
    case CX_LOSS_ARCTAN: {
      const double b_ = 1 / (a * a);
      const double sum = 1 + s * s * b_, inv = 1 / sum;
      rho[0] = a * std::atan2(s, a);
      rho[1] = std::max(kMin, inv);
      rho[2] = -2.0 * s * b_ * (inv * inv);
      return;
    }
    case CX_LOSS_TOLERANT: {
      const double c_ = b * std::log(1.0 + std::exp(-a / b));
      const double x = (s - a) / b;
      if (x > 36.7) {
        rho[0] = s - a - c_; rho[1] = 1.0; rho[2] = 0.0;
      } else {
        const double e_x = std::exp(x);
        rho[0] = b * std::log(1.0 + e_x) - c_;
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

// Corrector (corrector.cc:41-155): jacobian (may be NULL) is corrected first, with the
// uncorrected residuals, then the residuals (residual_block.cc:176-196).
void orc_corrector_apply(double sq_norm, const double* rho, int num_rows, int num_cols,
                         double* residuals, double* jacobian) {
  const double sqrt_rho1 = std::sqrt(rho[1]);
  double residual_scaling, alpha_sq_norm;
  if (sq_norm == 0.0 || rho[2] <= 0.0) {
    residual_scaling = sqrt_rho1;
    alpha_sq_norm = 0.0;
  } else {
    const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
    const double alpha = 1.0 - std::sqrt(D);
    residual_scaling = sqrt_rho1 / (1 - alpha);
    alpha_sq_norm = alpha / sq_norm;
  }
  if (jacobian) {
    if (alpha_sq_norm == 0.0) {
      for (int i = 0; i < num_rows * num_cols; ++i) jacobian[i] *= sqrt_rho1;
    } else {
      for (int c = 0; c < num_cols; ++c) {
        double r_transpose_j = 0.0;
        for (int r = 0; r < num_rows; ++r) r_transpose_j += jacobian[r * num_cols + c] * residuals[r];
        for (int r = 0; r < num_rows; ++r)
          jacobian[r * num_cols + c] =
              sqrt_rho1 * (jacobian[r * num_cols + c] - alpha_sq_norm * residuals[r] * r_transpose_j);
      }
    }
  }
  for (int r = 0; r < num_rows; ++r) residuals[r] *= residual_scaling;
}

// rotation.h:315-388
void orc_angle_axis_to_quaternion(const double* aa, double* q) {
  const double theta = std::hypot(aa[0], aa[1], aa[2]);
  if (std::fpclassify(theta) != FP_ZERO) {
    const double half_theta = theta * 0.5;
    const double k = std::sin(half_theta) / theta;
    q[0] = std::cos(half_theta); q[1] = aa[0] * k; q[2] = aa[1] * k; q[3] = aa[2] * k;
  } else {
    q[0] = 1.0; q[1] = aa[0] * 0.5; q[2] = aa[1] * 0.5; q[3] = aa[2] * 0.5;
  }
}
void orc_quaternion_to_angle_axis(const double* q, double* aa) {
  const double sin_theta = std::hypot(q[1], q[2], q[3]);
  if (std::fpclassify(sin_theta) != FP_ZERO) {
    const double cos_theta = q[0];
    const double two_theta = 2.0 * ((cos_theta < 0.0) ? std::atan2(-sin_theta, -cos_theta) : std::atan2(sin_theta, cos_theta));
    const double k = two_theta / sin_theta;
    aa[0] = q[1] * k; aa[1] = q[2] * k; aa[2] = q[3] * k;
  } else {
    aa[0] = q[1] * 2.0; aa[1] = q[2] * 2.0; aa[2] = q[3] * 2.0;
  }
}
void orc_quaternion_plus(const double* x, const double* delta, double* x_plus_delta) { QuaternionPlus(x, delta, x_plus_delta); }
void orc_quaternion_plus_jacobian(const double* x, double* jacobian) { QuaternionPlusJacobian(x, jacobian); }
void orc_snavely_quaternion(const double* cam10, const double* pt, const double* obs, double* res, double* jc9, double* jp) {
  SnavelyQuaternionAutoDiff(cam10, pt, obs, res, jc9, jp);
}

// Evaluator::Plus for the BAL program: points Euclidean; cameras Euclidean (camera_model 0: 9 parameters) or
// ProductManifold<QuaternionManifold, EuclideanManifold<6>> (camera_model 1: 10 parameters, 9 tangent)
void orc_bal_plus(int C, int P, int camera_model, const double* x, const double* delta, double* out) {
  for (int i = 0; i < 3 * P; ++i) out[i] = x[i] + delta[i];
  if (camera_model == CX_CAMERA_ANGLE_AXIS) {
    for (int i = 0; i < 9 * C; ++i) out[3 * P + i] = x[3 * P + i] + delta[3 * P + i];
    return;
  }
  for (int c = 0; c < C; ++c) {
    const double* xc = x + 3 * P + 10 * c;
    const double* dc = delta + 3 * P + 9 * c;
    double* oc = out + 3 * P + 10 * c;
    QuaternionPlus(xc, dc, oc);
    for (int i = 0; i < 6; ++i) oc[4 + i] = xc[4 + i] + dc[3 + i];
  }
}

void orc_bal_evaluate_model(const cx_block_structure* s, int C, int P, int64_t O, const int32_t* cam,
                            const int32_t* pt, const double* observations, const int64_t* order,
                            const double* state, int camera_model, int loss_type, double loss_a, double loss_b,
                            double* cost, double* residuals, double* gradient, double* values) {
  (void)s;
  const int cam_size = camera_model == CX_CAMERA_ANGLE_AXIS ? 9 : 10;
  const int threads = orc_get_num_threads();
  double total = 0.0;
  const int num_cols = 3 * P + 9 * C;
  std::vector<std::vector<double>> grads;
  if (gradient) grads.assign(threads, std::vector<double>(num_cols, 0.0));
#pragma omp parallel for schedule(static) num_threads(threads) reduction(+ : total)
  for (int64_t k = 0; k < O; ++k) {
    const int64_t i = order[k];
    const double* camera = state + 3 * P + cam_size * cam[i];
    const double* point = state + 3 * pt[i];
    double r[2], jc[18], jp[6];
    const bool need_j = values || gradient;
    if (camera_model == CX_CAMERA_ANGLE_AXIS)
      SnavelyAutoDiff(camera, point, observations + 2 * i, r, need_j ? jc : nullptr, need_j ? jp : nullptr);
    else
      SnavelyQuaternionAutoDiff(camera, point, observations + 2 * i, r, need_j ? jc : nullptr, need_j ? jp : nullptr);
    const double sq = r[0] * r[0] + r[1] * r[1];
    if (loss_type == CX_LOSS_NONE) {
      total += 0.5 * sq;                       // residual_block.cc:160-163
    } else {
      double rho[3];
      orc_loss_evaluate(loss_type, loss_a, loss_b, sq, rho);
      total += 0.5 * rho[0];                   // residual_block.cc:165-167
      if (need_j) {
        double r_copy[2] = {r[0], r[1]};
        orc_corrector_apply(sq, rho, 2, 9, r_copy, jc);
        r_copy[0] = r[0]; r_copy[1] = r[1];
        orc_corrector_apply(sq, rho, 2, 3, r_copy, jp);
      }
      orc_corrector_apply(sq, rho, 2, 0, r, nullptr);
    }
    if (residuals) { residuals[2 * k] = r[0]; residuals[2 * k + 1] = r[1]; }
    if (values) {
      std::copy(jp, jp + 6, values + 6 * k);
      std::copy(jc, jc + 18, values + 6 * O + 18 * k);
    }
    if (gradient) {
      double* g = grads[omp_get_thread_num()].data();
      MatTVec(jp, 2, 3, r, g + 3 * pt[i], 1);
      MatTVec(jc, 2, 9, r, g + 3 * P + 9 * cam[i], 1);
    }
  }
  if (cost) *cost = total;
  if (gradient) {
    std::fill(gradient, gradient + num_cols, 0.0);
    for (auto& g : grads) for (int i = 0; i < num_cols; ++i) gradient[i] += g[i];
  }
}

void orc_bal_evaluate_robust(const cx_block_structure* s, int C, int P, int64_t O, const int32_t* cam,
                             const int32_t* pt, const double* observations, const int64_t* order,
                             const double* state, int loss_type, double loss_a, double loss_b,
                             double* cost, double* residuals, double* gradient, double* values) {
  orc_bal_evaluate_model(s, C, P, O, cam, pt, observations, order, state, CX_CAMERA_ANGLE_AXIS, loss_type, loss_a, loss_b,
                         cost, residuals, gradient, values);
}

void orc_bal_evaluate(const cx_block_structure* s, int C, int P, int64_t O, const int32_t* cam,
                      const int32_t* pt, const double* observations, const int64_t* order,
                      const double* state, double* cost, double* residuals, double* gradient,
                      double* values) {
  orc_bal_evaluate_robust(s, C, P, O, cam, pt, observations, order, state, CX_LOSS_NONE, 0.0, 0.0, cost,
                          residuals, gradient, values);
}

}  // extern "C"
