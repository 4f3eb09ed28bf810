// Bundle-adjustment Evaluator on gfx950: residuals, cost, gradient and the block
// sparse Jacobian of SnavelyReprojectionError written straight into the device
// matrix (ProgramEvaluator::Evaluate, program_evaluator.h:137-304;
// BlockEvaluatePreparer::Prepare, block_evaluate_preparer.cc:50-78;
// examples/snavely_reprojection_error.h:53-104; rotation.h:792-857).
//
// One thread per residual block runs the reference's forward-mode arithmetic
// (Jet<double,12>: camera in dual slots 0..8, point in 9..11, jet.h formulas), the
// 2x3 and 2x9 cells are transposed through LDS and stored with coalesced 16-byte
// writes into the reference cell layout.  Algorithmic traffic: 208 B written and
// 24 B read per residual block plus the parameter gathers (L2 resident).
#include <algorithm>
#include <cstdlib>
#include <numeric>

#include "cx_internal.h"
#include "cx_kernels.h"


namespace {

struct LossParams {
  int32_t type;
  double a, b;
};

// LossFunction::Evaluate of the built-in losses (loss_function.cc:46-144)
__device__ __forceinline__ void loss_evaluate(const LossParams& L, double s, double rho[3]) {
  const double kMin = 2.2250738585072014e-308;  // std::numeric_limits<double>::min()
  rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  switch (L.type) {
    case CX_LOSS_HUBER: {
      const double b = L.a * L.a;
      if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * L.a * r - b;
        rho[1] = fmax(kMin, L.a / r);
        rho[2] = -rho[1] / (2.0 * s);
      }
      break;
    }
    case CX_LOSS_SOFT_L_ONE: {
      const double b = L.a * L.a, c = 1 / b;
      const double sum = 1.0 + s * c, tmp = sqrt(sum);
      rho[0] = 2.0 * b * (tmp - 1.0);
      rho[1] = fmax(kMin, 1.0 / tmp);
      rho[2] = -(c * rho[1]) / (2.0 * sum);
      break;
    }
    case CX_LOSS_CAUCHY: {
      const double b = L.a * L.a, c = 1 / b;
      const double sum = 1.0 + s * c, inv = 1.0 / sum;
      rho[0] = b * log(sum);
      rho[1] = fmax(kMin, inv);
      rho[2] = -c * (inv * inv);
      break;
    }
    case CX_LOSS_ARCTAN: {
      const double b = 1 / (L.a * L.a);
      const double sum = 1 + s * s * b, inv = 1 / sum;
      rho[0] = L.a * atan2(s, L.a);
      rho[1] = fmax(kMin, inv);
      rho[2] = -2.0 * s * b * (inv * inv);
      break;
    }
    case CX_LOSS_TOLERANT: {
      const double c = L.b * log(1.0 + exp(-L.a / L.b));
      const double x = (s - L.a) / L.b;
      if (x > 36.7) {
        rho[0] = s - L.a - c;
      } else {
        const double e_x = exp(x);
        rho[0] = L.b * log(1.0 + e_x) - c;
        rho[1] = fmax(kMin, e_x / (1.0 + e_x));
        rho[2] = 0.5 / (L.b * (1.0 + cosh(x)));
      }
      break;
    }
    case CX_LOSS_TUKEY: {
      const double a2 = L.a * L.a;
      if (s <= a2) {
        const double value = 1.0 - s / a2, value_sq = value * value;
        rho[0] = a2 / 3.0 * (1.0 - value_sq * value);
        rho[1] = value_sq;
        rho[2] = -2.0 / a2 * value;
      } else {
        rho[0] = a2 / 3.0; rho[1] = 0.0; rho[2] = 0.0;
      }
      break;
    }
    default: break;
  }
}

// Corrector::CorrectJacobian for a 2 x N row-major block (corrector.cc:118-153)
template <int N>
__device__ __forceinline__ void correct_jacobian(double sqrt_rho1, double alpha_sq_norm, double r0, double r1, double* j) {
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const double rtj = j[c] * r0 + j[N + c] * r1;
    j[c] = sqrt_rho1 * (j[c] - alpha_sq_norm * r0 * rtj);
    j[N + c] = sqrt_rho1 * (j[N + c] - alpha_sq_norm * r1 * rtj);
  }
}

template <int N>
struct Jet {
  double a;
  double v[N];
};
template <int N> __device__ __forceinline__ Jet<N> jconst(double s) { Jet<N> r; r.a = s;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = 0.0; return r; }
template <int N> __device__ __forceinline__ Jet<N> jvar(double s, int k) { Jet<N> r = jconst<N>(s); r.v[k] = 1.0; return r; }
template <int N> __device__ __forceinline__ Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a + g.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i]; return r; }
template <int N> __device__ __forceinline__ Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a - g.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i]; return r; }
template <int N> __device__ __forceinline__ Jet<N> operator-(const Jet<N>& f) { Jet<N> r; r.a = -f.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }
template <int N> __device__ __forceinline__ Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> r = f; r.a = f.a + s; return r; }
template <int N> __device__ __forceinline__ Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> r = f; r.a = f.a - s; return r; }
template <int N> __device__ __forceinline__ Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> r; r.a = s - f.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }
// jet.h:349-352
template <int N> __device__ __forceinline__ Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a * g.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
// jet.h:367-379
template <int N> __device__ __forceinline__ Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  const double gi = 1.0 / g.a, q = f.a * gi;
  Jet<N> r; r.a = q;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - q * g.v[i]) * gi;
  return r;
}
// jet.h:382-386
template <int N> __device__ __forceinline__ Jet<N> operator/(double s, const Jet<N>& g) {
  const double m = -s / (g.a * g.a);
  Jet<N> r; r.a = s / g.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = g.v[i] * m;
  return r;
}
template <int N> __device__ __forceinline__ Jet<N> jcos(const Jet<N>& f) { Jet<N> r; r.a = cos(f.a); const double m = -sin(f.a);
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = m * f.v[i]; return r; }
template <int N> __device__ __forceinline__ Jet<N> jsin(const Jet<N>& f) { Jet<N> r; r.a = sin(f.a); const double m = cos(f.a);
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = m * f.v[i]; return r; }
// jet.h:733-748
template <int N> __device__ __forceinline__ Jet<N> jhypot3(const Jet<N>& x, const Jet<N>& y, const Jet<N>& z) {
  const double t = norm3d(x.a, y.a, z.a);
  const double cx_ = x.a / t, cy_ = y.a / t, cz_ = z.a / t;
  Jet<N> r; r.a = t;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = cx_ * x.v[i] + cy_ * y.v[i] + cz_ * z.v[i];
  return r;
}

// value-only path (plain doubles, true divisions) used when no Jacobian is asked for
__device__ __forceinline__ void snavely_value(const double* cam, const double* pt, double ox, double oy, double& r0, double& r1) {
  const double theta = norm3d(cam[0], cam[1], cam[2]);
  double p[3];
  if (theta != 0.0) {
    const double ct = cos(theta), st = sin(theta), ti = 1.0 / theta;
    const double w[3] = {cam[0] * ti, cam[1] * ti, cam[2] * ti};
    const double wx[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const double tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - ct);
    p[0] = pt[0] * ct + wx[0] * st + w[0] * tmp;
    p[1] = pt[1] * ct + wx[1] * st + w[1] * tmp;
    p[2] = pt[2] * ct + wx[2] * st + w[2] * tmp;
  } else {
    p[0] = pt[0] + (cam[1] * pt[2] - cam[2] * pt[1]);
    p[1] = pt[1] + (cam[2] * pt[0] - cam[0] * pt[2]);
    p[2] = pt[2] + (cam[0] * pt[1] - cam[1] * pt[0]);
  }
  p[0] += cam[3]; p[1] += cam[4]; p[2] += cam[5];
  const double xp = -p[0] / p[2], yp = -p[1] / p[2];
  const double r2 = xp * xp + yp * yp;
  const double dist = 1.0 + r2 * (cam[7] + cam[8] * r2);
  r0 = cam[6] * dist * xp - ox;
  r1 = cam[6] * dist * yp - oy;
}

// dual-number path: residual and the 2x9 / 2x3 Jacobian cells
__device__ __forceinline__ void snavely_jet(const double* camv, const double* ptv, double ox, double oy, double& r0, double& r1,
                                            double (&jc)[18], double (&jp)[6]) {
  using J = Jet<12>;
  J c[9], x[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) c[i] = jvar<12>(camv[i], i);
#pragma unroll
  for (int i = 0; i < 3; ++i) x[i] = jvar<12>(ptv[i], 9 + i);
  J p[3];
  const J theta = jhypot3(c[0], c[1], c[2]);
  if (theta.a != 0.0) {
    const J ct = jcos(theta), st = jsin(theta), ti = 1.0 / theta;
    const J w[3] = {c[0] * ti, c[1] * ti, c[2] * ti};
    const J wx[3] = {w[1] * x[2] - w[2] * x[1], w[2] * x[0] - w[0] * x[2], w[0] * x[1] - w[1] * x[0]};
    const J tmp = (w[0] * x[0] + w[1] * x[1] + w[2] * x[2]) * (1.0 - ct);
    p[0] = x[0] * ct + wx[0] * st + w[0] * tmp;
    p[1] = x[1] * ct + wx[1] * st + w[1] * tmp;
    p[2] = x[2] * ct + wx[2] * st + w[2] * tmp;
  } else {
    p[0] = x[0] + (c[1] * x[2] - c[2] * x[1]);
    p[1] = x[1] + (c[2] * x[0] - c[0] * x[2]);
    p[2] = x[2] + (c[0] * x[1] - c[1] * x[0]);
  }
  p[0] = p[0] + c[3]; p[1] = p[1] + c[4]; p[2] = p[2] + c[5];
  const J xp = -p[0] / p[2], yp = -p[1] / p[2];
  const J r2 = xp * xp + yp * yp;
  const J dist = 1.0 + r2 * (c[7] + c[8] * r2);
  const J px = c[6] * dist * xp, py = c[6] * dist * yp;
  r0 = px.a - ox;
  r1 = py.a - oy;
#pragma unroll
  for (int i = 0; i < 9; ++i) { jc[i] = px.v[i]; jc[9 + i] = py.v[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { jp[i] = px.v[9 + i]; jp[3 + i] = py.v[9 + i]; }
}

// The same residual and Jacobian cells in closed form (round 2; the default with the Jacobian, CX_EVAL_VARIANT=1 selects the
// dual numbers above).  The dual-number path is 13 x the scalar work (1 891 fp64 instructions in the kernel's ISA against
// 1 380 with this form, most of the rest being sin / cos) and needs 172 VGPRs inside the pipelined loop of k_bal_evaluate
// against 133: three workgroups per CU instead of two, 1.55 against 1.91 ms on the Final shape (in round 1's one-tile kernel,
// bound by its gathers, the two ran the same 2.0 ms).  All parity tests are green with either; the dual numbers follow the
// reference's autodiff operation by operation.  Differentiating by hand block by block:
//   p = R(w) X + t;  dp/dX = R = c I + s [k]x + (1 - c) k k',  k = w / |w|;
//   dp/dw = -[R X]x J_l(w),  J_l = (s / th) I + ((1 - c) / th) [k]x + (1 - s / th) k k'   (left Jacobian of SO(3));
//   |w| = 0 exactly (the reference's first-order branch, rotation.h:836-856): R X = X + w x X, dp/dw = -[X]x, dp/dX = I + [w]x;
//   (xp, yp) = -(p0, p1) / p2;  r = f d (xp, yp) - obs,  d = 1 + r2 (k1 + k2 r2);  chain rule through the 2 x 2 and 2 x 3 blocks.
// Same function, other rounding: against the oracle's dual numbers 3e-14 relative for |w| ~ 1, 1e-13 at 1e-3, 1e-16 / |w|
// below (the dual numbers divide by |w| where this form has no cancellation) -- inside the 1e-11 the parity tests state.
__device__ __forceinline__ void snavely_closed_form(const double* cam, const double* X, double ox, double oy, double& r0, double& r1,
                                                    double (&jc)[18], double (&jp)[6]) {
  const double w0 = cam[0], w1 = cam[1], w2 = cam[2];
  const double th = norm3d(w0, w1, w2);
  double q[3], R[9], M[9];  // q = R X; R = dq/dX; M = dq/dw (row-major 3 x 3)
  if (th != 0.0) {
    const double c = cos(th), s = sin(th), ti = 1.0 / th;
    const double k0 = w0 * ti, k1 = w1 * ti, k2 = w2 * ti;
    const double kx0 = k1 * X[2] - k2 * X[1], kx1 = k2 * X[0] - k0 * X[2], kx2 = k0 * X[1] - k1 * X[0];
    const double omc = 1.0 - c;
    const double tmp = (k0 * X[0] + k1 * X[1] + k2 * X[2]) * omc;
    q[0] = X[0] * c + kx0 * s + k0 * tmp;
    q[1] = X[1] * c + kx1 * s + k1 * tmp;
    q[2] = X[2] * c + kx2 * s + k2 * tmp;
    R[0] = c + omc * k0 * k0;       R[1] = omc * k0 * k1 - s * k2;  R[2] = omc * k0 * k2 + s * k1;
    R[3] = omc * k1 * k0 + s * k2;  R[4] = c + omc * k1 * k1;       R[5] = omc * k1 * k2 - s * k0;
    R[6] = omc * k2 * k0 - s * k1;  R[7] = omc * k2 * k1 + s * k0;  R[8] = c + omc * k2 * k2;
    const double al = s * ti;
    const double sh = sin(0.5 * th);
    const double be = 2.0 * sh * sh * ti;  // (1 - cos th) / th without the cancellation
    const double ga = 1.0 - al;
    double J[9];                            // J_l
    J[0] = al + ga * k0 * k0;       J[1] = ga * k0 * k1 - be * k2;  J[2] = ga * k0 * k2 + be * k1;
    J[3] = ga * k1 * k0 + be * k2;  J[4] = al + ga * k1 * k1;       J[5] = ga * k1 * k2 - be * k0;
    J[6] = ga * k2 * k0 - be * k1;  J[7] = ga * k2 * k1 + be * k0;  J[8] = al + ga * k2 * k2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {           // M = -[q]x J_l
      M[j] = q[2] * J[3 + j] - q[1] * J[6 + j];
      M[3 + j] = q[0] * J[6 + j] - q[2] * J[j];
      M[6 + j] = q[1] * J[j] - q[0] * J[3 + j];
    }
  } else {
    q[0] = X[0] + (w1 * X[2] - w2 * X[1]);
    q[1] = X[1] + (w2 * X[0] - w0 * X[2]);
    q[2] = X[2] + (w0 * X[1] - w1 * X[0]);
    R[0] = 1.0; R[1] = -w2; R[2] = w1;
    R[3] = w2;  R[4] = 1.0; R[5] = -w0;
    R[6] = -w1; R[7] = w0;  R[8] = 1.0;
    M[0] = 0.0;   M[1] = X[2];  M[2] = -X[1];
    M[3] = -X[2]; M[4] = 0.0;   M[5] = X[0];
    M[6] = X[1];  M[7] = -X[0]; M[8] = 0.0;
  }
  const double p0 = q[0] + cam[3], p1 = q[1] + cam[4], p2 = q[2] + cam[5];
  const double ip2 = 1.0 / p2;               // the dual-number division: multiply by the reciprocal (jet.h:367-379)
  const double xp = -p0 * ip2, yp = -p1 * ip2;
  const double f = cam[6], l1 = cam[7], l2 = cam[8];
  const double r2 = xp * xp + yp * yp;
  const double dist = 1.0 + r2 * (l1 + l2 * r2);
  const double fd = f * dist;
  r0 = fd * xp - ox;
  r1 = fd * yp - oy;
  // d(px, py) / d(xp, yp)
  const double dd = 2.0 * f * (l1 + 2.0 * l2 * r2);   // f * d dist / d r2 * 2
  const double a00 = fd + dd * xp * xp, a01 = dd * xp * yp, a11 = fd + dd * yp * yp;
  // times d(xp, yp) / dp = [[-ip2, 0, -xp ip2], [0, -ip2, -yp ip2]]
  double Jp[6];
  Jp[0] = -a00 * ip2; Jp[1] = -a01 * ip2; Jp[2] = -(a00 * xp + a01 * yp) * ip2;
  Jp[3] = -a01 * ip2; Jp[4] = -a11 * ip2; Jp[5] = -(a01 * xp + a11 * yp) * ip2;
#pragma unroll
  for (int row = 0; row < 2; ++row) {
    const double* g = Jp + 3 * row;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      jc[9 * row + j] = g[0] * M[j] + g[1] * M[3 + j] + g[2] * M[6 + j];
      jc[9 * row + 3 + j] = g[j];
      jp[3 * row + j] = g[0] * R[j] + g[1] * R[3 + j] + g[2] * R[6 + j];
    }
  }
  jc[6] = dist * xp;          jc[15] = dist * yp;
  jc[7] = f * xp * r2;        jc[16] = f * yp * r2;
  jc[8] = f * xp * r2 * r2;   jc[17] = f * yp * r2 * r2;
}

// jet.h:483-487
template <int N> __device__ __forceinline__ Jet<N> jsqrt(const Jet<N>& f) {
  const double t = sqrt(f.a);
  const double m = 1.0 / (2.0 * t);
  Jet<N> r; r.a = t;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * m;
  return r;
}
template <int N> __device__ __forceinline__ Jet<N> operator*(double s, const Jet<N>& f) { Jet<N> r; r.a = s * f.a;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = s * f.v[i]; return r; }

// ---- quaternion cameras: SnavelyReprojectionErrorWithQuaternions (snavely_reprojection_error.h:111-170) with
// QuaternionRotatePoint / UnitQuaternionRotatePoint (rotation.h:722-760).  T is double or Jet<13>.
template <typename T>
__device__ __forceinline__ void unit_quaternion_rotate(const T (&q)[4], const T (&pt)[3], T (&result)[3]) {
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

__device__ __forceinline__ void snavely_value_quat(const double* cam, const double* ptv, double ox, double oy, double& r0, double& r1) {
  const double scale = 1.0 / sqrt(cam[0] * cam[0] + cam[1] * cam[1] + cam[2] * cam[2] + cam[3] * cam[3]);
  const double unit[4] = {scale * cam[0], scale * cam[1], scale * cam[2], scale * cam[3]};
  const double pt[3] = {ptv[0], ptv[1], ptv[2]};
  double p[3];
  unit_quaternion_rotate(unit, pt, p);
  p[0] += cam[4]; p[1] += cam[5]; p[2] += cam[6];
  const double xp = -p[0] / p[2], yp = -p[1] / p[2];
  const double r2 = xp * xp + yp * yp;
  const double dist = 1.0 + r2 * (cam[8] + cam[9] * r2);
  r0 = cam[7] * dist * xp - ox;
  r1 = cam[7] * dist * yp - oy;
}

// Dual numbers over the 10 + 3 ambient parameters, then the projection with the PlusJacobian of
// ProductManifold<QuaternionManifold, EuclideanManifold<6>> (residual_block.cc:136-159, manifold.cc:62-78):
// jc is the 2x9 tangent-space cell the static <2,3,9> layout stores.
__device__ __forceinline__ void snavely_jet_quat(const double* camv, const double* ptv, double ox, double oy, double& r0,
                                                 double& r1, double (&jc)[18], double (&jp)[6]) {
  using J = Jet<13>;
  J c[10], x[3];
#pragma unroll
  for (int i = 0; i < 10; ++i) c[i] = jvar<13>(camv[i], i);
#pragma unroll
  for (int i = 0; i < 3; ++i) x[i] = jvar<13>(ptv[i], 10 + i);
  const J scale = 1.0 / jsqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]);
  const J unit[4] = {scale * c[0], scale * c[1], scale * c[2], scale * c[3]};
  J p[3];
  unit_quaternion_rotate(unit, x, p);
  p[0] = p[0] + c[4]; p[1] = p[1] + c[5]; p[2] = p[2] + c[6];
  const J xp = -p[0] / p[2], yp = -p[1] / p[2];
  const J r2 = xp * xp + yp * yp;
  const J dist = 1.0 + r2 * (c[8] + c[9] * r2);
  const J px = c[7] * dist * xp, py = c[7] * dist * yp;
  r0 = px.a - ox;
  r1 = py.a - oy;
  // 4x3 PlusJacobian of the quaternion (w x y z): rows {-x -y -z; w z -y; -z w x; y -x w}
  const double w = camv[0], qx = camv[1], qy = camv[2], qz = camv[3];
  jc[0] = -px.v[0] * qx + px.v[1] * w - px.v[2] * qz + px.v[3] * qy;
  jc[1] = -px.v[0] * qy + px.v[1] * qz + px.v[2] * w - px.v[3] * qx;
  jc[2] = -px.v[0] * qz - px.v[1] * qy + px.v[2] * qx + px.v[3] * w;
  jc[9] = -py.v[0] * qx + py.v[1] * w - py.v[2] * qz + py.v[3] * qy;
  jc[10] = -py.v[0] * qy + py.v[1] * qz + py.v[2] * w - py.v[3] * qx;
  jc[11] = -py.v[0] * qz - py.v[1] * qy + py.v[2] * qx + py.v[3] * w;
#pragma unroll
  for (int i = 0; i < 6; ++i) { jc[3 + i] = px.v[4 + i]; jc[12 + i] = py.v[4 + i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { jp[i] = px.v[10 + i]; jp[3 + i] = py.v[10 + i]; }
}

// Tile k of this workgroup under the XCD-aware map of a grid that may be smaller than the number of tiles
// (gridDim.x a multiple of 8): XCD x = blockIdx % 8 owns the contiguous tiles [x * per, (x + 1) * per), its
// workgroups take them round-robin.  With gridDim.x >= 8 * per every workgroup has exactly one tile (xcd_segment).
__device__ __forceinline__ int eval_tile(int k, int num_tiles) {
  const int per = (num_tiles + 7) >> 3;
  const int j = int(blockIdx.x >> 3) + k * int(gridDim.x >> 3);
  const int t = int(blockIdx.x & 7) * per + j;
  return (j < per && t < num_tiles) ? t : -1;
}

// One residual block per thread, kBlock per tile; 232 bytes per residual block, 208 of them written.
// Round 2, what bounds it (Final shape, tools/eval_ab.py, profiles/r02_evaluator_probes.json).  Round 1's kernel (one tile
// per workgroup, dual numbers, every lane gathering its own camera) took 1.9 ms.  Probes of that kernel: the arithmetic
// alone 1.07 ms (0.59 ms in closed form), the stores alone 1.02 ms (5.5 TB/s), every load and store WITHOUT the
// arithmetic 1.8 ms -- it is bound by its memory requests, and more by their number than by their bytes: a lane-per-camera
// gather is 64 cache-line requests per load instruction, 340 of the kernel's 450 requests per wavefront.  Hence
//   * the cooperative gather (gather_by_row): 1.8 -> 1.5 ms for the loads and stores alone;
//   * the closed-form Jacobian (snavely_closed_form) as the default: 133 instead of 172 VGPRs, three workgroups per CU;
//   * a PERSISTENT, SOFTWARE-PIPELINED loop: a workgroup walks over its tiles and issues the gather of tile t + 1 and the
//     index load of tile t + 2 before the arithmetic of tile t, and waits for them after the arithmetic, before it issues
//     the stores of tile t (a CU's loads queue behind its stores): 1.73 -> 1.55 ms.
// 1.9 -> 1.55 ms with the Jacobian, 0.53 -> 0.46 ms for values and residuals.
// VARIANT (A/B, CX_EVAL_VARIANT): bit 0 dual numbers (Jet<12>) instead of the closed-form Jacobian, bit 2 plain instead of
// cooperative camera gather, bit 1 probe: every load and store, no arithmetic.
// SCALED (round 3): the Jacobi scaling of TrustRegionMinimizer (trust_region_minimizer.cc:263-279) folded in.  The
// scaling vector is computed once, at iteration 0, and ScaleColumns re-applies it after EVERY evaluation -- a second pass
// over all of J per LM iteration (416 B per residual block, 3.3 ms on the Final shape).  With col_scale set the kernel
// multiplies the blocks by the column scales in registers, after the Corrector and before they are stored: the values are
// bit for bit those of evaluate -> ScaleColumns (one rounding per product either way), and the camera-major copy is
// written in the same pass.
template <bool WITH_J, int MODEL, int VARIANT = 0, bool SCALED = false>
__global__ __launch_bounds__(kBlock, !WITH_J ? 4 : ((MODEL == CX_CAMERA_ANGLE_AXIS && !(VARIANT & 1)) ? 3 : 2)) void k_bal_evaluate(const double* __restrict__ state,
                                                         const double* __restrict__ obs,
                                                         const int32_t* __restrict__ row_pt,
                                                         const int32_t* __restrict__ row_cam, int64_t O,
                                                         int64_t cam_off, double* __restrict__ residuals,
                                                         double* __restrict__ E, double* __restrict__ F,
                                                         double* __restrict__ cost_partial, LossParams loss_in,
                                                         double* __restrict__ Ft, const int32_t* __restrict__ cam_pos,
                                                         int num_tiles, const double* __restrict__ col_scale) {
  static_assert(!SCALED || WITH_J, "column scales apply to the Jacobian");
  constexpr bool kClosed = (VARIANT & 1) == 0;
  constexpr bool kCoop = (VARIANT & 4) == 0 && MODEL == CX_CAMERA_ANGLE_AXIS;  // (64 x 9 doubles per wavefront = the LDS buffer)
  constexpr int kCam = (MODEL == CX_CAMERA_ANGLE_AXIS) ? 9 : 10;
  __shared__ double lds[kBlock * 9];
  __shared__ double lds_scale[SCALED ? kBlock * 9 : 1];  // SCALED: the wavefronts' slices for the gathered column scales
  __shared__ double red[4];
  __shared__ int cpos[kBlock];
  const int tid0 = threadIdx.x;
  int tid = tid0;
  int tile = eval_tile(0, num_tiles);
  if (tile < 0) return;
  // prologue: indices and parameters of the first tile, indices of the second
  int32_t ci = 0, pi = 0, cp_n = 0;  // cp_n: position of the row's F cell in the camera-major copy
  {
    const int64_t r = int64_t(tile) * kBlock + tid;
    if (r < O) {
      ci = row_cam[r];
      pi = row_pt[r];
      if (WITH_J && Ft != nullptr) cp_n = cam_pos[r];
    }
  }
  double cam_n[kCam], pt_n[3];
  int32_t ci_cur = ci, pi_cur = pi;  // SCALED: block ids of the tile being processed (its scales are requested at the loop's top)
  double2 o_n = make_double2(0.0, 0.0);
  {
    gather_by_row<kCam, kCoop>(state + cam_off, ci, tid0 & 63, cam_n);  // (rows past O read block 0: valid memory, never used)
    const double* pp = state + 3 * int64_t(pi);
#pragma unroll
    for (int i = 0; i < 3; ++i) pt_n[i] = pp[i];
    const int64_t r = int64_t(tile) * kBlock + tid;
    if (r < O) o_n = reinterpret_cast<const double2*>(obs)[r];
  }
  int tile1 = eval_tile(1, num_tiles);
  ci = 0; pi = 0;
  if (tile1 >= 0) {
    const int64_t r = int64_t(tile1) * kBlock + tid;
    if (r < O) { ci = row_cam[r]; pi = row_pt[r]; }
  }
  // (nothing of the prologue is left in flight at the loop's entry: the compiler merges what is pending on the two
  // edges into the loop header, and a load pending on the entry edge became a vmcnt wait at the top of every
  // iteration -- behind the previous iteration's stores)
#pragma unroll
  for (int i = 0; i < kCam; ++i) asm volatile("" : "+v"(cam_n[i]));
#pragma unroll
  for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(pt_n[i]));
  asm volatile("" : "+v"(o_n.x), "+v"(o_n.y), "+v"(ci), "+v"(pi), "+v"(cp_n));
  if constexpr (SCALED) asm volatile("" : "+v"(ci_cur), "+v"(pi_cur));
  for (int k = 0; tile >= 0; ++k) {
    // the thread index and the loss parameters pass through an opaque statement per tile.  Left visible as loop
    // invariants, everything derived from them (LDS and global addresses, the index arithmetic of the stores; every
    // uniform subexpression of every loss type) is hoisted out of the loop and kept in vector registers: 62 -> 175
    // VGPRs for the value-only kernel, 138 -> 292 with the Jacobian.  (cx_eval.hip is also compiled with
    // -mllvm -disable-machine-licm: the literal constants of sin / cos were hoisted into 77 more.)
    tid = tid0;
    asm volatile("" : "+v"(tid));
    const int64_t r0i = int64_t(tile) * kBlock;
    const int nvalid = int(min(int64_t(kBlock), O - r0i));
    const int64_t r = r0i + tid;
    double cam[kCam], pt[3];
#pragma unroll
    for (int i = 0; i < kCam; ++i) cam[i] = cam_n[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) pt[i] = pt_n[i];
    const double2 o = o_n;
    const int32_t cp = cp_n;
    // SCALED: the column scales of THIS tile, requested first of the iteration's loads (they are the oldest when the
    // multiplications at the end of the arithmetic wait for them, so the younger requests of the next tiles stay in
    // flight) and not a tile ahead: prefetching them like the parameters costs 24 more live registers and spills.
    // The 9 scales of a camera come by the cooperative gather (consecutive lanes read consecutive words) and reach their
    // rows through a slice of LDS of their own just before they are used.
    double sc[SCALED ? 9 : 1], sp[SCALED ? 3 : 1];
    if constexpr (SCALED) {
      gather_by_row<9, true>(col_scale + cam_off, ci_cur, tid & 63, sc);  // tangent columns: points first, 9 per camera
      const double* spp = col_scale + 3 * int64_t(pi_cur);
#pragma unroll
      for (int i = 0; i < 3; ++i) sp[i] = spp[i];
      ci_cur = ci;  // (ci, pi still name the next tile's blocks here)
      pi_cur = pi;
    }
    if constexpr (kCoop) {
      transpose_gathered<kCam>(lds + (tid >> 6) * 64 * kCam, tid & 63, cam);
      __syncthreads();  // the slices are the staging buffer of the stores below
    }
    // requests of the tiles to come, in front of this tile's stores
    const int tile2 = eval_tile(k + 2, num_tiles);
    if (tile1 >= 0) {
      gather_by_row<kCam, kCoop>(state + cam_off, ci, tid & 63, cam_n);
      const double* pp = state + 3 * int64_t(pi);
#pragma unroll
      for (int i = 0; i < 3; ++i) pt_n[i] = pp[i];
      const int64_t r1 = int64_t(tile1) * kBlock + tid;
      if (r1 < O) {
        o_n = reinterpret_cast<const double2*>(obs)[r1];
        if (WITH_J && Ft != nullptr) cp_n = cam_pos[r1];
      }
      ci = 0; pi = 0;
      if (tile2 >= 0) {
        const int64_t r2 = int64_t(tile2) * kBlock + tid;
        if (r2 < O) { ci = row_cam[r2]; pi = row_pt[r2]; }
      }
    }
    double res0 = 0.0, res1 = 0.0, cost_term = 0.0;
    double jc[18], jp[6];
    LossParams loss = loss_in;
    asm volatile("" : "+s"(loss.type), "+s"(loss.a), "+s"(loss.b));
    if (tid < nvalid) {
      if constexpr ((VARIANT & 2) != 0) {  // probe: every load and store of the kernel, no arithmetic
        res0 = o.x; res1 = o.y;
#pragma unroll
        for (int i = 0; i < 18; ++i) jc[i] = cam[i % kCam] + double(i);
#pragma unroll
        for (int i = 0; i < 6; ++i) jp[i] = pt[i % 3] + double(i);
      } else if constexpr (MODEL == CX_CAMERA_ANGLE_AXIS) {
        if (WITH_J && kClosed) snavely_closed_form(cam, pt, o.x, o.y, res0, res1, jc, jp);
        else if (WITH_J) snavely_jet(cam, pt, o.x, o.y, res0, res1, jc, jp);
        else snavely_value(cam, pt, o.x, o.y, res0, res1);
      } else {
        if (WITH_J) snavely_jet_quat(cam, pt, o.x, o.y, res0, res1, jc, jp);
        else snavely_value_quat(cam, pt, o.x, o.y, res0, res1);
      }
      const double sq = res0 * res0 + res1 * res1;
      cost_term = 0.5 * sq;
      if (loss.type != CX_LOSS_NONE) {
        // residual_block.cc:165-196 with Corrector (corrector.cc:41-116): Jacobians first, with
        // the uncorrected residuals, then the residuals
        double rho[3];
        loss_evaluate(loss, sq, rho);
        cost_term = 0.5 * rho[0];
        const double sqrt_rho1 = sqrt(rho[1]);
        double residual_scaling = sqrt_rho1, alpha_sq_norm = 0.0;
        if (sq != 0.0 && rho[2] > 0.0) {
          const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
          const double alpha = 1.0 - sqrt(D);
          residual_scaling = sqrt_rho1 / (1 - alpha);
          alpha_sq_norm = alpha / sq;
        }
        if (WITH_J) {
          correct_jacobian<9>(sqrt_rho1, alpha_sq_norm, res0, res1, jc);
          correct_jacobian<3>(sqrt_rho1, alpha_sq_norm, res0, res1, jp);
        }
        res0 *= residual_scaling;
        res1 *= residual_scaling;
      }
    }
    if constexpr (SCALED) {  // ScaleColumns (block_sparse_matrix.cc:403-450): every value times the scale of its column
      // every lane of the wavefront hands on the words it gathered (also the lanes past the last row of the last tile), so
      // the exchange sits outside the `tid < nvalid` block; the slice is the wavefront's own: no barrier
      transpose_gathered<9>(lds_scale + (tid >> 6) * 64 * 9, tid & 63, sc);
      if (tid < nvalid) {
#pragma unroll
        for (int i = 0; i < 9; ++i) { jc[i] *= sc[i]; jc[9 + i] *= sc[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { jp[i] *= sp[i]; jp[3 + i] *= sp[i]; }
      }
    }
    // The requests of the next tiles are waited for HERE, before this tile's stores are issued: vmcnt counts loads and
    // stores in one in-order counter, and the number of stores below depends on nvalid, so a wait placed after them
    // (where the values are first read, at the top of the next iteration) could only be vmcnt(0) -- the loads would
    // wait for this tile's stores to reach memory, which is the serialisation the loop exists to remove.
#pragma unroll
    for (int i = 0; i < kCam; ++i) asm volatile("" : "+v"(cam_n[i]));
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(pt_n[i]));
    asm volatile("" : "+v"(o_n.x), "+v"(o_n.y), "+v"(ci), "+v"(pi), "+v"(cp_n));
    if (residuals && tid < nvalid) reinterpret_cast<double2*>(residuals)[r] = make_double2(res0, res1);
    if (WITH_J) {
      // F cells: row-major run and (Ft != nullptr) the camera-major copy in the same pass, see unstage_f_cells_two
      unstage_cells_halves<18>(F + 18 * r0i, Ft, cp, cpos, nvalid, lds, jc, tid);
      unstage_cells<6>(E + 6 * r0i, nvalid, lds, jp, tid);
    }
    if (cost_partial) {
      double c[1] = {cost_term};
      block_sum<1>(c, red, tid);
      if (tid == 0) cost_partial[tile] = c[0];
    }
    tile = tile1;
    tile1 = tile2;
  }
}

// deterministic sum of the per-workgroup cost partials, two stages: 256 workgroups add contiguous slices (each in a
// fixed order), the last of them to finish -- told by an agent-scope ticket -- adds the 256 slice sums in index order.
// (One workgroup walking 113 k partials alone took 183 us per evaluation on the Final shape.)
__global__ __launch_bounds__(256) void k_sum_partials(const double* __restrict__ partial, int64_t n, double* __restrict__ slice_sum,
                                                      unsigned* __restrict__ ticket, double* __restrict__ out) {
  __shared__ double red[4];
  __shared__ int is_last;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = per * blockIdx.x, hi = min(n, lo + per);
  double s[1] = {0.0};
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s[0] += partial[i];
  block_sum<1>(s, red);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&slice_sum[blockIdx.x], s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slice sum is complete before the ticket is drawn (cx_solver.hip: dot2_finish)
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  double v[1] = {0.0};
  for (int i = threadIdx.x; i < int(gridDim.x); i += 256) v[0] += __hip_atomic_load(&slice_sum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  block_sum<1>(v, red);
  if (threadIdx.x == 0) {
    *out = v[0];
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void k_divide_by(double* __restrict__ v, const double* __restrict__ d, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) v[i] = v[i] / d[i];
}

// Evaluator::Plus (program_evaluator.h:306-320): out = x + sign * delta on Euclidean blocks ...
__global__ __launch_bounds__(256) void k_plus_euclidean(const double* __restrict__ x, const double* __restrict__ delta, double sign,
                                                        double* __restrict__ out, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) out[i] = x[i] + sign * delta[i];
}

// ... and ProductManifold<QuaternionManifold, EuclideanManifold<6>>::Plus per camera (manifold.cc:27-59):
// 10 ambient values from 10 + 9 (tangent) values
__global__ __launch_bounds__(256) void k_plus_quaternion_cameras(const double* __restrict__ x, const double* __restrict__ delta,
                                                                 double sign, double* __restrict__ out, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double* xc = x + 10 * int64_t(c);
  const double* dc = delta + 9 * int64_t(c);
  double* oc = out + 10 * int64_t(c);
  const double d0 = sign * dc[0], d1 = sign * dc[1], d2 = sign * dc[2];
  const double norm_delta = norm3d(d0, d1, d2);
  if (norm_delta == 0.0) {
    oc[0] = xc[0]; oc[1] = xc[1]; oc[2] = xc[2]; oc[3] = xc[3];
  } else {
    const double s = sin(norm_delta) / norm_delta;
    const double q0 = cos(norm_delta), q1 = s * d0, q2 = s * d1, q3 = s * d2;
    oc[0] = q0 * xc[0] - q1 * xc[1] - q2 * xc[2] - q3 * xc[3];
    oc[1] = q0 * xc[1] + q1 * xc[0] + q2 * xc[3] - q3 * xc[2];
    oc[2] = q0 * xc[2] - q1 * xc[3] + q2 * xc[0] + q3 * xc[1];
    oc[3] = q0 * xc[3] + q1 * xc[2] - q2 * xc[1] + q3 * xc[0];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) oc[4 + i] = xc[4 + i] + sign * dc[3 + i];
}

}  // namespace

int cxe_plus(cx_evaluator* e, const double* x, const double* delta, double sign, double* out) {
  hipStream_t st = e->ctx->stream;
  const int64_t np = 3 * int64_t(e->P);
  if (e->camera_model == CX_CAMERA_ANGLE_AXIS) {
    const int64_t n = np + 9 * int64_t(e->C);
    hipLaunchKernelGGL(k_plus_euclidean, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, x, delta, sign, out, n);
  } else {
    if (np > 0) hipLaunchKernelGGL(k_plus_euclidean, dim3(unsigned((np + 255) / 256)), dim3(256), 0, st, x, delta, sign, out, np);
    if (e->C > 0)
      hipLaunchKernelGGL(k_plus_quaternion_cameras, dim3(unsigned((e->C + 255) / 256)), dim3(256), 0, st, x + np, delta + np, sign,
                         out + np, e->C);
  }
  CX_HIP(hipGetLastError());
  return CX_OK;
}

extern "C" {

int cx_evaluator_create_bal(cx_context* ctx, int32_t C, int32_t P, int64_t O, const int32_t* cam, const int32_t* pt,
                            const double* obs, cx_evaluator** out) {
  CX_CHECK_ARG(ctx && out && C > 0 && P > 0 && O > 0 && cam && pt && obs);
  CX_CHECK_ARG(24 * O < (int64_t(1) << 31));  // int32 cell positions (block_jacobian_writer.cc:95-99,157-161)
  for (int64_t i = 0; i < O; ++i) CX_CHECK_ARG(cam[i] >= 0 && cam[i] < C && pt[i] >= 0 && pt[i] < P);
  if (cxm_is_front(ctx)) return cxm_evaluator_create_bal(ctx, C, P, O, cam, pt, obs, out);
  // LexicographicallyOrderResidualBlocks (reorder_program.cc:256-338): bucket by point,
  // every bucket filled from its back, so a chunk lists its residuals in reverse input order
  std::vector<int64_t> offsets(size_t(P) + 1, 0), order(O);
  for (int64_t i = 0; i < O; ++i) offsets[pt[i]]++;
  std::partial_sum(offsets.begin(), offsets.end(), offsets.begin());
  for (int64_t i = 0; i < O; ++i) order[--offsets[pt[i]]] = i;
  // BuildJacobianLayout (block_jacobian_writer.cc:68-167)
  std::vector<cx_block> rows(O), cols(size_t(P) + C);
  std::vector<int32_t> rcb(O + 1);
  std::vector<cx_cell> cells(2 * O);
  for (int j = 0; j < P; ++j) cols[j] = cx_block{3, 3 * j};
  for (int i = 0; i < C; ++i) cols[P + i] = cx_block{9, 3 * P + 9 * i};
  auto e = new cx_evaluator;
  e->ctx = ctx;
  e->C = C; e->P = P; e->O = O;
  e->row_of_obs.resize(O);
  std::vector<double> obs_rows(2 * O);
  for (int64_t k = 0; k < O; ++k) {
    const int64_t i = order[k];
    rows[k] = cx_block{2, int32_t(2 * k)};
    rcb[k] = int32_t(2 * k);
    cells[2 * k] = cx_cell{pt[i], int32_t(6 * k)};
    cells[2 * k + 1] = cx_cell{P + cam[i], int32_t(6 * O + 18 * k)};
    e->row_of_obs[i] = k;
    obs_rows[2 * k] = obs[2 * i];
    obs_rows[2 * k + 1] = obs[2 * i + 1];
  }
  rcb[O] = int32_t(2 * O);
  cx_block_structure bs{int32_t(O), P + C, rows.data(), cols.data(), rcb.data(), cells.data()};
  int rc = cx_matrix_create(ctx, &bs, P, &e->J);
  if (rc == CX_OK && !e->J->is239) {
    cx_set_error("BAL problem does not map onto the static <2,3,9> layout (duplicate observation of a point by a camera?)");
    rc = CX_ERR_UNSUPPORTED;
  }
  if (rc == CX_OK) rc = e->d_obs.upload(obs_rows, ctx->stream);
  if (rc == CX_OK) rc = e->d_partial.alloc(size_t((O + kBlock - 1) / kBlock) + 1 + 256 + 1);
  if (rc == CX_OK && hipMemset(e->d_partial.p, 0, e->d_partial.n * sizeof(double)) != hipSuccess) rc = CX_ERR_HIP;
  if (rc != CX_OK) {
    if (e->J) cx_matrix_destroy(e->J);
    delete e;
    return rc;
  }
  *out = e;
  return CX_OK;
}

void cx_evaluator_destroy(cx_evaluator* e) {
  if (!e) return;
  if (!e->parts.empty() || cxm_is_front(e->ctx)) return cxm_evaluator_destroy(e);
  cx_matrix_destroy(e->J);
  delete e;
}

cx_matrix* cx_evaluator_jacobian(cx_evaluator* e) { return e ? e->J : nullptr; }

int cx_evaluator_row_of_observation(const cx_evaluator* e, int64_t* out) {
  CX_CHECK_ARG(e && out);
  std::copy(e->row_of_obs.begin(), e->row_of_obs.end(), out);
  return CX_OK;
}

int cx_evaluator_evaluate(cx_evaluator* e, const double* state, double* cost, double* residuals, double* gradient,
                          int32_t evaluate_jacobian, int32_t memspace) {
  CX_CHECK_ARG(e && state);
  if (!e->parts.empty()) return cxm_evaluator_evaluate(e, state, cost, residuals, gradient, evaluate_jacobian, memspace);
  cx_context* ctx = e->ctx;
  cx_matrix* A = e->J;
  hipStream_t st = ctx->stream;
  CX_HIP(hipSetDevice(ctx->device));
  const int64_t ncols = A->num_cols, nrows = A->num_rows;
  HostOrDevice hs(ctx), hg(ctx);
  const int64_t nstate = 3 * int64_t(e->P) + (e->camera_model == CX_CAMERA_ANGLE_AXIS ? 9 : 10) * int64_t(e->C);
  int staged = hs.in(state, size_t(nstate), memspace);
  if (staged == CX_OK) staged = hg.inout(gradient, size_t(ncols), memspace, false);
  // sharded: cost and gradient are summed over the ranks below; a rank that could not stage its inputs says so first
  if (cost != nullptr || gradient != nullptr) CX_TRY(cx_comm_agree(ctx, staged));
  else CX_TRY(staged);
  const bool with_j = evaluate_jacobian != 0 || gradient != nullptr;
  // Residuals asked for in host memory are produced in the evaluator's own device buffer and copied out, so that
  // they also stay available in HBM (cx_evaluator_device_residuals) for the linear solve that follows.
  const bool res_to_host = residuals != nullptr && cx_is_host_space(memspace);
  double* res_host = residuals;
  if (res_to_host && memspace == CX_HOST_SLICES) res_host = reinterpret_cast<const cx_host_slices*>(residuals)->head;  // a front's row slice
  double* res_dev = (memspace == CX_DEVICE) ? residuals : nullptr;
  if (res_to_host || (gradient && !res_dev)) {
    CX_TRY(e->d_res.alloc(size_t(nrows)));
    res_dev = e->d_res.p;
  }
  // d_res is "the residuals last handed out in host memory" only when this call hands some out; a gradient-only call
  // borrows it as scratch for the residuals of ITS state, after which the copy no longer matches the caller's host array
  if (residuals != nullptr || res_dev == e->d_res.p) e->res_valid = res_to_host;
  const int grid = int((e->O + kBlock - 1) / kBlock);
  // A gradient without a Jacobian (evaluate_jacobian == 0, gradient != NULL) still needs J for g = J'r, but must leave
  // the evaluator's matrix alone: that storage is what CreateJacobian handed to the caller, possibly column-scaled
  // since (ProgramEvaluator uses per-thread scratch for this case, program_evaluator.h:186-206).  The values of such a
  // call go to scratch arrays of the same layout, which stand in for the matrix' arrays during the product below.
  const bool scratch_j = with_j && evaluate_jacobian == 0;
  if (scratch_j) {
    CX_TRY(e->d_scratch_values.alloc(size_t(24 * e->O)));
    CX_TRY(e->d_scratch_Ft.alloc(size_t(18 * e->O)));
  }
  double* E = scratch_j ? e->d_scratch_values.p : A->d_values.p;
  double* F = E + 6 * e->O;
  const LossParams loss{e->loss_type, e->loss_a, e->loss_b};
  // the camera-major copy of F is written by the same kernel unless the caller said that a ScaleColumns follows
  // (which rewrites it anyway) -- cx_evaluator_set_emit_camera_major
  static const bool emit_allowed = std::getenv("CX_NO_FT_EMIT") == nullptr;  // A/B switch
  // a column scale registered with the evaluator (cx_evaluator_set_column_scale) is applied by the kernel itself: no
  // ScaleColumns follows such an evaluation.  Whether the kernel also writes the camera-major copy stays the caller's
  // choice (emit_ft): measured on the Final shape, scattering the 144-byte cells from this kernel costs 2.5 ms, while
  // the gather pass that rebuilds the copy at its first use (k_permute_ft: scattered READS, streamed writes) costs 1.75.
  const bool scaled_j = with_j && !scratch_j && e->has_col_scale;
  double* ft_out = nullptr;
  if (scratch_j) {
    ft_out = e->d_scratch_Ft.p;
  } else if (with_j && e->emit_ft && emit_allowed) {
    CX_TRY(A->d_Ft.alloc(size_t(A->O) * 18));
    ft_out = A->d_Ft.p;
  }
  // persistent grid: as many workgroups as the chip holds at once (a multiple of 8, see eval_tile); CX_EVAL_PERSISTENT=0
  // launches one workgroup per tile instead (round 1's form, kept for the A/B of tools/eval_ab.py)
  static const bool persistent = !(std::getenv("CX_EVAL_PERSISTENT") && std::atoi(std::getenv("CX_EVAL_PERSISTENT")) == 0);
  static const int variant = std::getenv("CX_EVAL_VARIANT") ? std::atoi(std::getenv("CX_EVAL_VARIANT")) : 0;  // A/B switch
  CX_HIP(hipEventRecord(ctx->ev[6], st));
#define CX_LAUNCH_EVAL_VS(WJ, MODEL, V, S)                                                                             \
  do {                                                                                                                 \
    static int occ = 0;                                                                                                \
    if (occ == 0) {                                                                                                    \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bal_evaluate<WJ, MODEL, V, S>, kBlock, 0) != hipSuccess || occ < 1) occ = 1; \
    }                                                                                                                  \
    const int resident = std::max(8, (ctx->num_cus * occ) / 8 * 8);                                                    \
    const int launch_grid = persistent ? std::min(xcd_grid(grid), resident) : xcd_grid(grid);                          \
    hipLaunchKernelGGL((k_bal_evaluate<WJ, MODEL, V, S>), dim3(launch_grid), dim3(kBlock), 0, st, (const double*)hs.dptr, \
                       (const double*)e->d_obs.p, (const int32_t*)A->d_row_pt.p, (const int32_t*)A->d_row_cam.p, e->O,  \
                       3 * int64_t(e->P), res_dev, E, F, cost ? e->d_partial.p : nullptr, loss, ft_out,                \
                       (const int32_t*)A->d_cam_pos.p, grid, (const double*)(S ? e->d_col_scale.p : nullptr));         \
  } while (0)
#define CX_LAUNCH_EVAL_V(WJ, MODEL, V) CX_LAUNCH_EVAL_VS(WJ, MODEL, V, false)
#define CX_LAUNCH_EVAL(WJ, MODEL) CX_LAUNCH_EVAL_V(WJ, MODEL, 0)
  if (scaled_j) {
    if (e->camera_model == CX_CAMERA_ANGLE_AXIS) CX_LAUNCH_EVAL_VS(true, CX_CAMERA_ANGLE_AXIS, 0, true);
    else CX_LAUNCH_EVAL_VS(true, CX_CAMERA_QUATERNION_MANIFOLD, 0, true);
  } else if (e->camera_model == CX_CAMERA_ANGLE_AXIS) {
    if (with_j && variant == 1) CX_LAUNCH_EVAL_V(true, CX_CAMERA_ANGLE_AXIS, 1);
    else if (with_j && variant == 2) CX_LAUNCH_EVAL_V(true, CX_CAMERA_ANGLE_AXIS, 2);
    else if (with_j && variant == 4) CX_LAUNCH_EVAL_V(true, CX_CAMERA_ANGLE_AXIS, 4);
    else if (with_j && variant == 5) CX_LAUNCH_EVAL_V(true, CX_CAMERA_ANGLE_AXIS, 5);
    else if (with_j) CX_LAUNCH_EVAL(true, CX_CAMERA_ANGLE_AXIS);
    else CX_LAUNCH_EVAL(false, CX_CAMERA_ANGLE_AXIS);
  } else {
    if (with_j) CX_LAUNCH_EVAL(true, CX_CAMERA_QUATERNION_MANIFOLD);
    else CX_LAUNCH_EVAL(false, CX_CAMERA_QUATERNION_MANIFOLD);
  }
#undef CX_LAUNCH_EVAL
#undef CX_LAUNCH_EVAL_V
#undef CX_LAUNCH_EVAL_VS
  if (cost) {
    // layout of d_partial: [grid] per-workgroup costs | total | [256] slice sums | ticket (kept zero between launches)
    const int slices = std::min(256, grid);
    hipLaunchKernelGGL(k_sum_partials, dim3(slices), dim3(256), 0, st, (const double*)e->d_partial.p, int64_t(grid),
                       e->d_partial.p + grid + 1, reinterpret_cast<unsigned*>(e->d_partial.p + grid + 1 + 256), e->d_partial.p + grid);
  }
  CX_HIP(hipGetLastError());
  if (with_j && !scratch_j) { A->ft_valid = ft_out != nullptr; A->f32_valid = false; }
  // the residuals are final once the evaluation kernel has run: their way back to the caller (the largest copy of an LM
  // iteration, 16 B per residual block) starts now, on the copy stream, beside the camera-major gather pass and the
  // gradient product below
  hipStream_t cs = st;
  if (res_to_host) {
    cs = cx_copy_stream(ctx);
    if (cs != st) {
      CX_HIP(hipEventRecord(ctx->ev[4], st));
      CX_HIP(hipStreamWaitEvent(cs, ctx->ev[4], 0));
    }
    CX_TRY(cx_copy_d2h(ctx, res_host, res_dev, size_t(nrows) * sizeof(double), cs));
  }
  // an evaluation that applied the column scale is not followed by a ScaleColumns that would rebuild the camera-major
  // copy: it is rebuilt here, by the gather pass, inside the evaluation's own time
  if (scaled_j && !A->ft_valid) CX_TRY(cx_matrix_ensure_ft(A));
  CX_HIP(hipEventRecord(ctx->ev[7], st));
  if (gradient) {
    // g = J' r (program_evaluator.h:242-258), written outright (the products' kernels cover every entry)
    if (scratch_j) {  // the product reads the scratch copies (pointers are taken at launch), the matrix keeps its own
      const bool ft_valid = A->ft_valid, use_f32 = A->use_f32;
      std::swap(A->d_values.p, e->d_scratch_values.p);
      std::swap(A->d_Ft.p, e->d_scratch_Ft.p);
      A->ft_valid = true;
      A->use_f32 = false;
      const int rc = cxk_left_multiply(A, res_dev, hg.dptr, false);
      std::swap(A->d_values.p, e->d_scratch_values.p);
      std::swap(A->d_Ft.p, e->d_scratch_Ft.p);
      A->ft_valid = ft_valid;
      A->use_f32 = use_f32;
      CX_TRY(rc);
    } else {
      CX_TRY(cxk_left_multiply(A, res_dev, hg.dptr, false));
    }
    if (ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, hg.dptr + 3 * int64_t(e->P), 9 * int64_t(e->C)));
    // the stored J carries the column scales: (J S)'r = S J'r, and the gradient of the caller's (unscaled) problem is
    // that divided by the scales (trust_region_minimizer.cc:263-279 evaluates the gradient before it scales J)
    if (scaled_j)
      hipLaunchKernelGGL(k_divide_by, dim3(unsigned((ncols + 255) / 256)), dim3(256), 0, st, hg.dptr, (const double*)e->d_col_scale.p, ncols);
  }
  if (cost) {
    if (ctx->nranks > 1) CX_TRY(cx_allreduce_device(ctx, e->d_partial.p + grid, 1));  // shards sum their costs
    CX_TRY(cx_read_back(ctx, cost, e->d_partial.p + grid, sizeof(double), st));
  }
  CX_TRY(hg.out_async(st));
  CX_TRY(cx_stream_sync(ctx, st));
  if (cs != st) CX_TRY(cx_stream_sync(ctx, cs));
  CX_HIP(hipEventElapsedTime(&e->last_ms, ctx->ev[6], ctx->ev[7]));
  return CX_OK;
}

const double* cx_evaluator_device_residuals(const cx_evaluator* e) {
  if (e && !e->parts.empty()) return cxm_evaluator_device_residuals(e);
  return (e && e->res_valid) ? e->d_res.p : nullptr;
}

// 64 entries of the device copy, evenly spread (first and last included), for the check below
__global__ void k_sample_vector(const double* __restrict__ v, int64_t n, double* __restrict__ out) {
  const int t = threadIdx.x;
  if (t < 64) out[t] = v[n <= 64 ? min(int64_t(t), n - 1) : (int64_t(t) * (n - 1)) / 63];
}

int cx_evaluator_device_residuals_match(cx_evaluator* e, const double* host_residuals, int32_t* match) {
  CX_CHECK_ARG(e != nullptr && host_residuals != nullptr && match != nullptr);
  *match = 0;
  if (!e->parts.empty()) {  // a front: every shard checks its own rows
    int32_t all = 1;
    const cx_matrix* A = e->J;
    for (size_t i = 0; i < e->parts.size() && all; ++i) {
      int32_t m = 0;
      CX_TRY(cx_evaluator_device_residuals_match(e->parts[i], host_residuals + A->part_row0[i], &m));
      all = all && m;
    }
    *match = all;
    return CX_OK;
  }
  if (!e->res_valid || e->d_res.p == nullptr) return CX_OK;
  cx_context* ctx = e->ctx;
  CX_HIP(hipSetDevice(ctx->device));
  const int64_t n = e->J->num_rows;
  if (n == 0) { *match = 1; return CX_OK; }
  CX_TRY(e->d_sample.alloc(64));
  hipLaunchKernelGGL(k_sample_vector, dim3(1), dim3(64), 0, ctx->stream, (const double*)e->d_res.p, n, e->d_sample.p);
  double got[64];
  CX_TRY(cx_read_back(ctx, got, e->d_sample.p, sizeof(got)));
  int32_t same = 1;
  for (int t = 0; t < 64 && same; ++t) {
    const int64_t i = n <= 64 ? std::min<int64_t>(t, n - 1) : (int64_t(t) * (n - 1)) / 63;
    same = std::memcmp(&got[t], &host_residuals[i], sizeof(double)) == 0;
  }
  *match = same;
  return CX_OK;
}

int cx_evaluator_set_emit_camera_major(cx_evaluator* e, int32_t on) {
  CX_CHECK_ARG(e != nullptr);
  e->emit_ft = on != 0;
  return e->parts.empty() ? CX_OK : cxm_evaluator_forward_settings(e);
}

int cx_evaluator_set_column_scale(cx_evaluator* e, const double* scale, int32_t memspace) {
  CX_CHECK_ARG(e != nullptr);
  if (!e->parts.empty()) return cxm_evaluator_set_column_scale(e, scale, memspace);
  if (scale == nullptr) {
    e->has_col_scale = false;
    return CX_OK;
  }
  cx_context* ctx = e->ctx;
  CX_HIP(hipSetDevice(ctx->device));
  const size_t n = size_t(3 * int64_t(e->P) + 9 * int64_t(e->C));
  CX_TRY(e->d_col_scale.alloc(n));
  CX_TRY(cx_vector_in(ctx, e->d_col_scale.p, scale, n, memspace));
  CX_TRY(cx_stream_sync(ctx, ctx->stream));
  e->has_col_scale = true;
  return CX_OK;
}

int cx_evaluator_set_camera_model(cx_evaluator* e, int32_t camera_model) {
  CX_CHECK_ARG(e != nullptr && (camera_model == CX_CAMERA_ANGLE_AXIS || camera_model == CX_CAMERA_QUATERNION_MANIFOLD));
  e->camera_model = camera_model;
  return e->parts.empty() ? CX_OK : cxm_evaluator_forward_settings(e);
}

int64_t cx_evaluator_num_parameters(const cx_evaluator* e) {
  if (!e) return 0;
  return 3 * int64_t(e->P) + (e->camera_model == CX_CAMERA_ANGLE_AXIS ? 9 : 10) * int64_t(e->C);
}

int64_t cx_evaluator_num_effective_parameters(const cx_evaluator* e) { return e ? 3 * int64_t(e->P) + 9 * int64_t(e->C) : 0; }

int cx_evaluator_plus(cx_evaluator* e, const double* x, const double* delta, double* x_plus_delta, int32_t memspace) {
  CX_CHECK_ARG(e && x && delta && x_plus_delta);
  if (!e->parts.empty()) return cxm_evaluator_plus(e, x, delta, x_plus_delta, memspace);
  cx_context* ctx = e->ctx;
  CX_HIP(hipSetDevice(ctx->device));
  HostOrDevice hx(ctx), hd(ctx), ho(ctx);
  CX_TRY(hx.in(x, size_t(cx_evaluator_num_parameters(e)), memspace));
  CX_TRY(hd.in(delta, size_t(cx_evaluator_num_effective_parameters(e)), memspace));
  CX_TRY(ho.inout(x_plus_delta, size_t(cx_evaluator_num_parameters(e)), memspace, false));
  CX_TRY(cxe_plus(e, hx.dptr, hd.dptr, 1.0, ho.dptr));
  CX_TRY(ho.out());
  CX_TRY(cx_stream_sync(ctx, ctx->stream));
  return CX_OK;
}

int cx_evaluator_set_loss(cx_evaluator* e, int32_t loss_type, double a, double b) {
  CX_CHECK_ARG(e != nullptr && loss_type >= CX_LOSS_NONE && loss_type <= CX_LOSS_TUKEY);
  if (loss_type != CX_LOSS_NONE) CX_CHECK_ARG(a > 0.0 || (loss_type == CX_LOSS_TOLERANT && a >= 0.0));
  if (loss_type == CX_LOSS_TOLERANT) CX_CHECK_ARG(b > 0.0);
  e->loss_type = loss_type;
  e->loss_a = a;
  e->loss_b = b;
  return e->parts.empty() ? CX_OK : cxm_evaluator_forward_settings(e);
}

double cx_evaluator_last_kernel_ms(const cx_evaluator* e) {
  if (e && !e->parts.empty()) return cxm_evaluator_last_kernel_ms(e);
  return e ? double(e->last_ms) : 0.0;
}

}  // extern "C"
