// Small fixed-size fp64 algebra shared by the APD-GICP kernels (device) and the host-loop debug
// path (host).  Everything is written with static indices so that it stays in registers on gfx950
// (runtime-indexed private arrays go to scratch).  Compiled with -ffp-contract=off: no expression in this
// project is fused.  (Tried in r01: local FMA contraction of the fp64 helpers buys 4 % on k_linearize and
// moves an ill-conditioned far-range LM run by 2e-4 rad, beyond the 1e-4 parity bar -- not worth it.)
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/apd_atan2f.h"  // apd::apd_atan2f: the C library's algorithm (fdlibm), the same bits on host and device

#define APD_HD __host__ __device__ __forceinline__

namespace apd {

// symmetric 3x3, upper triangle
struct Sym3 {
  double xx, xy, xz, yy, yz, zz;
};

// rigid transform, row-major 3x4 [R | t]
struct Rigid {
  double m[12];
};

APD_HD Rigid rigid_identity() {
  Rigid r;
  r.m[0] = 1, r.m[1] = 0, r.m[2] = 0, r.m[3] = 0;
  r.m[4] = 0, r.m[5] = 1, r.m[6] = 0, r.m[7] = 0;
  r.m[8] = 0, r.m[9] = 0, r.m[10] = 1, r.m[11] = 0;
  return r;
}

// Isometry3d * Isometry3d (x0 = delta * x0, lsq_registration_impl.hpp:119,144)
APD_HD Rigid rigid_mul(const Rigid& a, const Rigid& b) {
  Rigid r;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) r.m[4 * i + j] = a.m[4 * i] * b.m[j] + a.m[4 * i + 1] * b.m[4 + j] + a.m[4 * i + 2] * b.m[8 + j];
    r.m[4 * i + 3] = a.m[4 * i] * b.m[3] + a.m[4 * i + 1] * b.m[7] + a.m[4 * i + 2] * b.m[11] + a.m[4 * i + 3];
  }
  return r;
}

APD_HD Sym3 sym3_add(const Sym3& a, const Sym3& b) { return Sym3{a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yy + b.yy, a.yz + b.yz, a.zz + b.zz}; }

// sin and cos of an angle |x| <= pi (the three angles of the APD sensor model are atan2f results, A:170-177): one Cody-Waite
// step (n = rint(x 2/pi) in -2 .. 2, n * pio2_1 exact in 35 bits) and the two fdlibm kernels (k_sin.c / k_cos.c, S1..S6 /
// C1..C6) on [-pi/4, pi/4] with explicit FMAs -- about 45 fp64 instructions for the pair against ~135 of the library's
// sincos(), whose reduction also covers arguments up to 1e308.  Measured on the host against glibc over 22 M float-valued
// arguments in [-pi, pi]: at most 0.77 ulp from the exact value, the same double as glibc's sin / cos in 99.83 % of them
// (the library call this replaces is itself a ~1 ulp implementation that differs from the reference's libm).
APD_HD void sincos_pi(double x, double* so, double* co) {
  const double fn = rint(x * 6.36619772367581382433e-01);
  const double r = fma(-fn, 1.57079632673412561417e+00, x);
  const double w = fn * 6.07710050650619224932e-11;
  const double y0 = r - w, y1 = (r - y0) - w;  // reduced argument, head and tail
  const double z = y0 * y0, v = z * y0;
  const double rs = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                        8.33333333332248946124e-03);
  const double s = y0 - (fma(z, fma(0.5, y1, -(v * rs)), -y1) - v * -1.66666666666666324348e-01);
  const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                   -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double hz = 0.5 * z, ww = 1.0 - hz;
  const double c = ww + (((1.0 - ww) - hz) + fma(z, rc, -(y0 * y1)));
  const int n = (int)fn;
  const double sv = (n & 1) ? c : s, cv = (n & 1) ? s : c;
  *so = (n & 2) ? -sv : sv;
  *co = ((n + 1) & 2) ? -cv : cv;
}

// sin and cos of an fp32-valued angle |x| <= pi from a table: x = k / 64 + d with k = rint(64 |x|) (d is EXACT: x is a float, k / 64 a
// multiple of 2^-6 below 4), sin(k / 64) and cos(k / 64) correctly rounded (apd_sincos_tab.hpp, 202 rows, generated in 60-digit decimal
// arithmetic), sin d and 1 - cos d from three-term series (|d| <= 2^-7: the next terms are below 4e-22), and the addition theorems written
// around the table value: sin x = S + (C sin d - S u), cos x = C - (S sin d + C u), u = 1 - cos d.  About 23 fp64 operations and one
// 16-byte load against ~42 operations of sincos_pi: the three pairs of a point were 42 us of a 0.73 ms step (measured by evaluating
// them twice, docs/experiments.md round 5).  Absolute error ~1.5e-16; RELATIVE accuracy is lost next to a zero of the result (cos of
// an angle a few ulps from pi / 2: 5e-11) -- which is why the angle of arrival, whose cosine is a DIVISOR (A:170-171), keeps sincos_pi
// and only elevation and azimuth, whose sines and cosines are rotation-matrix entries, come from the table.
#include "apd_sincos_tab.hpp"
static __device__ const double apd_sincos_table[2 * APD_SINCOS_TAB_N] = APD_SINCOS_TAB_INIT;
__device__ __forceinline__ void sincos_tab(double x, double* so, double* co) {
#if defined(__HIP_DEVICE_COMPILE__)
#define APD_TAB_AS1 __attribute__((address_space(1)))
#else
#define APD_TAB_AS1  // (the host pass only parses device functions)
#endif
  const double ax = fabs(x);
  const double fk = rint(ax * 64.0);
  const double d = fma(-fk, 0.015625, ax);
  const int k = min((int)fk, APD_SINCOS_TAB_N - 1);
  const double2 sc = ((const APD_TAB_AS1 double2*)apd_sincos_table)[k];
  const double S = sc.x, C = sc.y;
  const double z = d * d;
  const double sd = fma(d * z, fma(z, fma(z, -1.98412698412698412698e-04, 8.33333333333333333333e-03), -1.66666666666666666667e-01), d);
  const double u = z * fma(z, fma(z, 1.38888888888888888889e-03, -4.16666666666666666667e-02), 0.5);
  const double s = S + fma(C, sd, -(S * u));
  const double c = C - fma(S, sd, C * u);
  *so = x < 0.0 ? -s : s;
  *co = c;
#undef APD_TAB_AS1
}

// R * C * R^T for symmetric C (R = rows r0,r1,r2 of a Rigid)
APD_HD Sym3 sym3_rotate(const Rigid& T, const Sym3& c) {
  // RC = R * C
  double rc[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double a = T.m[4 * i], b = T.m[4 * i + 1], d = T.m[4 * i + 2];
    rc[3 * i + 0] = a * c.xx + b * c.xy + d * c.xz;
    rc[3 * i + 1] = a * c.xy + b * c.yy + d * c.yz;
    rc[3 * i + 2] = a * c.xz + b * c.yz + d * c.zz;
  }
  Sym3 o;
  o.xx = rc[0] * T.m[0] + rc[1] * T.m[1] + rc[2] * T.m[2];
  o.xy = rc[0] * T.m[4] + rc[1] * T.m[5] + rc[2] * T.m[6];
  o.xz = rc[0] * T.m[8] + rc[1] * T.m[9] + rc[2] * T.m[10];
  o.yy = rc[3] * T.m[4] + rc[4] * T.m[5] + rc[5] * T.m[6];
  o.yz = rc[3] * T.m[8] + rc[4] * T.m[9] + rc[5] * T.m[10];
  o.zz = rc[6] * T.m[8] + rc[7] * T.m[9] + rc[8] * T.m[10];
  return o;
}

// inverse of a symmetric 3x3 by cofactors (stands in for Matrix4d::inverse() of blkdiag(C,1),
// fast_apdgicp_impl.hpp:191)
APD_HD Sym3 sym3_inverse(const Sym3& a) {
  const double c00 = a.yy * a.zz - a.yz * a.yz;
  const double c01 = a.yz * a.xz - a.xy * a.zz;
  const double c02 = a.xy * a.yz - a.yy * a.xz;
  const double det = a.xx * c00 + a.xy * c01 + a.xz * c02;
  const double id = 1.0 / det;
  Sym3 r;
  r.xx = c00 * id;
  r.xy = c01 * id;
  r.xz = c02 * id;
  r.yy = (a.xx * a.zz - a.xz * a.xz) * id;
  r.yz = (a.xy * a.xz - a.xx * a.yz) * id;
  r.zz = (a.xx * a.yy - a.xy * a.xy) * id;
  return r;
}

// 1 / sqrt(x) for a finite x >= 1 to within an ulp or two: v_rsq_f64 and two Newton steps (the IEEE route is a square root AND a
// division: 23 fp64 instructions against 9)
APD_HD double rsqrt_pos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const double e = __builtin_fma(-x * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
  }
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}

// One Jacobi rotation zeroing a_pq of a symmetric 3x3; r is the third index.
// (app, aqq, apq, arp, arq) are the affected entries, (v?p, v?q) the two eigenvector columns.
// t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq), written without theta: numerator and denominator
// times |2 apq| -- one division and one square root per rotation instead of three and two; c = 1 / sqrt(t^2 + 1) by rsqrt_pos.
// (theta = 0 takes the root t = +1 like `theta >= 0` did.)  The algebra is allowed to contract (fma): the result is compared by
// tolerance (the reference runs Eigen's JacobiSVD), and every covariance kernel runs this one function, so they stay bitwise equal.
APD_HD void jacobi_rot(double& app, double& aqq, double& apq, double& arp, double& arq, double& v0p, double& v0q, double& v1p, double& v1q,
                       double& v2p, double& v2q) {
#pragma clang fp contract(fast)
  if (apq == 0.0) return;
  const double d = aqq - app, h = 2.0 * apq;
  const bool pos = d == 0.0 || ((d > 0.0) == (apq > 0.0));
  const double t = (pos ? fabs(h) : -fabs(h)) / (fabs(d) + sqrt(d * d + h * h));
  const double c = rsqrt_pos(t * t + 1.0), s = t * c;
  app = app - t * apq;
  aqq = aqq + t * apq;
  apq = 0.0;
  const double rp = arp, rq = arq;
  arp = c * rp - s * rq;
  arq = s * rp + c * rq;
  double a, b;
  a = v0p, b = v0q, v0p = c * a - s * b, v0q = s * a + c * b;
  a = v1p, b = v1q, v1p = c * a - s * b, v1q = s * a + c * b;
  a = v2p, b = v2q, v2p = c * a - s * b, v2q = s * a + c * b;
}

// Symmetric eigen-decomposition, eigenvalues descending (w0>=w1>=w2), U columns = eigenvectors
// (u[3*row+col]).  Stands in for Eigen::JacobiSVD<Matrix3d> on a symmetric PSD input
// (fast_apdgicp_impl.hpp:337).
APD_HD void sym3_eig(const Sym3& A, double w[3], double u[9]) {
  double a00 = A.xx, a01 = A.xy, a02 = A.xz, a11 = A.yy, a12 = A.yz, a22 = A.zz;
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  for (int sweep = 0; sweep < 32; sweep++) {
#pragma clang fp contract(fast)
    const double off = a01 * a01 + a02 * a02 + a12 * a12;
    const double diag = a00 * a00 + a11 * a11 + a22 * a22;
    if (off == 0.0 || off <= 1e-40 * diag) break;
    jacobi_rot(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21);  // (p,q)=(0,1), r=2
    jacobi_rot(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22);  // (0,2), r=1
    jacobi_rot(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22);  // (1,2), r=0
  }
  // sort descending, carrying columns
#define APD_SWAPCOL(wa, wb, c0a, c1a, c2a, c0b, c1b, c2b) \
  if (wa < wb) {                                          \
    double t_;                                            \
    t_ = wa, wa = wb, wb = t_;                            \
    t_ = c0a, c0a = c0b, c0b = t_;                        \
    t_ = c1a, c1a = c1b, c1b = t_;                        \
    t_ = c2a, c2a = c2b, c2b = t_;                        \
  }
  APD_SWAPCOL(a00, a11, v00, v10, v20, v01, v11, v21)
  APD_SWAPCOL(a00, a22, v00, v10, v20, v02, v12, v22)
  APD_SWAPCOL(a11, a22, v01, v11, v21, v02, v12, v22)
#undef APD_SWAPCOL
  w[0] = a00, w[1] = a11, w[2] = a22;
  u[0] = v00, u[1] = v01, u[2] = v02, u[3] = v10, u[4] = v11, u[5] = v12, u[6] = v20, u[7] = v21, u[8] = v22;
}

// U * diag(vals) * U^T
APD_HD Sym3 sym3_from_eig(const double u[9], double l0, double l1, double l2) {
#pragma clang fp contract(fast)
  Sym3 o;
  o.xx = u[0] * l0 * u[0] + u[1] * l1 * u[1] + u[2] * l2 * u[2];
  o.xy = u[0] * l0 * u[3] + u[1] * l1 * u[4] + u[2] * l2 * u[5];
  o.xz = u[0] * l0 * u[6] + u[1] * l1 * u[7] + u[2] * l2 * u[8];
  o.yy = u[3] * l0 * u[3] + u[4] * l1 * u[4] + u[5] * l2 * u[5];
  o.yz = u[3] * l0 * u[6] + u[4] * l1 * u[7] + u[5] * l2 * u[8];
  o.zz = u[6] * l0 * u[6] + u[7] * l1 * u[7] + u[8] * l2 * u[8];
  return o;
}

// fast_apdgicp_impl.hpp:326-357 applied to the population covariance `cov`.
// returns false for an unknown mode (the reference aborts, :341-343)
APD_HD bool regularize_cov(int mode, const Sym3& cov, Sym3& out) {
  if (mode == 0) {  // NONE
    out = cov;
    return true;
  }
  if (mode == 4) {  // FROBENIUS: ((C+lI)^-1 / ||(C+lI)^-1||_F)^-1 = (C+lI) * ||(C+lI)^-1||_F
    Sym3 C = cov;
    C.xx += 1e-3, C.yy += 1e-3, C.zz += 1e-3;
    Sym3 Ci = sym3_inverse(C);
    const double nf = sqrt(Ci.xx * Ci.xx + Ci.yy * Ci.yy + Ci.zz * Ci.zz + 2.0 * (Ci.xy * Ci.xy + Ci.xz * Ci.xz + Ci.yz * Ci.yz));
    Ci.xx /= nf, Ci.xy /= nf, Ci.xz /= nf, Ci.yy /= nf, Ci.yz /= nf, Ci.zz /= nf;
    out = sym3_inverse(Ci);
    return true;
  }
  double w[3], u[9];
  sym3_eig(cov, w, u);
  double l0, l1, l2;
  if (mode == 3) {  // PLANE
    l0 = 1.0, l1 = 1.0, l2 = 1e-3;
  } else if (mode == 1) {  // MIN_EIG
    l0 = fmax(w[0], 1e-3), l1 = fmax(w[1], 1e-3), l2 = fmax(w[2], 1e-3);
  } else if (mode == 2) {  // NORMALIZED_MIN_EIG
    const double mx = fmax(w[0], fmax(w[1], w[2]));
    l0 = fmax(w[0] / mx, 1e-3), l1 = fmax(w[1] / mx, 1e-3), l2 = fmax(w[2] / mx, 1e-3);
  } else {
    out = Sym3{0, 0, 0, 0, 0, 0};
    return false;
  }
  out = sym3_from_eig(u, l0, l1, l2);
  return true;
}

// so3_exp (so3/so3.hpp:59-78) followed by Quaterniond::toRotationMatrix(); writes the rotation
// block of `delta` and d[3..5] into its translation (lsq_registration_impl.hpp:115-117,140-142).
APD_HD Rigid make_delta(const double d[6]) {
  const double theta_sq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    const double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    const double theta = sqrt(theta_sq), half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  const double w = real, x = imag * d[0], y = imag * d[1], z = imag * d[2];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Rigid r;
  r.m[0] = 1 - (tyy + tzz), r.m[1] = txy - twz, r.m[2] = txz + twy, r.m[3] = d[3];
  r.m[4] = txy + twz, r.m[5] = 1 - (txx + tzz), r.m[6] = tyz - twx, r.m[7] = d[4];
  r.m[8] = txz - twy, r.m[9] = tyz + twx, r.m[10] = 1 - (txx + tyy), r.m[11] = d[5];
  return r;
}

// is_converged (lsq_registration_impl.hpp:83-92): element-wise on R-I and t, not an angle.
APD_HD bool is_converged(const Rigid& delta, double rot_eps, double trans_eps) {
  double rmax = 0.0, tmax = 0.0;
  bool nan = false;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double v = 1.0 / rot_eps * fabs(delta.m[4 * i + j] - (i == j ? 1.0 : 0.0));
      nan |= (v != v);
      rmax = fmax(rmax, v);
    }
    const double v = 1.0 / trans_eps * fabs(delta.m[4 * i + 3]);
    nan |= (v != v);
    tmax = fmax(tmax, v);
  }
  if (nan) return false;  // epsilon == 0 with a zero entry: inf*0; the reference's maxCoeff is then unspecified
  return fmax(rmax, tmax) < 1.0;
}

// Solve (H + lambda I) x = -b for a symmetric positive semi-definite 6x6 (H column-major).
// LDL^T without pivoting (fully unrolled, register resident).  Eigen::LDLT pivots on the diagonal
// (lsq_registration_impl.hpp:112,137); for the SPD systems of this path both are backward stable
// and agree to ~cond*eps.  A vanishing pivot zeroes that unknown (LDLT's pseudo-inverse rule), so
// H == 0 (no correspondences) gives x == 0 like the reference.
APD_HD void solve6_spd(const double* H, double lambda, const double* b, double* x) {
  double L[6][6];
  double D[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double dj = H[j + 6 * j] + lambda;
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < j) dj -= L[j][k] * L[j][k] * D[k];
    D[j] = dj;
    const bool ok = fabs(dj) > 1e-300;
    const double inv = ok ? 1.0 / dj : 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      if (i > j) {
        double v = H[i + 6 * j];
#pragma unroll
        for (int k = 0; k < 6; k++)
          if (k < j) v -= L[i][k] * L[j][k] * D[k];
        L[i][j] = v * inv;
      }
    }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double v = -b[i];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < i) v -= L[i][k] * y[k];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = fabs(D[i]) > 1e-300 ? y[i] / D[i] : 0.0;
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double v = y[i];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k > i) v -= L[k][i] * x[k];
    x[i] = v;
  }
}

// The same elimination with its triangular factor, pivots and intermediate vector in caller-provided memory (48 doubles of
// LDS in the last block of k_linearize / k_error, where ONE lane runs it): held in registers they were 48 doubles that the
// per-point pass of the kernel around it had to spill to scratch for.  Same operations in the same order: the same bits.
APD_HD void solve6_spd_ws(const double* H, double lambda, const double* b, double* x, double* ws) {
  double* L = ws;        // [6][6], strictly lower part used
  double* D = ws + 36;   // [6]
  double* y = ws + 42;   // [6]
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double dj = H[j + 6 * j] + lambda;
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < j) dj -= L[6 * j + k] * L[6 * j + k] * D[k];
    D[j] = dj;
    const bool ok = fabs(dj) > 1e-300;
    const double inv = ok ? 1.0 / dj : 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      if (i > j) {
        double v = H[i + 6 * j];
#pragma unroll
        for (int k = 0; k < 6; k++)
          if (k < j) v -= L[6 * i + k] * L[6 * j + k] * D[k];
        L[6 * i + j] = v * inv;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double v = -b[i];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < i) v -= L[6 * i + k] * y[k];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = fabs(D[i]) > 1e-300 ? y[i] / D[i] : 0.0;
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double v = y[i];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k > i) v -= L[6 * k + i] * x[k];
    x[i] = v;
  }
}

}  // namespace apd
