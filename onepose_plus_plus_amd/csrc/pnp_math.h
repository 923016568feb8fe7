// Perspective-three-point solver + pose utilities shared by the HIP PnP-RANSAC kernels
// (pnp.hip) and the host unit tests (compiled with g++: tests/test_pnp_cpu.py).
//
// Replaces, on the GPU, the pose step that follows the matcher on every caller of the hot path:
//   ransac_PnP   /root/reference/src/utils/metric_utils.py:121-204
//   (cv2.solvePnPRansac(EPNP, 10000 iterations) -> R|t ; accuracy-level parity only: OpenCV's
//    RANSAC draws from its own RNG)
// Minimal solver: Grunert's P3P (quartic in the depth ratio, as reviewed by Haralick et al.,
// IJCV 1994), all arithmetic in double.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define OPP_HD __host__ __device__ inline
#else
#define OPP_HD inline
#endif

struct OppPose {
  double R[9];  // row-major, X_cam = R * X_world + t
  double t[3];
};

OPP_HD double opp_dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
OPP_HD void opp_cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
OPP_HD double opp_norm3(const double* a) { return sqrt(opp_dot3(a, a)); }

// real roots of x^4 + b x^3 + c x^2 + d x + e = 0 (Ferrari via the resolvent cubic), polished by
// Newton steps on the original polynomial.  Returns the number of real roots written.
OPP_HD int opp_solve_quartic(double b, double c, double d, double e, double* roots) {
  // depressed quartic y^4 + p y^2 + q y + r, x = y - b/4
  const double b2 = b * b;
  const double p = c - 0.375 * b2;
  const double q = d - 0.5 * b * c + 0.125 * b2 * b;
  const double r = e - 0.25 * b * d + 0.0625 * b2 * c - (3.0 / 256.0) * b2 * b2;
  int n = 0;
  double ys[4];
  if (fabs(q) < 1e-14 * (1.0 + fabs(p) + fabs(r))) {  // biquadratic
    const double disc = p * p - 4.0 * r;
    if (disc >= 0.0) {
      const double sq = sqrt(disc);
      const double z1 = 0.5 * (-p + sq), z2 = 0.5 * (-p - sq);
      if (z1 >= 0.0) {
        ys[n++] = sqrt(z1);
        ys[n++] = -sqrt(z1);
      }
      if (z2 >= 0.0) {
        ys[n++] = sqrt(z2);
        ys[n++] = -sqrt(z2);
      }
    }
  } else {
    // resolvent cubic  m^3 + p m^2 + (p^2/4 - r) m - q^2/8 = 0 ; take a positive real root
    const double A = p, B = 0.25 * p * p - r, C = -0.125 * q * q;
    // depressed cubic t^3 + P t + Q, m = t - A/3
    const double P = B - A * A / 3.0;
    const double Q = 2.0 * A * A * A / 27.0 - A * B / 3.0 + C;
    const double disc = 0.25 * Q * Q + P * P * P / 27.0;
    double m;
    if (disc >= 0.0) {
      const double sq = sqrt(disc);
      m = cbrt(-0.5 * Q + sq) + cbrt(-0.5 * Q - sq) - A / 3.0;
    } else {
      const double rr = sqrt(-P * P * P / 27.0);
      const double phi = acos(fmax(-1.0, fmin(1.0, -0.5 * Q / rr)));
      const double mag = 2.0 * sqrt(-P / 3.0);
      double best = -1e300;
      for (int k = 0; k < 3; ++k) {
        const double cand = mag * cos((phi + 2.0 * M_PI * k) / 3.0) - A / 3.0;
        if (cand > best) best = cand;
      }
      m = best;
    }
    if (m <= 0.0) return 0;
    const double s = sqrt(2.0 * m);
    const double t1 = -(2.0 * p + 2.0 * m) - 2.0 * q / s;  // discriminants of the two quadratics
    const double t2 = -(2.0 * p + 2.0 * m) + 2.0 * q / s;
    if (t1 >= 0.0) {
      ys[n++] = 0.5 * (s + sqrt(t1));
      ys[n++] = 0.5 * (s - sqrt(t1));
    }
    if (t2 >= 0.0) {
      ys[n++] = 0.5 * (-s + sqrt(t2));
      ys[n++] = 0.5 * (-s - sqrt(t2));
    }
  }
  for (int i = 0; i < n; ++i) {
    double x = ys[i] - 0.25 * b;
    for (int it = 0; it < 3; ++it) {  // Newton polish
      const double f = (((x + b) * x + c) * x + d) * x + e;
      const double fp = ((4.0 * x + 3.0 * b) * x + 2.0 * c) * x + d;
      if (fabs(fp) < 1e-300) break;
      x -= f / fp;
    }
    roots[i] = x;
  }
  return n;
}

// P3P.  y[3][3]: unit bearing vectors in the camera frame; x[3][3]: world points.
// Writes up to 4 poses; returns their number.
OPP_HD int opp_p3p_grunert(const double y[3][3], const double x[3][3], OppPose* out) {
  double d12[3], d13[3], d23[3];
  for (int k = 0; k < 3; ++k) {
    d12[k] = x[0][k] - x[1][k];
    d13[k] = x[0][k] - x[2][k];
    d23[k] = x[1][k] - x[2][k];
  }
  const double a2 = opp_dot3(d23, d23), b2 = opp_dot3(d13, d13), c2 = opp_dot3(d12, d12);  // a=|x2x3| b=|x1x3| c=|x1x2|
  if (a2 < 1e-24 || b2 < 1e-24 || c2 < 1e-24) return 0;
  const double ca = opp_dot3(y[1], y[2]), cb = opp_dot3(y[0], y[2]), cg = opp_dot3(y[0], y[1]);
  const double k1 = (a2 - c2) / b2, k2 = (a2 + c2) / b2, k3 = (b2 - c2) / b2, k4 = (b2 - a2) / b2;
  const double A4 = (k1 - 1.0) * (k1 - 1.0) - 4.0 * (c2 / b2) * ca * ca;
  const double A3 = 4.0 * (k1 * (1.0 - k1) * cb - (1.0 - k2) * ca * cg + 2.0 * (c2 / b2) * ca * ca * cb);
  const double A2 = 2.0 * (k1 * k1 - 1.0 + 2.0 * k1 * k1 * cb * cb + 2.0 * k3 * ca * ca - 4.0 * k2 * ca * cb * cg + 2.0 * k4 * cg * cg);
  const double A1 = 4.0 * (-k1 * (1.0 + k1) * cb + 2.0 * (a2 / b2) * cg * cg * cb - (1.0 - k2) * ca * cg);
  const double A0 = (1.0 + k1) * (1.0 + k1) - 4.0 * (a2 / b2) * cg * cg;
  if (fabs(A4) < 1e-14) return 0;
  double vs[4];
  const int nr = opp_solve_quartic(A3 / A4, A2 / A4, A1 / A4, A0 / A4, vs);
  int n = 0;
  for (int i = 0; i < nr; ++i) {
    const double v = vs[i];
    if (!(v > 0.0)) continue;
    const double den = 2.0 * (cg - v * ca);
    if (fabs(den) < 1e-12) continue;
    const double u = ((k1 - 1.0) * v * v - 2.0 * k1 * cb * v + 1.0 + k1) / den;
    if (!(u > 0.0)) continue;
    const double s1sq = b2 / (1.0 + v * v - 2.0 * v * cb);
    if (!(s1sq > 0.0)) continue;
    const double s1 = sqrt(s1sq), s2 = u * s1, s3 = v * s1;
    // camera-frame points
    double p[3][3];
    for (int k = 0; k < 3; ++k) {
      p[0][k] = s1 * y[0][k];
      p[1][k] = s2 * y[1][k];
      p[2][k] = s3 * y[2][k];
    }
    // orthonormal frames of the two triangles -> R = Fc * Fw^T
    double fw[3][3], fc[3][3], tmp[3];
    for (int f = 0; f < 2; ++f) {
      const double(*q)[3] = f == 0 ? x : p;
      double(*F)[3] = f == 0 ? fw : fc;
      double e1[3], e2[3], e3[3], w[3];
      for (int k = 0; k < 3; ++k) {
        e1[k] = q[1][k] - q[0][k];
        w[k] = q[2][k] - q[0][k];
      }
      const double n1 = opp_norm3(e1);
      for (int k = 0; k < 3; ++k) e1[k] /= n1;
      opp_cross3(e1, w, e3);
      const double n3 = opp_norm3(e3);
      if (n3 < 1e-18) return n;
      for (int k = 0; k < 3; ++k) e3[k] /= n3;
      opp_cross3(e3, e1, e2);
      for (int k = 0; k < 3; ++k) {
        F[0][k] = e1[k];
        F[1][k] = e2[k];
        F[2][k] = e3[k];
      }
    }
    OppPose P;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) P.R[r * 3 + c] = fc[0][r] * fw[0][c] + fc[1][r] * fw[1][c] + fc[2][r] * fw[2][c];
    for (int r = 0; r < 3; ++r) {
      tmp[r] = P.R[r * 3 + 0] * x[0][0] + P.R[r * 3 + 1] * x[0][1] + P.R[r * 3 + 2] * x[0][2];
      P.t[r] = p[0][r] - tmp[r];
    }
    out[n++] = P;
  }
  return n;
}

// squared reprojection error (pixels) of world point X under pose P and intrinsics K (fx, fy, cx, cy);
// returns a huge value for points behind the camera
OPP_HD double opp_reproj_err2(const OppPose& P, const double* K4, const double* X, const double* uv) {
  const double xc = P.R[0] * X[0] + P.R[1] * X[1] + P.R[2] * X[2] + P.t[0];
  const double yc = P.R[3] * X[0] + P.R[4] * X[1] + P.R[5] * X[2] + P.t[1];
  const double zc = P.R[6] * X[0] + P.R[7] * X[1] + P.R[8] * X[2] + P.t[2];
  if (!(zc > 1e-12)) return 1e300;
  const double du = K4[0] * xc / zc + K4[2] - uv[0];
  const double dv = K4[1] * yc / zc + K4[3] - uv[1];
  return du * du + dv * dv;
}

// R <- exp([w]x) * R  (Rodrigues)
OPP_HD void opp_rot_update(double* R, const double* w) {
  const double th = opp_norm3(w);
  double E[9];
  if (th < 1e-12) {
    E[0] = 1; E[1] = -w[2]; E[2] = w[1];
    E[3] = w[2]; E[4] = 1; E[5] = -w[0];
    E[6] = -w[1]; E[7] = w[0]; E[8] = 1;
  } else {
    const double k[3] = {w[0] / th, w[1] / th, w[2] / th};
    const double c = cos(th), s = sin(th), v = 1.0 - c;
    E[0] = c + k[0] * k[0] * v;        E[1] = k[0] * k[1] * v - k[2] * s; E[2] = k[0] * k[2] * v + k[1] * s;
    E[3] = k[1] * k[0] * v + k[2] * s; E[4] = c + k[1] * k[1] * v;        E[5] = k[1] * k[2] * v - k[0] * s;
    E[6] = k[2] * k[0] * v - k[1] * s; E[7] = k[2] * k[1] * v + k[0] * s; E[8] = c + k[2] * k[2] * v;
  }
  double N[9];
  for (int r = 0; r < 3; ++r)
    for (int c2 = 0; c2 < 3; ++c2) N[r * 3 + c2] = E[r * 3] * R[c2] + E[r * 3 + 1] * R[3 + c2] + E[r * 3 + 2] * R[6 + c2];
  for (int i = 0; i < 9; ++i) R[i] = N[i];
}

// solves the symmetric positive (semi-)definite 6x6 system H d = g in place (Gaussian elimination with
// partial pivoting); returns false if singular
OPP_HD bool opp_solve6(double* H, double* g) {
  for (int i = 0; i < 6; ++i) {
    int piv = i;
    for (int r = i + 1; r < 6; ++r)
      if (fabs(H[r * 6 + i]) > fabs(H[piv * 6 + i])) piv = r;
    if (fabs(H[piv * 6 + i]) < 1e-18) return false;
    if (piv != i) {
      for (int c = 0; c < 6; ++c) {
        const double t = H[i * 6 + c];
        H[i * 6 + c] = H[piv * 6 + c];
        H[piv * 6 + c] = t;
      }
      const double t = g[i];
      g[i] = g[piv];
      g[piv] = t;
    }
    for (int r = i + 1; r < 6; ++r) {
      const double f = H[r * 6 + i] / H[i * 6 + i];
      for (int c = i; c < 6; ++c) H[r * 6 + c] -= f * H[i * 6 + c];
      g[r] -= f * g[i];
    }
  }
  for (int i = 5; i >= 0; --i) {
    double s = g[i];
    for (int c = i + 1; c < 6; ++c) s -= H[i * 6 + c] * g[c];
    g[i] = s / H[i * 6 + i];
  }
  return true;
}

// accumulates one correspondence into the Gauss-Newton normal equations (upper triangle of H 6x6, g 6)
// for the update (w, dt): X_cam' = exp([w]x) (R X + t) + dt ; residual in pixels
OPP_HD void opp_gn_accumulate(const OppPose& P, const double* K4, const double* X, const double* uv, double* H, double* g) {
  const double xc = P.R[0] * X[0] + P.R[1] * X[1] + P.R[2] * X[2] + P.t[0];
  const double yc = P.R[3] * X[0] + P.R[4] * X[1] + P.R[5] * X[2] + P.t[1];
  const double zc = P.R[6] * X[0] + P.R[7] * X[1] + P.R[8] * X[2] + P.t[2];
  const double iz = 1.0 / zc;
  const double ru = K4[0] * xc * iz + K4[2] - uv[0];
  const double rv = K4[1] * yc * iz + K4[3] - uv[1];
  // d(u,v)/d(Xc)
  const double ju[3] = {K4[0] * iz, 0.0, -K4[0] * xc * iz * iz};
  const double jv[3] = {0.0, K4[1] * iz, -K4[1] * yc * iz * iz};
  // d(Xc)/d(w) = -[Xc]x ; d(Xc)/d(dt) = I
  double Ju[6], Jv[6];
  // -[Xc]x = [[0, zc, -yc], [-zc, 0, xc], [yc, -xc, 0]]
  Ju[0] = ju[0] * 0.0 + ju[1] * (-zc) + ju[2] * yc;
  Ju[1] = ju[0] * zc + ju[1] * 0.0 + ju[2] * (-xc);
  Ju[2] = ju[0] * (-yc) + ju[1] * xc + ju[2] * 0.0;
  Jv[0] = jv[0] * 0.0 + jv[1] * (-zc) + jv[2] * yc;
  Jv[1] = jv[0] * zc + jv[1] * 0.0 + jv[2] * (-xc);
  Jv[2] = jv[0] * (-yc) + jv[1] * xc + jv[2] * 0.0;
  for (int k = 0; k < 3; ++k) {
    Ju[3 + k] = ju[k];
    Jv[3 + k] = jv[k];
  }
  for (int r = 0; r < 6; ++r) {
    for (int c = r; c < 6; ++c) H[r * 6 + c] += Ju[r] * Ju[c] + Jv[r] * Jv[c];
    g[r] -= Ju[r] * ru + Jv[r] * rv;
  }
}
