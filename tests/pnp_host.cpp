// Host build of the PnP math (onepose_plus_plus_amd/csrc/pnp_math.h) for CPU unit tests.
#include "../onepose_plus_plus_amd/csrc/pnp_math.h"

extern "C" {
int t_quartic(double b, double c, double d, double e, double* roots) { return opp_solve_quartic(b, c, d, e, roots); }

// y [3][3] unit bearings, x [3][3] world points -> poses as 12 doubles each (R row-major | t)
int t_p3p(const double* y, const double* x, double* out) {
  double yy[3][3], xx[3][3];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      yy[i][k] = y[i * 3 + k];
      xx[i][k] = x[i * 3 + k];
    }
  OppPose P[4];
  const int n = opp_p3p_grunert(yy, xx, P);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 9; ++k) out[i * 12 + k] = P[i].R[k];
    for (int k = 0; k < 3; ++k) out[i * 12 + 9 + k] = P[i].t[k];
  }
  return n;
}

// Gauss-Newton refinement over n correspondences; pose in/out as 12 doubles
int t_refine(double* pose, const double* K4, const double* X, const double* uv, int n, int iters) {
  OppPose P;
  for (int k = 0; k < 9; ++k) P.R[k] = pose[k];
  for (int k = 0; k < 3; ++k) P.t[k] = pose[9 + k];
  for (int it = 0; it < iters; ++it) {
    double H[36] = {0}, g[6] = {0};
    for (int i = 0; i < n; ++i) opp_gn_accumulate(P, K4, X + 3 * i, uv + 2 * i, H, g);
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < r; ++c) H[r * 6 + c] = H[c * 6 + r];
    if (!opp_solve6(H, g)) return -1;
    opp_rot_update(P.R, g);
    double nt[3];
    double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    opp_rot_update(E, g);
    for (int r = 0; r < 3; ++r) nt[r] = E[r * 3] * P.t[0] + E[r * 3 + 1] * P.t[1] + E[r * 3 + 2] * P.t[2] + g[3 + r];
    for (int r = 0; r < 3; ++r) P.t[r] = nt[r];
  }
  for (int k = 0; k < 9; ++k) pose[k] = P.R[k];
  for (int k = 0; k < 3; ++k) pose[9 + k] = P.t[k];
  return 0;
}

double t_reproj(const double* pose, const double* K4, const double* X, const double* uv) {
  OppPose P;
  for (int k = 0; k < 9; ++k) P.R[k] = pose[k];
  for (int k = 0; k < 3; ++k) P.t[k] = pose[9 + k];
  return opp_reproj_err2(P, K4, X, uv);
}
}
