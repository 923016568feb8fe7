// PnP-RANSAC on the GPU: the pose step that follows the 2D-3D matcher on every caller of the hot
// path.
//
// Replaces  ransac_PnP / cv2.solvePnPRansac(EPNP, 10000 iterations)
//           /root/reference/src/utils/metric_utils.py:121-204 (called from :251-259, demo.py:132)
// so that the matches never leave the device: at a few hundred images/s the reference's CPU
// RANSAC + D2H sync is the end-to-end limiter.  Accuracy-level parity only (OpenCV draws its
// samples from its own RNG); validated against synthetic ground-truth poses (tests/test_pnp_gpu.py).
//
//   1. hypotheses : one thread per hypothesis: 4 distinct matches from a counter-based hash RNG,
//                   Grunert P3P on three (double), the 4th disambiguates
//   2. scoring    : one wave per hypothesis counts reprojection inliers over all matches
//   3. selection  : arg-max inlier count (ties -> lowest hypothesis index: deterministic)
//   4. refinement : Gauss-Newton on the inlier set (re-evaluated every iteration), one workgroup
#include "opp_common.h"
#include "pnp_math.h"

namespace {

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

struct PnpParams {
  double K4[4];   // fx, fy, cx, cy
  double thr2;    // squared reprojection threshold (px^2)
  double scale;   // world-point scale (reference: configs["point_cloud_rescale"])
  int n, iters;
  unsigned seed;
};

__global__ __launch_bounds__(256) void pnp_hypotheses_kernel(const float* __restrict__ p2, const float* __restrict__ p3,
                                                             PnpParams prm, double* __restrict__ hyp, int* __restrict__ valid) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= prm.iters) return;
  int idx[4];
  unsigned s = hash32(prm.seed ^ (0x9e3779b9u * (unsigned)(h + 1)));
  for (int k = 0; k < 4; ++k) {
    for (int tries = 0; tries < 64; ++tries) {
      s = hash32(s + 0x632be5abu);
      const int c = (int)(s % (unsigned)prm.n);
      bool dup = false;
      for (int j = 0; j < k; ++j) dup |= idx[j] == c;
      idx[k] = c;
      if (!dup) break;
    }
  }
  double y[3][3], x[3][3];
  for (int k = 0; k < 3; ++k) {
    const double u = ((double)p2[2 * idx[k]] - prm.K4[2]) / prm.K4[0];
    const double v = ((double)p2[2 * idx[k] + 1] - prm.K4[3]) / prm.K4[1];
    const double inv = 1.0 / sqrt(u * u + v * v + 1.0);
    y[k][0] = u * inv;
    y[k][1] = v * inv;
    y[k][2] = inv;
    for (int c = 0; c < 3; ++c) x[k][c] = (double)p3[3 * idx[k] + c] * prm.scale;
  }
  OppPose sol[4];
  const int ns = opp_p3p_grunert(y, x, sol);
  const double X4[3] = {(double)p3[3 * idx[3]] * prm.scale, (double)p3[3 * idx[3] + 1] * prm.scale,
                        (double)p3[3 * idx[3] + 2] * prm.scale};
  const double uv4[2] = {(double)p2[2 * idx[3]], (double)p2[2 * idx[3] + 1]};
  int best = -1;
  double best_e = 1e300;
  for (int i = 0; i < ns; ++i) {
    const double e = opp_reproj_err2(sol[i], prm.K4, X4, uv4);
    if (e < best_e) {
      best_e = e;
      best = i;
    }
  }
  const bool ok = best >= 0 && best_e <= prm.thr2 * 4.0;   // the 4th match must roughly agree
  valid[h] = ok ? 1 : 0;
  if (ok) {
    for (int k = 0; k < 9; ++k) hyp[(size_t)h * 12 + k] = sol[best].R[k];
    for (int k = 0; k < 3; ++k) hyp[(size_t)h * 12 + 9 + k] = sol[best].t[k];
  }
}

__global__ __launch_bounds__(256) void pnp_score_kernel(const float* __restrict__ p2, const float* __restrict__ p3, PnpParams prm,
                                                        const double* __restrict__ hyp, const int* __restrict__ valid,
                                                        int* __restrict__ score) {
  const int lane = threadIdx.x & 63;
  const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (h >= prm.iters) return;
  int cnt = 0;
  if (valid[h]) {
    OppPose P;
    for (int k = 0; k < 9; ++k) P.R[k] = hyp[(size_t)h * 12 + k];
    for (int k = 0; k < 3; ++k) P.t[k] = hyp[(size_t)h * 12 + 9 + k];
    for (int i = lane; i < prm.n; i += 64) {
      const double X[3] = {(double)p3[3 * i] * prm.scale, (double)p3[3 * i + 1] * prm.scale, (double)p3[3 * i + 2] * prm.scale};
      const double uv[2] = {(double)p2[2 * i], (double)p2[2 * i + 1]};
      cnt += opp_reproj_err2(P, prm.K4, X, uv) <= prm.thr2 ? 1 : 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) score[h] = valid[h] ? cnt : -1;   // -1: degenerate sample, never the best
}

// single workgroup: best hypothesis, Gauss-Newton refinement on its inliers, final inlier mask
__global__ __launch_bounds__(256) void pnp_refine_kernel(const float* __restrict__ p2, const float* __restrict__ p3, PnpParams prm,
                                                         const double* __restrict__ hyp, const int* __restrict__ score,
                                                         int gn_iters, double* __restrict__ pose_out, int* __restrict__ mask,
                                                         int* __restrict__ n_inl, int* __restrict__ ok_out) {
  __shared__ int s_best[256], s_idx[256];
  __shared__ double s_red[4][27];
  __shared__ OppPose s_pose;
  __shared__ int s_cnt[4];
  __shared__ int s_fail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bs = -1, bi = 0x7fffffff;
  for (int h = tid; h < prm.iters; h += 256)
    if (score[h] > bs) {
      bs = score[h];
      bi = h;
    }
  s_best[tid] = bs;
  s_idx[tid] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      const int ob = s_best[tid + o], oi = s_idx[tid + o];
      if (ob > s_best[tid] || (ob == s_best[tid] && oi < s_idx[tid])) {
        s_best[tid] = ob;
        s_idx[tid] = oi;
      }
    }
    __syncthreads();
  }
  const int best_score = s_best[0], best_h = s_idx[0];
  if (tid == 0) {
    // fail only without any valid hypothesis (fewer than 4 matches, degenerate geometry): the reference's cv2.error
    // branch.  A best hypothesis with fewer than 4 inliers is still returned (state True, its few inliers), like
    // cv2.solvePnPRansac, which then reports success with an empty / tiny inlier set (metric_utils.py:194-196)
    s_fail = best_score < 0 ? 1 : 0;
    if (!s_fail) {
      for (int k = 0; k < 9; ++k) s_pose.R[k] = hyp[(size_t)best_h * 12 + k];
      for (int k = 0; k < 3; ++k) s_pose.t[k] = hyp[(size_t)best_h * 12 + 9 + k];
    }
  }
  __syncthreads();
  if (s_fail) {   // same failure convention as the reference's cv2.error branch: identity pose, no inliers
    if (tid < 12) pose_out[tid] = (tid == 0 || tid == 5 || tid == 10) ? 1.0 : 0.0;
    for (int i = tid; i < prm.n; i += 256) mask[i] = 0;
    if (tid == 0) {
      *n_inl = 0;
      *ok_out = 0;
    }
    return;
  }
  for (int it = 0; it <= gn_iters; ++it) {
    const OppPose P = s_pose;
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    int cnt = 0;
    const bool last = it == gn_iters;
    for (int i = tid; i < prm.n; i += 256) {
      const double X[3] = {(double)p3[3 * i] * prm.scale, (double)p3[3 * i + 1] * prm.scale, (double)p3[3 * i + 2] * prm.scale};
      const double uv[2] = {(double)p2[2 * i], (double)p2[2 * i + 1]};
      const bool in = opp_reproj_err2(P, prm.K4, X, uv) <= prm.thr2;
      cnt += in ? 1 : 0;
      if (last) {
        mask[i] = in ? 1 : 0;
      } else if (in) {
        double H[36], g[6];
#pragma unroll
        for (int k = 0; k < 36; ++k) H[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = 0.0;
        opp_gn_accumulate(P, prm.K4, X, uv, H, g);
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = r; c < 6; ++c) acc[q++] += H[r * 6 + c];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[21 + k] += g[k];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) s_cnt[wave] = cnt;
    if (!last) {
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) s_red[wave][k] = v;
      }
    }
    __syncthreads();
    if (tid == 0) {
      const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
      if (last) {
        *n_inl = total;
        *ok_out = 1;
      } else if (total >= 4) {
        double H[36], g[6];
        int q = 0;
        for (int r = 0; r < 6; ++r)
          for (int c = r; c < 6; ++c) {
            const double v = s_red[0][q] + s_red[1][q] + s_red[2][q] + s_red[3][q];
            H[r * 6 + c] = v;
            H[c * 6 + r] = v;
            ++q;
          }
        for (int k = 0; k < 6; ++k) g[k] = s_red[0][21 + k] + s_red[1][21 + k] + s_red[2][21 + k] + s_red[3][21 + k];
        if (opp_solve6(H, g)) {
          OppPose N = s_pose;
          opp_rot_update(N.R, g);
          double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
          opp_rot_update(E, g);
          for (int r = 0; r < 3; ++r)
            N.t[r] = E[r * 3] * s_pose.t[0] + E[r * 3 + 1] * s_pose.t[1] + E[r * 3 + 2] * s_pose.t[2] + g[3 + r];
          s_pose = N;
        }
      }
    }
    __syncthreads();
  }
  // [R | t / scale] row-major 3x4 (the reference divides tvec by the scale again, metric_utils.py:192)
  if (tid < 12) {
    const int r = tid / 4, c = tid % 4;
    pose_out[tid] = c < 3 ? s_pose.R[r * 3 + c] : s_pose.t[r] / prm.scale;
  }
}

}  // namespace

extern "C" size_t opp_pnp_workspace_bytes(int iterations) {
  return opp_align((size_t)iterations * 12 * sizeof(double)) + 2 * opp_align((size_t)iterations * sizeof(int)) + 1024;
}

extern "C" int opp_pnp_ransac(const float* pts2d, const float* pts3d, int n_points, const double* K4, double reproj_error_px,
                              double scale, int iterations, unsigned seed, int refine_iters, double* pose_out,
                              int* inlier_mask, int* n_inliers, int* ok, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OPP_CHECK_ARG(pts2d && pts3d && K4 && pose_out && inlier_mask && n_inliers && ok && ws, "pnp: null argument");
  OPP_CHECK_ARG(iterations > 0 && iterations <= (1 << 20) && reproj_error_px > 0 && scale > 0, "pnp: bad parameters");
  OPP_CHECK_ARG(ws_bytes >= opp_pnp_workspace_bytes(iterations), "pnp: workspace too small");
  PnpParams prm;
  for (int k = 0; k < 4; ++k) prm.K4[k] = K4[k];
  prm.thr2 = reproj_error_px * reproj_error_px;
  prm.scale = scale;
  prm.n = n_points;
  prm.iters = iterations;
  prm.seed = seed;
  char* base = (char*)ws;
  double* hyp = (double*)base;
  int* valid = (int*)(base + opp_align((size_t)iterations * 12 * sizeof(double)));
  int* score = (int*)((char*)valid + opp_align((size_t)iterations * sizeof(int)));
  if (n_points < 4) {   // cv2.solvePnPRansac raises -> reference returns identity / no inliers / state False
    prm.iters = 0;
    (void)hipMemsetAsync(score, 0, sizeof(int), stream);
  } else {
    hipLaunchKernelGGL(pnp_hypotheses_kernel, dim3(opp_cdiv(iterations, 256)), dim3(256), 0, stream, pts2d, pts3d, prm, hyp, valid);
    hipLaunchKernelGGL(pnp_score_kernel, dim3(opp_cdiv(iterations, 4)), dim3(256), 0, stream, pts2d, pts3d, prm, hyp, valid, score);
  }
  hipLaunchKernelGGL(pnp_refine_kernel, dim3(1), dim3(256), 0, stream, pts2d, pts3d, prm, hyp, score, refine_iters, pose_out,
                     inlier_mask, n_inliers, ok);
  OPP_CHECK_LAUNCH("pnp kernels");
  return OPP_OK;
}
