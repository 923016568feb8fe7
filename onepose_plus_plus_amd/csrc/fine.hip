// Fine stage: 5x5 window gather at 1/2 resolution (no unfold materialisation) and the
// sub-pixel expectation head.
//
// Reference:
//   FinePreprocess._forward  src/models/OnePosePlus/loftr_module/fine_preprocess.py:41-55
//   FineMatching             src/models/OnePosePlus/utils/fine_matching.py:28-110
//   kornia 0.4.1 dsnt.spatial_expectation2d / create_meshgrid (normalised {-1..1} grid, x fastest)
//
// The gather is HBM-bound: M * W*W * C * 4 bytes (12.8 KB per match) read from the NHWC fine
// feature map with lanes along the channel axis (512 B contiguous per window cell).
#include "opp_common.h"
#include "opp_internal.h"

namespace {

// win [M][WW][C] (token-major: M*WW rows of C) ; f3 [M][C] ; feat NHWC [Hf][Wf][ldf] ; bank [C][N]
__global__ __launch_bounds__(128) void fine_gather_kernel(const float* __restrict__ feat, int Hf, int Wf, int ldf,
                                                          const float* __restrict__ bank, int n_points,
                                                          const long long* __restrict__ i_ids,
                                                          const long long* __restrict__ j_ids, int wc, int stride,
                                                          int Wwin, int C, float* __restrict__ win, int ldw,
                                                          float* __restrict__ f3, int ld3) {
  const int m = blockIdx.x;
  const int j = (int)j_ids[m];
  const int jy = j / wc, jx = j - jy * wc;
  const int cy = jy * stride - Wwin / 2, cx = jx * stride - Wwin / 2;
  const int WW = Wwin * Wwin;
  // one 16-byte load per lane: a window cell is C * 4 contiguous bytes of the NHWC map (C % 4 == 0 checked by the launcher)
  const int c4n = C >> 2;
  for (int e = threadIdx.x; e < WW * c4n; e += blockDim.x) {
    const int r = e / c4n, c = (e - r * c4n) * 4;
    const int ky = r / Wwin, kx = r - ky * Wwin;
    const int y = cy + ky, x = cx + kx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)y < (unsigned)Hf && (unsigned)x < (unsigned)Wf)
      v = *reinterpret_cast<const float4*>(feat + ((size_t)y * Wf + x) * ldf + c);
    *reinterpret_cast<float4*>(win + ((size_t)m * WW + r) * ldw + c) = v;
  }
  const long long i = i_ids[m];
  for (int c = threadIdx.x; c < C; c += blockDim.x) f3[(size_t)m * ld3 + c] = bank[(size_t)c * n_points + i];
}

// one wave per match: heatmap = softmax(<f3, win_r> / sqrt(C)) over WW cells; expectation + std
__global__ __launch_bounds__(256) void fine_head_kernel(const float* __restrict__ f3, int ld3,
                                                        const float* __restrict__ win, int ldw, int M, int Wwin,
                                                        int C, float temp, const float* __restrict__ mkpts_c, float base_scale,
                                                        const float* __restrict__ qscale,
                                                        float* __restrict__ expec, float* __restrict__ mkpts_f) {
  const int lane = threadIdx.x & 63;
  const int m_raw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool live = m_raw < M;              // waves past the end redo the last match and store nothing
  const int m = live ? m_raw : M - 1;
  const int WW = Wwin * Wwin;   // <= 64 handled by one lane per cell
  // similarities with 16 lanes per window cell (8 channels each, 16-byte loads: a wave reads 4 cells = 2 KB per pass),
  // then handed to lane = cell through LDS for the softmax / expectation below
  __shared__ float sim_sh[4][64];
  const int wv = threadIdx.x >> 6, grp = lane >> 4, sub = lane & 15;
  for (int r0 = 0; r0 < WW; r0 += 4) {
    const int r = r0 + grp;
    float acc = 0.f;
    if (r < WW) {
      const float* w = win + ((size_t)m * WW + r) * ldw;
      const float* f = f3 + (size_t)m * ld3;
      for (int c = sub * 4; c < C; c += 64) {
        const float4 w4 = *reinterpret_cast<const float4*>(w + c);
        const float4 f4 = *reinterpret_cast<const float4*>(f + c);
        acc = fmaf(f4.x, w4.x, acc);
        acc = fmaf(f4.y, w4.y, acc);
        acc = fmaf(f4.z, w4.z, acc);
        acc = fmaf(f4.w, w4.w, acc);
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0 && r < WW) sim_sh[wv][r] = acc;
  }
  __syncthreads();
  float sim = -INFINITY;
  if (lane < WW) sim = temp * sim_sh[wv][lane];   // softmax_temp * sim_matrix, fine_matching.py:82-83
  float mx = sim;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float e = lane < WW ? expf(sim - mx) : 0.f;
  float tot = e;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
  const float p = e / tot;
  // normalised grid: (k / (W-1) - 0.5) * 2, x fastest
  float gx = 0.f, gy = 0.f;
  if (lane < WW) {
    const int ky = lane / Wwin, kx = lane - ky * Wwin;
    gx = ((float)kx / (float)(Wwin - 1) - 0.5f) * 2.f;
    gy = ((float)ky / (float)(Wwin - 1) - 0.5f) * 2.f;
  }
  float ex = gx * p, ey = gy * p, exx = gx * gx * p, eyy = gy * gy * p;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ex += __shfl_xor(ex, o, 64);
    ey += __shfl_xor(ey, o, 64);
    exx += __shfl_xor(exx, o, 64);
    eyy += __shfl_xor(eyy, o, 64);
  }
  if (lane == 0 && live) {
    const float vx = fmaxf(exx - ex * ex, 1e-10f);
    const float vy = fmaxf(eyy - ey * ey, 1e-10f);
    expec[3 * m + 0] = ex;
    expec[3 * m + 1] = ey;
    expec[3 * m + 2] = sqrtf(vx) + sqrtf(vy);
    const float half_w = (float)(Wwin / 2);
    // scale * query_image_scale[b][[1, 0]]  (fine_matching.py:104)
    const float qsx = qscale ? base_scale * qscale[1] : base_scale;
    const float qsy = qscale ? base_scale * qscale[0] : base_scale;
    mkpts_f[2 * m + 0] = mkpts_c[2 * m + 0] + (ex * half_w) * qsx;
    mkpts_f[2 * m + 1] = mkpts_c[2 * m + 1] + (ey * half_w) * qsy;
  }
}

// Backward of fine_head_kernel's expectation (the training step; fine_matching.py:63-94 under autograd): one wave per match.
//   sim_r = temp <f3, win_r>, p = softmax(sim), ex = sum gx p, ey = sum gy p, vx = max(sum gx^2 p - ex^2, 1e-10), std = sqrt(vx) + sqrt(vy)
//   g = d loss / d (ex, ey, std):  a_r = gx_r Dex + gy_r Dey + gx_r^2 Dxx + gy_r^2 Dyy  with  Dxx = g_std / (2 sqrt(vx)) [var >= 1e-10],
//   Dex = g_ex - 2 ex Dxx (y alike);  d sim_r = p_r (a_r - sum_s p_s a_s);  d f3 = temp sum_r d sim_r win_r;  d win_r = temp d sim_r f3
__global__ __launch_bounds__(256) void fine_head_bwd_kernel(const float* __restrict__ f3, int ld3, const float* __restrict__ win, int ldw, int M, int Wwin,
                                                            int C, float temp, const float* __restrict__ gexp, float* __restrict__ gf3,
                                                            float* __restrict__ gwin) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  __shared__ float sim_sh[4][64];
  const int wv = threadIdx.x >> 6, grp = lane >> 4, sub = lane & 15;
  const int WW = Wwin * Wwin;
  const bool live = m < M;
  const int mm = live ? m : M - 1;
  const float* f = f3 + (size_t)mm * ld3;
  for (int r0 = 0; r0 < WW; r0 += 4) {
    const int r = r0 + grp;
    float acc = 0.f;
    if (r < WW) {
      const float* w = win + ((size_t)mm * WW + r) * ldw;
      for (int c = sub * 4; c < C; c += 64) {
        const float4 w4 = *reinterpret_cast<const float4*>(w + c);
        const float4 f4 = *reinterpret_cast<const float4*>(f + c);
        acc = fmaf(f4.x, w4.x, acc);
        acc = fmaf(f4.y, w4.y, acc);
        acc = fmaf(f4.z, w4.z, acc);
        acc = fmaf(f4.w, w4.w, acc);
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0 && r < WW) sim_sh[wv][r] = acc;
  }
  __syncthreads();
  float sim = -INFINITY;
  if (lane < WW) sim = temp * sim_sh[wv][lane];
  float mx = sim;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float e = lane < WW ? expf(sim - mx) : 0.f;
  float tot = e;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
  const float p = e / tot;
  float gx = 0.f, gy = 0.f;
  if (lane < WW) {
    const int ky = lane / Wwin, kx = lane - ky * Wwin;
    gx = ((float)kx / (float)(Wwin - 1) - 0.5f) * 2.f;
    gy = ((float)ky / (float)(Wwin - 1) - 0.5f) * 2.f;
  }
  float ex = gx * p, ey = gy * p, exx = gx * gx * p, eyy = gy * gy * p;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ex += __shfl_xor(ex, o, 64);
    ey += __shfl_xor(ey, o, 64);
    exx += __shfl_xor(exx, o, 64);
    eyy += __shfl_xor(eyy, o, 64);
  }
  const float g_ex = gexp[3 * mm + 0], g_ey = gexp[3 * mm + 1], g_sd = gexp[3 * mm + 2];
  const float varx = exx - ex * ex, vary = eyy - ey * ey;
  const float dxx = varx >= 1e-10f ? g_sd * 0.5f / sqrtf(varx) : 0.f;     // torch.clamp(min) passes the gradient where var >= min
  const float dyy = vary >= 1e-10f ? g_sd * 0.5f / sqrtf(vary) : 0.f;
  const float dex = g_ex - 2.f * ex * dxx, dey = g_ey - 2.f * ey * dyy;
  const float a = gx * dex + gy * dey + gx * gx * dxx + gy * gy * dyy;
  float pa = p * a;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pa += __shfl_xor(pa, o, 64);
  const float dsim = lane < WW ? temp * (p * (a - pa)) : 0.f;               // includes the temperature factor of sim
  if (!live) return;
  // lanes over channels from here on: d f3[c] = sum_r dsim_r win[r][c],  d win[r][c] = dsim_r f3[c]
  // (every lane of the wave stays in the loop -- the shuffle reads dsim of lanes r < WW, which must be active -- and only the
  // loads / stores are guarded: C need not be a multiple of 64)
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    const bool okc = c < C;
    const float fc = okc ? f[c] : 0.f;
    float acc = 0.f;
    for (int r = 0; r < WW; ++r) {
      const float dr = __shfl(dsim, r, 64);
      if (okc) {
        acc = fmaf(dr, win[((size_t)m * WW + r) * ldw + c], acc);
        gwin[((size_t)m * WW + r) * ldw + c] = dr * fc;
      }
    }
    if (okc) gf3[(size_t)m * ld3 + c] = acc;
  }
}

// ---- match-driven fine branch (api.hip: opp_fine_patches) -------------------------------------------------------------------
// In eval mode the last three convolutions of the FPN fine branch (layer1_outconv, layer1_outconv2: backbone/resnet.py:154-157) are
// consumed only through the W x W windows around the M matches (fine_preprocess.py:41-55), so they are evaluated on a per-match patch
// pyramid instead of the whole 1/2-resolution map: (W+4)^2 pixels of l1 = conv1x1(x1) + up2x(x2_out) -> (W+2)^2 of u1 -> W^2 of the map.
// This kernel gathers, per match, the (W+4)^2 x1 rows and the bilinear x2 (align_corners=True) upsampling of x2_out at those pixels
// (resnet.py:155), with the arithmetic of the GEMM epilogue it replaces (gemm_mfma.hip, OPP_RES_BILINEAR2X) so that the patch path
// reproduces the dense map bit for bit.  Pixels outside the image are exact zeros (the zero padding of the convolutions that follow).
__global__ __launch_bounds__(256) void fine_patch_gather_kernel(const float* __restrict__ x1, int Hf, int Wf, int c1,
                                                                const float* __restrict__ x2o, int Hr, int Wr, int c2, float res_sy,
                                                                float res_sx, const long long* __restrict__ j_ids, int wc, int stride, int org,
                                                                int P, float* __restrict__ xa, float* __restrict__ up) {
  const int m = blockIdx.x;
  const int j = (int)j_ids[m];
  const int jy = j / wc, jx = j - jy * wc;
  const int y0p = jy * stride + org, x0p = jx * stride + org;
  const int PP = P * P;
  const int q1 = c1 >> 2, q2 = c2 >> 2;
  for (int e = threadIdx.x; e < PP * q1; e += blockDim.x) {
    const int r = e / q1, c = (e - r * q1) * 4;
    const int py = r / P, px = r - py * P;
    const int y = y0p + py, x = x0p + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)y < (unsigned)Hf && (unsigned)x < (unsigned)Wf) v = *reinterpret_cast<const float4*>(x1 + ((size_t)y * Wf + x) * c1 + c);
    *reinterpret_cast<float4*>(xa + ((size_t)m * PP + r) * c1 + c) = v;
  }
  for (int e = threadIdx.x; e < PP * q2; e += blockDim.x) {
    const int r = e / q2, c = (e - r * q2) * 4;
    const int py = r / P, px = r - py * P;
    const int oy = y0p + py, ox = x0p + px;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)oy < (unsigned)Hf && (unsigned)ox < (unsigned)Wf) {
      // same statements, same order, no FMA contraction: the residual term of the dense convolution's epilogue
#pragma clang fp contract(off)
      const float sy = res_sy * (float)oy;
      const float sx = res_sx * (float)ox;
      int y0 = (int)sy;
      if (y0 > Hr - 1) y0 = Hr - 1;
      int x0 = (int)sx;
      if (x0 > Wr - 1) x0 = Wr - 1;
      const int y1 = y0 + (y0 < Hr - 1 ? 1 : 0);
      const int x1i = x0 + (x0 < Wr - 1 ? 1 : 0);
      const float wy1 = fminf(fmaxf(sy - (float)y0, 0.f), 1.f);
      const float wx1 = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
      const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
      const float4 t0 = *reinterpret_cast<const float4*>(x2o + ((size_t)y0 * Wr + x0) * c2 + c);
      const float4 t1 = *reinterpret_cast<const float4*>(x2o + ((size_t)y0 * Wr + x1i) * c2 + c);
      const float4 t2 = *reinterpret_cast<const float4*>(x2o + ((size_t)y1 * Wr + x0) * c2 + c);
      const float4 t3 = *reinterpret_cast<const float4*>(x2o + ((size_t)y1 * Wr + x1i) * c2 + c);
      const float a00[4] = {t0.x, t0.y, t0.z, t0.w}, a01[4] = {t1.x, t1.y, t1.z, t1.w};
      const float a10[4] = {t2.x, t2.y, t2.z, t2.w}, a11[4] = {t3.x, t3.y, t3.z, t3.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float top = wx0 * a00[k] + wx1 * a01[k];
        const float bot = wx0 * a10[k] + wx1 * a11[k];
        o[k] = wy0 * top + wy1 * bot;
      }
    }
    *reinterpret_cast<float4*>(up + ((size_t)m * PP + r) * c2 + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// rows of a per-match patch buffer [M][P * P][ld] whose pixel lies outside the image -> exact zeros (what the dense map's consumers
// read there: the zero padding of the next convolution / of the window unfold).  Interior matches return at once.
__global__ __launch_bounds__(64) void patch_zero_oob_kernel(float* __restrict__ buf, int ld, const long long* __restrict__ j_ids, int wc, int stride,
                                                            int org, int P, int Hf, int Wf) {
  const int m = blockIdx.x;
  const int j = (int)j_ids[m];
  const int jy = j / wc, jx = j - jy * wc;
  const int y0p = jy * stride + org, x0p = jx * stride + org;
  if (y0p >= 0 && x0p >= 0 && y0p + P <= Hf && x0p + P <= Wf) return;
  const int q = ld >> 2;
  for (int e = threadIdx.x; e < P * P * q; e += blockDim.x) {
    const int r = e / q, c = (e - r * q) * 4;
    const int py = r / P, px = r - py * P;
    const int y = y0p + py, x = x0p + px;
    if (!((unsigned)y < (unsigned)Hf && (unsigned)x < (unsigned)Wf))
      *reinterpret_cast<float4*>(buf + ((size_t)m * P * P + r) * ld + c) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// f3 [M][ld3] = bank[:, i_ids[m]]   (the RAW fine bank, quirk q8; the point half of fine_gather_kernel)
__global__ __launch_bounds__(128) void fine_points_gather_kernel(const float* __restrict__ bank, int n_points, const long long* __restrict__ i_ids, int C,
                                                                 float* __restrict__ f3, int ld3) {
  const int m = blockIdx.x;
  const long long i = i_ids[m];
  for (int c = threadIdx.x; c < C; c += blockDim.x) f3[(size_t)m * ld3 + c] = bank[(size_t)c * n_points + i];
}

}  // namespace

int opp_fine_patch_gather(const float* x1, int Hf, int Wf, int c1, const float* x2o, int c2, const long long* j_ids, int M, int wc, int stride, int org, int P,
                          float* xa, float* up, hipStream_t stream) {
  if (M <= 0) return OPP_OK;
  OPP_CHECK_ARG(c1 % 4 == 0 && c2 % 4 == 0 && Hf % 2 == 0 && Wf % 2 == 0, "fine patch gather: channel counts must be multiples of 4, the map size even");
  const int Hr = Hf / 2, Wr = Wf / 2;
  // align_corners=True source index scale (in - 1) / (out - 1), as run_conv sets it for the dense convolution (resnet.py:155)
  const float res_sy = Hf > 1 ? (float)(Hr - 1) / (float)(Hf - 1) : 0.f;
  const float res_sx = Wf > 1 ? (float)(Wr - 1) / (float)(Wf - 1) : 0.f;
  hipLaunchKernelGGL(fine_patch_gather_kernel, dim3(M), dim3(256), 0, stream, x1, Hf, Wf, c1, x2o, Hr, Wr, c2, res_sy, res_sx, j_ids, wc, stride, org, P, xa, up);
  OPP_CHECK_LAUNCH("fine_patch_gather_kernel");
  return OPP_OK;
}

int opp_patch_zero_oob(float* buf, int ld, const long long* j_ids, int M, int wc, int stride, int org, int P, int Hf, int Wf, hipStream_t stream) {
  if (M <= 0) return OPP_OK;
  OPP_CHECK_ARG(ld % 4 == 0, "patch_zero_oob: row stride must be a multiple of 4");
  hipLaunchKernelGGL(patch_zero_oob_kernel, dim3(M), dim3(64), 0, stream, buf, ld, j_ids, wc, stride, org, P, Hf, Wf);
  OPP_CHECK_LAUNCH("patch_zero_oob_kernel");
  return OPP_OK;
}

int opp_fine_points_gather(const float* bank, int n_points, const long long* i_ids, int M, int C, float* f3, int ld3, hipStream_t stream) {
  if (M <= 0) return OPP_OK;
  hipLaunchKernelGGL(fine_points_gather_kernel, dim3(M), dim3(128), 0, stream, bank, n_points, i_ids, C, f3, ld3);
  OPP_CHECK_LAUNCH("fine_points_gather_kernel");
  return OPP_OK;
}

int opp_fine_head_bwd(const float* f3, int ld3, const float* win, int ldw, int M, int Wwin, int C, float temp, const float* gexp, float* gf3, float* gwin,
                      hipStream_t stream) {
  if (M <= 0) return OPP_OK;
  OPP_CHECK_ARG(f3 && win && gexp && gf3 && gwin && Wwin * Wwin <= 64 && Wwin > 1 && C % 4 == 0 && ldw % 4 == 0 && ld3 % 4 == 0, "fine head backward: bad argument");
  hipLaunchKernelGGL(fine_head_bwd_kernel, dim3(opp_cdiv(M, 4)), dim3(256), 0, stream, f3, ld3, win, ldw, M, Wwin, C, temp, gexp, gf3, gwin);
  OPP_CHECK_LAUNCH("fine_head_bwd_kernel");
  return OPP_OK;
}

int opp_fine_gather(const float* feat, int Hf, int Wf, int ldf, const float* bank, int n_points,
                    const long long* i_ids, const long long* j_ids, int M, int wc, int stride, int Wwin, int C,
                    float* win, int ldw, float* f3, int ld3, hipStream_t stream) {
  if (M <= 0) return OPP_OK;
  OPP_CHECK_ARG(C % 4 == 0 && ldf % 4 == 0 && ldw % 4 == 0, "fine gather: channel counts / strides must be multiples of 4");
  // algorithmic bytes: windows read + written, point descriptors read + written
  OppProfScope prof(OPP_PROF_FINE_GATHER, stream, (double)M * (Wwin * Wwin + 1) * C * 4.0 * 2.0);
  hipLaunchKernelGGL(fine_gather_kernel, dim3(M), dim3(128), 0, stream, feat, Hf, Wf, ldf, bank, n_points, i_ids, j_ids, wc,
                     stride, Wwin, C, win, ldw, f3, ld3);
  OPP_CHECK_LAUNCH("fine_gather_kernel");
  return OPP_OK;
}

int opp_fine_head(const float* f3, int ld3, const float* win, int ldw, int M, int Wwin, int C, float temp,
                  const float* mkpts_c, float base_scale, const float* qscale, float* expec, float* mkpts_f,
                  hipStream_t stream) {
  if (M <= 0) return OPP_OK;
  OPP_CHECK_ARG(Wwin * Wwin <= 64, "fine head: window %d too large", Wwin);
  OPP_CHECK_ARG(C % 4 == 0 && ldw % 4 == 0 && ld3 % 4 == 0, "fine head: channel counts / strides must be multiples of 4");
  OppProfScope prof(OPP_PROF_FINE_HEAD, stream, (double)M * (Wwin * Wwin + 1) * C * 4.0);   // windows + point tokens read once
  hipLaunchKernelGGL(fine_head_kernel, dim3(opp_cdiv(M, 4)), dim3(256), 0, stream, f3, ld3, win, ldw, M, Wwin, C, temp, mkpts_c,
                     base_scale, qscale, expec, mkpts_f);
  OPP_CHECK_LAUNCH("fine_head_kernel");
  return OPP_OK;
}
