// Small nodes of the training step's graph that are neither GEMMs nor convolutions:
//   LayerNorm forward (saving mean / rstd) and backward           loftr_module/transformer.py:87-88, :92-94 (nn.LayerNorm)
//   row / column log-sum-exp of the score matrix                  utils/coarse_matching.py:115 (the two softmaxes)
//   dual-softmax confidences from the score matrix and its LSEs   utils/coarse_matching.py:115
//   fine-window gather of a BATCH of images + its backward         loftr_module/fine_preprocess.py:41-55 (F.unfold + indexing)
// All HBM-bound; reductions are fixed-order (deterministic) except the window scatter, which adds overlapping windows with
// fp32 atomics like torch's own index_put / col2im backward does.
#include "opp_internal.h"

namespace {

int grid_for(size_t total, int block = 256, int cap = 16384) {
  const size_t b = (total + block - 1) / block;
  return (int)(b < 1 ? 1 : (b < (size_t)cap ? b : (size_t)cap));
}

// ---- LayerNorm -----------------------------------------------------------------------------------------------------------
// one wave per row, VPT = C / 64 values per lane (C = 256 / 128 / 64): y = (x - mean) * rstd * gamma + beta (+ res)
template <int VPT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ res, int rows, float eps, float* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  constexpr int C = VPT * 64;
  typedef float vec_t __attribute__((ext_vector_type(VPT)));
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const vec_t v = *reinterpret_cast<const vec_t*>(x + (size_t)row * C + lane * VPT);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) s += v[i];
  const float mean = opp_wave_sum_dpp(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const float d = v[i] - mean;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(opp_wave_sum_dpp(q) / (float)C + eps);
  const vec_t g = *reinterpret_cast<const vec_t*>(gamma + lane * VPT);
  const vec_t b = *reinterpret_cast<const vec_t*>(beta + lane * VPT);
  vec_t o;
#pragma unroll
  for (int i = 0; i < VPT; ++i) o[i] = (v[i] - mean) * rstd * g[i] + b[i];
  if (res) {
    const vec_t r = *reinterpret_cast<const vec_t*>(res + (size_t)row * C + lane * VPT);
#pragma unroll
    for (int i = 0; i < VPT; ++i) o[i] = r[i] + o[i];
  }
  *reinterpret_cast<vec_t*>(y + (size_t)row * C + lane * VPT) = o;
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// dx = rstd * (g gamma - mean(g gamma) - xhat mean(g gamma xhat)); block partials of dgamma = sum g xhat, dbeta = sum g
// 4 waves x kLnRowsPerWave rows per block; part [blocks][2][C]
constexpr int kLnRowsPerWave = 32;
template <int VPT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, int rows,
                                                     float* __restrict__ dx, float* __restrict__ part) {
  constexpr int C = VPT * 64;
  typedef float vec_t __attribute__((ext_vector_type(VPT)));
  __shared__ float red[4][2][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const vec_t gm = *reinterpret_cast<const vec_t*>(gamma + lane * VPT);
  float dgam[VPT], dbet[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) dgam[i] = dbet[i] = 0.f;
  const int r0 = (blockIdx.x * 4 + wave) * kLnRowsPerWave;
  for (int r = r0; r < min(rows, r0 + kLnRowsPerWave); ++r) {
    const vec_t gv = *reinterpret_cast<const vec_t*>(g + (size_t)r * C + lane * VPT);
    const vec_t xv = *reinterpret_cast<const vec_t*>(x + (size_t)r * C + lane * VPT);
    const float mu = mean[r], rs = rstd[r];
    float xh[VPT], gg[VPT], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      xh[i] = (xv[i] - mu) * rs;
      gg[i] = gv[i] * gm[i];
      s1 += gg[i];
      s2 += gg[i] * xh[i];
      dgam[i] += gv[i] * xh[i];
      dbet[i] += gv[i];
    }
    const float m1 = opp_wave_sum_dpp(s1) / (float)C;
    const float m2 = opp_wave_sum_dpp(s2) / (float)C;
    vec_t o;
#pragma unroll
    for (int i = 0; i < VPT; ++i) o[i] = rs * (gg[i] - m1 - xh[i] * m2);
    *reinterpret_cast<vec_t*>(dx + (size_t)r * C + lane * VPT) = o;
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    red[wave][0][lane * VPT + i] = dgam[i];
    red[wave][1][lane * VPT + i] = dbet[i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * C; e += 256) {
    const int k = e / C, c = e - k * C;
    part[((size_t)blockIdx.x * 2 + k) * C + c] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
  }
}

// out[k][c] = sum over the blocks of part[b][k][c] (k = 0 dgamma, 1 dbeta): 16 columns x 16 slices per workgroup, slice s adds
// blocks s, s + 16, ... in ascending order, the 16 slice sums are then added in slice order (fixed order = deterministic)
__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(const float* __restrict__ part, int blocks, int C, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
  __shared__ double red[16][16];
  const int cx = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + cx;                 // column of the [2 C] row (k, c)
  double s = 0.0;
  if (e < 2 * C) {
    const int k = e / C, c = e - k * C;
    for (int b = sl; b < blocks; b += 16) s += (double)part[((size_t)b * 2 + k) * C + c];
  }
  red[sl][cx] = s;
  __syncthreads();
  if (sl != 0 || e >= 2 * C) return;
  s = 0.0;
  for (int k = 0; k < 16; ++k) s += red[k][cx];
  (e < C ? dgamma : dbeta)[e < C ? e : e - C] = (float)s;
}

// ---- log-sum-exp of the rows and columns of S [B][N][L] ------------------------------------------------------------------
// rows: one wave per row, online (max, sum) per lane merged by shuffles
__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ S, long long rows, int L, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = S + (size_t)row * L;
  float m = -INFINITY, s = 0.f;
  for (int j = lane; j < L; j += 64) {
    const float v = p[j];
    if (v > m) {
      s = s * __expf(m - v) + 1.f;
      m = v;
    } else {
      s += __expf(v - m);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mm = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mm));
    m = mm;
  }
  if (lane == 0) out[row] = m + __logf(s);
}

// columns: thread = column, block walks kLseRows rows -> partial (max, sum) [B][chunks][2][L]; merged in chunk order
constexpr int kLseRows = 128;
__global__ __launch_bounds__(256) void lse_cols_partial_kernel(const float* __restrict__ S, int N, int L, int chunks, float* __restrict__ part) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int chunk = blockIdx.y, b = blockIdx.z;
  if (col >= L) return;
  const int r0 = chunk * kLseRows, r1 = min(N, r0 + kLseRows);
  const float* p = S + ((size_t)b * N) * L + col;
  float m = -INFINITY;
  for (int r = r0; r < r1; ++r) m = fmaxf(m, p[(size_t)r * L]);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += __expf(p[(size_t)r * L] - m);
  float* o = part + (((size_t)b * chunks + chunk) * 2) * L + col;
  o[0] = m;
  o[L] = s;
}
__global__ __launch_bounds__(256) void lse_cols_merge_kernel(const float* __restrict__ part, int L, int chunks, float* __restrict__ out) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (col >= L) return;
  const float* p = part + ((size_t)b * chunks * 2) * L + col;
  float m = -INFINITY;
  for (int c = 0; c < chunks; ++c) m = fmaxf(m, p[(size_t)c * 2 * L]);
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += p[(size_t)c * 2 * L + L] * __expf(p[(size_t)c * 2 * L] - m);
  out[(size_t)b * L + col] = m + __logf(s);
}

// conf = exp(S - lse_col[j]) * exp(S - lse_row[i])   (softmax over the points x softmax over the cells)
__global__ void dual_softmax_conf_kernel(const float4* __restrict__ S, const float* __restrict__ lse_row, const float* __restrict__ lse_col,
                                         long long rows, int N, int L4, float4* __restrict__ conf) {
  const size_t total = (size_t)rows * L4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const long long row = (long long)(i / L4);
    const int q = (int)(i - (size_t)row * L4);
    const int b = (int)(row / N);
    const float lr = lse_row[row];
    const float4 lc = *reinterpret_cast<const float4*>(lse_col + (size_t)b * L4 * 4 + q * 4);
    const float4 v = S[i];
    float4 o;
    o.x = __expf(v.x - lc.x) * __expf(v.x - lr);
    o.y = __expf(v.y - lc.y) * __expf(v.y - lr);
    o.z = __expf(v.z - lc.z) * __expf(v.z - lr);
    o.w = __expf(v.w - lc.w) * __expf(v.w - lr);
    conf[i] = o;
  }
}

// ---- fine windows of a batch ----------------------------------------------------------------------------------------------
// win [M][WW][C] <- feat [B][Hf][Wf][C] around cell j_ids[m] of image b_ids[m] (zero outside the image)
__global__ __launch_bounds__(128) void fine_gather_batch_kernel(const float* __restrict__ feat, int Hf, int Wf, int C, const long long* __restrict__ b_ids,
                                                                const long long* __restrict__ j_ids, int wc, int stride, int Wwin,
                                                                float* __restrict__ win) {
  const int m = blockIdx.x;
  const int b = (int)b_ids[m], j = (int)j_ids[m];
  const int jy = j / wc, jx = j - jy * wc;
  const int cy = jy * stride - Wwin / 2, cx = jx * stride - Wwin / 2;
  const int WW = Wwin * Wwin, c4n = C >> 2;
  for (int e = threadIdx.x; e < WW * c4n; e += blockDim.x) {
    const int r = e / c4n, c = (e - r * c4n) * 4;
    const int ky = r / Wwin, kx = r - ky * Wwin;
    const int y = cy + ky, x = cx + kx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)y < (unsigned)Hf && (unsigned)x < (unsigned)Wf) v = *reinterpret_cast<const float4*>(feat + (((size_t)b * Hf + y) * Wf + x) * C + c);
    *reinterpret_cast<float4*>(win + ((size_t)m * WW + r) * C + c) = v;
  }
}
// dfeat (zeroed by the launcher) += windows' gradients
__global__ __launch_bounds__(128) void fine_scatter_batch_kernel(const float* __restrict__ gwin, int Hf, int Wf, int C, const long long* __restrict__ b_ids,
                                                                 const long long* __restrict__ j_ids, int wc, int stride, int Wwin,
                                                                 float* __restrict__ dfeat) {
  const int m = blockIdx.x;
  const int b = (int)b_ids[m], j = (int)j_ids[m];
  const int jy = j / wc, jx = j - jy * wc;
  const int cy = jy * stride - Wwin / 2, cx = jx * stride - Wwin / 2;
  const int WW = Wwin * Wwin;
  for (int e = threadIdx.x; e < WW * C; e += blockDim.x) {
    const int r = e / C, c = e - r * C;
    const int ky = r / Wwin, kx = r - ky * Wwin;
    const int y = cy + ky, x = cx + kx;
    if ((unsigned)y < (unsigned)Hf && (unsigned)x < (unsigned)Wf)
      atomicAdd(dfeat + (((size_t)b * Hf + y) * Wf + x) * C + c, gwin[((size_t)m * WW + r) * C + c]);
  }
}

}  // namespace

int opp_ln_forward(const float* x, const float* gamma, const float* beta, const float* res, int rows, int C, float eps, float* y, float* mean,
                   float* rstd, hipStream_t stream) {
  OPP_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0, "layer_norm_train_forward: null / empty argument");
  const dim3 grid(opp_cdiv(rows, 4)), block(256);
  if (C == 256) hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, block, 0, stream, x, gamma, beta, res, rows, eps, y, mean, rstd);
  else if (C == 128) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, block, 0, stream, x, gamma, beta, res, rows, eps, y, mean, rstd);
  else if (C == 64) hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, block, 0, stream, x, gamma, beta, res, rows, eps, y, mean, rstd);
  else {
    opp_set_error("layer_norm_train: C must be 64, 128 or 256 (got %d)", C);
    return OPP_ERR_UNSUPPORTED;
  }
  OPP_CHECK_LAUNCH("ln_fwd_kernel");
  return OPP_OK;
}

size_t opp_ln_backward_ws_bytes(int rows, int C) { return opp_align((size_t)opp_cdiv(rows, 4 * kLnRowsPerWave) * 2 * C * sizeof(float)); }

int opp_ln_backward(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, int rows, int C, float* dx,
                    float* dgamma, float* dbeta, void* ws, size_t ws_bytes, hipStream_t stream) {
  OPP_CHECK_ARG(g && x && gamma && mean && rstd && dx && dgamma && dbeta && ws && rows > 0, "layer_norm_train_backward: null / empty argument");
  OPP_CHECK_ARG(ws_bytes >= opp_ln_backward_ws_bytes(rows, C), "layer_norm_train_backward: workspace too small");
  const int blocks = opp_cdiv(rows, 4 * kLnRowsPerWave);
  float* part = static_cast<float*>(ws);
  if (C == 256) hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(blocks), dim3(256), 0, stream, g, x, gamma, mean, rstd, rows, dx, part);
  else if (C == 128) hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(blocks), dim3(256), 0, stream, g, x, gamma, mean, rstd, rows, dx, part);
  else if (C == 64) hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(blocks), dim3(256), 0, stream, g, x, gamma, mean, rstd, rows, dx, part);
  else {
    opp_set_error("layer_norm_train: C must be 64, 128 or 256 (got %d)", C);
    return OPP_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3(opp_cdiv(2 * C, 16)), dim3(256), 0, stream, part, blocks, C, dgamma, dbeta);
  OPP_CHECK_LAUNCH("ln_bwd kernels");
  return OPP_OK;
}

size_t opp_lse_ws_bytes(int B, int N, int L) { return opp_align((size_t)B * opp_cdiv(N, kLseRows) * 2 * L * sizeof(float)); }

// lse_row [B][N] = logsumexp_j S[b][i][j]; lse_col [B][L] = logsumexp_i S[b][i][j]; conf (optional, may alias S) = the dual softmax
int opp_dual_softmax_lse(const float* S, int B, int N, int L, float* lse_row, float* lse_col, float* conf, void* ws, size_t ws_bytes,
                         hipStream_t stream) {
  OPP_CHECK_ARG(S && lse_row && lse_col && ws && B > 0 && N > 0 && L > 0, "dual_softmax_forward: null / empty argument");
  OPP_CHECK_ARG(ws_bytes >= opp_lse_ws_bytes(B, N, L), "dual_softmax_forward: workspace too small");
  OPP_CHECK_ARG(!conf || L % 4 == 0, "dual_softmax_forward: L must be a multiple of 4");
  const long long rows = (long long)B * N;
  const int chunks = opp_cdiv(N, kLseRows);
  float* part = static_cast<float*>(ws);
  hipLaunchKernelGGL(lse_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, S, rows, L, lse_row);
  hipLaunchKernelGGL(lse_cols_partial_kernel, dim3(opp_cdiv(L, 256), chunks, B), dim3(256), 0, stream, S, N, L, chunks, part);
  hipLaunchKernelGGL(lse_cols_merge_kernel, dim3(opp_cdiv(L, 256), B), dim3(256), 0, stream, part, L, chunks, lse_col);
  if (conf)
    hipLaunchKernelGGL(dual_softmax_conf_kernel, dim3(grid_for((size_t)rows * (L / 4))), dim3(256), 0, stream, reinterpret_cast<const float4*>(S), lse_row,
                       lse_col, rows, N, L / 4, reinterpret_cast<float4*>(conf));
  OPP_CHECK_LAUNCH("dual_softmax_lse kernels");
  return OPP_OK;
}

int opp_fine_gather_batch(const float* feat, int Hf, int Wf, int C, const long long* b_ids, const long long* j_ids, int M, int wc, int stride, int Wwin,
                          float* win, hipStream_t stream) {
  OPP_CHECK_ARG(feat && b_ids && j_ids && win && C % 4 == 0 && M >= 0, "fine_gather_batch: bad argument");
  if (M == 0) return OPP_OK;
  hipLaunchKernelGGL(fine_gather_batch_kernel, dim3(M), dim3(128), 0, stream, feat, Hf, Wf, C, b_ids, j_ids, wc, stride, Wwin, win);
  OPP_CHECK_LAUNCH("fine_gather_batch_kernel");
  return OPP_OK;
}

int opp_fine_scatter_batch(const float* gwin, int B, int Hf, int Wf, int C, const long long* b_ids, const long long* j_ids, int M, int wc, int stride,
                           int Wwin, float* dfeat, hipStream_t stream) {
  OPP_CHECK_ARG(gwin && b_ids && j_ids && dfeat && M >= 0, "fine_scatter_batch: bad argument");
  if (hipMemsetAsync(dfeat, 0, (size_t)B * Hf * Wf * C * sizeof(float), stream) != hipSuccess) {
    opp_set_error("fine_scatter_batch: memset failed");
    return OPP_ERR_LAUNCH;
  }
  if (M == 0) return OPP_OK;
  hipLaunchKernelGGL(fine_scatter_batch_kernel, dim3(M), dim3(128), 0, stream, gwin, Hf, Wf, C, b_ids, j_ids, wc, stride, Wwin, dfeat);
  OPP_CHECK_LAUNCH("fine_scatter_batch_kernel");
  return OPP_OK;
}


// ----------------------------------------------------------------------------------------------------------------------------
// Ground-truth matrices of one training sample on the device: OnePosePlusDataset.build_assignmatrix
// (/root/reference/src/datasets/OnePosePlus_dataset.py:174-236).  The loader builds conf_matrix_gt [N][L] int16 and
// fine_location_matrix_gt [N][L][2] fp32 (-50 filled) on the host -- 229 MB per sample at N = 7000 -- from k <= a few thousand
// (2D keypoint, 3D point) pairs; here only the pairs travel and the matrices are formed where they are consumed.
//   key[t] = i * L + j of pair t, or -1 when the reference drops it (padded 3D index >= N, j > L); status bit 0: an index the
//   reference would raise IndexError on (j == L, indices out of range), bit 1: duplicate (i, j) pairs ("Keypoints duplicate!").
// ----------------------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void assign_keys_kernel(const float* __restrict__ kc, int n2d, const long long* __restrict__ assign, int k, int N, int L,
                                                          int wc, float sx, float sy, float cs, long long* __restrict__ keys, int* __restrict__ status) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= k) return;
  long long kp = assign[t], i = assign[(size_t)k + t];
  long long key = -1;
  if (i < N) {                                     // valid = assign_matrix[1] < self.shape3d  (:195-196)
    if (kp < 0) kp += n2d;                         // torch indexing wraps negative indices once
    if (i < 0) i += N;
    if (kp < 0 || kp >= n2d || i < 0) {
      atomicOr(status, 1);
    } else {
      // keypoints2D_coarse_selected / query_img_scale[[1, 0]] * coarse_scale, rounded half to even (:205-212)
      const float x = rintf(kc[2 * kp] / sx * cs), y = rintf(kc[2 * kp + 1] / sy * cs);
      const float jf = y * (float)wc + x;
      // NaN / inf / beyond int64: `.long()` upstream gives INT64_MIN and the indexing raises; here the float -> integer conversion would be
      // undefined, so such keypoints are reported through the status word (the host raises IndexError) instead of landing in some cell
      const bool finite = jf == jf && fabsf(jf) < 9.0e18f;
      long long j = finite ? (long long)jf : 0;      // (:219-223)
      if (!finite) {
        atomicOr(status, 1);
      } else if (!(j > L)) {                              // invalid_mask = j_ids > conf_matrix.shape[1]  (:225): j == L survives the mask ...
        if (j < 0) j += L;
        if (j < 0 || j >= L) atomicOr(status, 1);   // ... and is an IndexError upstream
        else key = i * (long long)L + j;
      }
    }
  }
  keys[t] = key;
}

// index_put with repeated indices on the CPU keeps the LAST value: pair t writes unless a later pair has the same (i, j)
__global__ __launch_bounds__(256) void assign_scatter_kernel(const float* __restrict__ kf, int n2d, const long long* __restrict__ assign, int k,
                                                             const long long* __restrict__ keys, short* __restrict__ conf, float* __restrict__ floc,
                                                             int* __restrict__ status) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= k) return;
  const long long key = keys[t];
  if (key < 0) return;
  for (int u = t + 1; u < k; ++u)
    if (keys[u] == key) {
      atomicOr(status, 2);
      return;
    }
  long long kp = assign[t];
  if (kp < 0) kp += n2d;
  conf[key] = 1;
  floc[2 * key] = kf[2 * kp];
  floc[2 * key + 1] = kf[2 * kp + 1];
}

__global__ __launch_bounds__(256) void fill4_kernel(float4* __restrict__ out, size_t n4, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4(v, v, v, v);
}

}  // namespace

int opp_assignmatrix(const float* kp2d_coarse, const float* kp2d_fine, int n2d, const long long* assign, int k, int N, int L, int w_c, float scale_x,
                     float scale_y, float coarse_scale, short* conf_gt, float* fine_loc_gt, long long* keys, int* status, hipStream_t stream) {
  OPP_CHECK_ARG(conf_gt && fine_loc_gt && N > 0 && L > 0 && w_c > 0 && k >= 0 && n2d >= 0, "build_assignmatrix: bad argument");
  OPP_CHECK_ARG(k == 0 || (kp2d_coarse && kp2d_fine && assign && keys && status), "build_assignmatrix: null pair data");
  OPP_CHECK_ARG((reinterpret_cast<uintptr_t>(fine_loc_gt) & 15) == 0 && ((size_t)N * L * 2) % 4 == 0, "build_assignmatrix: fine_location_matrix must be 16-byte aligned");
  if (hipMemsetAsync(conf_gt, 0, (size_t)N * L * sizeof(short), stream) != hipSuccess || (status && hipMemsetAsync(status, 0, sizeof(int), stream) != hipSuccess)) {
    opp_set_error("build_assignmatrix: memset failed");
    return OPP_ERR_LAUNCH;
  }
  const size_t n4 = (size_t)N * L * 2 / 4;
  hipLaunchKernelGGL(fill4_kernel, dim3(grid_for(n4)), dim3(256), 0, stream, reinterpret_cast<float4*>(fine_loc_gt), n4, -50.0f);
  if (k > 0) {
    hipLaunchKernelGGL(assign_keys_kernel, dim3(opp_cdiv(k, 256)), dim3(256), 0, stream, kp2d_coarse, n2d, assign, k, N, L, w_c, scale_x, scale_y, coarse_scale,
                       keys, status);
    hipLaunchKernelGGL(assign_scatter_kernel, dim3(opp_cdiv(k, 256)), dim3(256), 0, stream, kp2d_fine, n2d, assign, k, keys, conf_gt, fine_loc_gt, status);
  }
  OPP_CHECK_LAUNCH("build_assignmatrix kernels");
  return OPP_OK;
}
