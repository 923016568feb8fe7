// Coarse-level focal loss of the training step over the B x N x L confidence matrix, forward and backward.
//
// Reference: Loss.compute_coarse_loss, src/lightning_model/losses.py:18-55 (`coarse_type: focal`):
//   conf   = clamp(conf, 1e-6, 1 - 1e-6)
//   pos    = -alpha       * (1 - conf)^gamma * log(conf)        over the entries with conf_gt == 1
//   neg    = -(1 - alpha) * conf^gamma       * log(1 - conf)    over the entries with conf_gt == 0
//   (both times an optional per-entry weight), loss = pos_weight * mean(pos) + neg_weight * mean(neg).
// The reference materialises four boolean-indexed copies of the 115 M-entry matrix (B = 4, N = 7000, L = 4096) and
// autograd walks them back; here one pass reads conf (fp32) + conf_gt (int16) = 6 B per entry and leaves the four
// sums (sum pos, sum neg, #pos, #neg) in fp64, and the backward is one elementwise pass writing d loss / d conf.
// Both are HBM-bound: 6 B (forward) / 10 B (backward) per entry.
//
// Deterministic: per-thread fp64 accumulators, wave shuffles, block partials, fixed-order final sum.
#include "opp_internal.h"

namespace {

constexpr int kLossThreads = 256;
constexpr int kLossMaxBlocks = 2048;
constexpr float kConfLo = 1e-6f, kConfHi = (float)(1.0 - 1e-6);

__device__ __forceinline__ float pow_gamma(float x, float gamma) { return gamma == 2.0f ? x * x : powf(x, gamma); }
// d/dx x^gamma
__device__ __forceinline__ float dpow_gamma(float x, float gamma) { return gamma == 2.0f ? 2.0f * x : gamma * powf(x, gamma - 1.0f); }

struct Acc {
  double pos = 0.0, neg = 0.0, npos = 0.0, nneg = 0.0;
};

__device__ __forceinline__ void focal_add(float conf, int gt, float w, float alpha, float gamma, Acc& a) {
  const float c = fminf(fmaxf(conf, kConfLo), kConfHi);
  if (gt == 1) {
    a.pos += (double)(-alpha * pow_gamma(1.0f - c, gamma) * logf(c) * w);
    a.npos += 1.0;
  } else if (gt == 0) {
    a.neg += (double)(-(1.0f - alpha) * pow_gamma(c, gamma) * logf(1.0f - c) * w);
    a.nneg += 1.0;
  }
}

// ground truth as the data loader / caller holds it: int16 (the reference casts to it), fp32 (the reference's dataset emits
// float zeros and ones) or 8-bit (bool); values other than 0 / 1 are ignored like the reference's `== 1` / `== 0` masks.
// The optional per-entry weight is either a full array or -- Loss.compute_c_weight, losses.py:103-111 -- the outer product
// mask0[b][i] * mask1[b][j], formed on the fly from the two vectors (no B x N x L weight tensor is ever materialised).
template <typename GT>
struct GtVec;
template <>
struct GtVec<short> {
  typedef short4 v4;
  static __device__ __forceinline__ int cls(short g) { return g == 1 ? 1 : (g == 0 ? 0 : -1); }
};
template <>
struct GtVec<float> {
  typedef float4 v4;
  static __device__ __forceinline__ int cls(float g) { return g == 1.0f ? 1 : (g == 0.0f ? 0 : -1); }
};
template <>
struct GtVec<unsigned char> {
  typedef uchar4 v4;
  static __device__ __forceinline__ int cls(unsigned char g) { return g == 1 ? 1 : (g == 0 ? 0 : -1); }
};
struct FocalW {          // weight source: w (full array) | m0 / m1 (outer product; N rows per sample, L columns) | none
  const float* w;
  const float* m0;
  const float* m1;
  int N, L;
};
__device__ __forceinline__ float4 focal_w4(const FocalW& fw, size_t i4) {   // weights of entries 4 i4 .. 4 i4 + 3 (L % 4 == 0 if masks)
  if (fw.w) return reinterpret_cast<const float4*>(fw.w)[i4];
  if (fw.m0) {
    const size_t i = i4 << 2;
    const size_t row = i / (size_t)fw.L;
    const int l = (int)(i - row * fw.L);
    const size_t bsmp = row / (size_t)fw.N;
    const float a = fw.m0[row];
    const float4 c = *reinterpret_cast<const float4*>(fw.m1 + bsmp * fw.L + l);
    return make_float4(a * c.x, a * c.y, a * c.z, a * c.w);
  }
  return make_float4(1.f, 1.f, 1.f, 1.f);
}
__device__ __forceinline__ float focal_w1(const FocalW& fw, size_t i) {
  if (fw.w) return fw.w[i];
  if (fw.m0) {
    const size_t row = i / (size_t)fw.L;
    const int l = (int)(i - row * fw.L);
    return fw.m0[row] * fw.m1[(row / (size_t)fw.N) * fw.L + l];
  }
  return 1.f;
}

template <typename GT>
__global__ __launch_bounds__(kLossThreads) void focal_fwd_kernel(const float* __restrict__ conf, const GT* __restrict__ gt, const FocalW fw,
                                                                 size_t n, int vec_ok, float alpha, float gamma, double* __restrict__ part) {
  Acc a;
  const size_t n4 = vec_ok ? (n >> 2) : 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 c = reinterpret_cast<const float4*>(conf)[i];
    const typename GtVec<GT>::v4 g = reinterpret_cast<const typename GtVec<GT>::v4*>(gt)[i];
    const float4 w = focal_w4(fw, i);
    focal_add(c.x, GtVec<GT>::cls(g.x), w.x, alpha, gamma, a);
    focal_add(c.y, GtVec<GT>::cls(g.y), w.y, alpha, gamma, a);
    focal_add(c.z, GtVec<GT>::cls(g.z), w.z, alpha, gamma, a);
    focal_add(c.w, GtVec<GT>::cls(g.w), w.w, alpha, gamma, a);
  }
  // ragged tail (or everything when the vector path does not apply): grid-stride scalar loop
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    focal_add(conf[i], GtVec<GT>::cls(gt[i]), focal_w1(fw, i), alpha, gamma, a);
  double v[4] = {a.pos, a.neg, a.npos, a.nneg};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[e] += __shfl_xor(v[e], o, 64);
  __shared__ double red[kLossThreads / 64][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][e] = v[e];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0.0;
    for (int w = 0; w < kLossThreads / 64; ++w) t += red[w][threadIdx.x];
    part[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
  }
}

// sums[e] = sum over the block partials, fixed order: 64 interleaved slices per quantity (slice s adds blocks s, s + 64, ... in order), then a
// pairwise tree over the slices -- deterministic, and 64 x shorter chains than one thread per quantity (r04: 220 us per call at 16 384 partials)
__global__ __launch_bounds__(256) void focal_finalize_kernel(const double* __restrict__ part, int blocks, double* __restrict__ sums) {
  __shared__ double red[64][4];
  const int e = threadIdx.x & 3, s = threadIdx.x >> 2;
  double t = 0.0;
  for (int b = s; b < blocks; b += 64) t += part[(size_t)b * 4 + e];
  red[s][e] = t;
  __syncthreads();
  for (int half = 32; half > 0; half >>= 1) {
    if (s < half) red[s][e] += red[s + half][e];
    __syncthreads();
  }
  if (s == 0) sums[e] = red[0][e];
}

// d loss / d conf; scales = {g * pos_weight / #pos, g * neg_weight / #neg} (device).  torch.clamp passes the gradient
// where lo <= conf <= hi (inclusive) and blocks it outside.
__device__ __forceinline__ float focal_grad(float conf, int gt, float w, float alpha, float gamma, float s_pos, float s_neg) {
  if (!(conf >= kConfLo && conf <= kConfHi)) return 0.f;
  if (gt == 1) {
    const float om = 1.0f - conf;
    const float d = -alpha * (-dpow_gamma(om, gamma) * logf(conf) + pow_gamma(om, gamma) / conf);
    return s_pos * w * d;
  }
  if (gt == 0) {
    const float om = 1.0f - conf;
    const float d = -(1.0f - alpha) * (dpow_gamma(conf, gamma) * logf(om) - pow_gamma(conf, gamma) / om);
    return s_neg * w * d;
  }
  return 0.f;
}

template <typename GT>
__global__ __launch_bounds__(kLossThreads) void focal_bwd_kernel(const float* __restrict__ conf, const GT* __restrict__ gt, const FocalW fw,
                                                                 size_t n, int vec_ok, float alpha, float gamma,
                                                                 const float* __restrict__ scales, float* __restrict__ grad) {
  const float s_pos = scales[0], s_neg = scales[1];
  const size_t n4 = vec_ok ? (n >> 2) : 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 c = reinterpret_cast<const float4*>(conf)[i];
    const typename GtVec<GT>::v4 g = reinterpret_cast<const typename GtVec<GT>::v4*>(gt)[i];
    const float4 w = focal_w4(fw, i);
    float4 o;
    o.x = focal_grad(c.x, GtVec<GT>::cls(g.x), w.x, alpha, gamma, s_pos, s_neg);
    o.y = focal_grad(c.y, GtVec<GT>::cls(g.y), w.y, alpha, gamma, s_pos, s_neg);
    o.z = focal_grad(c.z, GtVec<GT>::cls(g.z), w.z, alpha, gamma, s_pos, s_neg);
    o.w = focal_grad(c.w, GtVec<GT>::cls(g.w), w.w, alpha, gamma, s_pos, s_neg);
    reinterpret_cast<float4*>(grad)[i] = o;
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    grad[i] = focal_grad(conf[i], GtVec<GT>::cls(gt[i]), focal_w1(fw, i), alpha, gamma, s_pos, s_neg);
}

// ---------------------------------------------------------------------------------------
// Backward of the dual-softmax confidence matrix (CoarseMatching.forward, utils/coarse_matching.py:115):
//   conf = A * B,  A = softmax(S) over the N points (per cell j),  B = softmax(S) over the L cells (per point i)
//   dS_ij = 2 conf_ij g_ij - A_ij c_j - B_ij r_i,   c_j = sum_i g_ij conf_ij,   r_i = sum_j g_ij conf_ij
// (softmax backward P (g' - sum g' P) with g' = g times the other factor; both inner sums are sums of g conf).
// A and B are re-evaluated from S and the log-sum-exps lse_col [B][L], lse_row [B][N] (A = exp(S - lse_col_j)), so only
// S is kept for the backward - autograd keeps both softmax outputs and the product (3 x 460 MB at B = 4, N = 7000) and
// makes ~10 passes.  Here: one pass for the row / column sums (fixed-order partials, deterministic), one elementwise.
// ---------------------------------------------------------------------------------------
constexpr int kDsmRows = 64;                 // rows per block (16 per wave)
template <int VEC>
__global__ __launch_bounds__(256) void dsm_sums_kernel(const float* __restrict__ g, const float* __restrict__ sim,
                                                       const float* __restrict__ lse_row, const float* __restrict__ lse_col,
                                                       int N, int L, int col_tiles, float* __restrict__ rpart,
                                                       float* __restrict__ cpart) {
  __shared__ float csh[4][64 * VEC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, rb = blockIdx.x, ct = blockIdx.y;
  const int j0 = ct * 64 * VEC + lane * VEC;
  const size_t base = (size_t)b * N * L;
  float cs[VEC], lc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    cs[e] = 0.f;
    lc[e] = (j0 + e < L) ? lse_col[(size_t)b * L + j0 + e] : 0.f;
  }
  for (int rr = wave; rr < kDsmRows; rr += 4) {
    const int row = rb * kDsmRows + rr;
    if (row >= N) break;                     // wave-uniform
    const float lr = lse_row[(size_t)b * N + row];
    float rs = 0.f;
    if (j0 < L) {
      const size_t o = base + (size_t)row * L + j0;
      float gv[VEC], sv[VEC];
      if (VEC == 4) {
        const float4 a = *reinterpret_cast<const float4*>(g + o), s4 = *reinterpret_cast<const float4*>(sim + o);
        gv[0] = a.x; gv[1 % VEC] = a.y; gv[2 % VEC] = a.z; gv[3 % VEC] = a.w;
        sv[0] = s4.x; sv[1 % VEC] = s4.y; sv[2 % VEC] = s4.z; sv[3 % VEC] = s4.w;
      } else {
        gv[0] = g[o];
        sv[0] = sim[o];
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float p = gv[e] * (expf(sv[e] - lc[e]) * expf(sv[e] - lr));      // g * conf
        cs[e] += p;
        rs += p;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rs += __shfl_xor(rs, o, 64);
    if (lane == 0) rpart[((size_t)b * N + row) * col_tiles + ct] = rs;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) csh[wave][lane * VEC + e] = cs[e];
  __syncthreads();
  const int nrb = gridDim.x;
  for (int t = threadIdx.x; t < 64 * VEC; t += 256) {
    const int j = ct * 64 * VEC + t;
    if (j < L) cpart[((size_t)b * nrb + rb) * L + j] = ((csh[0][t] + csh[1][t]) + csh[2][t]) + csh[3][t];
  }
}

// r[b][row] = sum over column tiles ; c[b][col] = sum over row blocks (fixed order)
__global__ void dsm_reduce_kernel(const float* __restrict__ rpart, const float* __restrict__ cpart, int B, int N, int L,
                                  int col_tiles, int row_blocks, float* __restrict__ r, float* __restrict__ c) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nr = (size_t)B * N, nc = (size_t)B * L;
  if (i < nr) {
    float s = 0.f;
    for (int t = 0; t < col_tiles; ++t) s += rpart[i * col_tiles + t];
    r[i] = s;
  } else if (i < nr + nc) {
    const size_t k = i - nr;
    const size_t b = k / L, j = k - b * L;
    float s = 0.f;
    for (int q = 0; q < row_blocks; ++q) s += cpart[(b * row_blocks + q) * L + j];
    c[k] = s;
  }
}

__global__ __launch_bounds__(256) void dsm_apply_kernel(const float* __restrict__ g, const float* __restrict__ sim,
                                                        const float* __restrict__ lse_row, const float* __restrict__ lse_col,
                                                        int N, int L, size_t total, const float* __restrict__ r,
                                                        const float* __restrict__ c, float* __restrict__ ds) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t row = i / L;                // = b * N + row
    const size_t col = (row / N) * L + (i - row * L);
    const float s = sim[i];
    const float A = expf(s - lse_col[col]), Bv = expf(s - lse_row[row]);
    ds[i] = 2.0f * (A * Bv) * g[i] - A * c[col] - Bv * r[row];
  }
}

int loss_blocks(size_t n) {
  const size_t want = (n / 4 + kLossThreads * 8 - 1) / (kLossThreads * 8);
  return (int)(want < 1 ? 1 : (want > kLossMaxBlocks ? kLossMaxBlocks : want));
}

bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

}  // namespace

size_t opp_focal_loss_ws_bytes(size_t n) { return (size_t)loss_blocks(n) * 4 * sizeof(double); }

namespace {
// gt_kind: 0 int16, 1 fp32, 2 uint8 / bool.  The 16-byte vector path needs aligned operands and, with the outer-product
// weight, rows of a multiple of four entries.
int focal_check(const float* conf, const void* gt, int gt_kind, const float* weight, const float* mask0, const float* mask1, int N, int L, size_t n,
                FocalW& fw, int& vec_ok) {
  OPP_CHECK_ARG(conf && gt && n > 0, "focal_loss: null argument / empty input");
  OPP_CHECK_ARG(gt_kind >= 0 && gt_kind <= 2, "focal_loss: conf_gt kind must be 0 (int16), 1 (fp32) or 2 (uint8)");
  OPP_CHECK_ARG(!(weight && (mask0 || mask1)) && ((mask0 == nullptr) == (mask1 == nullptr)), "focal_loss: pass a weight array OR both mask vectors");
  OPP_CHECK_ARG(!mask0 || (N > 0 && L > 0 && n % ((size_t)N * L) == 0), "focal_loss: mask vectors need n = B * N * L");
  fw.w = weight;
  fw.m0 = mask0;
  fw.m1 = mask1;
  fw.N = N;
  fw.L = L;
  const size_t gt_align = gt_kind == 0 ? 8 : (gt_kind == 1 ? 16 : 4);
  vec_ok = aligned16(conf) && (reinterpret_cast<size_t>(gt) % gt_align) == 0 && (!weight || aligned16(weight)) &&
           (!mask0 || (L % 4 == 0 && aligned16(mask1)));
  return OPP_OK;
}
}  // namespace

int opp_focal_loss_fwd_ex(const float* conf, const void* gt, int gt_kind, const float* weight, const float* mask0, const float* mask1, int N, int L,
                          size_t n, float alpha, float gamma, double* sums, void* ws, size_t ws_bytes, hipStream_t stream) {
  FocalW fw;
  int vec_ok;
  OPP_TRY(focal_check(conf, gt, gt_kind, weight, mask0, mask1, N, L, n, fw, vec_ok));
  OPP_CHECK_ARG(sums && ws, "focal_loss: null output");
  const int blocks = loss_blocks(n);
  OPP_CHECK_ARG(ws_bytes >= (size_t)blocks * 4 * sizeof(double), "focal_loss: workspace too small");
  double* part = static_cast<double*>(ws);
  {
    const double gt_b = gt_kind == 0 ? 2.0 : (gt_kind == 1 ? 4.0 : 1.0);
    OppProfScope prof(OPP_PROF_FOCAL_FWD, stream, (double)n * (4.0 + gt_b + (weight ? 4.0 : 0.0)));
    if (gt_kind == 0)
      hipLaunchKernelGGL(focal_fwd_kernel<short>, dim3(blocks), dim3(kLossThreads), 0, stream, conf, static_cast<const short*>(gt), fw, n, vec_ok, alpha, gamma, part);
    else if (gt_kind == 1)
      hipLaunchKernelGGL(focal_fwd_kernel<float>, dim3(blocks), dim3(kLossThreads), 0, stream, conf, static_cast<const float*>(gt), fw, n, vec_ok, alpha, gamma, part);
    else
      hipLaunchKernelGGL(focal_fwd_kernel<unsigned char>, dim3(blocks), dim3(kLossThreads), 0, stream, conf, static_cast<const unsigned char*>(gt), fw, n, vec_ok, alpha, gamma, part);
  }
  hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(256), 0, stream, part, blocks, sums);
  OPP_CHECK_LAUNCH("focal_loss forward");
  return OPP_OK;
}

int opp_focal_loss_bwd_ex(const float* conf, const void* gt, int gt_kind, const float* weight, const float* mask0, const float* mask1, int N, int L,
                          size_t n, float alpha, float gamma, const float* scales, float* grad, hipStream_t stream) {
  FocalW fw;
  int vec_ok;
  OPP_TRY(focal_check(conf, gt, gt_kind, weight, mask0, mask1, N, L, n, fw, vec_ok));
  OPP_CHECK_ARG(scales && grad, "focal_loss backward: null argument");
  vec_ok = vec_ok && aligned16(grad);
  const double gt_b = gt_kind == 0 ? 2.0 : (gt_kind == 1 ? 4.0 : 1.0);
  OppProfScope prof(OPP_PROF_FOCAL_BWD, stream, (double)n * (8.0 + gt_b + (weight ? 4.0 : 0.0)));
  const int blocks = loss_blocks(n);
  if (gt_kind == 0)
    hipLaunchKernelGGL(focal_bwd_kernel<short>, dim3(blocks), dim3(kLossThreads), 0, stream, conf, static_cast<const short*>(gt), fw, n, vec_ok, alpha, gamma, scales, grad);
  else if (gt_kind == 1)
    hipLaunchKernelGGL(focal_bwd_kernel<float>, dim3(blocks), dim3(kLossThreads), 0, stream, conf, static_cast<const float*>(gt), fw, n, vec_ok, alpha, gamma, scales, grad);
  else
    hipLaunchKernelGGL(focal_bwd_kernel<unsigned char>, dim3(blocks), dim3(kLossThreads), 0, stream, conf, static_cast<const unsigned char*>(gt), fw, n, vec_ok, alpha, gamma, scales, grad);
  OPP_CHECK_LAUNCH("focal_loss backward");
  return OPP_OK;
}

int opp_focal_loss_fwd(const float* conf, const short* gt, const float* weight, size_t n, float alpha, float gamma, double* sums, void* ws,
                       size_t ws_bytes, hipStream_t stream) {
  return opp_focal_loss_fwd_ex(conf, gt, 0, weight, nullptr, nullptr, 0, 0, n, alpha, gamma, sums, ws, ws_bytes, stream);
}

int opp_focal_loss_bwd(const float* conf, const short* gt, const float* weight, size_t n, float alpha, float gamma, const float* scales, float* grad,
                       hipStream_t stream) {
  return opp_focal_loss_bwd_ex(conf, gt, 0, weight, nullptr, nullptr, 0, 0, n, alpha, gamma, scales, grad, stream);
}

namespace {
struct DsmPlan {
  int vec, col_tiles, row_blocks;
  size_t rpart, cpart, r, c, total;   // float counts
};
DsmPlan dsm_plan(int B, int N, int L) {
  DsmPlan p;
  p.vec = (L % 4 == 0) ? 4 : 1;
  p.col_tiles = (L + 64 * p.vec - 1) / (64 * p.vec);
  p.row_blocks = (N + kDsmRows - 1) / kDsmRows;
  p.rpart = (size_t)B * N * p.col_tiles;
  p.cpart = (size_t)B * p.row_blocks * L;
  p.r = (size_t)B * N;
  p.c = (size_t)B * L;
  p.total = p.rpart + p.cpart + p.r + p.c;
  return p;
}
}  // namespace

size_t opp_dual_softmax_bwd_ws_bytes(int B, int N, int L) { return dsm_plan(B, N, L).total * sizeof(float) + 64; }

int opp_dual_softmax_bwd(const float* g, const float* sim, const float* lse_row, const float* lse_col, int B, int N, int L,
                         float* ds, void* ws, size_t ws_bytes, hipStream_t stream) {
  OPP_CHECK_ARG(g && sim && lse_row && lse_col && ds && ws && B > 0 && N > 0 && L > 0, "dual_softmax backward: null argument / empty input");
  const DsmPlan p = dsm_plan(B, N, L);
  OPP_CHECK_ARG(ws_bytes >= p.total * sizeof(float), "dual_softmax backward: workspace too small");
  OPP_CHECK_ARG(p.vec == 1 || (aligned16(g) && aligned16(sim)), "dual_softmax backward: operands must be 16-byte aligned");
  OPP_CHECK_ARG(B <= 65535 && p.col_tiles <= 65535, "dual_softmax backward: grid too large");
  float* rpart = static_cast<float*>(ws);
  float* cpart = rpart + p.rpart;
  float* r = cpart + p.cpart;
  float* c = r + p.r;
  const dim3 grid(p.row_blocks, p.col_tiles, B);
  if (p.vec == 4)
    hipLaunchKernelGGL(dsm_sums_kernel<4>, grid, dim3(256), 0, stream, g, sim, lse_row, lse_col, N, L, p.col_tiles, rpart, cpart);
  else
    hipLaunchKernelGGL(dsm_sums_kernel<1>, grid, dim3(256), 0, stream, g, sim, lse_row, lse_col, N, L, p.col_tiles, rpart, cpart);
  const size_t nred = p.r + p.c;
  hipLaunchKernelGGL(dsm_reduce_kernel, dim3((unsigned)((nred + 255) / 256)), dim3(256), 0, stream, rpart, cpart, B, N, L, p.col_tiles,
                     p.row_blocks, r, c);
  const size_t total = (size_t)B * N * L;
  const size_t want = (total + 256 * 8 - 1) / (256 * 8);
  hipLaunchKernelGGL(dsm_apply_kernel, dim3((unsigned)(want > 4096 ? 4096 : (want < 1 ? 1 : want))), dim3(256), 0, stream, g, sim, lse_row,
                     lse_col, N, L, total, r, c, ds);
  OPP_CHECK_LAUNCH("dual_softmax backward");
  return OPP_OK;
}
