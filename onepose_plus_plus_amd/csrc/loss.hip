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

__global__ __launch_bounds__(kLossThreads) void focal_fwd_kernel(const float* __restrict__ conf, const short* __restrict__ gt,
                                                                 const float* __restrict__ weight, size_t n, float alpha,
                                                                 float gamma, double* __restrict__ part) {
  Acc a;
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 c = reinterpret_cast<const float4*>(conf)[i];
    const short4 g = reinterpret_cast<const short4*>(gt)[i];
    float4 w = make_float4(1.f, 1.f, 1.f, 1.f);
    if (weight) w = reinterpret_cast<const float4*>(weight)[i];
    focal_add(c.x, g.x, w.x, alpha, gamma, a);
    focal_add(c.y, g.y, w.y, alpha, gamma, a);
    focal_add(c.z, g.z, w.z, alpha, gamma, a);
    focal_add(c.w, g.w, w.w, alpha, gamma, a);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // ragged tail
    const size_t i = (n4 << 2) + threadIdx.x;
    focal_add(conf[i], gt[i], weight ? weight[i] : 1.f, alpha, gamma, a);
  }
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

__global__ void focal_finalize_kernel(const double* __restrict__ part, int blocks, double* __restrict__ sums) {
  const int e = threadIdx.x;
  if (e >= 4) return;
  double t = 0.0;
  for (int b = 0; b < blocks; ++b) t += part[(size_t)b * 4 + e];
  sums[e] = t;
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

__global__ __launch_bounds__(kLossThreads) void focal_bwd_kernel(const float* __restrict__ conf, const short* __restrict__ gt,
                                                                 const float* __restrict__ weight, size_t n, float alpha,
                                                                 float gamma, const float* __restrict__ scales,
                                                                 float* __restrict__ grad) {
  const float s_pos = scales[0], s_neg = scales[1];
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 c = reinterpret_cast<const float4*>(conf)[i];
    const short4 g = reinterpret_cast<const short4*>(gt)[i];
    float4 w = make_float4(1.f, 1.f, 1.f, 1.f);
    if (weight) w = reinterpret_cast<const float4*>(weight)[i];
    float4 o;
    o.x = focal_grad(c.x, g.x, w.x, alpha, gamma, s_pos, s_neg);
    o.y = focal_grad(c.y, g.y, w.y, alpha, gamma, s_pos, s_neg);
    o.z = focal_grad(c.z, g.z, w.z, alpha, gamma, s_pos, s_neg);
    o.w = focal_grad(c.w, g.w, w.w, alpha, gamma, s_pos, s_neg);
    reinterpret_cast<float4*>(grad)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = (n4 << 2) + threadIdx.x;
    grad[i] = focal_grad(conf[i], gt[i], weight ? weight[i] : 1.f, alpha, gamma, s_pos, s_neg);
  }
}

int loss_blocks(size_t n) {
  const size_t want = (n / 4 + kLossThreads * 8 - 1) / (kLossThreads * 8);
  return (int)(want < 1 ? 1 : (want > kLossMaxBlocks ? kLossMaxBlocks : want));
}

bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

}  // namespace

size_t opp_focal_loss_ws_bytes(size_t n) { return (size_t)loss_blocks(n) * 4 * sizeof(double); }

int opp_focal_loss_fwd(const float* conf, const short* gt, const float* weight, size_t n, float alpha, float gamma,
                       double* sums, void* ws, size_t ws_bytes, hipStream_t stream) {
  OPP_CHECK_ARG(conf && gt && sums && ws && n > 0, "focal_loss: null argument / empty input");
  OPP_CHECK_ARG(aligned16(conf) && (reinterpret_cast<size_t>(gt) & 7) == 0 && (!weight || aligned16(weight)),
                "focal_loss: conf / weight must be 16-byte aligned, conf_gt 8-byte aligned");
  const int blocks = loss_blocks(n);
  OPP_CHECK_ARG(ws_bytes >= (size_t)blocks * 4 * sizeof(double), "focal_loss: workspace too small");
  double* part = static_cast<double*>(ws);
  {
    OppProfScope prof(OPP_PROF_FOCAL_FWD, stream, (double)n * (weight ? 10.0 : 6.0));
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(blocks), dim3(kLossThreads), 0, stream, conf, gt, weight, n, alpha, gamma, part);
  }
  hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(64), 0, stream, part, blocks, sums);
  OPP_CHECK_LAUNCH("focal_loss forward");
  return OPP_OK;
}

int opp_focal_loss_bwd(const float* conf, const short* gt, const float* weight, size_t n, float alpha, float gamma,
                       const float* scales, float* grad, hipStream_t stream) {
  OPP_CHECK_ARG(conf && gt && scales && grad && n > 0, "focal_loss backward: null argument / empty input");
  OPP_CHECK_ARG(aligned16(conf) && aligned16(grad) && (reinterpret_cast<size_t>(gt) & 7) == 0 && (!weight || aligned16(weight)),
                "focal_loss backward: conf / grad / weight must be 16-byte aligned, conf_gt 8-byte aligned");
  OppProfScope prof(OPP_PROF_FOCAL_BWD, stream, (double)n * (weight ? 14.0 : 10.0));
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(loss_blocks(n)), dim3(kLossThreads), 0, stream, conf, gt, weight, n, alpha, gamma, scales,
                     grad);
  OPP_CHECK_LAUNCH("focal_loss backward");
  return OPP_OK;
}
