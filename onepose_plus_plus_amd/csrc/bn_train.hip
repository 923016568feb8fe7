// BatchNorm2d in TRAINING mode (batch statistics) for the ResNet-FPN backbone: the reference trains with plain
// nn.BatchNorm2d layers (src/models/OnePosePlus/backbone/resnet.py:25-26, :101-102, :110-113), i.e. every BN
// normalises with the mean / biased variance of the current batch over (B, H, W) and updates its running statistics
// with momentum 0.1 and the UNBIASED variance (torch.nn.BatchNorm2d defaults; no SyncBN upstream).
//
// The convolution writes its raw output (no folded scale / shift); then
//   bn_partial_kernel   one pass over the [rows][ld] NHWC tensor: per-channel sum and sum of squares in fp64
//                       (HBM-bound: rows * ld * 4 bytes read once)
//   bn_finalize_kernel  mean, 1/sqrt(var + eps) (fp32, like at::native batch_norm), batch mean / unbiased variance
//                       for the running-statistics update
//   bn_apply_kernel     y = (x - mean) * invstd * gamma + beta (+ residual) -> ReLU / LeakyReLU, in place or out of place
#include "opp_internal.h"

namespace {

constexpr int kBnRowsPerBlock = 256;

__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ y, int rows, int ld, double* __restrict__ part) {
  // blockDim = (64, 4): x = channel quad, y = row lane
  __shared__ double red[4][64][8];
  const int q = threadIdx.x, lane_r = threadIdx.y, Q = ld >> 2;
  const int r0 = blockIdx.x * kBnRowsPerBlock;
  const int r1 = min(rows, r0 + kBnRowsPerBlock);
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  if (q < Q) {
    for (int r = r0 + lane_r; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(y + (size_t)r * ld + q * 4);
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y; ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[lane_r][q][e] = s[e];
    red[lane_r][q][4 + e] = ss[e];
  }
  __syncthreads();
  if (lane_r == 0 && q < Q) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double t = red[0][q][e] + red[1][q][e] + red[2][q][e] + red[3][q][e];
      // part [blocks][2][ld]
      part[((size_t)blockIdx.x * 2 + (e >> 2)) * ld + q * 4 + (e & 3)] = t;
    }
  }
}

// 16 channels x 16 slices per block: slice k sums the block partials k, k + 16, ... in ascending order, then the 16 slice
// sums are added in slice order (fixed order = deterministic; a single thread per channel walking 1024 partials of a
// 256 x 256 x B = 4 map took 125 us per layer)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ part, int blocks, int rows, int ld, int C,
                                                          float eps, float* __restrict__ mean, float* __restrict__ invstd,
                                                          float* __restrict__ stat_out) {
  __shared__ double red[16][16][2];
  const int cx = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  double s = 0.0, ss = 0.0;
  if (c < ld) {
    for (int b = sl; b < blocks; b += 16) {
      s += part[((size_t)b * 2) * ld + c];
      ss += part[((size_t)b * 2 + 1) * ld + c];
    }
  }
  red[sl][cx][0] = s;
  red[sl][cx][1] = ss;
  __syncthreads();
  if (sl != 0 || c >= ld) return;
  s = 0.0;
  ss = 0.0;
  for (int k = 0; k < 16; ++k) {
    s += red[k][cx][0];
    ss += red[k][cx][1];
  }
  const double m = s / (double)rows;
  double var = ss / (double)rows - m * m;
  var = var < 0.0 ? 0.0 : var;
  mean[c] = c < C ? (float)m : 0.f;
  invstd[c] = c < C ? 1.0f / sqrtf((float)var + eps) : 0.f;
  if (stat_out && c < C) {   // [2][C]: batch mean, unbiased batch variance (what the running statistics absorb)
    stat_out[c] = (float)m;
    stat_out[C + c] = (float)(rows > 1 ? var * (double)rows / (double)(rows - 1) : var);
  }
}

__global__ void bn_apply_kernel(const float* __restrict__ y, int rows, int ld, int C, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ res, int act, float* __restrict__ out) {
  const int Q = ld >> 2;
  const size_t total = (size_t)rows * Q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    float x[4] = {v.x, v.y, v.z, v.w}, o[4];
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (res) {
      const float4 rv = reinterpret_cast<const float4*>(res)[i];
      r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = q * 4 + e;
      float t = 0.f;
      if (c < C) {
        t = (x[e] - mean[c]) * invstd[c] * gamma[c] + beta[c];
        t += r[e];
        if (act == OPP_ACT_RELU) t = t < 0.f ? 0.f : t;
        else if (act == OPP_ACT_LEAKY) t = t > 0.f ? t : 0.01f * t;
      }
      o[e] = t;      // padded channels stay exactly zero
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

size_t opp_bn_train_scratch_bytes(int rows, int ld) {
  const size_t blocks = (size_t)opp_cdiv(rows, kBnRowsPerBlock);
  return opp_align(blocks * 2 * ld * sizeof(double)) + 2 * opp_align((size_t)ld * sizeof(float));
}

int opp_bn_train(const float* y, int rows, int ld, int C, const float* gamma, const float* beta, float eps, const float* res,
                 int act, float* out, float* stat_out, void* scratch, hipStream_t stream, float* save_mean, float* save_invstd) {
  OPP_CHECK_ARG(y && gamma && beta && out && scratch && rows > 0 && ld % 4 == 0 && ld <= 256 && C <= ld, "bn_train: bad argument");
  const int blocks = opp_cdiv(rows, kBnRowsPerBlock);
  double* part = static_cast<double*>(scratch);
  float* mean = reinterpret_cast<float*>(static_cast<char*>(scratch) + opp_align((size_t)blocks * 2 * ld * sizeof(double)));
  float* invstd = reinterpret_cast<float*>(reinterpret_cast<char*>(mean) + opp_align((size_t)ld * sizeof(float)));
  if (save_mean && save_invstd) {   // the training step's tape keeps the batch statistics [ld] for the backward (conv_bwd.hip)
    mean = save_mean;
    invstd = save_invstd;
  }
  hipLaunchKernelGGL(bn_partial_kernel, dim3(blocks), dim3(64, 4), 0, stream, y, rows, ld, part);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(opp_cdiv(ld, 16)), dim3(256), 0, stream, part, blocks, rows, ld, C, eps, mean, invstd, stat_out);
  const size_t total = (size_t)rows * (ld / 4);
  const int ablocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ablocks), dim3(256), 0, stream, y, rows, ld, C, mean, invstd, gamma, beta, res, act, out);
  OPP_CHECK_LAUNCH("bn_train kernels");
  return OPP_OK;
}
