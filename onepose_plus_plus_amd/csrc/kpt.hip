// 3D keypoint normalisation + MLP positional encoding of the SfM point cloud, fused with the
// add into the channel-major coarse descriptor bank and the [C][N] -> [N][C] re-layout.
//
// Reference:
//   normalize_3d_keypoints    src/models/OnePosePlus/utils/normalize.py:16-26
//   KeypointEncoding_linear   src/models/OnePosePlus/utils/position_encoding.py:46-79
//   call site                 src/models/OnePosePlus/OnePosePlusModel.py:144-156
//   transpose to [N,C]        src/models/OnePosePlus/loftr_module/transformer.py:145
//
// The descriptor bank read ([256][N] fp32, channel-major) is the HBM-bound part: it is read
// with lanes along N (128 B per 32 points) and transposed through LDS.
#include "opp_common.h"

namespace {

__device__ __forceinline__ float wave_sum_k(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// stats[0..2] = per-batch mean (B == 1), stats[3] = 0.6 * max bbox extent of batch 0
__global__ __launch_bounds__(1024) void kpt_stats_kernel(const float* __restrict__ kpts, int n, float* __restrict__ stats) {
  __shared__ float red[16][9];
  const int tid = threadIdx.x;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, sm[3] = {0.f, 0.f, 0.f};
  for (int i = tid; i < n; i += 1024) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = kpts[i * 3 + a];
      mn[a] = fminf(mn[a], v);
      mx[a] = fmaxf(mx[a], v);
      sm[a] += v;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
      sm[a] += __shfl_xor(sm[a], o, 64);
    }
  }
  if ((tid & 63) == 0) {
    const int w = tid >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[w][a] = mn[a];
      red[w][3 + a] = mx[a];
      red[w][6 + a] = sm[a];
    }
  }
  __syncthreads();
  if (tid == 0) {
    float ext = 0.f;
    for (int a = 0; a < 3; ++a) {
      float lo = INFINITY, hi = -INFINITY, s = 0.f;
      for (int w = 0; w < 16; ++w) {
        lo = fminf(lo, red[w][a]);
        hi = fmaxf(hi, red[w][3 + a]);
        s += red[w][6 + a];
      }
      stats[a] = s / (float)n;
      ext = fmaxf(ext, hi - lo);
    }
    stats[3] = ext * 0.6f;
  }
}

constexpr int kPts = 32;       // points per block
constexpr int kStrideA = 257;  // LDS row strides (odd -> conflict-free column access)
constexpr int kStrideB = 129;

template <int CIN, int COUT>
__device__ __forceinline__ void mlp_layer(const float* __restrict__ in, int in_stride, float* __restrict__ out,
                                          int out_stride, const float* __restrict__ wt,
                                          const float* __restrict__ bias) {
  constexpr int G = 256 / COUT;      // point groups
  constexpr int PPT = kPts / G;      // points per thread
  const int c = threadIdx.x % COUT;
  const int pg = threadIdx.x / COUT;
  float acc[PPT];
  const float bv = bias[c];
#pragma unroll
  for (int i = 0; i < PPT; ++i) acc[i] = bv;
  for (int k = 0; k < CIN; ++k) {
    const float w = wt[k * COUT + c];
#pragma unroll
    for (int i = 0; i < PPT; ++i) acc[i] = fmaf(w, in[(pg + i * G) * in_stride + k], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) out[(pg + i * G) * out_stride + c] = acc[i];
}

// per-point channel norm (InstanceNorm1d applied to [B,L,C]: quirk q3) + ReLU, in place
template <int C>
__device__ __forceinline__ void point_norm_relu(float* __restrict__ buf, int stride, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p = wave * (kPts / 4); p < (wave + 1) * (kPts / 4); ++p) {
    float* row = buf + p * stride;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += row[c];
    const float mean = wave_sum_k(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float d = row[c] - mean;
      q += d * d;
    }
    const float var = wave_sum_k(q) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int c = lane; c < C; c += 64) {
      const float y = (row[c] - mean) * rstd;
      row[c] = y < 0.f ? 0.f : y;   // ReLU that propagates NaN like torch (a zero-extent cloud gives NaN upstream)
    }
  }
}

// MLP 3 -> H1 -> H2 -> H3 -> 256 with the shipped widths (32, 64, 128).
// wt*: transposed weights [Cin][Cout]; tokens[n][c] = bank[c][n] + enc[n][c]
__global__ __launch_bounds__(256) void kpt_encode_kernel(const float* __restrict__ kpts, const float* __restrict__ stats,
                                                         const float* __restrict__ bank, int n,
                                                         const float* __restrict__ wt0, const float* __restrict__ b0,
                                                         const float* __restrict__ wt1, const float* __restrict__ b1,
                                                         const float* __restrict__ wt2, const float* __restrict__ b2,
                                                         const float* __restrict__ wt3, const float* __restrict__ b3,
                                                         float* __restrict__ tokens, int ldo, float eps) {
  __shared__ float bufA[kPts * kStrideA];
  __shared__ float bufB[kPts * kStrideB];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * kPts;
  if (tid < kPts * 3) {
    const int p = tid / 3, a = tid - p * 3;
    const int idx = n0 + p;
    float v = 0.f;
    if (idx < n) v = (kpts[idx * 3 + a] - stats[a]) / stats[3];
    bufA[p * kStrideA + a] = v;
  }
  __syncthreads();
  mlp_layer<3, 32>(bufA, kStrideA, bufB, kStrideB, wt0, b0);
  __syncthreads();
  point_norm_relu<32>(bufB, kStrideB, eps);
  __syncthreads();
  mlp_layer<32, 64>(bufB, kStrideB, bufA, kStrideA, wt1, b1);
  __syncthreads();
  point_norm_relu<64>(bufA, kStrideA, eps);
  __syncthreads();
  mlp_layer<64, 128>(bufA, kStrideA, bufB, kStrideB, wt2, b2);
  __syncthreads();
  point_norm_relu<128>(bufB, kStrideB, eps);
  __syncthreads();
  mlp_layer<128, 256>(bufB, kStrideB, bufA, kStrideA, wt3, b3);
  __syncthreads();
  {  // descriptors + encoding (position_encoding.py:60), lanes along the point axis of the bank
    const int p = tid & 31;
    const int idx = n0 + p;
    if (idx < n) {
      for (int c = tid >> 5; c < 256; c += 8) bufA[p * kStrideA + c] += bank[(size_t)c * n + idx];
    }
  }
  __syncthreads();
  for (int p = 0; p < kPts; ++p) {
    const int idx = n0 + p;
    if (idx < n) tokens[(size_t)idx * ldo + tid] = bufA[p * kStrideA + tid];
  }
}

// tokens[n][c] = bank[c][n]  (keypoints_encoding disabled)
__global__ __launch_bounds__(256) void bank_transpose_kernel(const float* __restrict__ bank, int n, int C,
                                                             float* __restrict__ tokens, int ldo) {
  __shared__ float tile[32][33];
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, idx = n0 + tx;
    tile[j][tx] = (c < C && idx < n) ? bank[(size_t)c * n + idx] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int idx = n0 + j, c = c0 + tx;
    if (idx < n && c < C) tokens[(size_t)idx * ldo + c] = tile[tx][j];
  }
}

}  // namespace

int opp_kpt_stats(const float* kpts, int n, float* stats, hipStream_t stream) {
  OPP_CHECK_ARG(n > 0, "kpt_stats: empty cloud");
  hipLaunchKernelGGL(kpt_stats_kernel, dim3(1), dim3(1024), 0, stream, kpts, n, stats);
  OPP_CHECK_LAUNCH("kpt_stats_kernel");
  return OPP_OK;
}

int opp_kpt_encode(const float* kpts, const float* stats, const float* bank, int n, const float* const* wt,
                   const float* const* bias, float* tokens, int ldo, hipStream_t stream) {
  hipLaunchKernelGGL(kpt_encode_kernel, dim3(opp_cdiv(n, kPts)), dim3(256), 0, stream, kpts, stats, bank, n, wt[0], bias[0],
                     wt[1], bias[1], wt[2], bias[2], wt[3], bias[3], tokens, ldo, 1e-5f);
  OPP_CHECK_LAUNCH("kpt_encode_kernel");
  return OPP_OK;
}

int opp_bank_transpose(const float* bank, int n, int C, float* tokens, int ldo, hipStream_t stream) {
  hipLaunchKernelGGL(bank_transpose_kernel, dim3(opp_cdiv(n, 32), opp_cdiv(C, 32)), dim3(256), 0, stream, bank, n, C, tokens, ldo);
  OPP_CHECK_LAUNCH("bank_transpose_kernel");
  return OPP_OK;
}
