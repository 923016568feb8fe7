// LayerNorm and the linear-attention core (elu+1 kernel feature map) for gfx950.
//
// Reference:
//   LinearAttention.forward   src/models/OnePosePlus/loftr_module/linear_attention.py:29-61
//   LoFTREncoderLayer.forward src/models/OnePosePlus/loftr_module/transformer.py:65-94
//
// The q/k/v projections, merge and MLP are GEMMs (gemm_mfma.hip); their epilogue already
// applied phi(x) = elu(x)+1 to Q and K and divided V by the source length.  What is left
// here is HBM-bound: the per-head reduction KV = sum_s phi(K_s)^T V_s, Ksum = sum_s phi(K_s)
// over the S source tokens (reads K,V once: S * 2 * C * 4 bytes), and the per-token apply
// out = (phi(Q) KV) / (phi(Q).Ksum + eps) * S.
//
// Token streams are regular: a stream is [n_seg][seg_len][ld] (coarse: one segment of 4096
// image cells and one of N points; fine: M segments of 25 window cells and M of 1 point).
#include "opp_internal.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------
// LayerNorm over the last dim (C = 64*VPT), one wave per row, optional residual add:
//   out = (res ? res : 0) + (x - mean) / sqrt(var + eps) * gamma + beta
// transformer.py:88 (norm1) and :93-94 (norm2 + residual).
// ---------------------------------------------------------------------------------------
template <int VPT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ res, int ldres,
                                                        float* __restrict__ out, int ldo, int rows,
                                                        float eps) {
  constexpr int RPW = 4;   // rows per wave: all loads issued before the first reduction
  constexpr int C = 64 * VPT;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  float v[RPW][VPT], rv[RPW][VPT], gm[VPT], bt[VPT];
  typedef float vec_t __attribute__((ext_vector_type(VPT)));     // one 8 / 16 B access per lane and row
  auto ldv = [](const float* p, float (&dst)[VPT]) {
    const vec_t t = *reinterpret_cast<const vec_t*>(p);
#pragma unroll
    for (int i = 0; i < VPT; ++i) dst[i] = t[i];
  };
  ldv(gamma + lane * VPT, gm);
  ldv(beta + lane * VPT, bt);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const bool ok = row0 + r < rows;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      v[r][i] = 0.f;
      rv[r][i] = 0.f;
    }
    if (ok) ldv(x + (size_t)(row0 + r) * ldx + lane * VPT, v[r]);
    if (ok && res) ldv(res + (size_t)(row0 + r) * ldres + lane * VPT, rv[r]);
  }
  // the four rows' reductions advance in lock step: four independent shuffle chains per step
  float sm[RPW], mean[RPW], rstd[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    sm[r] = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) sm[r] += v[r][i];
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) sm[r] = opp_wave_sum_dpp(sm[r]);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    mean[r] = sm[r] / (float)C;
    sm[r] = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const float d = v[r][i] - mean[r];
      sm[r] = opp_ln_sq_acc(d, sm[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) sm[r] = opp_wave_sum_dpp(sm[r]);
#pragma unroll
  for (int r = 0; r < RPW; ++r) rstd[r] = 1.0f / sqrtf(sm[r] / (float)C + eps);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    if (row0 + r < rows) {
      vec_t o;
#pragma unroll
      for (int i = 0; i < VPT; ++i) {
        float y = opp_ln_affine(v[r][i], mean[r], rstd[r], gm[i], bt[i]);
        if (res) y = rv[r][i] + y;
        o[i] = y;
      }
      *reinterpret_cast<vec_t*>(out + (size_t)(row0 + r) * ldo + lane * VPT) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------
// KV / Ksum partial reduction.  Block = C threads; thread t <-> (head h = t / D, column v = t % D)
// keeps KV[h][0..D-1][v] in registers.  grid = (chunks_per_seg, n_seg).
//   kv_part [seg][chunk][h][d][v]   ks_part [seg][chunk][h*D + d]
// ---------------------------------------------------------------------------------------
template <int D, int C>
__global__ __launch_bounds__(C) void linattn_kv_partial_kernel(const float* __restrict__ kmat,
                                                               const float* __restrict__ vmat, int ld,
                                                               int seg_len, int chunk_len,
                                                               float* __restrict__ kv_part,
                                                               float* __restrict__ ks_part) {
  constexpr int TB = 8;  // tokens staged per barrier
  __shared__ __attribute__((aligned(16))) float ksh[TB][C];
  const int t = threadIdx.x;
  const int h = t / D;
  const int chunk = blockIdx.x;
  const int seg = blockIdx.y;
  const int s_begin = chunk * chunk_len;
  const int s_end = min(seg_len, s_begin + chunk_len);
  const size_t base = (size_t)seg * seg_len;

  float acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.f;
  float ksum = 0.f;

  for (int s0 = s_begin; s0 < s_end; s0 += TB) {
    float vv[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int s = s0 + j;
      float kval = 0.f, vval = 0.f;
      if (s < s_end) {
        kval = kmat[(base + s) * ld + t];
        vval = vmat[(base + s) * ld + t];
      }
      ksh[j][t] = kval;
      vv[j] = vval;
      ksum += kval;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const float4* kr = reinterpret_cast<const float4*>(&ksh[j][h * D]);
#pragma unroll
      for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 k4 = kr[d4];
        acc[d4 * 4 + 0] = fmaf(k4.x, vv[j], acc[d4 * 4 + 0]);
        acc[d4 * 4 + 1] = fmaf(k4.y, vv[j], acc[d4 * 4 + 1]);
        acc[d4 * 4 + 2] = fmaf(k4.z, vv[j], acc[d4 * 4 + 2]);
        acc[d4 * 4 + 3] = fmaf(k4.w, vv[j], acc[d4 * 4 + 3]);
      }
    }
    __syncthreads();
  }
  const size_t p = (size_t)seg * gridDim.x + chunk;
  float* kvp = kv_part + p * (size_t)(C * D);
  const int v = t % D;
#pragma unroll
  for (int d = 0; d < D; ++d) kvp[(h * D + d) * D + v] = acc[d];
  ks_part[p * C + t] = ksum;
}

// fixed-order sum of the chunk partials (deterministic): out[seg][i] = sum_c part[seg][c][i]
__global__ void linattn_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                      int n_chunks, int width) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int seg = blockIdx.y;
  if (i >= width) return;
  const float* p = part + (size_t)seg * n_chunks * width + i;
  float s = 0.f;
  for (int c = 0; c < n_chunks; ++c) s += p[(size_t)c * width];
  out[(size_t)seg * width + i] = s;
}

// ---------------------------------------------------------------------------------------
// apply: msg[l][h*D + v] = (sum_d Q[l,h,d] KV[h,d,v]) * (1 / (sum_d Q[l,h,d] Ksum[h,d] + eps)) * S
// grid = (chunks_per_seg, n_seg); the segment index selects the KV of the SOURCE stream.
// ---------------------------------------------------------------------------------------
template <int D, int C>
__global__ __launch_bounds__(C) void linattn_apply_kernel(const float* __restrict__ qmat, int ldq,
                                                          const float* __restrict__ kv,
                                                          const float* __restrict__ ks,
                                                          float* __restrict__ out, int ldo,
                                                          int seg_len, int chunk_len, float src_len,
                                                          float eps) {
  constexpr int TB = 8;
  __shared__ __attribute__((aligned(16))) float qsh[TB][C];
  const int t = threadIdx.x;
  const int h = t / D;
  const int v = t % D;
  const int seg = blockIdx.y;
  const int s_begin = blockIdx.x * chunk_len;
  const int s_end = min(seg_len, s_begin + chunk_len);
  const size_t base = (size_t)seg * seg_len;

  float kvr[D], ksr[D];
  const float* kvp = kv + (size_t)seg * (C * D);
  const float* ksp = ks + (size_t)seg * C;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    kvr[d] = kvp[(h * D + d) * D + v];
    ksr[d] = ksp[h * D + d];
  }
  for (int s0 = s_begin; s0 < s_end; s0 += TB) {
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int s = s0 + j;
      qsh[j][t] = s < s_end ? qmat[(base + s) * ldq + t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int s = s0 + j;
      if (s >= s_end) break;
      const float4* qr = reinterpret_cast<const float4*>(&qsh[j][h * D]);
      float num = 0.f, den = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 q4 = qr[d4];
        num = fmaf(q4.x, kvr[d4 * 4 + 0], num);
        den = fmaf(q4.x, ksr[d4 * 4 + 0], den);
        num = fmaf(q4.y, kvr[d4 * 4 + 1], num);
        den = fmaf(q4.y, ksr[d4 * 4 + 1], den);
        num = fmaf(q4.z, kvr[d4 * 4 + 2], num);
        den = fmaf(q4.z, ksr[d4 * 4 + 2], den);
        num = fmaf(q4.w, kvr[d4 * 4 + 3], num);
        den = fmaf(q4.w, ksr[d4 * 4 + 3], den);
      }
      const float z = 1.0f / (den + eps);
      out[(base + s) * ldo + t] = (num * z) * src_len;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Fine level (n_seg = M matches, 25 window tokens + 1 point token each, C = 128, D = 16): the whole linear attention
// of one segment pair in one workgroup.  Q | K | V of the pair's tokens are staged in LDS with coalesced 16-byte
// loads (39 KB for 26 tokens), thread t <-> (head h = t / D, column v = t % D) keeps KV[h][.][v] of both streams in
// registers - it is exactly the slice the apply needs, so KV never goes to memory; Ksum crosses threads through LDS.
// Same arithmetic, in the same order, as linattn_kv_partial_kernel + linattn_apply_kernel (bit-identical results).
//   qkv rows: stream 0 = seg * len0 + l, stream 1 = n_seg * len0 + seg * len1 + l ; out rows likewise
// ---------------------------------------------------------------------------------------
template <int D, int C>
__global__ __launch_bounds__(C) void linattn_small_pair_kernel(const float* __restrict__ qkv, int ld, int n_seg, int len0,
                                                               int len1, int cross, float* __restrict__ out, int ldo,
                                                               float eps) {
  extern __shared__ __attribute__((aligned(16))) float sh[];   // K | V [len0 + len1][2 C], Ksum [2][C], z [rows][H]
  const int t = threadIdx.x;
  const int h = t / D;
  const int seg = blockIdx.x;
  const int rows = len0 + len1;
  const size_t row0 = (size_t)seg * len0, row1 = (size_t)n_seg * len0 + (size_t)seg * len1;
  auto grow = [&](int r) -> size_t { return r < len0 ? row0 + r : row1 + (r - len0); };
  constexpr int PER_ROW = 2 * C / 4;
  constexpr int H = C / D;
  float4* sh4 = reinterpret_cast<float4*>(sh);
  for (int e = t; e < rows * PER_ROW; e += C) {
    const int r = e / PER_ROW, c4 = e - r * PER_ROW;
    sh4[e] = *reinterpret_cast<const float4*>(qkv + grow(r) * ld + C + c4 * 4);
  }
  __syncthreads();
  float kv[2][D];
  float* ks_sh = sh + (size_t)rows * 2 * C;
  float* z_sh = ks_sh + 2 * C;
#pragma unroll
  for (int st = 0; st < 2; ++st) {
#pragma unroll
    for (int d = 0; d < D; ++d) kv[st][d] = 0.f;
    float ksum = 0.f;
    const int r_begin = st ? len0 : 0, r_end = st ? rows : len0;
#pragma unroll 5
    for (int r = r_begin; r < r_end; ++r) {
      const float* row = sh + (size_t)r * 2 * C;
      const float vv = row[C + t];
      ksum += row[t];
      const float4* kr = reinterpret_cast<const float4*>(row + h * D);
#pragma unroll
      for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 k4 = kr[d4];
        kv[st][d4 * 4 + 0] = fmaf(k4.x, vv, kv[st][d4 * 4 + 0]);
        kv[st][d4 * 4 + 1] = fmaf(k4.y, vv, kv[st][d4 * 4 + 1]);
        kv[st][d4 * 4 + 2] = fmaf(k4.z, vv, kv[st][d4 * 4 + 2]);
        kv[st][d4 * 4 + 3] = fmaf(k4.w, vv, kv[st][d4 * 4 + 3]);
      }
    }
    ks_sh[st * C + t] = ksum;
  }
  __syncthreads();
  // self: each stream attends to itself; cross: to the other stream's K, V.  The normaliser is the same for the D
  // columns of a head: one thread per (token, head) computes it.  Q comes straight from memory (the 16 lanes of a
  // head share one 64-byte segment): staging only K and V keeps the footprint at 27 KB -> every match of a
  // 1500-match image is resident at once
  for (int e = t; e < rows * H; e += C) {
    const int r = e / H, hh = e - r * H;
    const int src = (r < len0 ? 0 : 1) ^ (cross ? 1 : 0);
    const float4* qr = reinterpret_cast<const float4*>(qkv + grow(r) * ld + hh * D);
    const float4* kr = reinterpret_cast<const float4*>(ks_sh + src * C + hh * D);
    float den = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < D / 4; ++d4) {
      const float4 q4 = qr[d4], k4 = kr[d4];
      den = fmaf(q4.x, k4.x, den);
      den = fmaf(q4.y, k4.y, den);
      den = fmaf(q4.z, k4.z, den);
      den = fmaf(q4.w, k4.w, den);
    }
    z_sh[e] = 1.0f / (den + eps);
  }
  __syncthreads();
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int src = cross ? 1 - st : st;
    const float src_len = (float)(src ? len1 : len0);
    const int r_begin = st ? len0 : 0, r_end = st ? rows : len0;
    float kvs[D];   // compile-time register indices only
#pragma unroll
    for (int d = 0; d < D; ++d) kvs[d] = cross ? kv[1 - st][d] : kv[st][d];
#pragma unroll 5
    for (int r = r_begin; r < r_end; ++r) {
      const size_t g = grow(r);
      const float4* qr = reinterpret_cast<const float4*>(qkv + g * ld + h * D);
      float num = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 q4 = qr[d4];
        num = fmaf(q4.x, kvs[d4 * 4 + 0], num);
        num = fmaf(q4.y, kvs[d4 * 4 + 1], num);
        num = fmaf(q4.z, kvs[d4 * 4 + 2], num);
        num = fmaf(q4.w, kvs[d4 * 4 + 3], num);
      }
      out[g * ldo + t] = (num * z_sh[r * H + h]) * src_len;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Coarse level (one segment per stream, C = 256, D = 32): KV / Ksum on the fp32 MFMA.
// One block = one wave = one head of one chunk of kPairChunk tokens of ONE stream (grid = chunks x 8 heads,
// >= 4 waves per CU at 5k points: the reduction is bound by loads in flight, not by the MFMA):
//   KV_h[d][v] += sum_t K[t][h*32+d] * V[t][h*32+v]   ==  mfma_32x32x2(A = K^T, B = V), 2 tokens / MFMA
// Both streams are covered by one launch.  A workgroup = four waves = four consecutive chunks of ONE stream: their accumulators are summed
// through LDS in a fixed order ((w0 + w1) + (w2 + w3)), so one partial per 256 tokens leaves (r05: 36 instead of 142 partials at 4096 +
// 5000 tokens -- the fixed-order merge behind it, linattn_reduce_pair_kernel, reads a quarter of the data).  Deterministic.
// ---------------------------------------------------------------------------------------
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kPairChunk = 64;
constexpr int kPairGroup = 4;       // chunks (waves) per workgroup
__global__ __launch_bounds__(256) void linattn_kv_mfma_kernel(const float* __restrict__ qkv, int ld, int len0, int len1,
                                                               int groups0, float* __restrict__ kv_part,
                                                               float* __restrict__ ks_part) {
  constexpr int C = 256, CHUNK = kPairChunk;
  __shared__ float acc_sh[kPairGroup][1024];
  __shared__ float ks_sh[kPairGroup][32];
  const int grp = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  const int stream = grp >= groups0 ? 1 : 0;
  const int cidx = (stream ? grp - groups0 : grp) * kPairGroup + wave;      // chunk of the stream (past its end: an all-zero contribution)
  const int seg_len = stream ? len1 : len0;
  const int tok0 = stream ? len0 : 0;
  const int s_begin = cidx * CHUNK;
  const int s_end = min(seg_len, s_begin + CHUNK);
  const int lane = threadIdx.x & 63, h = blockIdx.y;
  const int half = lane >> 5, l31 = lane & 31;
  const float* kbase = qkv + (size_t)tok0 * ld + C + h * 32 + l31;
  const float* vbase = kbase + C;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float ksum = 0.f;
  for (int s = s_begin; s < s_end; s += 16) {
    float kk[8], vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = s + 2 * u + half;
      const bool ok = t < s_end;
      kk[u] = ok ? kbase[(size_t)t * ld] : 0.f;
      vv[u] = ok ? vbase[(size_t)t * ld] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[u], vv[u], acc, 0, 0, 0);
      ksum += kk[u];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
    acc_sh[wave][d * 32 + l31] = acc[r];
  }
  const float tot = ksum + __shfl_xor(ksum, 32, 64);
  if (half == 0) ks_sh[wave][l31] = tot;
  __syncthreads();
  float* kvp = kv_part + (size_t)grp * (C * 32) + h * 1024;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = threadIdx.x + i * 256;
    kvp[o] = (acc_sh[0][o] + acc_sh[1][o]) + (acc_sh[2][o] + acc_sh[3][o]);
  }
  if (threadIdx.x < 32) ks_part[(size_t)grp * C + h * 32 + threadIdx.x] = (ks_sh[0][threadIdx.x] + ks_sh[1][threadIdx.x]) + (ks_sh[2][threadIdx.x] + ks_sh[3][threadIdx.x]);
}

// out_kv [2][8192], out_ks [2][256] ; blockIdx.y = stream.  Block = 64 outputs x 4 chunk groups (group g sums
// chunks g, g+4, ... in order; the four group sums are added in group order): fixed order, 4x shorter chains.
__global__ __launch_bounds__(256) void linattn_reduce_pair_kernel(const float* __restrict__ kv_part,
                                                                 const float* __restrict__ ks_part, int chunks0,
                                                                 int chunks1, float* __restrict__ out_kv,
                                                                 float* __restrict__ out_ks) {
  constexpr int KV = 8192, C = 256;
  __shared__ float red[4][64];
  const int stream = blockIdx.y;
  const int c_begin = stream ? chunks0 : 0;
  const int c_end = stream ? chunks0 + chunks1 : chunks0;
  const int e = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  float s = 0.f;
  if (i < KV) {
    for (int c = c_begin + gq; c < c_end; c += 4) s += kv_part[(size_t)c * KV + i];
  } else if (i < KV + C) {
    for (int c = c_begin + gq; c < c_end; c += 4) s += ks_part[(size_t)c * C + (i - KV)];
  }
  red[gq][e] = s;
  __syncthreads();
  if (gq == 0) {
    const float tot = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    if (i < KV) out_kv[stream * KV + i] = tot;
    else if (i < KV + C) out_ks[stream * C + (i - KV)] = tot;
  }
}

// apply for both streams in one launch (C = 256, D = 32, one segment per stream) on the fp32 MFMA:
//   msg[t][h*32 + v] = (sum_d Q[t][h*32+d] KV_h[d][v]) / (sum_d Q[t][h*32+d] Ksum_h[d] + eps) * S
// Block = 32 tokens of one stream x 8 waves, wave = head.  The Q tile is staged in LDS with coalesced
// 16-byte loads (row stride 258 floats: the 64 lanes of an A-operand read hit 64 different banks);
// num = mfma_32x32x2(A = Q tile, B = KV_h) over 16 steps (the fp32 MFMA accumulates k in ascending order with one
// rounding per product-add, i.e. the chain of a scalar fmaf loop over d); the normaliser is one scalar chain per
// (token, head), shared by the 32 columns of the head through LDS.
constexpr int kApplyChunk = 32;
__global__ __launch_bounds__(512) void linattn_apply_pair_kernel(const float* __restrict__ qkv, int ld,
                                                                const float* __restrict__ kv,
                                                                const float* __restrict__ ks, int cross,
                                                                float* __restrict__ out, int ldo, int len0, int len1,
                                                                int chunks0, float eps) {
  constexpr int C = 256, D = 32, H = 8, CHUNK = kApplyChunk, QS = 258;
  __shared__ __attribute__((aligned(16))) float qsh[CHUNK * QS];
  __shared__ float z_sh[H][CHUNK];
  const int t = threadIdx.x;
  const int stream = blockIdx.x >= chunks0 ? 1 : 0;
  const int cidx = stream ? blockIdx.x - chunks0 : blockIdx.x;
  const int seg_len = stream ? len1 : len0;
  const int base = stream ? len0 : 0;
  const int src = cross ? 1 - stream : stream;          // quirk q6: both streams use pre-update K,V
  const float src_len = (float)(src ? len1 : len0);
  const int s_begin = cidx * CHUNK;
  const float* kvp = kv + (size_t)src * (C * D);
  const float* ksp = ks + (size_t)src * C;
  const int lane = t & 63, h = t >> 6;                  // wave = head
  const int half = lane >> 5, l31 = lane & 31;
  float bk[16];                                          // B operand: KV_h[d = 2 i + half][v = l31], in flight during the staging
#pragma unroll
  for (int i = 0; i < 16; ++i) bk[i] = kvp[(h * D + 2 * i + half) * D + l31];
#pragma unroll
  for (int i = 0; i < CHUNK * (C / 4) / 512; ++i) {
    const int e = t + i * 512;
    const int r = e / (C / 4), c4 = e - r * (C / 4);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s_begin + r < seg_len) q = *reinterpret_cast<const float4*>(qkv + (size_t)(base + s_begin + r) * ld + c4 * 4);
    float2* dst = reinterpret_cast<float2*>(qsh + r * QS + c4 * 4);
    dst[0] = make_float2(q.x, q.y);
    dst[1] = make_float2(q.z, q.w);
  }
  __syncthreads();
  if (t < H * CHUNK) {   // the normaliser of (token, head): one thread each, a sequential fmaf chain over d
    const int tok = t & (CHUNK - 1), hh = t / CHUNK;
    const float* q = qsh + tok * QS + hh * D;
    const float* k = ksp + hh * D;
    float den = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) den = fmaf(q[d], k[d], den);
    z_sh[hh][tok] = 1.0f / (den + eps);
  }
  f32x16_t num;
#pragma unroll
  for (int r = 0; r < 16; ++r) num[r] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) num = __builtin_amdgcn_mfma_f32_32x32x2f32(qsh[l31 * QS + h * D + 2 * i + half], bk[i], num, 0, 0, 0);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    const int tok = s_begin + row;
    if (tok < seg_len) out[(size_t)(base + tok) * ldo + h * D + l31] = (num[r] * z_sh[h][row]) * src_len;
  }
}

}  // namespace

int opp_layernorm(const float* x, int ldx, const float* gamma, const float* beta, const float* res,
                  int ldres, float* out, int ldo, int rows, int C, float eps, hipStream_t stream) {
  if (rows <= 0) return OPP_OK;
  {
    const int vb = (C / 64) * 4;   // bytes per lane access
    auto al = [&](const void* p) { return (reinterpret_cast<uintptr_t>(p) % vb) == 0; };
    OPP_CHECK_ARG(al(x) && al(gamma) && al(beta) && al(out) && (!res || al(res)) && (ldx * 4) % vb == 0 && (ldo * 4) % vb == 0 &&
                      (!res || (ldres * 4) % vb == 0), "layernorm: operands must be %d-byte aligned", vb);
  }
  dim3 grid(opp_cdiv(rows, 16)), block(256);
  if (C == 256)
    hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, stream, x, ldx, gamma, beta, res, ldres, out, ldo, rows, eps);
  else if (C == 128)
    hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, stream, x, ldx, gamma, beta, res, ldres, out, ldo, rows, eps);
  else {
    opp_set_error("layernorm: unsupported width %d", C);
    return OPP_ERR_UNSUPPORTED;
  }
  OPP_CHECK_LAUNCH("layernorm_kernel");
  return OPP_OK;
}

// number of chunk partials the kv reduction uses for a stream of `seg_len` tokens per segment
int opp_linattn_chunks(int seg_len) {
  const int chunk = 64;
  return seg_len <= chunk ? 1 : opp_cdiv(seg_len, chunk);
}

// KV/Ksum of one stream.  k/v point at the K / V columns of the stream's first token
// (row stride ld).  kv_out [n_seg][C*D], ks_out [n_seg][C]; scratch must hold
// n_seg * chunks * (C*D + C) floats when chunks > 1.
int opp_linattn_kv(const float* k, const float* v, int ld, int n_seg, int seg_len, int C, int D,
                   float* kv_out, float* ks_out, float* scratch, hipStream_t stream) {
  if (n_seg <= 0 || seg_len <= 0) return OPP_OK;
  const int chunks = opp_linattn_chunks(seg_len);
  const int chunk_len = chunks == 1 ? seg_len : 64;
  float* kvp = chunks == 1 ? kv_out : scratch;
  float* ksp = chunks == 1 ? ks_out : scratch + (size_t)n_seg * chunks * C * D;
  dim3 grid(chunks, n_seg);
  if (C == 256 && D == 32)
    hipLaunchKernelGGL((linattn_kv_partial_kernel<32, 256>), grid, dim3(256), 0, stream, k, v, ld, seg_len, chunk_len, kvp, ksp);
  else if (C == 128 && D == 16)
    hipLaunchKernelGGL((linattn_kv_partial_kernel<16, 128>), grid, dim3(128), 0, stream, k, v, ld, seg_len, chunk_len, kvp, ksp);
  else {
    opp_set_error("linattn: unsupported (C=%d, D=%d)", C, D);
    return OPP_ERR_UNSUPPORTED;
  }
  OPP_CHECK_LAUNCH("linattn_kv_partial_kernel");
  if (chunks > 1) {
    hipLaunchKernelGGL(linattn_reduce_kernel, dim3(opp_cdiv(C * D, 256), n_seg), dim3(256), 0, stream, kvp, kv_out, chunks, C * D);
    hipLaunchKernelGGL(linattn_reduce_kernel, dim3(opp_cdiv(C, 256), n_seg), dim3(256), 0, stream, ksp, ks_out, chunks, C);
    OPP_CHECK_LAUNCH("linattn_reduce_kernel");
  }
  return OPP_OK;
}

int opp_linattn_apply(const float* q, int ldq, const float* kv, const float* ks, float* out, int ldo,
                      int n_seg, int seg_len, int src_len, int C, int D, float eps, hipStream_t stream) {
  if (n_seg <= 0 || seg_len <= 0) return OPP_OK;
  const int chunk_len = 32;
  dim3 grid(opp_cdiv(seg_len, chunk_len), n_seg);
  if (C == 256 && D == 32)
    hipLaunchKernelGGL((linattn_apply_kernel<32, 256>), grid, dim3(256), 0, stream, q, ldq, kv, ks, out, ldo, seg_len, chunk_len, (float)src_len, eps);
  else if (C == 128 && D == 16)
    hipLaunchKernelGGL((linattn_apply_kernel<16, 128>), grid, dim3(128), 0, stream, q, ldq, kv, ks, out, ldo, seg_len, chunk_len, (float)src_len, eps);
  else {
    opp_set_error("linattn: unsupported (C=%d, D=%d)", C, D);
    return OPP_ERR_UNSUPPORTED;
  }
  OPP_CHECK_LAUNCH("linattn_apply_kernel");
  return OPP_OK;
}


bool opp_linattn_small_ok(int len0, int len1, int C, int D) {
  return C == 128 && D == 16 && len0 > 0 && len1 > 0 && len0 + len1 <= 32;
}

int opp_linattn_small_pair(const float* qkv, int ld, int n_seg, int len0, int len1, int cross, float* out, int ldo, int C, int D,
                           float eps, hipStream_t stream) {
  if (n_seg <= 0) return OPP_OK;
  OPP_CHECK_ARG(opp_linattn_small_ok(len0, len1, C, D) && ld % 4 == 0, "linattn_small_pair: unsupported shape");
  const size_t lds = ((size_t)(len0 + len1) * 2 * C + 2 * C + (size_t)(len0 + len1) * (C / D)) * sizeof(float);
  // algorithmic bytes: Q, K, V of every token read once, the message written once
  OppProfScope prof(OPP_PROF_LINATTN_SMALL, stream, (double)n_seg * (len0 + len1) * C * 4.0 * 4.0);
  hipLaunchKernelGGL((linattn_small_pair_kernel<16, 128>), dim3(n_seg), dim3(128), lds, stream, qkv, ld, n_seg, len0, len1, cross, out,
                     ldo, eps);
  OPP_CHECK_LAUNCH("linattn_small_pair_kernel");
  return OPP_OK;
}

// ---- coarse level: both streams in three launches ------------------------------------------
size_t opp_linattn_pair_scratch_floats(int len0, int len1) {
  const size_t chunks = (size_t)opp_cdiv(len0, kPairChunk) + opp_cdiv(len1, kPairChunk);
  return chunks * (8192 + 256);
}

// qkv [len0+len1][ld] with Q | K | V column blocks of 256.  kv [2][8192], ks [2][256].
int opp_linattn_kv_pair(const float* qkv, int ld, int len0, int len1, float* kv, float* ks, float* scratch,
                        hipStream_t stream) {
  // partials: one per workgroup = per kPairGroup chunks of a stream
  const int c0 = opp_cdiv(opp_cdiv(len0, kPairChunk), kPairGroup), c1 = opp_cdiv(opp_cdiv(len1, kPairChunk), kPairGroup);
  float* kvp = scratch;
  float* ksp = scratch + (size_t)(c0 + c1) * 8192;
  if (c0 + c1 == 0) return OPP_OK;
  {  // algorithmic bytes: K and V of every token read once + the partials written
    OppProfScope prof(OPP_PROF_LINATTN_KV, stream, (double)(len0 + len1) * 512.0 * 4.0 + (double)(c0 + c1) * (8192 + 256) * 4.0);
    hipLaunchKernelGGL(linattn_kv_mfma_kernel, dim3(c0 + c1, 8), dim3(256), 0, stream, qkv, ld, len0, len1, c0, kvp, ksp);
  }
  {  // algorithmic bytes: every chunk partial read once, KV / Ksum of both streams written
    OppProfScope prof(OPP_PROF_LINATTN_REDUCE, stream, (double)(c0 + c1 + 2) * (8192 + 256) * 4.0);
    hipLaunchKernelGGL(linattn_reduce_pair_kernel, dim3(opp_cdiv(8192 + 256, 64), 2), dim3(256), 0, stream, kvp, ksp, c0, c1, kv, ks);
  }
  OPP_CHECK_LAUNCH("linattn_kv_pair");
  return OPP_OK;
}

int opp_linattn_apply_pair(const float* qkv, int ld, const float* kv, const float* ks, int cross, float* out, int ldo,
                           int len0, int len1, float eps, hipStream_t stream) {
  const int c0 = opp_cdiv(len0, kApplyChunk), c1 = opp_cdiv(len1, kApplyChunk);
  OppProfScope prof(OPP_PROF_LINATTN_APPLY, stream, (double)(len0 + len1) * 256.0 * 4.0 * 2.0);   // Q read + message written
  hipLaunchKernelGGL(linattn_apply_pair_kernel, dim3(c0 + c1), dim3(512), 0, stream, qkv, ld, kv, ks, cross, out, ldo, len0, len1,
                     c0, eps);
  OPP_CHECK_LAUNCH("linattn_apply_pair_kernel");
  return OPP_OK;
}
