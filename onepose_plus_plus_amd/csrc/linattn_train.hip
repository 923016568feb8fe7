// LinearAttention.forward (loftr_module/linear_attention.py:29-61) with its BACKWARD, for the training step
// (PL_OnePosePlus.training_step, src/lightning_model/OnePosePlus_lightning_model.py:54-81): the node the
// re-evaluated training graph uses instead of three torch.einsum calls and their autograd (bmm / einsum: 17 ms of a 120 ms
// step at B = 4, 7000 points).
//
// Per sample b and head h (D = C / nhead = 32 coarse, 16 fine), raw projections q [B][L][H][D], k, v [B][S][H][D]:
//   Qp = (elu(q) + 1) mq,  Kp = (elu(k) + 1) mk,  Vs = v mk / S
//   KV = sum_s Kp[s]^T Vs[s]  [D][D],  ks = sum_s Kp[s]  [D]
//   den[l] = Qp[l] . ks + eps,  num[l] = Qp[l] KV,  out[l] = num[l] / den[l] * S
// Backward, g = d loss / d out:
//   gnum = g S / den,  gden = -(g . num) S / den^2
//   gQp[l] = gnum[l] KV^T + gden[l] ks          gKV = sum_l Qp[l]^T gnum[l]       gks = sum_l gden[l] Qp[l]
//   gKp[s] = Vs[s] gKV^T + gks                  gVs[s] = Kp[s] gKV
//   gq = gQp phi'(q) mq,  gk = gKp phi'(k) mk,  gv = gVs mk / S            (phi'(x) = 1 for x > 0, exp(x) otherwise)
// All reductions over tokens are chunk partials summed in chunk order (deterministic).  fp32 fmaf arithmetic.
// Kernels are bandwidth-bound (every activation read once or twice per pass); a block = 64 tokens of one (sample, head).
#include "opp_internal.h"

namespace {

constexpr int kTok = 64;        // tokens per block
constexpr int kThreads = 256;

__device__ __forceinline__ float phi(float x) { return x > 0.f ? x + 1.f : __expf(x); }
__device__ __forceinline__ float dphi(float x) { return x > 0.f ? 1.f : __expf(x); }

// sum over the D (= 32 or 16) consecutive lanes that share one token: fixed butterfly order, every lane of the group gets the sum
template <int D>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = D / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// r03 ran the per-token matrix-vector products (num = Qp KV, g . num, the two D x D back-projections) on kTok = 64 of the 256
// threads, 1000+ dependent FMAs each: the kernels were latency-bound at 0.9 TB/s.  Since r04 a token is a GROUP OF D LANES
// (lane = output channel): every thread works in every phase, the D x D operand is read conflict-free from LDS in the layout each
// product needs (both KV and KV^T are staged), the per-token scalars are D-lane butterflies -- no per-token LDS round trips.

// partial KV / ks of one chunk of source tokens: a[t][d] (x) b[t][v] summed over t.  MODE 0: a = Kp, b = Vs (forward);
// MODE 1: a = Qp, b = gnum, and the vector partial is sum_t gden[t] Qp[t] (backward; needs KV, ks of the forward)
template <int D, int MODE>
__global__ __launch_bounds__(kThreads) void la_outer_kernel(const float* __restrict__ x,      // MODE 0: k ; MODE 1: q     [B][T][H][D]
                                                            const float* __restrict__ y,      // MODE 0: v ; MODE 1: g
                                                            const float* __restrict__ mask,   // [B][T] or null
                                                            const float* __restrict__ kv, const float* __restrict__ ks,   // MODE 1 only
                                                            int T, int H, float inv_s, float s_len, float eps,
                                                            float* __restrict__ part_m, float* __restrict__ part_v) {
  __shared__ float a_sh[kTok][D + 1], b_sh[kTok][D + 1], w_sh[kTok];
  __shared__ float kv_sh[MODE == 1 ? D * D : 1], ks_sh[MODE == 1 ? D : 1];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int nchunks = gridDim.x;
  const int t0 = chunk * kTok, nt = min(kTok, T - t0);
  const int tid = threadIdx.x;
  const size_t row = (size_t)H * D;
  if (MODE == 1) {
    for (int i = tid; i < D * D; i += kThreads) kv_sh[i] = kv[((size_t)b * H + h) * D * D + i];
    if (tid < D) ks_sh[tid] = ks[((size_t)b * H + h) * D + tid];
  }
  for (int i = tid; i < kTok * D; i += kThreads) {
    const int t = i / D, d = i - t * D;
    float av = 0.f, bv = 0.f;
    if (t < nt) {
      const size_t o = ((size_t)b * T + t0 + t) * row + (size_t)h * D + d;
      const float m = mask ? mask[(size_t)b * T + t0 + t] : 1.f;
      av = phi(x[o]) * m;
      bv = MODE == 0 ? y[o] * m * inv_s : y[o];
    }
    a_sh[t][d] = av;
    b_sh[t][d] = bv;
  }
  __syncthreads();
  if (MODE == 1) {
    // per token (group of D lanes, lane = v): num_v = Qp . KV[:, v], den = Qp . ks + eps, dot = g . num
    //   -> gnum = g S / den (overwrites b_sh), gden = -(dot S) / den^2
    for (int i = tid; i < kTok * D; i += kThreads) {
      const int t = i / D, v = i - t * D;
      float num = 0.f;
#pragma unroll 8
      for (int d = 0; d < D; ++d) num = fmaf(a_sh[t][d], kv_sh[d * D + v], num);
      const float gv = b_sh[t][v];
      const float dot = group_sum<D>(gv * num);
      const float den = group_sum<D>(a_sh[t][v] * ks_sh[v]) + eps;
      const float z = 1.0f / den;
      b_sh[t][v] = gv * (s_len * z);
      if (v == 0) w_sh[t] = -(dot * s_len) * z * z;
    }
    __syncthreads();
  }
  // outer-product sums: thread owns entries (d, v) with d * D + v = tid, tid + 256, ...
  float* pm = part_m + (((size_t)b * H + h) * nchunks + chunk) * (D * D);
  for (int e = tid; e < D * D; e += kThreads) {
    const int d = e / D, v = e - d * D;
    float acc = 0.f;
#pragma unroll 8
    for (int t = 0; t < kTok; ++t) acc = fmaf(a_sh[t][d], b_sh[t][v], acc);
    pm[e] = acc;
  }
  if (tid < D) {
    float acc = 0.f;
    for (int t = 0; t < kTok; ++t) acc = MODE == 0 ? acc + a_sh[t][tid] : fmaf(w_sh[t], a_sh[t][tid], acc);
    part_v[(((size_t)b * H + h) * nchunks + chunk) * D + tid] = acc;
  }
}

// out[bh][e] = sum over the chunks: four interleaved slices per output (slice s adds chunks s, s + 4, ... in ascending order), then
// (s0 + s1) + (s2 + s3) -- fixed order, deterministic; four lanes per output instead of one thread walking ~110 partials
__global__ __launch_bounds__(256) void la_chunk_reduce_kernel(const float* __restrict__ part, int nchunks, int per, size_t n_bh,
                                                              float* __restrict__ out) {
  const size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = gi >> 2;
  const int sl = (int)(gi & 3);
  float s = 0.f;
  const bool live = i < n_bh * per;
  if (live) {
    const size_t bh = i / per;
    const int e = (int)(i - bh * per);
    for (int c = sl; c < nchunks; c += 4) s += part[(bh * nchunks + c) * per + e];
  }
  const float s1 = __shfl_xor(s, 1, 64);
  const float pair = (sl & 1) ? s1 + s : s + s1;          // both lanes of a pair: (even slice) + (odd slice)
  const float other = __shfl_xor(pair, 2, 64);
  if (live && sl == 0) out[i] = pair + other;              // (s0 + s1) + (s2 + s3)
}

// per query token: MODE 0 forward out = num / den * S ; MODE 1 backward gq = (gnum KV^T + gden ks) phi'(q) mq
template <int D, int MODE>
__global__ __launch_bounds__(kThreads) void la_query_kernel(const float* __restrict__ q, const float* __restrict__ g, const float* __restrict__ mask,
                                                            const float* __restrict__ kv, const float* __restrict__ ks, int T, int H,
                                                            float s_len, float eps, float* __restrict__ out) {
  __shared__ float q_sh[kTok][D + 1], g_sh[MODE == 1 ? kTok : 1][D + 1], kv_sh[D * D], kvt_sh[MODE == 1 ? D * D : 1], ks_sh[D];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int t0 = chunk * kTok, nt = min(kTok, T - t0);
  const int tid = threadIdx.x;
  const size_t row = (size_t)H * D;
  for (int i = tid; i < D * D; i += kThreads) {
    const float val = kv[((size_t)b * H + h) * D * D + i];
    kv_sh[i] = val;
    if (MODE == 1) kvt_sh[(i % D) * D + i / D] = val;      // KV^T: the back-projection reads it with lane = row of KV
  }
  if (tid < D) ks_sh[tid] = ks[((size_t)b * H + h) * D + tid];
  for (int i = tid; i < kTok * D; i += kThreads) {
    const int t = i / D, d = i - t * D;
    float qv = 0.f, gv = 0.f;
    if (t < nt) {
      const size_t o = ((size_t)b * T + t0 + t) * row + (size_t)h * D + d;
      const float m = mask ? mask[(size_t)b * T + t0 + t] : 1.f;
      qv = phi(q[o]) * m;
      if (MODE == 1) gv = g[o];
    }
    q_sh[t][d] = qv;
    if (MODE == 1) g_sh[t][d] = gv;
  }
  __syncthreads();
  // a token = D consecutive lanes, lane = output channel e
  for (int i = tid; i < kTok * D; i += kThreads) {
    const int t = i / D, e = i - t * D;
    float num = 0.f;
#pragma unroll 8
    for (int d = 0; d < D; ++d) num = fmaf(q_sh[t][d], kv_sh[d * D + e], num);
    const float den = group_sum<D>(q_sh[t][e] * ks_sh[e]) + eps;
    const float z = 1.0f / den;
    const size_t o = ((size_t)b * T + t0 + min(t, nt - 1)) * row + (size_t)h * D + e;
    if (MODE == 0) {
      if (t < nt) out[o] = (num * z) * s_len;
    } else {
      const float dot = group_sum<D>(g_sh[t][e] * num);
      const float gd = -(dot * s_len) * z * z;
      const float sc = s_len * z;
      // gQp[e] = sum_v gnum[v] KV[e][v] + gden ks[e]
      float acc = gd * ks_sh[e];
#pragma unroll 8
      for (int v = 0; v < D; ++v) acc = fmaf(g_sh[t][v] * sc, kvt_sh[v * D + e], acc);
      if (t < nt) {
        const float m = mask ? mask[(size_t)b * T + t0 + t] : 1.f;
        out[o] = acc * dphi(q[o]) * m;
      }
    }
  }
}

// per source token (backward): gk = (Vs gKV^T + gks) phi'(k) mk ; gv = (Kp gKV) mk / S
template <int D>
__global__ __launch_bounds__(kThreads) void la_source_bwd_kernel(const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ mask,
                                                                 const float* __restrict__ gkv, const float* __restrict__ gks, int T, int H, float inv_s,
                                                                 float* __restrict__ gk, float* __restrict__ gv) {
  __shared__ float k_sh[kTok][D + 1], v_sh[kTok][D + 1], kv_sh[D * D], kvt_sh[D * D], ks_sh[D];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int t0 = chunk * kTok, nt = min(kTok, T - t0);
  const int tid = threadIdx.x;
  const size_t row = (size_t)H * D;
  for (int i = tid; i < D * D; i += kThreads) {
    const float val = gkv[((size_t)b * H + h) * D * D + i];
    kv_sh[i] = val;
    kvt_sh[(i % D) * D + i / D] = val;
  }
  if (tid < D) ks_sh[tid] = gks[((size_t)b * H + h) * D + tid];
  for (int i = tid; i < kTok * D; i += kThreads) {
    const int t = i / D, d = i - t * D;
    float kv_ = 0.f, vv = 0.f;
    if (t < nt) {
      const size_t o = ((size_t)b * T + t0 + t) * row + (size_t)h * D + d;
      const float m = mask ? mask[(size_t)b * T + t0 + t] : 1.f;
      kv_ = phi(k[o]) * m;
      vv = v[o] * m * inv_s;
    }
    k_sh[t][d] = kv_;
    v_sh[t][d] = vv;
  }
  __syncthreads();
  for (int i = tid; i < kTok * D; i += kThreads) {
    const int t = i / D, e = i - t * D;
    if (t >= nt) continue;
    const size_t o = ((size_t)b * T + t0 + t) * row + (size_t)h * D + e;
    const float m = mask ? mask[(size_t)b * T + t0 + t] : 1.f;
    float a = ks_sh[e], c = 0.f;
#pragma unroll 8
    for (int j = 0; j < D; ++j) {
      a = fmaf(v_sh[t][j], kvt_sh[j * D + e], a);    // gKp[e] = sum_v Vs[v] gKV[e][v] + gks[e]   (gKV^T read with lane = e)
      c = fmaf(k_sh[t][j], kv_sh[j * D + e], c);     // gVs[e] = sum_d Kp[d] gKV[d][e]
    }
    gk[o] = a * dphi(k[o]) * m;
    gv[o] = c * m * inv_s;
  }
}

// ---- short segments (the fine level: 25 window tokens against 1 point token per match, thousands of matches) ------------------
// The chunked kernels above spend a workgroup on a (sample, head) pair and three extra launches on partial sums that a short
// segment does not need.  Here ONE workgroup owns a sample with all its heads (H x 32 threads; a head = 32 lanes = 32 / D tokens
// in flight): K, V, Q (and the gradient) of the sample are staged once, KV / ks (forward) and gKV / gks (backward) live in LDS,
// every token sum runs in token order (deterministic), and a pass is one launch: forward 4 -> 1, backward 5 -> 1.
constexpr int kSmallTok = 32;      // L, S <= 32

struct SmallArgs {
  const float *q, *k, *v, *qm, *km, *kv, *ks, *g;
  float *out, *kv_out, *ks_out, *gq, *gk, *gv;
  int L, S, H;
  float inv_s, s_len, eps;
};

template <int D, bool BWD>
__global__ __launch_bounds__(256) void la_small_kernel(const SmallArgs a) {
  extern __shared__ float sm[];
  const int L = a.L, S = a.S, H = a.H, C = H * D, DD = D * D;
  float* q_sh = sm;                       // [L][C]   phi(q) mq
  float* k_sh = q_sh + L * C;             // [S][C]   phi(k) mk
  float* v_sh = k_sh + S * C;             // [S][C]   v mk / S
  float* kv_sh = v_sh + S * C;            // [H][D][D]
  float* ks_sh = kv_sh + H * DD;          // [H][D]
  float* g_sh = ks_sh + C;                // backward: [L][C] g, then gnum in place
  float* kvt_sh = g_sh + (BWD ? L * C : 0);      // backward: KV^T, later gKV^T
  float* gd_sh = kvt_sh + (BWD ? H * DD : 0);    // backward: [L][H] gden
  float* gks_sh = gd_sh + (BWD ? L * H : 0);     // backward: [H][D]
  const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const size_t qb = (size_t)b * L * C, sb = (size_t)b * S * C;
  for (int i = tid; i < L * C; i += nthr) {
    const float m = a.qm ? a.qm[(size_t)b * L + i / C] : 1.f;
    q_sh[i] = phi(a.q[qb + i]) * m;
    if (BWD) g_sh[i] = a.g[qb + i];
  }
  for (int i = tid; i < S * C; i += nthr) {
    const float m = a.km ? a.km[(size_t)b * S + i / C] : 1.f;
    k_sh[i] = phi(a.k[sb + i]) * m;
    v_sh[i] = a.v[sb + i] * m * a.inv_s;
  }
  if (BWD) {     // KV, ks of the forward come back from the caller
    for (int i = tid; i < H * DD; i += nthr) {
      const float val = a.kv[(size_t)b * H * DD + i];
      kv_sh[i] = val;
      const int h = i / DD, r = i - h * DD;
      kvt_sh[h * DD + (r % D) * D + r / D] = val;
    }
    for (int i = tid; i < C; i += nthr) ks_sh[i] = a.ks[(size_t)b * C + i];
  }
  __syncthreads();
  if (!BWD) {
    for (int i = tid; i < H * DD; i += nthr) {
      const int h = i / DD, r = i - h * DD, d = r / D, vv = r - d * D;
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc = fmaf(k_sh[s * C + h * D + d], v_sh[s * C + h * D + vv], acc);
      kv_sh[i] = acc;
      a.kv_out[(size_t)b * H * DD + i] = acc;
    }
    for (int i = tid; i < C; i += nthr) {
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc += k_sh[s * C + i];
      ks_sh[i] = acc;
      a.ks_out[(size_t)b * C + i] = acc;
    }
    __syncthreads();
    for (int i = tid; i < L * C; i += nthr) {
      const int l = i / C, c = i - l * C, h = c / D, e = c - h * D;
      const float* qr = q_sh + l * C + h * D;
      float num = 0.f, den = a.eps;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        num = fmaf(qr[d], kv_sh[h * DD + d * D + e], num);
        den = fmaf(qr[d], ks_sh[h * D + d], den);
      }
      a.out[qb + i] = (num * (1.0f / den)) * a.s_len;
    }
    return;
  }
  // ---- backward -------------------------------------------------------------------------------------------------------------
  // query tokens: item = (l, h, e) with the D channels of a (token, head) on D consecutive lanes
  const int items_q = ((L * C + 63) / 64) * 64;          // whole waves take part in the butterflies
  for (int i = tid; i < items_q; i += nthr) {
    const bool live = i < L * C;
    const int ii = live ? i : 0;
    const int l = ii / C, c = ii - l * C, h = c / D, e = c - h * D;
    const float* qr = q_sh + l * C + h * D;
    const float* gr = g_sh + l * C + h * D;
    float num = 0.f, den = a.eps;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      num = fmaf(qr[d], kv_sh[h * DD + d * D + e], num);
      den = fmaf(qr[d], ks_sh[h * D + d], den);
    }
    const float ge = gr[e];
    const float dot = group_sum<D>(live ? ge * num : 0.f);
    const float z = 1.0f / den;
    const float gd = -(dot * a.s_len) * z * z;
    const float sc = a.s_len * z;
    float acc = gd * ks_sh[h * D + e];
#pragma unroll
    for (int v = 0; v < D; ++v) acc = fmaf(gr[v] * sc, kvt_sh[h * DD + v * D + e], acc);
    if (live) {
      const float m = a.qm ? a.qm[(size_t)b * L + l] : 1.f;
      a.gq[qb + i] = acc * dphi(a.q[qb + i]) * m;
      if (e == 0) gd_sh[l * H + h] = gd;
    }
    // gnum overwrites g in place only after every lane of the group has read its row: the butterflies above are that point for
    // the lanes of this group, and no other group touches this (token, head) row
    __builtin_amdgcn_wave_barrier();
    if (live) g_sh[i] = ge * sc;
  }
  __syncthreads();
  // gKV = sum_l Qp[l]^T gnum[l], gks = sum_l gden[l] Qp[l]   (token order)
  for (int i = tid; i < H * DD; i += nthr) {
    const int h = i / DD, r = i - h * DD, d = r / D, vv = r - d * D;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc = fmaf(q_sh[l * C + h * D + d], g_sh[l * C + h * D + vv], acc);
    kv_sh[i] = acc;
    kvt_sh[h * DD + vv * D + d] = acc;
  }
  for (int i = tid; i < C; i += nthr) {
    const int h = i / D;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc = fmaf(gd_sh[l * H + h], q_sh[l * C + i], acc);
    gks_sh[i] = acc;
  }
  __syncthreads();
  for (int i = tid; i < S * C; i += nthr) {
    const int s = i / C, c = i - s * C, h = c / D, e = c - h * D;
    const float m = a.km ? a.km[(size_t)b * S + s] : 1.f;
    const float* kr = k_sh + s * C + h * D;
    const float* vr = v_sh + s * C + h * D;
    float ga = gks_sh[h * D + e], gc = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      ga = fmaf(vr[j], kvt_sh[h * DD + j * D + e], ga);      // gKp[e] = sum_v Vs[v] gKV[e][v] + gks[e]
      gc = fmaf(kr[j], kv_sh[h * DD + j * D + e], gc);       // gVs[e] = sum_d Kp[d] gKV[d][e]
    }
    a.gk[sb + i] = ga * dphi(a.k[sb + i]) * m;
    a.gv[sb + i] = gc * m * a.inv_s;
  }
}

bool small_ok(int L, int S, int H, int D) { return L <= kSmallTok && S <= kSmallTok && H * 32 <= 256 && H >= 1 && (D == 16 || D == 32); }

size_t small_lds(int L, int S, int H, int D, bool bwd) {
  const size_t C = (size_t)H * D, DD = (size_t)D * D;
  size_t f = L * C + 2 * S * C + H * DD + C;
  if (bwd) f += L * C + H * DD + (size_t)L * H + C;
  return f * sizeof(float);
}

template <int D, bool BWD>
int small_launch(const SmallArgs& a, int B, hipStream_t st) {
  const size_t lds = small_lds(a.L, a.S, a.H, D, BWD);
  static OppLdsOnce once;
  opp_lds_opt_in(reinterpret_cast<const void*>(la_small_kernel<D, BWD>), 160 * 1024, once);
  hipLaunchKernelGGL((la_small_kernel<D, BWD>), dim3(B), dim3(a.H * 32), lds, st, a);
  OPP_CHECK_LAUNCH("la_small_kernel");
  return OPP_OK;
}

struct Plan {
  int cq, cs;
  size_t off_pm, off_pv, total;    // floats
};
Plan make_plan(int B, int L, int S, int H, int D) {
  Plan p;
  p.cq = opp_cdiv(L, kTok);
  p.cs = opp_cdiv(S, kTok);
  const int cm = p.cq > p.cs ? p.cq : p.cs;
  p.off_pm = 0;
  p.off_pv = (size_t)B * H * cm * D * D;
  p.total = p.off_pv + (size_t)B * H * cm * D + 64;
  return p;
}

template <int D>
int fwd_impl(const float* q, const float* k, const float* v, const float* qm, const float* km, int B, int L, int S, int H, float eps, float* out,
             float* kv, float* ks, float* ws, hipStream_t st) {
  const Plan p = make_plan(B, L, S, H, D);
  const float inv_s = 1.0f / (float)S, s_len = (float)S;
  if (small_ok(L, S, H, D) && small_lds(L, S, H, D, false) <= 150 * 1024) {      // short segments: one workgroup per sample, one launch
    SmallArgs a{q, k, v, qm, km, nullptr, nullptr, nullptr, out, kv, ks, nullptr, nullptr, nullptr, L, S, H, inv_s, s_len, eps};
    return small_launch<D, false>(a, B, st);
  }
  hipLaunchKernelGGL((la_outer_kernel<D, 0>), dim3(p.cs, H, B), dim3(kThreads), 0, st, k, v, km, (const float*)nullptr, (const float*)nullptr, S, H,
                     inv_s, s_len, eps, ws + p.off_pm, ws + p.off_pv);
  const size_t nbh = (size_t)B * H;
  hipLaunchKernelGGL(la_chunk_reduce_kernel, dim3((unsigned)((nbh * D * D * 4 + 255) / 256)), dim3(256), 0, st, ws + p.off_pm, p.cs, D * D, nbh, kv);
  hipLaunchKernelGGL(la_chunk_reduce_kernel, dim3((unsigned)((nbh * D * 4 + 255) / 256)), dim3(256), 0, st, ws + p.off_pv, p.cs, D, nbh, ks);
  hipLaunchKernelGGL((la_query_kernel<D, 0>), dim3(p.cq, H, B), dim3(kThreads), 0, st, q, (const float*)nullptr, qm, kv, ks, L, H, s_len, eps, out);
  OPP_CHECK_LAUNCH("linattn_train forward");
  return OPP_OK;
}

template <int D>
int bwd_impl(const float* q, const float* k, const float* v, const float* qm, const float* km, const float* kv, const float* ks, const float* g, int B,
             int L, int S, int H, float eps, float* gq, float* gk, float* gv, float* ws, hipStream_t st) {
  const Plan p = make_plan(B, L, S, H, D);
  const float inv_s = 1.0f / (float)S, s_len = (float)S;
  const size_t nbh = (size_t)B * H;
  if (small_ok(L, S, H, D) && small_lds(L, S, H, D, true) <= 150 * 1024) {
    SmallArgs a{q, k, v, qm, km, kv, ks, g, nullptr, nullptr, nullptr, gq, gk, gv, L, S, H, inv_s, s_len, eps};
    return small_launch<D, true>(a, B, st);
  }
  float* gkv = ws + p.total;                    // [B][H][D][D] + [B][H][D] behind the partials
  float* gks = gkv + nbh * D * D;
  hipLaunchKernelGGL((la_outer_kernel<D, 1>), dim3(p.cq, H, B), dim3(kThreads), 0, st, q, g, qm, kv, ks, L, H, inv_s, s_len, eps, ws + p.off_pm,
                     ws + p.off_pv);
  hipLaunchKernelGGL(la_chunk_reduce_kernel, dim3((unsigned)((nbh * D * D * 4 + 255) / 256)), dim3(256), 0, st, ws + p.off_pm, p.cq, D * D, nbh, gkv);
  hipLaunchKernelGGL(la_chunk_reduce_kernel, dim3((unsigned)((nbh * D * 4 + 255) / 256)), dim3(256), 0, st, ws + p.off_pv, p.cq, D, nbh, gks);
  hipLaunchKernelGGL((la_query_kernel<D, 1>), dim3(p.cq, H, B), dim3(kThreads), 0, st, q, g, qm, kv, ks, L, H, s_len, eps, gq);
  hipLaunchKernelGGL((la_source_bwd_kernel<D>), dim3(p.cs, H, B), dim3(kThreads), 0, st, k, v, km, gkv, gks, S, H, inv_s, gk, gv);
  OPP_CHECK_LAUNCH("linattn_train backward");
  return OPP_OK;
}

}  // namespace

size_t opp_linattn_train_ws_bytes(int B, int L, int S, int H, int D) {
  if (B <= 0 || L <= 0 || S <= 0 || H <= 0 || D <= 0) return 256;
  return (make_plan(B, L, S, H, D).total + (size_t)B * H * (D * D + D) + 64) * sizeof(float);
}

int opp_linattn_train_fwd(const float* q, const float* k, const float* v, const float* q_mask, const float* kv_mask, int B, int L, int S, int H, int D,
                          float eps, float* out, float* kv, float* ks, void* ws, size_t ws_bytes, hipStream_t stream) {
  OPP_CHECK_ARG(q && k && v && out && kv && ks && ws && B > 0 && L > 0 && S > 0 && H > 0, "linattn_train: null / empty argument");
  OPP_CHECK_ARG(D == 32 || D == 16, "linattn_train: head width must be 32 or 16 (got %d)", D);
  OPP_CHECK_ARG(B <= 65535 && H <= 65535, "linattn_train: batch %d / heads %d exceed the grid limit of 65535", B, H);
  OPP_CHECK_ARG(ws_bytes >= opp_linattn_train_ws_bytes(B, L, S, H, D), "linattn_train: workspace too small");
  return D == 32 ? fwd_impl<32>(q, k, v, q_mask, kv_mask, B, L, S, H, eps, out, kv, ks, static_cast<float*>(ws), stream)
                 : fwd_impl<16>(q, k, v, q_mask, kv_mask, B, L, S, H, eps, out, kv, ks, static_cast<float*>(ws), stream);
}

int opp_linattn_train_bwd(const float* q, const float* k, const float* v, const float* q_mask, const float* kv_mask, const float* kv, const float* ks,
                          const float* grad_out, int B, int L, int S, int H, int D, float eps, float* gq, float* gk, float* gv, void* ws,
                          size_t ws_bytes, hipStream_t stream) {
  OPP_CHECK_ARG(q && k && v && kv && ks && grad_out && gq && gk && gv && ws && B > 0 && L > 0 && S > 0 && H > 0, "linattn_train backward: null / empty argument");
  OPP_CHECK_ARG(D == 32 || D == 16, "linattn_train backward: head width must be 32 or 16 (got %d)", D);
  OPP_CHECK_ARG(B <= 65535 && H <= 65535, "linattn_train backward: batch %d / heads %d exceed the grid limit of 65535", B, H);
  OPP_CHECK_ARG(ws_bytes >= opp_linattn_train_ws_bytes(B, L, S, H, D), "linattn_train backward: workspace too small");
  return D == 32 ? bwd_impl<32>(q, k, v, q_mask, kv_mask, kv, ks, grad_out, B, L, S, H, eps, gq, gk, gv, static_cast<float*>(ws), stream)
                 : bwd_impl<16>(q, k, v, q_mask, kv_mask, kv, ks, grad_out, B, L, S, H, eps, gq, gk, gv, static_cast<float*>(ws), stream);
}
