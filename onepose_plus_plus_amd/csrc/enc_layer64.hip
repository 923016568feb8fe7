// One coarse LoFTREncoderLayer behind its Q/K/V projection on 64-TOKEN tiles (C = 256, 8 heads, 8 waves):
//   attention apply -> merge -> norm1 -> mlp.0(cat[x, message]) -> ReLU -> mlp.2 -> norm2 -> x + .
// (/root/reference/src/models/OnePosePlus/loftr_module/transformer.py:65-94, linear_attention.py:55-61)
//
// Same arithmetic and accumulation order as enc_chain.hip (32-token tiles, bit-identical results), restructured around the
// one resource that decides its speed at 9096 tokens on 256 CUs: the tile count.  285 tiles of 32 rows run in TWO rounds with
// the second 11 % full; 143 tiles of 64 rows run in one, read every weight fragment once per TWO row blocks (half the L2 ->
// register weight stream per token) and amortise every phase transition over twice the MFMA work.
// What makes 64 rows fit 160 KB of LDS is the order of the chain -- only ONE bf16x3 operand tile [64][K = 256] (97 KB) exists:
//   1. x tile           -> operand tile -> mlp.0 accumulators += x . W1[:, 0:256]^T        (K order of mlp.0: x first)
//   2. attention apply  -> message      -> operand tile (16-bit scattered stores straight from the fp32-MFMA accumulators)
//   3. merge            -> norm1        -> operand tile (two 32-row halves through the fp32 staging tile)
//   4. mlp.0 accumulators += merged . W1[:, 256:512]^T, ReLU
//   5. hidden activation, 256 columns at a time -> operand tile -> mlp.2 accumulators (two K halves)
//   6. norm2, + x (re-read in fp32), rows written.
// The 64 x 512 hidden activation never exists outside the accumulators of the waves that own its columns.
// LDS: operand tile 97 KB + fp32 staging [32][260] 32.5 KB + 1 KB = 130.5 KB, one 8-wave workgroup per CU.
#include <stdlib.h>

#include <type_traits>

#include "enc_frag.h"

namespace {

constexpr int R64 = 64, C = 256, NW = 8, NT = 512, D = 32;
constexpr int SA = a_stride_bytes(C);          // 1552
constexpr int SS = C + 4;                      // fp32 staging row stride (floats)
constexpr int OFF_S = R64 * SA, OFF_Z = OFF_S + 32 * SS * 4;
constexpr size_t kLds64 = OFF_Z + 1024;

__global__ __launch_bounds__(NT) void enc_layer64_kernel(const OppEncChain a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* A = smem;
  float* S = reinterpret_cast<float*>(smem + OFF_S);
  float* z_sh = reinterpret_cast<float*>(smem + OFF_Z);   // [8][32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  // tile -> rows [row0, row0 + nrows) of ONE stream (the attention of its rows uses one KV)
  const int chunks0 = (a.len0 + R64 - 1) / R64;
  const int stream = (int)blockIdx.x >= chunks0 ? 1 : 0;
  const int cidx = stream ? blockIdx.x - chunks0 : blockIdx.x;
  const int seg_len = stream ? a.len1 : a.len0;
  const int row0 = (stream ? a.len0 : 0) + cidx * R64;
  const int nrows = min(R64, seg_len - cidx * R64);
  // phi(Q) rows of this tile: stream 1 may come from its own buffer (the per-object prefix keeps the image-independent projection)
  const float* qrows = (stream && a.q1 != nullptr) ? a.q1 + (size_t)(cidx * R64) * a.ldq : a.q + (size_t)row0 * a.ldq;

  // ---- GEMM over the operand tile: acc[i][j] += A[rows 32 i ..][k16-steps of the tile] * W[tile t0 + j][steps S0 .. S1)^T ----
  // weights fragment-major (opp_pack_frag_b3): ((t * KS + s) * 3 + part) * 1024 + lane * 16 bytes, KS = steps of the matrix;
  // operand-tile step = s - S0.  Six bf16 products per block, K ascending: the sequence of opp_gemm_kernel<bf16x3>.
  const char* a_lane = A + l31 * SA + half * 48;
  auto gemm = [&](auto ntile_c, auto depth_c, auto s0_c, auto s1_c, auto ks_c, const void* wf, int t0, f32x16 (&acc)[2][decltype(ntile_c)::value]) {
    constexpr int NTILE = decltype(ntile_c)::value, DEPTH = decltype(depth_c)::value;
    constexpr int S0 = decltype(s0_c)::value, S1 = decltype(s1_c)::value, KS = decltype(ks_c)::value;
    constexpr int NSTEP = S1 - S0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wf), 0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    u32x4 bq[DEPTH][NTILE][3];
    u32x4 af[2][2][3];
    auto load_b = [&](int s, int slot) {
#pragma unroll
      for (int j = 0; j < NTILE; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bq[slot][j][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (((t0 + j) * KS + S0 + s) * 3 + p) * 1024, 0);
    };
    auto load_a = [&](int s, int slot) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) af[slot][i][q] = *reinterpret_cast<const u32x4*>(a_lane + i * 32 * SA + s * 96 + q * 16);
    };
#pragma unroll
    for (int s = 0; s < DEPTH && s < NSTEP; ++s) load_b(s, s);
    load_a(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};   // A part of product pr (0 hi, 1 mid, 2 lo), smallest terms first
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + 1 < NSTEP) load_a(s + 1, (s + 1) & 1);
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NTILE; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s & 1][i][PA[pr]]),
                                                                 __builtin_bit_cast(bf16x8, bq[s % DEPTH][j][PB[pr]]), acc[i][j], 0, 0, 0);
      if (s + DEPTH < NSTEP) load_b(s + DEPTH, s % DEPTH);
      __builtin_amdgcn_sched_barrier(0);   // the ring refill must not sink below the next step
    }
  };
  auto ic = [](auto v) { return v; };
  (void)ic;
  // accumulator block (32 rows, columns col0 + l31) -> fp32 staging tile
  auto stage = [&](const f32x16& t, int col0, bool relu) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = t[r];
      if (relu) v = v < 0.f ? 0.f : v;   // NaN-propagating like torch.relu
      S[row * SS + col0 + l31] = v;
    }
  };
  // staged 32 x 256 fp32 rows -> operand-tile rows rb .. rb + 31, k-groups [g0, g0 + 32)
  auto split_staged = [&](int rb, int g0) {
#pragma unroll
    for (int i = 0; i < 32 * (C / 8) / NT; ++i) {
      const int p = tid + i * NT;
      const int r = p / (C / 8), g = p - r * (C / 8);
      const float4* s4 = reinterpret_cast<const float4*>(S + r * SS + g * 8);
      split8_store(s4[0], s4[1], A + (rb + r) * SA + (g0 + g) * 48);
    }
  };
  // LayerNorm of the 32 staged rows (tile rows rb ..): one wave per row, four rows per wave; the arithmetic of
  // layernorm_kernel / the GEMM epilogue.  MODE 0: -> operand tile; MODE 1: out[row] = x[row] + y (global)
  typedef float vec_t __attribute__((ext_vector_type(4)));
  auto layernorm_rows = [&](const float* gamma, const float* beta, float eps, auto mode_c, int rb) {
    constexpr int MODE = decltype(mode_c)::value;
    const vec_t gmv = *reinterpret_cast<const vec_t*>(gamma + lane * 4);
    const vec_t btv = *reinterpret_cast<const vec_t*>(beta + lane * 4);
    constexpr int RPW = 32 / NW;
    float v[RPW][4], sm[RPW], mean[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const vec_t t = *reinterpret_cast<const vec_t*>(S + (wave * RPW + r) * SS + lane * 4);
      sm[r] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[r][i] = t[i];
        sm[r] += t[i];
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) sm[r] = opp_wave_sum_dpp(sm[r]);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      mean[r] = sm[r] / (float)C;
      sm[r] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
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
      const int lr = rb + wave * RPW + r;
      float y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = opp_ln_affine(v[r][i], mean[r], rstd[r], gmv[i], btv[i]);
      if constexpr (MODE != 0) {                 // MODE 1 / 2: out[row] = x[row] + y (global); MODE 2 keeps the sum as the next operand tile
        if (lr < nrows) {
          const vec_t xr = *reinterpret_cast<const vec_t*>(a.X + (size_t)(row0 + lr) * a.ldx + lane * 4);
          vec_t o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = xr[i] + y[i];
          *reinterpret_cast<vec_t*>(a.out + (size_t)(row0 + lr) * a.ldo + lane * 4) = o;
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = o[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = 0.f;
        }
      }
      if constexpr (MODE != 1) {
        char* g = A + lr * SA + ((lane * 4) >> 3) * 48 + ((lane * 4) & 7) * 2;   // lane owns k = 4 lane .. 4 lane + 3
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          unsigned hi, mid, lo;
          split2(y[i], y[i + 1], hi, mid, lo);
          *reinterpret_cast<unsigned*>(g + i * 2) = hi;
          *reinterpret_cast<unsigned*>(g + 16 + i * 2) = mid;
          *reinterpret_cast<unsigned*>(g + 32 + i * 2) = lo;
        }
      }
    }
  };
  auto zero2 = [](auto& acc) {
#pragma unroll
    for (auto& row : acc)
#pragma unroll
      for (auto& t : row)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = 0.f;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  using I16 = std::integral_constant<int, 16>;
  using I32 = std::integral_constant<int, 32>;

  // ---- 1. x tile -> operand tile; mlp.0 first half of K (transformer.py:91: cat([x, message]), x first) ------------------
#pragma unroll
  for (int i = 0; i < R64 * (C / 8) / NT; ++i) {
    const int p = tid + i * NT;
    const int r = p / (C / 8), g = p - r * (C / 8);
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (r < nrows) {
      const float4* s4 = reinterpret_cast<const float4*>(a.X + (size_t)(row0 + r) * a.ldx + g * 8);
      v0 = s4[0];
      v1 = s4[1];
    }
    split8_store(v0, v1, A + r * SA + g * 48);
  }
  // B operand of the apply: KV_h[d = 2 i + half][v = l31] of this wave's head, in flight during the first GEMM
  const int src = a.cross ? 1 - stream : stream;          // quirk q6: both streams use pre-update K, V
  const float src_len = (float)(src ? a.len1 : a.len0);
  const float* kvp = (src && a.kv1 != nullptr) ? a.kv1 : a.kv + (size_t)src * (C * D);
  const float* ksp = (src && a.ks1 != nullptr) ? a.ks1 : a.ks + (size_t)src * C;
  const int h = wave;
  float bk[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bk[i] = kvp[(h * D + 2 * i + half) * D + l31];
  __syncthreads();
  f32x16 hid[2][2];
  zero2(hid);
  gemm(I2{}, I3{}, I0{}, I16{}, I32{}, a.w1, wave * 2, hid);
  __syncthreads();     // the x operand tile is dead

  // ---- 2. attention message, 32 rows at a time: fp32 MFMA over d (linear_attention.py:57-61), exactly as
  //         linattn_apply_pair_kernel; the message goes straight from the accumulators into the operand tile ------------
  constexpr int QS = 258;
  float* qsh = S;
#pragma unroll
  for (int rh = 0; rh < 2; ++rh) {
#pragma unroll
    for (int i = 0; i < 32 * (C / 4) / NT; ++i) {
      const int e = tid + i * NT;
      const int r = e / (C / 4), c4 = e - r * (C / 4);
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rh * 32 + r < nrows) q = *reinterpret_cast<const float4*>(qrows + (size_t)(rh * 32 + r) * a.ldq + c4 * 4);
      float2* dst = reinterpret_cast<float2*>(qsh + r * QS + c4 * 4);
      dst[0] = make_float2(q.x, q.y);
      dst[1] = make_float2(q.z, q.w);
    }
    __syncthreads();
    if (tid < 8 * 32) {   // the normaliser of (token, head): a sequential fmaf chain over d
      const int tok = tid & 31, hh = tid >> 5;
      const float* q = qsh + tok * QS + hh * D;
      const float* k = ksp + hh * D;
      float den = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) den = fmaf(q[d], k[d], den);
      z_sh[hh * 32 + tok] = 1.0f / (den + a.eps_attn);
    }
    f32x16 num;
#pragma unroll
    for (int r = 0; r < 16; ++r) num[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) num = __builtin_amdgcn_mfma_f32_32x32x2f32(qsh[l31 * QS + h * D + 2 * i + half], bk[i], num, 0, 0, 0);
    __syncthreads();
    // message value of (row, k = h * 32 + l31) -> its hi / mid / lo bf16 in the operand tile
    {
      char* col = A + (h * 4 + (l31 >> 3)) * 48 + (l31 & 7) * 2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float v = (num[r] * z_sh[h * 32 + row]) * src_len;
        unsigned hi, mid, lo;
        split2(v, 0.f, hi, mid, lo);
        char* d = col + (rh * 32 + row) * SA;
        *reinterpret_cast<unsigned short*>(d) = (unsigned short)hi;
        *reinterpret_cast<unsigned short*>(d + 16) = (unsigned short)mid;
        *reinterpret_cast<unsigned short*>(d + 32) = (unsigned short)lo;
      }
    }
    __syncthreads();   // the Q half and its normalisers are consumed
  }

  // ---- 3. merge -> norm1 (transformer.py:86-87) -> operand tile ---------------------------------------------------------
  {
    f32x16 acc[2][1];
    zero2(acc);
    gemm(I1{}, I4{}, I0{}, I16{}, I16{}, a.wm, wave, acc);
    __syncthreads();   // every wave is done reading the message tile
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      stage(acc[rh][0], wave * 32, false);
      __syncthreads();
      layernorm_rows(a.g1, a.b1, a.eps_ln, I0{}, rh * 32);
      __syncthreads();
    }
  }

  // ---- 4. mlp.0, second half of K, + ReLU (applied when the accumulators are staged) ----------------------------------
  gemm(I2{}, I3{}, I16{}, I32{}, I32{}, a.w1, wave * 2, hid);
  __syncthreads();     // the merged operand tile is dead

  // ---- 5. mlp.2 over the hidden activation, 256 of its 512 columns (= K of mlp.2) at a time ---------------------------
  f32x16 out[2][1];
  zero2(out);
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      if (wave / (NW / 2) == kh) {
        const int c0 = (wave - kh * (NW / 2)) * 64;
        stage(hid[rh][0], c0, true);
        stage(hid[rh][1], c0 + 32, true);
      }
      __syncthreads();
      split_staged(rh * 32, 0);
      __syncthreads();
    }
    if (kh == 0) gemm(I1{}, I4{}, I0{}, I16{}, I32{}, a.w2, wave, out);
    else gemm(I1{}, I4{}, I16{}, I32{}, I32{}, a.w2, wave, out);
    __syncthreads();   // before the operand tile is rewritten / the staging tile reused
  }

  // ---- 6. norm2 -> x + . (transformer.py:92-94) ------------------------------------------------------------------------
  const bool fold = a.wq_next != nullptr;             // (kernel argument: uniform)
#pragma unroll
  for (int rh = 0; rh < 2; ++rh) {
    stage(out[rh][0], wave * 32, false);
    __syncthreads();
    if (fold) layernorm_rows(a.g2, a.b2, a.eps_ln, I2{}, rh * 32);     // ... and the finished rows become the operand tile of step 7
    else layernorm_rows(a.g2, a.b2, a.eps_ln, I1{}, rh * 32);
    __syncthreads();
  }
  if (!fold) return;

  // ---- 7. the NEXT layer's q | k | v projection of this tile (transformer.py:76-79 of layer i + 1; r05): 24 column tiles of 32, three per
  //         wave, K = 256; same six-product sequence and epilogue arithmetic as the stand-alone GEMM (gemm_mfma.hip, OPP_ACT_QKV) ---------
  float* qdst = (stream && a.qkv_out1 != nullptr) ? a.qkv_out1 + (size_t)(cidx * R64) * (3 * C) : a.qkv_out + (size_t)row0 * (3 * C);
  const float vdiv = (float)seg_len;                  // values / v_length of the stream that produces them (linear_attention.py:55-56)
  auto store_qkv = [&](const f32x16& t, int rb, int tile) {
    const int col = tile * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rb + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row < nrows) {
        float v = t[r];
        if (col < 2 * C) v = v > 0.f ? v + 1.f : __expf(v);      // elu(x) + 1, linear_attention.py:10-11
        else v = v / vdiv;
        if (stream == 0 && a.qmask != nullptr) v *= a.qmask[row0 + row];   // padded image tokens: q, k, v rows -> 0 (linear_attention.py:49-53)
        qdst[(size_t)row * (3 * C) + col] = v;
      }
    }
  };
  {
    f32x16 acc[2][2];
    zero2(acc);
    gemm(I2{}, I3{}, I0{}, I16{}, I16{}, a.wq_next, wave * 3, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) store_qkv(acc[i][j], i * 32, wave * 3 + j);
  }
  {
    f32x16 acc[2][1];
    zero2(acc);
    gemm(I1{}, I4{}, I0{}, I16{}, I16{}, a.wq_next, wave * 3 + 2, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i) store_qkv(acc[i][0], i * 32, wave * 3 + 2);
  }
}

}  // namespace

int opp_enc_layer64(const OppEncChain& a, hipStream_t stream) {
  OPP_CHECK_ARG(a.C == 256 && a.apply && a.X && a.out && a.q && a.kv && a.ks && a.wm && a.w1 && a.w2 && a.g1 && a.b1 && a.g2 && a.b2,
                "enc_layer64: coarse level (C = 256) with the fused attention apply only");
  OPP_CHECK_ARG(a.len0 >= 0 && a.len1 >= 0 && a.len0 + a.len1 > 0, "enc_layer64: empty token set");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  OPP_CHECK_ARG(al16(a.X) && al16(a.out) && al16(a.q) && a.ldx % 4 == 0 && a.ldo % 4 == 0 && a.ldq % 4 == 0 && al16(a.g1) && al16(a.b1) &&
                    al16(a.g2) && al16(a.b2) && al16(a.wm) && al16(a.w1) && al16(a.w2) && al16(a.q1), "enc_layer64: operands must be 16-byte aligned");
  OPP_CHECK_ARG(a.wq_next == nullptr || (a.qkv_out != nullptr && al16(a.wq_next)), "enc_layer64: folded projection needs its output buffer");
  static OppLdsOnce lds_once;            // per device (opp_common.h)
  opp_lds_opt_in(reinterpret_cast<const void*>(enc_layer64_kernel), kLds64, lds_once);
  const int tiles = opp_cdiv(a.len0, R64) + opp_cdiv(a.len1, R64);
  OppProfScope prof(OPP_PROF_ENC_CHAIN, stream, 2.0 * (double)(a.len0 + a.len1) * ((a.wq_next ? 10.0 : 7.0) * C * C + 32.0 * C));
  hipLaunchKernelGGL(enc_layer64_kernel, dim3(tiles), dim3(NT), kLds64, stream, a);
  OPP_CHECK_LAUNCH("enc_layer64_kernel");
  return OPP_OK;
}
