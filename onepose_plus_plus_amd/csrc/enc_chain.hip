// One LoFTREncoderLayer per launch, after the Q/K/V projection and the KV reduction:
//   message = LinearAttention apply  ->  merge  ->  norm1  ->  mlp.0(cat[x, message])  ->  ReLU  ->  mlp.2  ->  norm2  ->  x + .
// (/root/reference/src/models/OnePosePlus/loftr_module/transformer.py:65-94, linear_attention.py:55-61)
//
// Why one kernel: at 9096 tokens (4096 image cells + 5000 points) each of these Linears is 1.2 - 4.8 GFLOP, i.e. 3 - 11 us
// of MFMA time on the whole chip; as separate launches they are dominated by launch gaps, prologue latency, tile
// quantisation (143 - 285 workgroups on 256 CUs) and the round trips of [T][256..512] fp32 activations through HBM.
// Here a workgroup owns a tile of 32 token rows for the whole chain:
//   * the activations of the tile live in LDS in the MFMA A-operand form of the bf16x3 arithmetic
//     (x = hi + mid + lo bf16 exactly, 48 B per 8 k: [hi x8 | mid x8 | lo x8]; row stride K * 6 + 16 B so that the 16 rows
//     of a ds_read_b128 lane group fall on 16 distinct 4-bank groups);
//   * every wave owns its own 32 (merge, mlp.2) or 64 (mlp.0) output columns, so the weight (B) fragments are not shared
//     between waves and go L2 -> registers directly: the weights are packed at load time in fragment-major order
//     (opp_pack_frag_b3: one k16-step of one 32-column tile and one split part = 1 KiB contiguous = one fully coalesced
//     buffer_load_dwordx4 per wave), prefetched kDepth k16-steps ahead in a register ring;
//   * six v_mfma_f32_32x32x16_bf16 per product (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; fp32 accumulate), K walked
//     in ascending k16-steps: the same accumulation sequence as opp_gemm_kernel<bf16x3>, so the results are bit-identical
//     to the launch-per-Linear path (tests/test_kernels_gpu.py::test_encoder_chain_*);
//   * between the GEMMs the accumulators are staged once through an fp32 LDS tile (LayerNorm needs whole rows), normalised /
//     activated, split to bf16x3 and written back as the next A operand; the 512-wide hidden activation overwrites the dead
//     x / message operand tiles.
// LDS: C = 256: 2 x 48.5 KB operand tiles + 32.5 KB fp32 staging + 1 KB = 131 KB, one 8-wave workgroup per CU;
//      C = 128 (fine level): 66 KB, 4 waves, two workgroups per CU.
// Bound: MFMA (672 MFMAs per wave and tile at C = 256) co-limited by the L2 -> CU weight stream (2.75 MB of split weights
// per 32-row tile, 64 B/clk/CU at the MFMA rate).
#include <stdlib.h>

#include "enc_frag.h"


namespace {

constexpr int kRows = 32;      // token rows per workgroup
constexpr int kDepthDefault = 4;   // k16-steps of weight fragments in flight per wave

template <int C, bool APPLY, int kDepth = kDepthDefault, int ABL = 0>
__global__ __launch_bounds__(C * 2) void enc_chain_kernel(const OppEncChain a) {
  constexpr int NW = C / 32;                 // waves: one 32-column tile of the C-wide outputs each
  constexpr int NT = NW * 64;
  constexpr int SA = a_stride_bytes(C);      // operand tile row stride, K = C
  constexpr int SH = a_stride_bytes(2 * C);  // hidden tile row stride, K = 2 C
  constexpr int TILE_A = kRows * SA;
  constexpr int SS = C + 4;                  // fp32 staging row stride (floats)
  constexpr int OFF_AX = 0, OFF_AG = TILE_A, OFF_S = 2 * TILE_A, OFF_Z = OFF_S + kRows * SS * 4;
  static_assert(kRows * SH <= 2 * TILE_A, "hidden tile must fit the two dead operand tiles");
  static_assert(!APPLY || kRows * 258 * 4 <= TILE_A, "Q tile aliases the message operand tile");
  static_assert(!APPLY || C == 256, "apply prologue: coarse level only (wave = head)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* AX = smem + OFF_AX;
  char* AG = smem + OFF_AG;
  float* S = reinterpret_cast<float*>(smem + OFF_S);
  float* z_sh = reinterpret_cast<float*>(smem + OFF_Z);   // [8][32] (APPLY)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  // tile -> rows [row0, row0 + nrows) of ONE stream (a tile never straddles the stream boundary: the attention of
  // its rows uses one KV)
  const int chunks0 = (a.len0 + kRows - 1) / kRows;
  const int stream = (int)blockIdx.x >= chunks0 ? 1 : 0;
  const int cidx = stream ? blockIdx.x - chunks0 : blockIdx.x;
  const int seg_len = stream ? a.len1 : a.len0;
  const int row0 = (stream ? a.len0 : 0) + cidx * kRows;
  const int nrows = min(kRows, seg_len - cidx * kRows);

  // ---- x tile -> bf16x3 operand tile AX -----------------------------------------------------------------
  auto load_split = [&](const float* src, int ld, char* dst) {
#pragma unroll
    for (int i = 0; i < kRows * (C / 8) / NT; ++i) {
      const int p = tid + i * NT;
      const int r = p / (C / 8), g = p - r * (C / 8);
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (r < nrows) {
        const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)(row0 + r) * ld + g * 8);
        v0 = s4[0];
        v1 = s4[1];
      }
      split8_store(v0, v1, dst + r * SA + g * 48);
    }
  };
  load_split(a.X, a.ldx, AX);

  // ---- attention message of the tile -> operand tile AG -------------------------------------------------
  if constexpr (APPLY) {
    // msg[t][h*32 + v] = (sum_d Q[t][h*32+d] KV_h[d][v]) / (sum_d Q[t][h*32+d] Ksum_h[d] + eps) * S  (linear_attention.py:57-61)
    // exactly as linattn_apply_pair_kernel (attention.hip): wave = head, fp32 MFMA over d in ascending order
    constexpr int D = 32, QS = 258;
    float* qsh = reinterpret_cast<float*>(AG);
    const int src = a.cross ? 1 - stream : stream;          // quirk q6: both streams use pre-update K, V
    const float src_len = (float)(src ? a.len1 : a.len0);
    const float* kvp = a.kv + (size_t)src * (C * D);
    const float* ksp = a.ks + (size_t)src * C;
    const int h = wave;
    float bk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bk[i] = kvp[(h * D + 2 * i + half) * D + l31];
#pragma unroll
    for (int i = 0; i < kRows * (C / 4) / NT; ++i) {
      const int e = tid + i * NT;
      const int r = e / (C / 4), c4 = e - r * (C / 4);
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < nrows) q = *reinterpret_cast<const float4*>(a.q + (size_t)(row0 + r) * a.ldq + c4 * 4);
      float2* dst = reinterpret_cast<float2*>(qsh + r * QS + c4 * 4);
      dst[0] = make_float2(q.x, q.y);
      dst[1] = make_float2(q.z, q.w);
    }
    __syncthreads();
    if (tid < 8 * kRows) {   // the normaliser of (token, head): a sequential fmaf chain over d
      const int tok = tid & (kRows - 1), hh = tid / kRows;
      const float* q = qsh + tok * QS + hh * D;
      const float* k = ksp + hh * D;
      float den = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) den = fmaf(q[d], k[d], den);
      z_sh[hh * kRows + tok] = 1.0f / (den + a.eps_attn);
    }
    f32x16 num;
#pragma unroll
    for (int r = 0; r < 16; ++r) num[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) num = __builtin_amdgcn_mfma_f32_32x32x2f32(qsh[l31 * QS + h * D + 2 * i + half], bk[i], num, 0, 0, 0);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      S[row * SS + h * D + l31] = (num[r] * z_sh[h * kRows + row]) * src_len;
    }
    __syncthreads();   // Q tile (aliasing AG) is dead, the message tile is staged
    // staged fp32 message -> operand tile AG
#pragma unroll
    for (int i = 0; i < kRows * (C / 8) / NT; ++i) {
      const int p = tid + i * NT;
      const int r = p / (C / 8), g = p - r * (C / 8);
      const float4* s4 = reinterpret_cast<const float4*>(S + r * SS + g * 8);
      split8_store(s4[0], s4[1], AG + r * SA + g * 48);
    }
  } else {
    load_split(a.msg, a.ldm, AG);
  }
  __syncthreads();

  // ---- GEMM over the LDS operand tile(s): acc[j] += A[32 rows][K] * W[tile t0 + j][K]^T -------------------------
  // a_of(s) = LDS byte address of this lane's fragment group of k16-step s (row l31, k-group 2 s + half)
  // wf = fragment-major weights: ((t * KS + s) * 3 + part) * 1024 + lane * 16 bytes
  auto gemm = [&](auto ntile_c, auto ks_c, auto&& a_of, const void* wf, size_t wf_bytes, int t0, f32x16 (&acc)[decltype(ntile_c)::value]) {
    constexpr int NTILE = decltype(ntile_c)::value;
    constexpr int KS = decltype(ks_c)::value;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wf), 0, (int)wf_bytes, 0x00020000);
    const int voff = lane * 16;
    u32x4 bq[kDepth][NTILE][3];
    u32x4 af[2][3];
    auto load_b = [&](int s, int slot) {
#pragma unroll
      for (int j = 0; j < NTILE; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bq[slot][j][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (((t0 + j) * KS + s) * 3 + p) * 1024, 0);
    };
    auto load_a = [&](int s, int slot) {
      const char* p = a_of(s);
#pragma unroll
      for (int q = 0; q < 3; ++q) af[slot][q] = *reinterpret_cast<const u32x4*>(p + q * 16);
    };
#pragma unroll
    for (int s = 0; s < kDepth && s < KS; ++s) load_b(s, s);
    load_a(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};   // A part of product pr (0 hi, 1 mid, 2 lo), smallest terms first
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    if (ABL == 3) return;   // tuning: no GEMM at all (prologue + phase transitions only)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) load_a(s + 1, (s + 1) & 1);
      if (ABL != 2) {       // tuning 2: weight stream without the MFMAs
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
          for (int j = 0; j < NTILE; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s & 1][PA[pr]]),
                                                             __builtin_bit_cast(bf16x8, bq[s % kDepth][j][PB[pr]]), acc[j], 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < NTILE; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) asm volatile("" ::"v"(bq[s % kDepth][j][p]), "v"(af[s & 1][p]));
      }
      if (ABL != 1 && s + kDepth < KS) load_b(s + kDepth, s % kDepth);   // tuning 1: MFMAs without the weight stream
      // the refill of this step's ring slot must not sink below the next step (the scheduler would otherwise keep
      // one or two loads in flight instead of kDepth - 1 steps)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto zero = [](auto& acc) {
#pragma unroll
    for (auto& t : acc)
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = 0.f;
  };
  // accumulator tile (columns col0 + l31) -> fp32 staging tile
  auto stage = [&](const f32x16& t, int col0, bool relu) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = t[r];
      if (relu) v = v < 0.f ? 0.f : v;   // NaN-propagating like torch.relu
      S[row * SS + col0 + l31] = v;
    }
  };
  // LayerNorm of the staged rows (one wave per row; same arithmetic as layernorm_kernel / the GEMM epilogue):
  //   mode 0: y -> operand tile dst (bf16x3) ; mode 1: out[row] = x[row] + y (global)
  constexpr int VPT = C / 64;
  typedef float vec_t __attribute__((ext_vector_type(VPT)));
  auto layernorm_rows = [&](const float* gamma, const float* beta, float eps, auto mode_c, char* dst) {
    constexpr int MODE = decltype(mode_c)::value;
    const vec_t gmv = *reinterpret_cast<const vec_t*>(gamma + lane * VPT);
    const vec_t btv = *reinterpret_cast<const vec_t*>(beta + lane * VPT);
    constexpr int RPW = kRows / NW;
    float v[RPW][VPT], sm[RPW], mean[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int lr = wave * RPW + r;
      const vec_t t = *reinterpret_cast<const vec_t*>(S + lr * SS + lane * VPT);
      sm[r] = 0.f;
#pragma unroll
      for (int i = 0; i < VPT; ++i) {
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
      const int lr = wave * RPW + r;
      float y[VPT];
#pragma unroll
      for (int i = 0; i < VPT; ++i) y[i] = opp_ln_affine(v[r][i], mean[r], rstd[r], gmv[i], btv[i]);
      if constexpr (MODE == 0) {
        // lane owns k = lane * VPT .. + VPT - 1 of the row: VPT / 2 packed dwords per part
        char* g = dst + lr * SA + ((lane * VPT) >> 3) * 48 + ((lane * VPT) & 7) * 2;
#pragma unroll
        for (int i = 0; i < VPT; i += 2) {
          unsigned hi, mid, lo;
          split2(y[i], y[i + 1], hi, mid, lo);
          *reinterpret_cast<unsigned*>(g + i * 2) = hi;
          *reinterpret_cast<unsigned*>(g + 16 + i * 2) = mid;
          *reinterpret_cast<unsigned*>(g + 32 + i * 2) = lo;
        }
      } else {
        if (lr < nrows) {
          const vec_t xr = *reinterpret_cast<const vec_t*>(a.X + (size_t)(row0 + lr) * a.ldx + lane * VPT);
          vec_t o;
#pragma unroll
          for (int i = 0; i < VPT; ++i) o[i] = xr[i] + y[i];
          *reinterpret_cast<vec_t*>(a.out + (size_t)(row0 + lr) * a.ldo + lane * VPT) = o;
        }
      }
    }
  };

  const char* a_row_x = AX + l31 * SA + half * 48;
  const char* a_row_g = AG + l31 * SA + half * 48;
  const char* a_row_h = smem + l31 * SH + half * 48;
  const size_t wbytes_cc = (size_t)C * C * 6, wbytes_h = (size_t)2 * C * 2 * C * 6, wbytes_2 = (size_t)C * 2 * C * 6;

  // ---- merge -> norm1 (transformer.py:86-87) -------------------------------------------------------------
  {
    f32x16 acc[1];
    zero(acc);
    gemm(std::integral_constant<int, 1>{}, std::integral_constant<int, C / 16>{}, [&](int s) { return a_row_g + s * 96; }, a.wm, wbytes_cc, wave, acc);
    stage(acc[0], wave * 32, false);
  }
  __syncthreads();     // staged rows complete; every wave is done reading the message tile
  layernorm_rows(a.g1, a.b1, a.eps_ln, std::integral_constant<int, 0>{}, AG);
  __syncthreads();

  // ---- mlp.0 on cat([x, message]) + ReLU (transformer.py:91) ---------------------------------------------------
  f32x16 hid[2];
  zero(hid);
  gemm(std::integral_constant<int, 2>{}, std::integral_constant<int, 2 * C / 16>{},
       [&](int s) { return s < C / 16 ? a_row_x + s * 96 : a_row_g + (s - C / 16) * 96; }, a.w1, wbytes_h, wave * 2, hid);
  __syncthreads();     // x / message operand tiles are dead: the hidden tile takes their place
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (wave / (NW / 2) == pass) {
      const int c0 = (wave - pass * (NW / 2)) * 64;
      stage(hid[0], c0, true);
      stage(hid[1], c0 + 32, true);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kRows * (C / 8) / NT; ++i) {
      const int p = tid + i * NT;
      const int r = p / (C / 8), g = p - r * (C / 8);
      const float4* s4 = reinterpret_cast<const float4*>(S + r * SS + g * 8);
      split8_store(s4[0], s4[1], smem + r * SH + (pass * (C / 8) + g) * 48);
    }
    __syncthreads();
  }

  // ---- mlp.2 -> norm2 -> x + . (transformer.py:92-94) ----------------------------------------------------------
  {
    f32x16 acc[1];
    zero(acc);
    gemm(std::integral_constant<int, 1>{}, std::integral_constant<int, 2 * C / 16>{}, [&](int s) { return a_row_h + s * 96; }, a.w2, wbytes_2, wave, acc);
    stage(acc[0], wave * 32, false);
  }
  __syncthreads();
  layernorm_rows(a.g2, a.b2, a.eps_ln, std::integral_constant<int, 1>{}, nullptr);
}

// fp32 W [N][K] (PyTorch Linear layout) -> fragment-major bf16x3: for 32-column tile t, k16-step s, part p (hi, mid, lo),
// lane (l31, half): the 8 bf16 of W[t*32 + l31][(2 s + half) * 8 .. + 8) at ((t * K/16 + s) * 3 + p) * 1024 + lane * 16
__global__ __launch_bounds__(256) void pack_frag_b3_kernel(const float* __restrict__ w, int N, int K, char* __restrict__ out) {
  const int KS = K / 16;
  const size_t n_items = (size_t)(N / 32) * KS * 64;       // (t, s, lane)
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < n_items; it += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(it & 63);
    const size_t ts = it >> 6;
    const int s = (int)(ts % KS), t = (int)(ts / KS);
    const int n = t * 32 + (lane & 31), k0 = (2 * s + (lane >> 5)) * 8;
    const float4* src = reinterpret_cast<const float4*>(w + (size_t)n * K + k0);
    const float4 v0 = src[0], v1 = src[1];
    u32x4 hi, mid, lo;
    split8(v0, v1, hi, mid, lo);
    char* dst = out + ts * 3 * 1024 + lane * 16;
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + 1024) = mid;
    *reinterpret_cast<u32x4*>(dst + 2048) = lo;
  }
}

template <int C, bool APPLY, int DEPTH = kDepthDefault, int ABL = 0>
int launch_chain(const OppEncChain& a, hipStream_t stream) {
  constexpr int TILE_A = kRows * a_stride_bytes(C);
  constexpr size_t lds = 2 * TILE_A + kRows * (C + 4) * 4 + 1024;
  auto k = enc_chain_kernel<C, APPLY, DEPTH, ABL>;
  static OppLdsOnce lds_once;            // per device (opp_common.h)
  opp_lds_opt_in(reinterpret_cast<const void*>(k), lds, lds_once);
  const int tiles = opp_cdiv(a.len0, kRows) + opp_cdiv(a.len1, kRows);
  // algorithmic FLOPs: merge C*C + mlp.0 2C*2C + mlp.2 2C*C per token (+ the apply: C*32 per token)
  OppProfScope prof(OPP_PROF_ENC_CHAIN, stream, 2.0 * (double)(a.len0 + a.len1) * (7.0 * C * C + (APPLY ? 32.0 * C : 0.0)));
  hipLaunchKernelGGL(k, dim3(tiles), dim3(C * 2), lds, stream, a);
  OPP_CHECK_LAUNCH("enc_chain_kernel");
  return OPP_OK;
}

}  // namespace

size_t opp_frag_b3_bytes(int N, int K) { return (size_t)N * K * 6; }

int opp_pack_frag_b3(const float* w, int N, int K, void* out, hipStream_t stream) {
  OPP_CHECK_ARG(w && out && N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0, "pack_frag_b3: N %% 32 / K %% 16 (got %d x %d)", N, K);
  const size_t items = (size_t)(N / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(pack_frag_b3_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, w, N, K, static_cast<char*>(out));
  OPP_CHECK_LAUNCH("pack_frag_b3_kernel");
  return OPP_OK;
}

bool opp_enc_chain_ok(int C, int nhead, bool apply) { return (C == 256 && nhead == 8) || (C == 128 && !apply); }

int opp_enc_chain(const OppEncChain& a, hipStream_t stream) {
  OPP_CHECK_ARG(a.X && a.out && a.wm && a.w1 && a.w2 && a.g1 && a.b1 && a.g2 && a.b2, "enc_chain: null argument");
  OPP_CHECK_ARG(a.len0 >= 0 && a.len1 >= 0 && a.len0 + a.len1 > 0, "enc_chain: empty token set");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  OPP_CHECK_ARG(al16(a.X) && al16(a.out) && a.ldx % 4 == 0 && a.ldo % 4 == 0 && al16(a.g1) && al16(a.b1) && al16(a.g2) && al16(a.b2) &&
                    al16(a.wm) && al16(a.w1) && al16(a.w2), "enc_chain: operands must be 16-byte aligned");
  if (a.apply) {
    OPP_CHECK_ARG(a.C == 256 && a.q && a.kv && a.ks && al16(a.q) && a.ldq % 4 == 0, "enc_chain: fused attention apply needs C = 256 and aligned phi(Q)");
#ifdef OPP_TUNING
    static const int abl = getenv("OPP_CHAIN_ABL") ? atoi(getenv("OPP_CHAIN_ABL")) : 0;
    static const int depth = getenv("OPP_CHAIN_DEPTH") ? atoi(getenv("OPP_CHAIN_DEPTH")) : kDepthDefault;
    if (abl == 1) return launch_chain<256, true, kDepthDefault, 1>(a, stream);
    if (abl == 2) return launch_chain<256, true, kDepthDefault, 2>(a, stream);
    if (abl == 3) return launch_chain<256, true, kDepthDefault, 3>(a, stream);
    if (depth == 2) return launch_chain<256, true, 2>(a, stream);
    if (depth == 3) return launch_chain<256, true, 3>(a, stream);
    if (depth == 6) return launch_chain<256, true, 6>(a, stream);
    if (depth == 8) return launch_chain<256, true, 8>(a, stream);
#endif
    return launch_chain<256, true>(a, stream);
  }
  OPP_CHECK_ARG(a.msg && al16(a.msg) && a.ldm % 4 == 0, "enc_chain: message operand missing / unaligned");
  if (a.C == 256) return launch_chain<256, false>(a, stream);
  if (a.C == 128) return launch_chain<128, false>(a, stream);
  opp_set_error("enc_chain: unsupported width %d", a.C);
  return OPP_ERR_UNSUPPORTED;
}
