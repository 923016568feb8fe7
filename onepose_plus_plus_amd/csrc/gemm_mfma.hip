// fp32 MFMA GEMM / implicit-GEMM convolution for gfx950.
//
// Replaces, on the hot path, what the reference gets from cuDNN/cuBLAS through nn.Conv2d,
// nn.Linear and torch.einsum:
//   backbone convs      src/models/OnePosePlus/backbone/resnet.py:10-45, :101-124, :141-164
//   transformer Linears src/models/OnePosePlus/loftr_module/transformer.py:26-47, :76-92
//   coarse score einsum src/models/OnePosePlus/utils/coarse_matching.py:102-107
//
// Design (MI355X-first, not a translated CUDA tiling):
//  * v_mfma_f32_32x32x2_f32: exact fp32 (the parity budget of 1e-4 on confidences rules out
//    bf16 operands), 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak.
//  * Both operands are K-contiguous ("TN").  A 32-wide K chunk of a tile row is 128 B: a wave
//    loads 8 rows x 128 B per instruction (full cache lines) and stores them to LDS as
//    [row][36] floats (4 floats of padding).  Within a chunk the k index is permuted so that
//    lane-half h of the wave owns k in [16h, 16h+16): every lane then pulls its MFMA operands
//    with 4 conflict-free ds_read_b128 per 32x32 sub-tile instead of 16 scalar reads.
//    (The MFMA sums over k, so a permutation applied to both operands is exact maths.)
//  * Implicit im2col: the A row of an output pixel for chunk (tap, c0) is one contiguous
//    128 B run of the NHWC input; zero rows are synthesised for the padding halo.
//  * Double-buffered LDS, next chunk's global loads issued before the current chunk's MFMAs
//    (register-staged prefetch), one barrier per chunk.
//  * Epilogue fused: folded-BN bias, residual (direct or bilinear x2 align_corners=True
//    upsample of a half-resolution NHWC tensor), ReLU / LeakyReLU / elu+1 feature map, value
//    scaling for the linear attention, temperature scaling for the score matrix.
#include <vector>

#include "opp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kLdsStride = 36;  // floats per LDS tile row: 32 + 4 pad (keeps 16 B alignment)

template <int BM, int BN, int WAVES_M, int WAVES_N, bool CONV>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void opp_gemm_kernel(const OppGemm g) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  constexpr int A_LD = BM * 8 / NT;
  constexpr int B_LD = BN * 8 / NT;
  static_assert(BM % (WAVES_M * 32) == 0 && BN % (WAVES_N * 32) == 0, "tile shape");
  static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "load split");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                         // [2][BM][36]
  float* Bs = smem + 2 * BM * kLdsStride;   // [2][BN][36]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  const int tiles_n = (g.n_store + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n;
  const int tile_n = blockIdx.x - tile_m * tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const int kq = tid & 7;     // which float4 of the 32-float chunk this thread moves
  const int lrow = tid >> 3;  // tile row of the first load slot; slot i adds i*(NT/8)

  // ---- per-thread A row descriptors -------------------------------------------------
  int a_iy0[A_LD], a_ix0[A_LD], a_pix[A_LD];   // conv mode
  int a_off0[A_LD], a_off1[A_LD];              // dense mode
  bool a_ok[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int r = m0 + lrow + i * (NT / 8);
    a_ok[i] = r < g.M;
    if (CONV) {
      const int ox = r % g.Wout;
      const int t = r / g.Wout;
      const int oy = t % g.Hout;
      const int b = t / g.Hout;
      a_iy0[i] = a_ok[i] ? oy * g.stride - g.pad : -(1 << 28);
      a_ix0[i] = ox * g.stride - g.pad;
      a_pix[i] = b * g.Hin * g.Win;
      a_off0[i] = a_off1[i] = 0;
    } else {
      a_off0[i] = r * g.lda0;
      a_off1[i] = r * g.lda1;
      a_iy0[i] = a_ix0[i] = a_pix[i] = 0;
    }
  }
  int b_off[B_LD];
  bool b_ok[B_LD];
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int n = n0 + lrow + i * (NT / 8);
    b_ok[i] = n < g.N;
    b_off[i] = n * g.ldw;
  }

  float4 a_reg[A_LD], b_reg[B_LD];

  auto load_global = [&](int kc) {
    const int k0 = kc * 32;
    if (CONV) {
      const int tap = k0 / g.Cin;
      const int c0 = k0 - tap * g.Cin;
      const int ky = tap / g.ksize;
      const int kx = tap - ky * g.ksize;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const int iy = a_iy0[i] + ky;
        const int ix = a_ix0[i] + kx;
        const bool inb = (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inb) {
          const size_t off = (size_t)(a_pix[i] + iy * g.Win + ix) * g.Cin + c0 + kq * 4;
          v = *reinterpret_cast<const float4*>(g.A0 + off);
        }
        a_reg[i] = v;
      }
    } else {
      const bool first = k0 < g.ksplit;
      const float* base = first ? g.A0 : g.A1;
      const int kk = (first ? k0 : k0 - g.ksplit) + kq * 4;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_ok[i]) v = *reinterpret_cast<const float4*>(base + (first ? a_off0[i] : a_off1[i]) + kk);
        a_reg[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b_ok[i]) v = *reinterpret_cast<const float4*>(g.W + b_off[i] + k0 + kq * 4);
      b_reg[i] = v;
    }
  };

  auto store_lds = [&](int buf) {
    float* as = As + buf * BM * kLdsStride;
    float* bs = Bs + buf * BN * kLdsStride;
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      *reinterpret_cast<float4*>(as + (lrow + i * (NT / 8)) * kLdsStride + kq * 4) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      *reinterpret_cast<float4*>(bs + (lrow + i * (NT / 8)) * kLdsStride + kq * 4) = b_reg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / 32;
  load_global(0);
  store_lds(0);
  __syncthreads();

  const int a_frag = (wm * TM * 32 + l31) * kLdsStride + half * 16;
  const int b_frag = (wn * TN * 32 + l31) * kLdsStride + half * 16;

  for (int kc = 0; kc < nk; ++kc) {
    const int cur = kc & 1;
    if (kc + 1 < nk) load_global(kc + 1);
    const float* as = As + cur * BM * kLdsStride + a_frag;
    const float* bs = Bs + cur * BN * kLdsStride + b_frag;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const float4*>(as + i * 32 * kLdsStride + k4 * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const float4*>(bs + j * 32 * kLdsStride + k4 * 4);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
    }
    if (kc + 1 < nk) store_lds(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------------
  const bool scale_on = (g.out_mul != 1.f) || (g.out_div != 1.f);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row >= g.M) continue;
      // bilinear x2 (align_corners=True) taps of the half-resolution residual, per row
      int p00 = 0, p01 = 0, p10 = 0, p11 = 0;
      float wy1 = 0.f, wx1 = 0.f;
      if (g.res_mode == OPP_RES_BILINEAR2X) {
        const int ox = row % g.Wout;
        const int t = row / g.Wout;
        const int oy = t % g.Hout;
        const int b = t / g.Hout;
        const float sy = g.res_sy * (float)oy;
        const float sx = g.res_sx * (float)ox;
        int y0 = (int)sy;
        if (y0 > g.Hr - 1) y0 = g.Hr - 1;
        int x0 = (int)sx;
        if (x0 > g.Wr - 1) x0 = g.Wr - 1;
        const int y1 = y0 + (y0 < g.Hr - 1 ? 1 : 0);
        const int x1 = x0 + (x0 < g.Wr - 1 ? 1 : 0);
        wy1 = fminf(fmaxf(sy - (float)y0, 0.f), 1.f);
        wx1 = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
        const int pb = b * g.Hr * g.Wr;
        p00 = (pb + y0 * g.Wr + x0) * g.ldr;
        p01 = (pb + y0 * g.Wr + x1) * g.ldr;
        p10 = (pb + y1 * g.Wr + x0) * g.ldr;
        p11 = (pb + y1 * g.Wr + x1) * g.ldr;
      }
      const float vdiv = row < g.split_row ? g.s0 : g.s1;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + l31;
        if (col >= g.n_store) continue;
        float v = acc[i][j][r];
        if (scale_on) v = (v * g.out_mul) / g.out_div;
        if (g.bias) v += g.bias[col];
        if (g.res_mode == OPP_RES_DIRECT) {
          v += g.R[(size_t)row * g.ldr + col];
        } else if (g.res_mode == OPP_RES_BILINEAR2X) {
          const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
          const float top = wx0 * g.R[p00 + col] + wx1 * g.R[p01 + col];
          const float bot = wx0 * g.R[p10 + col] + wx1 * g.R[p11 + col];
          v += wy0 * top + wy1 * bot;
        }
        if (g.act == OPP_ACT_RELU) {
          v = fmaxf(v, 0.f);
        } else if (g.act == OPP_ACT_LEAKY) {
          v = v > 0.f ? v : 0.01f * v;
        } else if (g.act == OPP_ACT_QKV) {
          if (col < g.qk_cols)
            v = v > 0.f ? v + 1.f : expm1f(v) + 1.f;   // elu(x) + 1, linear_attention.py:10-11
          else
            v = v / vdiv;                               // values / v_length, linear_attention.py:55-56
        }
        g.C[(size_t)row * g.ldc + col] = v;
      }
    }
  }
}

// ---- optional live profiling of one kernel symbol (tile config x conv/dense) with HIP events ----
struct GemmProfiler {
  bool on = false;
  int cfg = -1, conv = -1;
  std::vector<hipEvent_t> ev;   // pairs (start, stop)
  size_t used = 0;              // events used
  double flops = 0.0;
  long long dropped = 0;
} g_prof;

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(const OppGemm& g, hipStream_t stream) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  const size_t lds = (size_t)2 * (BM + BN) * kLdsStride * sizeof(float);
  const int tiles = opp_cdiv(g.M, BM) * opp_cdiv(g.n_store, BN);
  if (g.conv) {
    auto k = opp_gemm_kernel<BM, BN, WAVES_M, WAVES_N, true>;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), lds, stream, g);
  } else {
    auto k = opp_gemm_kernel<BM, BN, WAVES_M, WAVES_N, false>;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), lds, stream, g);
  }
  OPP_CHECK_LAUNCH("opp_gemm_kernel");
  return OPP_OK;
}

}  // namespace

int opp_gemm_launch_cfg(const OppGemm& g, int cfg, hipStream_t stream) {
  OPP_CHECK_ARG(g.M > 0 && g.N > 0 && g.K > 0 && g.K % 32 == 0, "gemm: bad M/N/K (%d,%d,%d)", g.M, g.N, g.K);
  OPP_CHECK_ARG(g.n_store >= g.N && g.C && g.W && g.A0, "gemm: bad output/operands");
  OPP_CHECK_ARG((size_t)g.M * (size_t)g.ldc < (1ull << 31), "gemm: output too large for 32-bit indexing");
  if (g.conv) {
    OPP_CHECK_ARG(g.Cin % 32 == 0 && g.K == g.ksize * g.ksize * g.Cin, "conv: Cin %% 32 / K mismatch");
    OPP_CHECK_ARG(g.M == g.Bn * g.Hout * g.Wout, "conv: M != B*Hout*Wout");
  } else {
    OPP_CHECK_ARG(g.ksplit % 32 == 0 && (g.ksplit >= g.K || g.A1), "gemm: bad ksplit");
    OPP_CHECK_ARG(g.res_mode != OPP_RES_BILINEAR2X, "gemm: bilinear residual needs conv mode");
  }
  if (cfg < 0) {
    const int cus = 256;
    if (g.n_store % 224 == 0) {
      cfg = (opp_cdiv(g.M, 128) * (g.n_store / 224) >= cus) ? 3 : 4;
    } else {
      const int t0 = opp_cdiv(g.M, 128) * opp_cdiv(g.n_store, 128);
      const int t1 = opp_cdiv(g.M, 64) * opp_cdiv(g.n_store, 128);
      if (t0 >= cus + cus / 2) cfg = 0;
      else if (t1 >= cus) cfg = 1;
      else cfg = 2;
    }
  }
  const bool prof = g_prof.on && g_prof.cfg == cfg && g_prof.conv == (g.conv ? 1 : 0);
  bool rec = false;
  if (prof) {
    if (g_prof.used + 2 <= g_prof.ev.size()) {
      (void)hipEventRecord(g_prof.ev[g_prof.used], stream);
      rec = true;
    } else {
      g_prof.dropped++;
    }
  }
  int rc;
  switch (cfg) {
    case 0: rc = launch_cfg<128, 128, 2, 2>(g, stream); break;
    case 1: rc = launch_cfg<64, 128, 2, 2>(g, stream); break;
    case 2: rc = launch_cfg<64, 64, 2, 2>(g, stream); break;
    case 3: rc = launch_cfg<128, 224, 4, 1>(g, stream); break;
    case 4: rc = launch_cfg<64, 224, 2, 1>(g, stream); break;
    default: opp_set_error("gemm: unknown tile config %d", cfg); return OPP_ERR_INVALID;
  }
  if (rec) {
    (void)hipEventRecord(g_prof.ev[g_prof.used + 1], stream);
    g_prof.used += 2;
    g_prof.flops += g.alg_flops > 0.0 ? g.alg_flops : 2.0 * (double)g.M * (double)g.N * (double)g.K;
  }
  return rc;
}

// Live measurement of one GEMM kernel symbol with HIP events recorded on the launch stream
// (bench.py roofline leg).  start: arm for (tile_cfg, conv) with room for `capacity` launches.
extern "C" int opp_profile_start(int tile_cfg, int conv, int capacity) {
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.ev.resize((size_t)capacity * 2);
  for (auto& e : g_prof.ev)
    if (hipEventCreate(&e) != hipSuccess) {
      opp_set_error("profile: hipEventCreate failed");
      return OPP_ERR_LAUNCH;
    }
  g_prof.used = 0;
  g_prof.flops = 0.0;
  g_prof.dropped = 0;
  g_prof.cfg = tile_cfg;
  g_prof.conv = conv;
  g_prof.on = true;
  return OPP_OK;
}

// stop: synchronises the recorded events; returns summed kernel time (ms), summed algorithmic
// FLOPs and the number of launches measured.
extern "C" int opp_profile_stop(double* total_ms, double* total_flops, int* launches) {
  g_prof.on = false;
  double ms = 0.0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) {
      opp_set_error("profile: hipEventSynchronize failed");
      return OPP_ERR_LAUNCH;
    }
    float t = 0.f;
    (void)hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]);
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = g_prof.flops;
  if (launches) *launches = (int)(g_prof.used / 2);
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.used = 0;
  return OPP_OK;
}

int opp_gemm_launch(const OppGemm& g, hipStream_t stream) { return opp_gemm_launch_cfg(g, -1, stream); }
