// Backward of the ResNet-FPN convolutions and BatchNorm layers for the training step.
//
// Reference: PL_OnePosePlus.training_step differentiates through ResNetFPN_8_2.forward
//   src/lightning_model/OnePosePlus_lightning_model.py:54-81 -> src/models/OnePosePlus/backbone/resnet.py:10-45, :101-164
// (nn.Conv2d without bias, nn.BatchNorm2d in train(), ReLU / LeakyReLU(0.01), F.interpolate(x2, bilinear, align_corners=True)).
// What autograd gets from cuDNN / MIOpen there is hand-written here:
//
//   conv weight gradient   dW[co][ci][ky][kx] = sum_p dY[p][co] * X[p shifted by (ky, kx)][ci]        (conv_wgrad_kernel)
//     a tiny output under a huge reduction over the B*Ho*Wo output pixels.  Both operands are PIXEL-major in memory (NHWC:
//     channel contiguous) while the MFMA wants 8 consecutive reduction indices per lane, so the loader transposes in
//     REGISTERS: a thread fetches 4 channels x 8 consecutive pixels (eight 16-byte loads, each coalesced over the channels of
//     one pixel), splits every value exactly into bf16 hi + mid + lo (the bf16x3 arithmetic of gemm_mfma.hip: six
//     v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate, not narrower than fp32) and writes, per channel, the 48-byte
//     [hi x8 | mid x8 | lo x8] group of those 8 pixels with three ds_write_b128 -- the LDS image is exactly the fragment
//     layout of opp_gemm_kernel<bf16x3> (52-float rows: conflict-free ds_read_b128), there are no 16-bit scattered stores and
//     no transposed copies of the activations in memory.  LDS rows are channel-permuted (row = (c % 4) * 32 + c / 4) so that
//     consecutive lanes write consecutive rows; the permutation is undone when the partial tiles are reduced.
//     8 waves on a 128 (co) x 128 (ci) tile of ONE tap; the two 4-wave halves of the workgroup alternate as loader of the next
//     32-pixel chunk (global loads one chunk ahead of the split + LDS hand-over, geometry one more chunk ahead), so that on
//     every SIMD one wave converts while its partner feeds the matrix pipe.  Split over the pixel range (grid = taps x tiles x
//     splits, the splits of one pixel range on one XCD so that the nine taps share the range through that L2), partial tiles
//     reduced in split order in fp64 (deterministic) straight into the PyTorch weight layout.
//     The same kernel is the weight gradient of a Linear (a 1 x 1 "convolution" over tokens): opp_wgrad_rows().
//   conv input gradient    = the forward implicit-GEMM kernel on the flipped / transposed weight (conv_flip_transpose_kernel),
//     stride 2 through a zero-inserted copy of dY (conv_dilate_kernel)                                    (api.hip drives it)
//   BatchNorm backward     two passes over the NHWC tensor: per-channel sums of dz and dz * xhat (fp64 partials, fixed order),
//     then dRaw = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)); the activation derivative (ReLU / LeakyReLU from
//     the sign of the saved output) and the residual branch's gradient (dz itself) are formed in the same passes.
//   bilinear x2 upsample   transposed gather with the forward's own tap arithmetic (upsample2x_backward_kernel).
#include <stdlib.h>

#include "opp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kOob = 0x80000000u;
constexpr int kS = 52;                 // floats per LDS row and 32-pixel chunk (48 + 4 pad, as gemm_mfma.hip)
constexpr int kBuf = 256 * kS;         // floats per LDS buffer: rows 0..127 = dY channels, 128..255 = X channels

struct WgradArgs {
  const float* dY = nullptr;   // [P][ldy]
  const float* X = nullptr;    // [pixels][ldx] (NHWC, pixel index from the geometry table)
  const int2* geo = nullptr;   // per output pixel: {input pixel index of window tap (0,0) (may be negative), valid-tap bits}; null = identity
  int ldy = 0, ldx = 0;
  int P = 0;                   // output pixels (reduction length)
  int Win = 0, ks = 1;
  int n_co_tiles = 1, n_ci_tiles = 1;
  int splits = 1, chunks_per_split = 0;
  float* part = nullptr;       // [splits][T][128][128], T = ks*ks*n_ci_tiles*n_co_tiles
  unsigned dy_bytes = 0, x_bytes = 0;
  int ablate = 0;              // tuning (OPP_WGRAD_ABLATE): 1 no split arithmetic, 2 no global loads, 3 no MFMAs, 4 no LDS hand-over -- wrong results
};

__device__ __forceinline__ unsigned b3_lvl(float a, float b, float& ra, float& rb) {
  const f32x2 t = {a, b};
  const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
  ra = a - __uint_as_float(p << 16);
  rb = b - __uint_as_float(p & 0xffff0000u);
  return p;
}

__global__ __launch_bounds__(512) void conv_wgrad_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;         // 4 x 2 waves, 32 (co) x 64 (ci) per wave
  const int set = wave >> 2;                        // loader half of the workgroup

  const int taps = a.ks * a.ks;
  const int T = taps * a.n_ci_tiles * a.n_co_tiles;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int s_local = jb / T, t = jb - s_local * T;
  const int split = s_local * 8 + xcd;
  if (split >= a.splits) return;
  const int tap = t % taps;
  const int rest = t / taps;
  const int ci_tile = rest % a.n_ci_tiles, co_tile = rest / a.n_ci_tiles;
  const int ky = tap / a.ks, kx = tap - ky * a.ks;
  const int tap_off = ky * a.Win + kx;
  const int total_chunks = (a.P + 31) >> 5;
  const int c_begin = split * a.chunks_per_split;
  const int n = max(0, min(a.chunks_per_split, total_chunks - c_begin));

  // loader role of this thread inside its half: 128 units for the dY rows, 128 for the X rows; a unit = 4 channels x 8 pixels
  const int lt = tid & 255;
  const int which = lt >> 7;
  const int u = lt & 127;
  const int cq = u & 31, pg = u >> 5;
  const int ch0 = (which == 0 ? co_tile : ci_tile) * 128 + cq * 4;
  const int ld = which == 0 ? a.ldy : a.ldx;
  const bool ch_ok = ch0 < ld;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(which == 0 ? a.dY : a.X), 0,
                                                                      which == 0 ? a.dy_bytes : a.x_bytes, 0x00020000);
  int2 geo[8] = {};   // geometry of the chunk this thread loads next (X rows with a table only)
  float4 v[8];        // its loaded values: pixel j, channels ch0 .. ch0 + 3
  const bool use_geo = (which == 1) && (a.geo != nullptr);

  auto load_geo = [&](int c) {       // c = chunk index inside this split
    if (use_geo && c < n) {
      const int4* gp = reinterpret_cast<const int4*>(a.geo + (size_t)(c_begin + c) * 32 + pg * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int4 g4 = gp[q];
        geo[2 * q] = make_int2(g4.x, g4.y);
        geo[2 * q + 1] = make_int2(g4.z, g4.w);
      }
    }
  };
  auto issue_loads = [&](int c) {
    if (a.ablate == 2) return;
    const int p0 = (c_begin + c) * 32 + pg * 8;
    const bool live = ch_ok && c < n;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned off;
      if (use_geo) {
        const bool ok = live && ((((unsigned)geo[j].y) >> tap) & 1u);
        off = ok ? (unsigned)((geo[j].x + tap_off) * ld + ch0) * 4u : kOob;
      } else {
        const int p = p0 + j;
        off = (live && p < a.P) ? (unsigned)(p * ld + ch0) * 4u : kOob;
      }
      const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
      v[j] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
    }
  };
  // registers -> LDS: per channel e the 8 pixels as [hi x8 | mid x8 | lo x8] (48 bytes) into row e * 32 + cq, pixel group pg
  auto split_store = [&](int buf) {
    if (a.ablate == 4) return;
    float* base = smem + buf * kBuf + (which * 128 + cq) * kS + pg * 12;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = e == 0 ? v[j].x : e == 1 ? v[j].y : e == 2 ? v[j].z : v[j].w;
      unsigned hi[4], mid[4], lo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float r0, r1, s0, s1, u0, u1;
        if (a.ablate == 1) {
          hi[k] = __float_as_uint(x[2 * k]);
          mid[k] = __float_as_uint(x[2 * k + 1]);
          lo[k] = hi[k];
          continue;
        }
        hi[k] = b3_lvl(x[2 * k], x[2 * k + 1], r0, r1);
        mid[k] = b3_lvl(r0, r1, s0, s1);
        lo[k] = b3_lvl(s0, s1, u0, u1);
      }
      float* row = base + e * 32 * kS;
      *reinterpret_cast<uint4*>(row) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      *reinterpret_cast<uint4*>(row + 4) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
      *reinterpret_cast<uint4*>(row + 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  auto as_b8 = [](const float4& f) { return *reinterpret_cast<const bf16x8*>(&f); };
  auto mfma_chunk = [&](int buf) {
    const float* As = smem + buf * kBuf + (wm * 32 + l31) * kS + half * 12;
    const float* Bs = smem + buf * kBuf + (128 + wn * 64 + l31) * kS + half * 12;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi: smallest terms first
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      float4 fa[3], fb[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[p] = *reinterpret_cast<const float4*>(As + st * 24 + p * 4);
        fb[0][p] = *reinterpret_cast<const float4*>(Bs + st * 24 + p * 4);
        fb[1][p] = *reinterpret_cast<const float4*>(Bs + 32 * kS + st * 24 + p * 4);
      }
      if (a.ablate == 3) {
        acc[0][0] += fa[0].x + fb[0][1].y + fb[1][2].z;
        continue;
      }
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b8(fa[PA[pr]]), as_b8(fb[j][PB[pr]]), acc[j], 0, 0, 0);
    }
  };

  // chunk c is handed to LDS by loader half (c & 1).  Prologue: half 0 brings chunk 0 in, half 1 has chunk 1 in flight.
  if (set == 0) {
    load_geo(0);
    issue_loads(0);
    split_store(0);
    load_geo(2);
  } else {
    load_geo(1);
    issue_loads(1);
  }
  __syncthreads();
  for (int c = 0; c < n; ++c) {
    if (set == ((c + 1) & 1)) {          // my chunk c + 1 is in registers: convert it while the other half runs MFMAs
      if (c + 1 < n) split_store((c + 1) & 1);
      load_geo(c + 3);
    } else {                             // my chunk c + 2: put its loads in flight under this chunk's MFMAs
      issue_loads(c + 2);
    }
    mfma_chunk(c & 1);
    __syncthreads();
  }

  // partial tile in the PERMUTED row / column order (lanes along columns: 128-byte stores); conv_wgrad_reduce_kernel un-permutes
  float* pt = a.part + ((size_t)split * T + t) * 16384;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      pt[row * 128 + wn * 64 + j * 32 + l31] = acc[j][r];
    }
}

// dW[co][ci][tap] (PyTorch [cout][cin][ks][ks]) = sum over the splits, in split order, in fp64
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, int splits, int taps, int n_ci_tiles,
                                                                int n_co_tiles, int cout, int cin, float* __restrict__ dW, int accumulate) {
  const int T = taps * n_ci_tiles * n_co_tiles;
  const size_t total = (size_t)T * 16384;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i >> 14);
    const int e = (int)(i & 16383);
    const int row = e >> 7, col = e & 127;
    const int tap = t % taps;
    const int rest = t / taps;
    const int ci_tile = rest % n_ci_tiles, co_tile = rest / n_ci_tiles;
    const int co = co_tile * 128 + 4 * (row & 31) + (row >> 5);
    const int ci = ci_tile * 128 + 4 * (col & 31) + (col >> 5);
    if (co >= cout || ci >= cin) continue;
    double s = 0.0;
    for (int k = 0; k < splits; ++k) s += (double)part[((size_t)k * T + t) * 16384 + e];
    float* o = dW + ((size_t)co * cin + ci) * taps + tap;
    *o = accumulate ? *o + (float)s : (float)s;
  }
}

// geometry of a convolution's output pixels for conv_wgrad_kernel: entries [P, P_pad) are invalid (mask 0)
__global__ void conv_geo_kernel(int B, int Ho, int Wo, int Hin, int Win, int ks, int stride, int pad, int P_pad, int2* __restrict__ geo) {
  const int P = B * Ho * Wo;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P_pad; p += gridDim.x * blockDim.x) {
    int2 g = make_int2(0, 0);
    if (p < P) {
      const int ox = p % Wo;
      const int t = p / Wo;
      const int oy = t % Ho;
      const int b = t / Ho;
      const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
      unsigned m = 0;
      for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx)
          if ((unsigned)(iy0 + ky) < (unsigned)Hin && (unsigned)(ix0 + kx) < (unsigned)Win) m |= 1u << (ky * ks + kx);
      g = make_int2((b * Hin + iy0) * Win + ix0, (int)m);
    }
    geo[p] = g;
  }
}

// wT[ci][co][ky][kx] = w[co][ci][ks-1-ky][ks-1-kx]: the weight of the convolution that computes the input gradient
__global__ void conv_flip_transpose_kernel(const float* __restrict__ w, int cout, int cin, int ks, float* __restrict__ out) {
  const int taps = ks * ks;
  const size_t total = (size_t)cout * cin * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const size_t r = i / taps;
    const int co = (int)(r % cout);
    const int ci = (int)(r / cout);
    out[i] = w[((size_t)co * cin + ci) * taps + (taps - 1 - tap)];
  }
}

// z[b][2 oy][2 ox][:] = dy[b][oy][ox][:], zero elsewhere (Hz = 2 Ho, Wz = 2 Wo): a stride-2 convolution's input gradient is the
// stride-1 input-gradient convolution over this zero-inserted tensor
__global__ void conv_dilate_kernel(const float4* __restrict__ dy, int B, int Ho, int Wo, int Q, float4* __restrict__ z) {
  const int Hz = 2 * Ho, Wz = 2 * Wo;
  const size_t total = (size_t)B * Hz * Wz * Q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const size_t p = i / Q;
    const int x = (int)(p % Wz);
    const size_t t = p / Wz;
    const int y = (int)(t % Hz);
    const int b = (int)(t / Hz);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(x & 1) && !(y & 1)) o = dy[(((size_t)b * Ho + (y >> 1)) * Wo + (x >> 1)) * Q + q];
    z[i] = o;
  }
}

// ---- BatchNorm + activation backward ---------------------------------------------------------------------------------
// forward (bn_train.hip): z = (raw - mean) * invstd * gamma + beta (+ res); y = act(z).
// dz = dy * act'(z): ReLU: y > 0 ? 1 : 0; LeakyReLU(0.01): y > 0 ? 1 : 0.01 (sign(y) = sign(z)); none: 1.
constexpr int kBnbRows = 256;

__device__ __forceinline__ float act_grad(float dy, float y, int act) {
  if (act == OPP_ACT_RELU) return y > 0.f ? dy : 0.f;
  if (act == OPP_ACT_LEAKY) return y > 0.f ? dy : 0.01f * dy;
  return dy;
}

// pass 1: per-channel partial sums of dz and dz * xhat over kBnbRows rows; part [blocks][2][ld] (fp64)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ raw,
                                                             int rows, int ld, int act, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, double* __restrict__ part) {
  __shared__ double red[4][64][8];
  const int q = threadIdx.x, lane_r = threadIdx.y, Q = ld >> 2;
  const int r0 = blockIdx.x * kBnbRows;
  const int r1 = min(rows, r0 + kBnbRows);
  double s[4] = {0, 0, 0, 0}, sx[4] = {0, 0, 0, 0};
  if (q < Q) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + q * 4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + q * 4);
    const float m[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
    for (int r = r0 + lane_r; r < r1; r += 4) {
      const size_t o = (size_t)r * ld + q * 4;
      const float4 g4 = *reinterpret_cast<const float4*>(dy + o);
      const float4 x4 = *reinterpret_cast<const float4*>(raw + o);
      float4 y4 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (act != OPP_ACT_NONE) y4 = *reinterpret_cast<const float4*>(y + o);
      const float g[4] = {g4.x, g4.y, g4.z, g4.w}, x[4] = {x4.x, x4.y, x4.z, x4.w}, yy[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dz = act_grad(g[e], yy[e], act);
        const float xh = (x[e] - m[e]) * iv[e];
        s[e] += (double)dz;
        sx[e] += (double)dz * (double)xh;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[lane_r][q][e] = s[e];
    red[lane_r][q][4 + e] = sx[e];
  }
  __syncthreads();
  if (lane_r == 0 && q < Q) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double tsum = red[0][q][e] + red[1][q][e] + red[2][q][e] + red[3][q][e];
      part[((size_t)blockIdx.x * 2 + (e >> 2)) * ld + q * 4 + (e & 3)] = tsum;
    }
  }
}

// fixed-order reduction of the block partials (as bn_finalize_kernel); coef [3][ld]: a = gamma * invstd, b = mean(dz), c = mean(dz * xhat);
// dgamma / dbeta [C] written (or accumulated)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, int blocks, int rows, int ld, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                              float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int accumulate) {
  __shared__ double red[16][16][2];
  const int cx = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  double s = 0.0, sx = 0.0;
  if (c < ld) {
    for (int b = sl; b < blocks; b += 16) {
      s += part[((size_t)b * 2) * ld + c];
      sx += part[((size_t)b * 2 + 1) * ld + c];
    }
  }
  red[sl][cx][0] = s;
  red[sl][cx][1] = sx;
  __syncthreads();
  if (sl != 0 || c >= ld) return;
  s = 0.0;
  sx = 0.0;
  for (int k = 0; k < 16; ++k) {
    s += red[k][cx][0];
    sx += red[k][cx][1];
  }
  const bool real = c < C;
  coef[c] = real ? gamma[c] * invstd[c] : 0.f;
  coef[ld + c] = real ? (float)(s / (double)rows) : 0.f;
  coef[2 * ld + c] = real ? (float)(sx / (double)rows) : 0.f;
  if (real) {
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)sx : (float)sx;
  }
}

// pass 2: draw = a * (dz - b - xhat * c) (zero on the padded channels); dres (optional) = dz.  draw may alias dy.
__global__ void bn_bwd_apply_kernel(const float* dy, const float* __restrict__ y, const float* __restrict__ raw, int rows, int ld,
                                    int act, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ coef,
                                    float* draw, float* dres) {
  const int Q = ld >> 2;
  const size_t total = (size_t)rows * Q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const float4 g4 = reinterpret_cast<const float4*>(dy)[i];
    const float4 x4 = reinterpret_cast<const float4*>(raw)[i];
    float4 y4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (act != OPP_ACT_NONE) y4 = reinterpret_cast<const float4*>(y)[i];
    const float g[4] = {g4.x, g4.y, g4.z, g4.w}, x[4] = {x4.x, x4.y, x4.z, x4.w}, yy[4] = {y4.x, y4.y, y4.z, y4.w};
    float o[4], d[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = q * 4 + e;
      const float dz = act_grad(g[e], yy[e], act);
      const float xh = (x[e] - mean[c]) * invstd[c];
      d[e] = dz;
      o[e] = coef[c] * (dz - coef[ld + c] - xh * coef[2 * ld + c]);
    }
    reinterpret_cast<float4*>(draw)[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (dres) reinterpret_cast<float4*>(dres)[i] = make_float4(d[0], d[1], d[2], d[3]);
  }
}

// ---- bilinear x2 (align_corners = True) upsample, transposed -----------------------------------------------------------
// forward (epilogue of opp_gemm_kernel, resnet.py:151,155): out[oy][ox] += wy0 wx0 r[y0][x0] + wy0 wx1 r[y0][x1] + wy1 wx0 r[y1][x0] + wy1 wx1 r[y1][x1]
// with sy = res_sy * oy, y0 = min(int(sy), Hr - 1), y1 = y0 + (y0 < Hr - 1), wy1 = clamp(sy - y0, 0, 1), wy0 = 1 - wy1 (same for x).
// backward: dr[y][x] = sum over the output pixels that touch (y, x), a gather over a small candidate window (no atomics, fixed order).
__device__ __forceinline__ void up_taps(int o, float s, int n_src, int& i0, int& i1, float& w0, float& w1) {
  const float sf = s * (float)o;
  i0 = (int)sf;
  if (i0 > n_src - 1) i0 = n_src - 1;
  i1 = i0 + (i0 < n_src - 1 ? 1 : 0);
  w1 = fminf(fmaxf(sf - (float)i0, 0.f), 1.f);
  w0 = 1.f - w1;
}

__global__ void upsample2x_backward_kernel(const float* __restrict__ g, int B, int Hr, int Wr, int ld, float sy, float sx,
                                           float* __restrict__ dr, int accumulate) {
  const int Ho = 2 * Hr, Wo = 2 * Wr, Q = ld >> 2;
  const size_t total = (size_t)B * Hr * Wr * Q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const size_t p = i / Q;
    const int x = (int)(p % Wr);
    const size_t t = p / Wr;
    const int y = (int)(t % Hr);
    const int b = (int)(t / Hr);
    // output rows oy with y0 == y or y1 == y lie in [2 y - 3, 2 y + 3] (sy = (Hr - 1) / (Ho - 1) in (0.33, 0.5])
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = max(0, 2 * y - 3); oy <= min(Ho - 1, 2 * y + 3); ++oy) {
      int y0, y1;
      float wy0, wy1;
      up_taps(oy, sy, Hr, y0, y1, wy0, wy1);
      const float wy = (y0 == y ? wy0 : 0.f) + (y1 == y ? wy1 : 0.f);
      if (y0 != y && y1 != y) continue;
      for (int ox = max(0, 2 * x - 3); ox <= min(Wo - 1, 2 * x + 3); ++ox) {
        int x0, x1;
        float wx0, wx1;
        up_taps(ox, sx, Wr, x0, x1, wx0, wx1);
        if (x0 != x && x1 != x) continue;
        const float wx = (x0 == x ? wx0 : 0.f) + (x1 == x ? wx1 : 0.f);
        const float4 gv = *reinterpret_cast<const float4*>(g + (((size_t)b * Ho + oy) * Wo + ox) * ld + q * 4);
        const float w = wy * wx;
        s.x += w * gv.x;
        s.y += w * gv.y;
        s.z += w * gv.z;
        s.w += w * gv.w;
      }
    }
    float4* o = reinterpret_cast<float4*>(dr) + i;
    if (accumulate) {
      const float4 old = *o;
      s.x += old.x;
      s.y += old.y;
      s.z += old.z;
      s.w += old.w;
    }
    *o = s;
  }
}

int grid_for(size_t total, int block = 256, int cap = 16384) {
  const size_t b = (total + block - 1) / block;
  return (int)(b < 1 ? 1 : (b < (size_t)cap ? b : (size_t)cap));
}

// splits of the pixel range: a multiple of 8 (one XCD per split) minimising rounds x (chunks per split + fixed cost)
void plan_splits(int T, int total_chunks, int& splits, int& cps) {
  long long best = -1;
  int best_s = 8;
  for (int s = 8; s <= 512; s += 8) {
    const int c = opp_cdiv(total_chunks, s);
    if (c < 4 && s > 8) break;
    const long long blocks = (long long)s * T;
    const long long rounds = (blocks + 255) / 256;
    const long long cost = rounds * (c + 8);
    if (best < 0 || cost < best) {
      best = cost;
      best_s = s;
    }
  }
  cps = opp_cdiv(total_chunks, best_s);
  splits = opp_cdiv(total_chunks, cps);
}

}  // namespace

size_t opp_conv_geo_entries(int P) { return ((size_t)(P + 31) / 32 + 4) * 32; }

int opp_conv_geo(int B, int Ho, int Wo, int Hin, int Win, int ks, int stride, int pad, void* geo, hipStream_t stream) {
  OPP_CHECK_ARG(geo && ks * ks <= 32, "conv_geo: bad argument");
  const int P_pad = (int)opp_conv_geo_entries(B * Ho * Wo);
  hipLaunchKernelGGL(conv_geo_kernel, dim3(grid_for(P_pad)), dim3(256), 0, stream, B, Ho, Wo, Hin, Win, ks, stride, pad, P_pad, static_cast<int2*>(geo));
  OPP_CHECK_LAUNCH("conv_geo_kernel");
  return OPP_OK;
}

size_t opp_conv_wgrad_ws_bytes(int P, int cout_pad, int cin_pad, int ks) {
  const int T = ks * ks * opp_cdiv(cin_pad, 128) * opp_cdiv(cout_pad, 128);
  int splits, cps;
  plan_splits(T, opp_cdiv(P, 32), splits, cps);
  return opp_align((size_t)splits * T * 16384 * sizeof(float));
}

// dW [cout][cin][ks][ks] = sum_p dY[p][co] X[geo(p) + tap][ci]; geo = null: X row p itself (Linear: dW[N][K] = dY^T X)
int opp_conv_wgrad(const float* dY, int ldy, const float* X, int ldx, size_t x_pixels, const void* geo, int P, int Win, int ks, int cout, int cin,
                   float* dW, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
  OPP_CHECK_ARG(dY && X && dW && ws && P > 0 && ldy % 4 == 0 && ldx % 4 == 0 && cout <= ldy && cin <= ldx && ks * ks <= 32, "conv_wgrad: bad argument");
  OPP_CHECK_ARG((size_t)P * ldy * 4 < (1ull << 31) && x_pixels * ldx * 4 < (1ull << 31), "conv_wgrad: operand too large for buffer addressing");
  WgradArgs a;
  a.dY = dY;
  a.X = X;
  a.geo = static_cast<const int2*>(geo);
  a.ldy = ldy;
  a.ldx = ldx;
  a.P = P;
  a.Win = Win;
  a.ks = ks;
  a.n_co_tiles = opp_cdiv(cout, 128);
  a.n_ci_tiles = opp_cdiv(cin, 128);
  const int T = ks * ks * a.n_ci_tiles * a.n_co_tiles;
  plan_splits(T, opp_cdiv(P, 32), a.splits, a.chunks_per_split);
  OPP_CHECK_ARG(ws_bytes >= (size_t)a.splits * T * 16384 * sizeof(float), "conv_wgrad: workspace too small");
  a.part = static_cast<float*>(ws);
  {
    static const int abl_env = getenv("OPP_WGRAD_ABLATE") ? atoi(getenv("OPP_WGRAD_ABLATE")) : 0;
    a.ablate = abl_env;
  }
  a.dy_bytes = (unsigned)((size_t)P * ldy * 4);
  a.x_bytes = (unsigned)(x_pixels * ldx * 4);
  const size_t lds = (size_t)2 * kBuf * sizeof(float);
  static OppLdsOnce lds_once;
  opp_lds_opt_in(reinterpret_cast<const void*>(conv_wgrad_kernel), lds, lds_once);
  const int blocks = opp_cdiv(a.splits, 8) * 8 * T;
  {
    OppProfScope prof(OPP_PROF_CONV_WGRAD, stream, 2.0 * (double)P * cout * cin * ks * ks);
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(blocks), dim3(512), lds, stream, a);
  }
  OPP_CHECK_LAUNCH("conv_wgrad_kernel");
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(grid_for((size_t)T * 16384)), dim3(256), 0, stream, a.part, a.splits, ks * ks, a.n_ci_tiles,
                     a.n_co_tiles, cout, cin, dW, accumulate);
  OPP_CHECK_LAUNCH("conv_wgrad_reduce_kernel");
  return OPP_OK;
}

int opp_conv_flip_transpose(const float* w, int cout, int cin, int ks, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(conv_flip_transpose_kernel, dim3(grid_for((size_t)cout * cin * ks * ks)), dim3(256), 0, stream, w, cout, cin, ks, out);
  OPP_CHECK_LAUNCH("conv_flip_transpose_kernel");
  return OPP_OK;
}

int opp_conv_dilate2(const float* dy, int B, int Ho, int Wo, int ld, float* z, hipStream_t stream) {
  OPP_CHECK_ARG(ld % 4 == 0, "conv_dilate2: ld %% 4");
  hipLaunchKernelGGL(conv_dilate_kernel, dim3(grid_for((size_t)B * Ho * Wo * ld)), dim3(256), 0, stream, reinterpret_cast<const float4*>(dy), B, Ho, Wo,
                     ld / 4, reinterpret_cast<float4*>(z));
  OPP_CHECK_LAUNCH("conv_dilate_kernel");
  return OPP_OK;
}

size_t opp_bn_bwd_scratch_bytes(int rows, int ld) {
  const size_t blocks = (size_t)opp_cdiv(rows, kBnbRows);
  return opp_align(blocks * 2 * ld * sizeof(double)) + opp_align((size_t)3 * ld * sizeof(float));
}

// BatchNorm (batch statistics) + activation backward over an NHWC tensor [rows][ld] with C real channels.
// dy: gradient of the block output y = act(bn(raw) [+ res]); y: that output (unused for act 0); raw: the convolution output;
// mean / invstd [ld]: the batch statistics of the forward.  draw (may alias dy) = gradient of raw; dres (optional) = gradient of
// the residual input (= dz); dgamma / dbeta [C] (optional).
int opp_bn_backward(const float* dy, const float* y, const float* raw, int rows, int ld, int C, int act, const float* gamma, const float* mean,
                    const float* invstd, float* draw, float* dres, float* dgamma, float* dbeta, int accumulate, void* scratch, hipStream_t stream) {
  OPP_CHECK_ARG(dy && raw && gamma && mean && invstd && draw && scratch && rows > 0 && ld % 4 == 0 && ld <= 256 && C <= ld && (act == OPP_ACT_NONE || y),
                "bn_backward: bad argument");
  const int blocks = opp_cdiv(rows, kBnbRows);
  double* part = static_cast<double*>(scratch);
  float* coef = reinterpret_cast<float*>(static_cast<char*>(scratch) + opp_align((size_t)blocks * 2 * ld * sizeof(double)));
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(blocks), dim3(64, 4), 0, stream, dy, y, raw, rows, ld, act, mean, invstd, part);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(opp_cdiv(ld, 16)), dim3(256), 0, stream, part, blocks, rows, ld, C, gamma, invstd, coef, dgamma, dbeta,
                     accumulate);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((size_t)rows * (ld / 4), 256, 8192)), dim3(256), 0, stream, dy, y, raw, rows, ld, act, mean, invstd,
                     coef, draw, dres);
  OPP_CHECK_LAUNCH("bn_backward kernels");
  return OPP_OK;
}

// dr [B][Hr][Wr][ld] (+)= transpose of the bilinear x2 upsample applied to g [B][2 Hr][2 Wr][ld]
int opp_upsample2x_backward(const float* g, int B, int Hr, int Wr, int ld, float* dr, int accumulate, hipStream_t stream) {
  OPP_CHECK_ARG(g && dr && ld % 4 == 0 && Hr > 0 && Wr > 0, "upsample2x_backward: bad argument");
  const int Ho = 2 * Hr, Wo = 2 * Wr;
  const float sy = Ho > 1 ? (float)(Hr - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(Wr - 1) / (float)(Wo - 1) : 0.f;
  hipLaunchKernelGGL(upsample2x_backward_kernel, dim3(grid_for((size_t)B * Hr * Wr * (ld / 4))), dim3(256), 0, stream, g, B, Hr, Wr, ld, sy, sx, dr, accumulate);
  OPP_CHECK_LAUNCH("upsample2x_backward_kernel");
  return OPP_OK;
}
