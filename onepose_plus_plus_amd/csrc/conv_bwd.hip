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
//     REGISTERS: a lane fetches 4 channels x 4 consecutive pixels (four 16-byte loads, each coalesced over the channels of
//     one pixel), splits every value exactly into bf16 hi + mid + lo (the bf16x3 arithmetic of gemm_mfma.hip: six
//     v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate, not narrower than fp32) and writes, per channel, its half of
//     the 48-byte [hi x8 | mid x8 | lo x8] group of 8 pixels with three ds_write_b64 -- the LDS image is exactly the fragment
//     layout of opp_gemm_kernel<bf16x3> (52-float rows: conflict-free ds_read_b128; the lane -> (row, half) map makes the
//     8-byte stores conflict-free too), there are no 16-bit scattered stores and no transposed copies of the activations in
//     memory.  LDS rows are channel-permuted (row = (c % 4) * 32 + c / 4) so that consecutive lanes write consecutive rows; the
//     permutation is undone when the partial tiles are reduced.
//     8 waves on a 128 (co) x 128 (ci) tile of ONE tap, 32 x 64 per wave.  The loader role is per WAVE (operand and 8-pixel
//     group in scalar registers: a per-lane buffer descriptor would turn every load into a waterfall loop) and identical for
//     all waves; the body between two barriers is branch-free: 24 MFMAs whose fragments were read half an interval earlier, and
//     in the 24 slots behind them the conversion of the chunk two ahead, the loads of the chunk four ahead and the geometry of
//     the chunk six ahead (three register sets, two LDS buffers, ONE barrier per 32-pixel chunk).  Split over the pixel range
//     (grid = taps x tiles x splits, the splits of one pixel range on one XCD so that the nine taps share the range through
//     that L2; the plan counts rounds per XCD), partial tiles reduced in a fixed order in fp64 (deterministic) straight into
//     the PyTorch weight layout.  Measured (profiles/r04_wgrad_*.txt): 174 TFLOP/s on layer1's shape with the GPU at its
//     1.3 kW limit (1.95 GHz); every structural variant of the loop lands within 2 % of that, so what is left is energy per
//     product (six MFMAs, the re-conversion of both operands for each of the nine taps), not scheduling.
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
constexpr int kDepth = 3;              // chunks of global loads in flight per thread (register sets)

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
};

__device__ __forceinline__ unsigned b3_lvl(float a, float b, float& ra, float& rb) {
  const f32x2 t = {a, b};
  const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
  ra = a - __uint_as_float(p << 16);
  rb = b - __uint_as_float(p & 0xffff0000u);
  return p;
}

template <int V>
struct WInt {
  static constexpr int value = V;
};

// ABL: timing-only ablations for tools/wgrad_ablate.sh (results wrong): 1 no split arithmetic, 2 no global loads, 3 no MFMAs,
// 4 no LDS hand-over, 5 no fragment reads, 6 no barrier in the loop, 7 every chunk re-loads chunk 0 (cache-resident operands)
template <int ABL>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;         // MFMA role: 4 x 2 waves, 32 (co) x 64 (ci) per wave

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

  // Loader role.  Per WAVE (scalar registers -- a per-lane buffer descriptor would turn every load into a waterfall loop):
  // the operand (waves 0-3 dY rows, 4-7 X rows) and the 8-pixel group pg of the 32-pixel chunk.  Per lane: 4 channels (cq)
  // and one half (ph) of the group: 4 pixels x 4 channels = four 16-byte loads and 8 pair conversions per chunk, the same for
  // every wave, so the body below has no roles and no branches.
  const int which = wave >> 2;
  const int pg = wave & 3;
  const int cq = (lane & 7) + 8 * (lane >> 4);      // a ds_write_b64 is served in groups of 16 contiguous lanes over 32 banks:
  const int ph = (lane >> 3) & 1;                   // 8 rows (52-float stride: 8 distinct bank quads) x 2 halves fill them
  const int ch0 = (which == 0 ? co_tile : ci_tile) * 128 + cq * 4;
  const int ld = which == 0 ? a.ldy : a.ldx;
  const bool ch_ok = ch0 < ld;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(which == 0 ? a.dY : a.X), 0,
                                                                      which == 0 ? a.dy_bytes : a.x_bytes, 0x00020000);
  const bool use_geo = (which == 1) && (a.geo != nullptr);
  // geometry through a descriptor of its own: 0 records when there is no table (every fetch returns 0, no branch)
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int2*>(use_geo ? a.geo : nullptr), 0,
                                                                       use_geo ? (unsigned)(((size_t)total_chunks * 32) * 8) : 0u, 0x00020000);
  u32x4 geo[2][2] = {};   // [set][half]: {origin, mask} of this lane's 4 pixels, two chunks' worth in flight
  float4 v[kDepth][4];    // kDepth chunks of loaded values in flight: [set][pixel], channels ch0 .. ch0 + 3

  auto load_geo = [&](int c, u32x4 (&g)[2]) {       // c = chunk index inside this split
    const int e0 = ((ABL == 7 ? 0 : (c_begin + c) * 32) + pg * 8 + ph * 4) * 8;
    g[0] = __builtin_amdgcn_raw_buffer_load_b128(grs, e0, 0, 0);
    g[1] = __builtin_amdgcn_raw_buffer_load_b128(grs, e0 + 16, 0, 0);
  };
  auto issue_load = [&](int c, int j, const u32x4 (&g)[2], float4& dst) {
    const int p = (ABL == 7 ? 0 : (c_begin + c) * 32) + pg * 8 + ph * 4 + j;
    const unsigned gx = j == 0 ? g[0].x : j == 1 ? g[0].z : j == 2 ? g[1].x : g[1].z;
    const unsigned gm = j == 0 ? g[0].y : j == 1 ? g[0].w : j == 2 ? g[1].y : g[1].w;
    const int pix = use_geo ? (int)gx + tap_off : p;                       // bitwise, not &&: no branches in the chunk body
    const unsigned ok = (unsigned)ch_ok & (unsigned)(c < n) & (use_geo ? (gm >> tap) & 1u : (unsigned)(p < a.P));
    const unsigned o = (unsigned)(pix * ld + ch0) * 4u;
    if (ABL == 2) return;
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? o : kOob), 0, 0);
    dst = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
  };
  // registers -> LDS in 24 steps (one per MFMA slot).  Step t = 3 * unit + level; unit = (channel e, pixel pair k) of this lane's
  // 4 channels x 4 pixels; level 0 / 1 / 2 peels the bf16 hi / mid / lo pair off the two values (exact: 3 x 8 significand bits).
  // After a channel's last step its three 8-byte pieces go into row e * 32 + cq at [hi x8 | mid x8 | lo x8] + this lane's half
  // (conflict-free: see cq / ph above).
  float* const wbase = smem + (which * 128 + cq) * kS + pg * 12 + ph * 2;
  unsigned hi[2], mid[2], lo[2];
  float cr0 = 0.f, cr1 = 0.f;   // residuals carried from one level to the next
  auto convert_step = [&](int t, const float4 (&src)[4], int buf) {
    const int unit = t / 3, lvl = t - 3 * unit;
    const int e = unit >> 1, k = unit & 1;
    if (ABL == 4) return;
    if (lvl == 0) {
      const float4& p0 = src[2 * k];
      const float4& p1 = src[2 * k + 1];
      const float x0 = e == 0 ? p0.x : e == 1 ? p0.y : e == 2 ? p0.z : p0.w;
      const float x1 = e == 0 ? p1.x : e == 1 ? p1.y : e == 2 ? p1.z : p1.w;
      if (ABL == 1) {
        hi[k] = __float_as_uint(x0);
        mid[k] = __float_as_uint(x1);
        lo[k] = hi[k];
      } else {
        hi[k] = b3_lvl(x0, x1, cr0, cr1);
      }
    } else if (lvl == 1) {
      if (ABL != 1) mid[k] = b3_lvl(cr0, cr1, cr0, cr1);
    } else {
      if (ABL != 1) {
        float u0, u1;
        lo[k] = b3_lvl(cr0, cr1, u0, u1);
      }
      if (k == 1) {
        float* row = wbase + buf * kBuf + e * 32 * kS;
        *reinterpret_cast<uint2*>(row) = make_uint2(hi[0], hi[1]);
        *reinterpret_cast<uint2*>(row + 4) = make_uint2(mid[0], mid[1]);
        *reinterpret_cast<uint2*>(row + 8) = make_uint2(lo[0], lo[1]);
      }
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // fragments of the two k16-steps of a chunk: [hi, mid, lo] of the wave's 32 dY rows and of its two 32-row X blocks
  float4 f0a[3], f0b[2][3], f1a[3], f1b[2][3];
  const float* const a_frag = smem + (wm * 32 + l31) * kS + half * 12;
  const float* const b_frag = smem + (128 + wn * 64 + l31) * kS + half * 12;
  auto read_frags = [&](int buf, int st, float4 (&fa)[3], float4 (&fb)[2][3]) {
    const float* As = a_frag + buf * kBuf + st * 24;
    const float* Bs = b_frag + buf * kBuf + st * 24;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      fa[p] = *reinterpret_cast<const float4*>(As + p * 4);
      fb[0][p] = *reinterpret_cast<const float4*>(Bs + p * 4);
      fb[1][p] = *reinterpret_cast<const float4*>(Bs + 32 * kS + p * 4);
    }
  };
  auto as_b8 = [](const float4& f) { return *reinterpret_cast<const bf16x8*>(&f); };
  // MFMAs first .. last - 1 of a k16-step (12 = 6 products x 2 column blocks), `between(i)` after the i-th: the slot in which
  // this wave's other work is issued while the matrix pipe runs
  auto mfma_range = [&](const float4 (&fa)[3], const float4 (&fb)[2][3], int first, int last, auto&& between) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi: smallest terms first
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int i = first; i < last; ++i) {
      const int pr = i >> 1, j = i & 1;
      if (ABL == 3) acc[j][i] += fa[PA[pr]].x + fb[j][PB[pr]].y;
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b8(fa[PA[pr]]), as_b8(fb[j][PB[pr]]), acc[j], 0, 0, 0);
      between(i);
    }
  };

  // Prologue: chunks 0 and 1 handed to LDS buffers 0 and 1, chunks 2 and 3 in flight, the geometry of chunks 4 and 5 on its way,
  // k16-step 0 of chunk 0 done.
#pragma unroll
  for (int d = 0; d < kDepth; ++d) {
    load_geo(d, geo[d & 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_load(d, j, geo[d & 1], v[d][j]);
  }
#pragma unroll
  for (int t = 0; t < 24; ++t) convert_step(t, v[0], 0);
  load_geo(3, geo[1]);
#pragma unroll
  for (int j = 0; j < 4; ++j) issue_load(3, j, geo[1], v[0][j]);
#pragma unroll
  for (int t = 0; t < 24; ++t) convert_step(t, v[1], 1);
  load_geo(4, geo[0]);
  load_geo(5, geo[1]);
  __syncthreads();
  read_frags(0, 0, f0a, f0b);
  read_frags(0, 1, f1a, f1b);
  mfma_range(f0a, f0b, 0, 12, [](int) {});

  // One interval = from the barrier of chunk c to the barrier of chunk c + 1; K = c % 6 at compile time gives the LDS buffer
  // (K & 1) and the register sets.  At the barrier every wave has ALL fragments of chunk c (buffer P) in registers and buffer
  // P ^ 1 (chunk c + 1) is complete, so between two barriers
  //   the matrix pipe runs k16-step 1 of chunk c, then k16-step 0 of chunk c + 1 -- 24 MFMAs whose operands were read from LDS
  //     half an interval earlier: no LDS latency and no load on either side of the barrier;
  //   the 24 slots behind those MFMAs carry, evenly, the conversion of chunk c + 2 (in registers since two intervals) into
  //     buffer P, which nobody reads any more, the four loads of chunk c + 4 and the geometry of chunk c + 6.
  // Neither the body nor the loop around six of them has a branch: the waitcnt pass counts the loads in flight instead of
  // draining them and nothing is sunk out of its slot into a successor block.  A chunk index past the end loads zeros
  // (out-of-range offsets): the host makes chunks_per_split a multiple of 6.
  auto interval = [&](auto kk, int c) {
    constexpr int K = decltype(kk)::value;
    constexpr int P = K & 1, CONV = (K + 2) % kDepth, LOAD = (K + 4) % kDepth, G = K & 1;
    if (ABL != 6) __syncthreads();
    if (ABL != 5) read_frags(P ^ 1, 0, f0a, f0b);
    __builtin_amdgcn_sched_barrier(0);
    auto slot = [&](int t) {
      convert_step(t, v[CONV], P);
      if (t % 6 == 2) issue_load(c + 4, t / 6, geo[G], v[LOAD][t / 6]);
      if (t == 21) load_geo(c + 6, geo[G]);
      __builtin_amdgcn_sched_barrier(0);
    };
    mfma_range(f1a, f1b, 0, 12, [&](int i) { slot(i); });
    if (ABL != 5) read_frags(P ^ 1, 1, f1a, f1b);
    __builtin_amdgcn_sched_barrier(0);
    mfma_range(f0a, f0b, 0, 12, [&](int i) { slot(12 + i); });
  };
  static_assert(kDepth == 3, "the six-interval body below is lcm(2 LDS buffers, kDepth register sets)");
  for (int c = 0; c < n; c += 6) {
    interval(WInt<0>{}, c);
    interval(WInt<1>{}, c + 1);
    interval(WInt<2>{}, c + 2);
    interval(WInt<3>{}, c + 3);
    interval(WInt<4>{}, c + 4);
    interval(WInt<5>{}, c + 5);
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

// dW[co][ci][tap] (PyTorch [cout][cin][ks][ks]) = sum over the splits in fp64 in a FIXED order: a workgroup owns 64 float4 of one
// partial tile; its four 64-thread groups each sum a contiguous quarter of the splits (in split order), then the quarter sums are
// added 0 + 1 + 2 + 3 -- the same tree on every run, four times shorter than one chain per output
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, int splits, int taps, int n_ci_tiles,
                                                                int n_co_tiles, int cout, int cin, float* __restrict__ dW, int accumulate) {
  __shared__ double sh[3][64][4];
  const int T = taps * n_ci_tiles * n_co_tiles;
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t e4 = (size_t)blockIdx.x * 64 + o;          // float4 index over [T][4096]
  const int t = (int)(e4 >> 12);
  const int e = (int)(e4 & 4095) * 4;                     // first of 4 consecutive columns of one row
  const int per = (splits + 3) >> 2;
  const int k0 = q * per, k1 = min(splits, k0 + per);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const float4* src = reinterpret_cast<const float4*>(part) + (size_t)t * 4096 + (e >> 2);
  for (int k = k0; k < k1; ++k) {
    const float4 v = src[(size_t)k * T * 4096];
    s0 += (double)v.x;
    s1 += (double)v.y;
    s2 += (double)v.z;
    s3 += (double)v.w;
  }
  if (q > 0) {
    sh[q - 1][o][0] = s0;
    sh[q - 1][o][1] = s1;
    sh[q - 1][o][2] = s2;
    sh[q - 1][o][3] = s3;
  }
  __syncthreads();
  if (q > 0) return;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    s0 += sh[r][o][0];
    s1 += sh[r][o][1];
    s2 += sh[r][o][2];
    s3 += sh[r][o][3];
  }
  const int row = e >> 7, col = e & 127;
  const int tap = t % taps;
  const int rest = t / taps;
  const int ci_tile = rest % n_ci_tiles, co_tile = rest / n_ci_tiles;
  const int co = co_tile * 128 + 4 * (row & 31) + (row >> 5);
  if (co >= cout) return;
  const double sv[4] = {s0, s1, s2, s3};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = col + i;
    const int ci = ci_tile * 128 + 4 * (c & 31) + (c >> 5);
    if (ci >= cin) continue;
    float* dst = dW + ((size_t)co * cin + ci) * taps + tap;
    *dst = accumulate ? *dst + (float)sv[i] : (float)sv[i];
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
// chunks per split (a multiple of the kernel's six-chunk body) and number of splits.  The T workgroups of a split sit on ONE XCD
// (32 CUs, one workgroup each: 106 KB of LDS), split k on XCD k % 8: cost = rounds of 32 workgroups on the fullest XCD times the
// length of one workgroup (+ 8 chunk-times of prologue, epilogue and reduce).
void plan_splits(int T, int total_chunks, int& splits, int& cps) {
  long long best = -1;
  int best_c = 6;
  for (int c = 6; c <= 6 * 4096; c += 6) {
    const int s = opp_cdiv(total_chunks, c);
    if (s > 512) continue;
    const long long rounds = ((long long)opp_cdiv(s, 8) * T + 31) / 32;
    const long long cost = rounds * (c + 8);
    if (best < 0 || cost < best) {
      best = cost;
      best_c = c;
    }
    if (s == 1) break;
  }
  cps = best_c;
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
  a.dy_bytes = (unsigned)((size_t)P * ldy * 4);
  a.x_bytes = (unsigned)(x_pixels * ldx * 4);
  const size_t lds = (size_t)2 * kBuf * sizeof(float);
  static OppLdsOnce lds_once;
  opp_lds_opt_in(reinterpret_cast<const void*>(conv_wgrad_kernel<0>), lds, lds_once);
  const int blocks = opp_cdiv(a.splits, 8) * 8 * T;
#ifdef OPP_TUNING
  static const int abl = getenv("OPP_WGRAD_ABLATE") ? atoi(getenv("OPP_WGRAD_ABLATE")) : 0;
  if (abl >= 1 && abl <= 7) {
    static OppLdsOnce abl_once[8];
    auto go = [&](auto k, int i) {
      opp_lds_opt_in(reinterpret_cast<const void*>(k), lds, abl_once[i]);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, stream, a);
    };
    switch (abl) {
      case 1: go(conv_wgrad_kernel<1>, 1); break;
      case 2: go(conv_wgrad_kernel<2>, 2); break;
      case 3: go(conv_wgrad_kernel<3>, 3); break;
      case 4: go(conv_wgrad_kernel<4>, 4); break;
      case 5: go(conv_wgrad_kernel<5>, 5); break;
      case 6: go(conv_wgrad_kernel<6>, 6); break;
      default: go(conv_wgrad_kernel<7>, 7); break;
    }
    return OPP_OK;
  }
#endif
  {
    OppProfScope prof(OPP_PROF_CONV_WGRAD, stream, 2.0 * (double)P * cout * cin * ks * ks);
    hipLaunchKernelGGL(conv_wgrad_kernel<0>, dim3(blocks), dim3(512), lds, stream, a);
  }
  OPP_CHECK_LAUNCH("conv_wgrad_kernel");
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(T * 64), dim3(256), 0, stream, a.part, a.splits, ks * ks, a.n_ci_tiles,
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
