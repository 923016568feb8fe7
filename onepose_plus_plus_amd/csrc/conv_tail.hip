// Column tail of the 196-channel convolutions (backbone/resnet.py:88-124: block_dims = [128, 196, 256]) for gfx950.
//
// 196 output channels are 6 MFMA column sub-tiles of 32 plus FOUR columns.  Carried as a seventh sub-tile they cost the matrix pipe
// 32 columns (on the 128 x 256 / 128 x 128 tiles: 256 columns for 196 useful ones, 23 % of the MFMA work of eight convolutions is
// padding).  Since round 5 the implicit-GEMM kernel (gemm_mfma.hip, tile config 24: 128 x 192) computes columns [0, 192) and this
// kernel the rest of the 224-channel NHWC row: columns 192 .. 195 as plain fp32 FMA chains on the vector ALU -- 4 x K multiply-adds per
// pixel, 2 % of the layer's work, exact fp32 (not narrower than the six-product bf16 arithmetic of the body) -- and exact zeros in the 28
// padding channels.  It runs for every tile policy and batch size, so results never depend on the tile shape of the body.
//
// Structure: a workgroup (4 waves) owns a 16-wide tile of output pixels of one image (8 rows x 16 columns; or one
// whole patch of the match-driven fine branch: VALID convolution, 7 x 7 / 5 x 5 outputs); a lane owns PPL = 2 vertically adjacent
// pixels, a wave 8 of the 32 channels of the current channel group.  Per group the input footprint x 32 channels AND the group's
// weights (taps x 32 channels x 4 columns) are staged once in LDS (coalesced 16-byte loads, zeros for the padding halo; the next
// group's loads fly under the arithmetic).  Per tap column kx a lane reads the 8 channels of its PPL + ks - 1 input rows once and
// reuses them for the ks taps above each other; the four weights of a (tap, channel) are one broadcast ds_read_b128 shared by the PPL
// pixels -- 0.19 LDS reads per packed FMA, so the vector ALU, not the LDS pipe, is the limit.  (The first version fetched the weights
// with scalar loads: 32 KB of weights cycling through the 16 KB scalar cache missed on every tap and, since scalar loads return out of
// order, could not be pipelined -- ~1000 cycles per tap, 250 us per forward; profiles/r05_conv_tail_v1_trace.txt.)  The four channel
// slices are summed through LDS in slice order, then bias (folded BatchNorm), residual (same tensor or bilinear x2, align_corners=True,
// the arithmetic of the GEMM epilogue), activation.  Deterministic.
#include "opp_internal.h"

namespace {

constexpr int TW = 16;                // output tile width
constexpr int LS = 36;                // LDS floats per staged pixel (32 channels + 4 pad: ds_read_b128 lane groups spread over the banks)

struct TailArgs {
  const float* x;                     // NHWC [Bn][Hin][Win][cin_pad]
  const float* wt;                    // [cin_pad / 32][taps][32][4]: per channel group, the four tail columns of a (tap, channel) contiguous
  const float* bias;                  // [cout_pad] or null
  const float* R;                     // residual or null
  float* y;                           // [Bn * Hout * Wout][ldc]
  int Hin, Win, cin_pad, ks, stride, pad, Hout, Wout, ldc, n0, ncols;   // n0 = first tail column, ncols <= 4 real columns
  int res_mode, ldr, Hr, Wr, act;
  float res_sy, res_sx;
};

template <int KS, int STRIDE>
__global__ __launch_bounds__(256, 3) void conv_tail_kernel(const TailArgs a) {      // <= 168 registers: three workgroups (12 waves) per CU hide the LDS latency
  extern __shared__ __attribute__((aligned(16))) float sh[];
  constexpr int PPL = 2;                                          // output pixels per lane (vertically adjacent): 8 x 16 tiles -> enough workgroups at 128 x 128 pixels
  constexpr int TH = 4 * PPL;                                     // output tile height
  constexpr int IWX = (TW - 1) * STRIDE + KS, IWY = (TH - 1) * STRIDE + KS;   // staged footprint
  constexpr int NITEM = IWX * IWY * 8;                            // float4 pieces of one staged channel group
  constexpr int NL = (NITEM + 255) / 256;                         // ... per thread
  constexpr bool PREFETCH = NL <= 8;                              // the stride-2 footprints (18 pieces per thread) are staged without the register stage
  constexpr int TAPS = KS * KS;
  constexpr int NR = (PPL - 1) * STRIDE + KS;                     // input rows a lane touches
  constexpr int WOFF = IWX * IWY * LS;                            // the group's weights behind the staged pixels
  const int tid = threadIdx.x, lane = tid & 63;
  const int part = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = channel slice [8 part, 8 part + 8) of the group
  const int ntx = (a.Wout + TW - 1) / TW;
  const int ty0 = ((int)blockIdx.x / ntx) * TH, tx0 = ((int)blockIdx.x % ntx) * TW;
  const int b = blockIdx.y;
  const int iy0 = ty0 * STRIDE - a.pad, ix0 = tx0 * STRIDE - a.pad;
  const int lpx = lane & 15, lpy = (lane >> 4) * PPL;             // this lane's pixels: (ty0 + lpy + j, tx0 + lpx), j < PPL
  const float* xb = a.x + (size_t)b * a.Hin * a.Win * a.cin_pad;
  // this thread's pieces of a staged group (NL float4 of the footprint, <= 2 float4 of the weights), loaded one group ahead
  float4 pre[NL];
  float4 wpre0 = make_float4(0.f, 0.f, 0.f, 0.f), wpre1 = wpre0;
  auto fetch = [&](int g0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = tid + i * 256;
      const int p = e >> 3, q = e & 7;
      const int py = p / IWX, px = p - py * IWX;
      const int iy = iy0 + py, ix = ix0 + px;
      pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);                    // padding halo / past the footprint
      if (e < NITEM && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win)
        pre[i] = *reinterpret_cast<const float4*>(xb + ((size_t)iy * a.Win + ix) * a.cin_pad + q * 4 + g0);
    }
    const float4* wg = reinterpret_cast<const float4*>(a.wt) + (size_t)(g0 >> 5) * (TAPS * 32);
    if (tid < TAPS * 32) wpre0 = wg[tid];
    if (TAPS * 32 > 256 && tid + 256 < TAPS * 32) wpre1 = wg[tid + 256];
  };
  float acc[PPL][4];
#pragma unroll
  for (int j = 0; j < PPL; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[j][c] = 0.f;
  if (PREFETCH) fetch(0);
  for (int g0 = 0; g0 < a.cin_pad; g0 += 32) {
    __syncthreads();                                               // the previous group's tile is consumed
    if constexpr (PREFETCH) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int e = tid + i * 256;
        if (e < NITEM) *reinterpret_cast<float4*>(sh + (e >> 3) * LS + (e & 7) * 4) = pre[i];
      }
      if (tid < TAPS * 32) *reinterpret_cast<float4*>(sh + WOFF + tid * 4) = wpre0;
      if (TAPS * 32 > 256 && tid + 256 < TAPS * 32) *reinterpret_cast<float4*>(sh + WOFF + (tid + 256) * 4) = wpre1;
    } else {
      // large (stride-2) footprints: global -> LDS four pieces at a time, no register stage
      const float4* wg = reinterpret_cast<const float4*>(a.wt) + (size_t)(g0 >> 5) * (TAPS * 32);
      for (int e = tid; e < TAPS * 32; e += 256) *reinterpret_cast<float4*>(sh + WOFF + e * 4) = wg[e];
#pragma unroll 4
      for (int e = tid; e < NITEM; e += 256) {
        const int p = e >> 3, q = e & 7;
        const int py = p / IWX, px = p - py * IWX;
        const int iy = iy0 + py, ix = ix0 + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win)
          v = *reinterpret_cast<const float4*>(xb + ((size_t)iy * a.Win + ix) * a.cin_pad + q * 4 + g0);
        *reinterpret_cast<float4*>(sh + p * LS + q * 4) = v;
      }
    }
    __syncthreads();
    if (PREFETCH && g0 + 32 < a.cin_pad) fetch(g0 + 32);           // the next group's loads fly under this group's arithmetic
    const float* wsh = sh + WOFF + part * 32;                      // [tap][32 channels][4]: this wave's 8 channels
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      float xr[NR][8];                                             // the 8 channels of the NR input rows this lane's pixels see in column kx
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float* src = sh + ((lpy * STRIDE + r) * IWX + lpx * STRIDE + kx) * LS + part * 8;
        const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
        xr[r][0] = x0.x; xr[r][1] = x0.y; xr[r][2] = x0.z; xr[r][3] = x0.w;
        xr[r][4] = x1.x; xr[r][5] = x1.y; xr[r][6] = x1.z; xr[r][7] = x1.w;
      }
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        const float* wt_ = wsh + (ky * KS + kx) * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 w = *reinterpret_cast<const float4*>(wt_ + i * 4);   // same address in every lane: an LDS broadcast
#pragma unroll
          for (int j = 0; j < PPL; ++j) {
            const float xv = xr[j * STRIDE + ky][i];
            acc[j][0] = fmaf(xv, w.x, acc[j][0]);
            acc[j][1] = fmaf(xv, w.y, acc[j][1]);
            acc[j][2] = fmaf(xv, w.z, acc[j][2]);
            acc[j][3] = fmaf(xv, w.w, acc[j][3]);
          }
        }
      }
    }
  }
  __syncthreads();
  float4* red = reinterpret_cast<float4*>(sh);                     // [4 slices][TH * TW pixels]
  constexpr int NPIX = TH * TW;
#pragma unroll
  for (int j = 0; j < PPL; ++j) red[part * NPIX + (lpy + j) * TW + lpx] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  __syncthreads();
  // thread = output pixel of the tile (stride 2: the upper half of the workgroup writes the padding channels of the same pixels)
  const int pix = tid % NPIX, role = tid / NPIX;                   // NPIX = 256 (one role) or 128 (role 0: values + half the zeros, 1: the rest)
  const int oy = ty0 + pix / TW, ox = tx0 + pix % TW;
  if (oy >= a.Hout || ox >= a.Wout) return;
  const size_t row = ((size_t)b * a.Hout + oy) * a.Wout + ox;
  float* out = a.y + row * a.ldc + a.n0;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (role != 0) {                                                 // the padding channels of the 32-column line: exact zeros
#pragma unroll
    for (int q = 4; q < 8; ++q) *reinterpret_cast<float4*>(out + 4 * q) = z;
    return;
  }
#pragma unroll
  for (int q = 1; q < (NPIX == 256 ? 8 : 4); ++q) *reinterpret_cast<float4*>(out + 4 * q) = z;
  const float4 s0 = red[pix], s1 = red[NPIX + pix], s2 = red[2 * NPIX + pix], s3 = red[3 * NPIX + pix];
  float v[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
  if (a.bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += a.bias[a.n0 + c];          // (the bias vector is cout_pad long: zeros behind the real channels)
  }
  if (a.res_mode == OPP_RES_DIRECT) {
    const float4 r = *reinterpret_cast<const float4*>(a.R + row * a.ldr + a.n0);
    v[0] += r.x;
    v[1] += r.y;
    v[2] += r.z;
    v[3] += r.w;
  } else if (a.res_mode == OPP_RES_BILINEAR2X) {
    // bilinear x2 (align_corners=True) taps of the half-resolution residual (resnet.py:151,155): the statements of the GEMM epilogue
#pragma clang fp contract(off)
    const float sy = a.res_sy * (float)oy;
    const float sx = a.res_sx * (float)ox;
    int y0 = (int)sy;
    if (y0 > a.Hr - 1) y0 = a.Hr - 1;
    int x0 = (int)sx;
    if (x0 > a.Wr - 1) x0 = a.Wr - 1;
    const int y1 = y0 + (y0 < a.Hr - 1 ? 1 : 0);
    const int x1 = x0 + (x0 < a.Wr - 1 ? 1 : 0);
    const float wy1 = fminf(fmaxf(sy - (float)y0, 0.f), 1.f);
    const float wx1 = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
    const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
    const size_t pb = (size_t)b * a.Hr * a.Wr;
    const float4 t0 = *reinterpret_cast<const float4*>(a.R + (pb + (size_t)y0 * a.Wr + x0) * a.ldr + a.n0);
    const float4 t1 = *reinterpret_cast<const float4*>(a.R + (pb + (size_t)y0 * a.Wr + x1) * a.ldr + a.n0);
    const float4 t2 = *reinterpret_cast<const float4*>(a.R + (pb + (size_t)y1 * a.Wr + x0) * a.ldr + a.n0);
    const float4 t3 = *reinterpret_cast<const float4*>(a.R + (pb + (size_t)y1 * a.Wr + x1) * a.ldr + a.n0);
    const float a00[4] = {t0.x, t0.y, t0.z, t0.w}, a01[4] = {t1.x, t1.y, t1.z, t1.w};
    const float a10[4] = {t2.x, t2.y, t2.z, t2.w}, a11[4] = {t3.x, t3.y, t3.z, t3.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float top = wx0 * a00[c] + wx1 * a01[c];
      const float bot = wx0 * a10[c] + wx1 * a11[c];
      v[c] += wy0 * top + wy1 * bot;
    }
  }
  if (a.act == OPP_ACT_RELU) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = v[c] < 0.f ? 0.f : v[c];     // NaN-propagating like torch.relu
  } else if (a.act == OPP_ACT_LEAKY) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : 0.01f * v[c];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c >= a.ncols) v[c] = 0.f;                                  // fewer than four real tail columns: the rest is padding
  *reinterpret_cast<float4*>(out) = make_float4(v[0], v[1], v[2], v[3]);
}

// w [cout][cin][ks][ks] (PyTorch) -> wt [cin_pad / 32][taps][32][4]: wt[((g * taps + t) * 32 + (ci & 31)) * 4 + c] = w[n0 + c][ci][t] * scale[n0 + c]
// (zeros beyond cout / cin); the same fp32 product as opp_pack_conv's folded BatchNorm scale
__global__ __launch_bounds__(256) void pack_conv_tail_kernel(const float* __restrict__ w, const float* __restrict__ scale, int cout, int cin, int ks, int n0,
                                                             int cin_pad, float* __restrict__ wt) {
  const int taps = ks * ks;
  const int n = taps * cin_pad * 4;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int c = e & 3, cl = (e >> 2) & 31, t = ((e >> 7) % taps), ci = (e >> 7) / taps * 32 + cl;
    const int co = n0 + c;
    float v = 0.f;
    if (co < cout && ci < cin) {
      v = w[((size_t)co * cin + ci) * taps + t];
      if (scale) v *= scale[co];
    }
    wt[e] = v;
  }
}

}  // namespace

size_t opp_conv_tail_weight_floats(int cin_pad, int ks) { return (size_t)ks * ks * cin_pad * 4; }

int opp_pack_conv_tail(const float* w, const float* scale, int cout, int cin, int ks, int n0, int cin_pad, float* wt, hipStream_t stream) {
  OPP_CHECK_ARG(w && wt && cout > n0 && cout - n0 <= 4 && n0 % 32 == 0 && cin_pad % 32 == 0, "pack_conv_tail: bad shape (cout %d, first tail column %d)", cout, n0);
  const int n = ks * ks * cin_pad * 4;
  hipLaunchKernelGGL(pack_conv_tail_kernel, dim3(opp_cdiv(n, 256) < 256 ? opp_cdiv(n, 256) : 256), dim3(256), 0, stream, w, scale, cout, cin, ks, n0, cin_pad, wt);
  OPP_CHECK_LAUNCH("pack_conv_tail_kernel");
  return OPP_OK;
}

// the tail columns [n0, n0 + 32) of the convolution described by g (its body runs on opp_gemm_kernel with N = n_store = n0)
int opp_conv_tail(const OppGemm& g, const float* wt, int n0, int ncols, hipStream_t stream) {
  OPP_CHECK_ARG(g.conv && g.A0 && g.C && wt && n0 % 32 == 0 && ncols >= 1 && ncols <= 4 && g.ldc >= n0 + 32 && g.ldc % 4 == 0, "conv_tail: bad arguments");
  OPP_CHECK_ARG((g.ksize == 1 || g.ksize == 3) && (g.stride == 1 || g.stride == 2) && g.Cin % 32 == 0, "conv_tail: 1x1 / 3x3, stride 1 / 2 only");
  OPP_CHECK_ARG(g.Bn >= 1 && g.Bn <= 65535, "conv_tail: batch / patch count %d outside the grid limit", g.Bn);
  OPP_CHECK_ARG((reinterpret_cast<uintptr_t>(g.C) & 15) == 0 && (g.res_mode == OPP_RES_NONE || ((reinterpret_cast<uintptr_t>(g.R) & 15) == 0 && g.ldr % 4 == 0)),
                "conv_tail: output / residual must be 16-byte aligned");
  OPP_CHECK_ARG(g.act == OPP_ACT_NONE || g.act == OPP_ACT_RELU || g.act == OPP_ACT_LEAKY, "conv_tail: activation %d", g.act);
  TailArgs a;
  a.x = g.A0;
  a.wt = wt;
  a.bias = g.bias;
  a.R = g.R;
  a.y = g.C;
  a.Hin = g.Hin;
  a.Win = g.Win;
  a.cin_pad = g.Cin;
  a.ks = g.ksize;
  a.stride = g.stride;
  a.pad = g.pad;
  a.Hout = g.Hout;
  a.Wout = g.Wout;
  a.ldc = g.ldc;
  a.n0 = n0;
  a.ncols = ncols;
  a.res_mode = g.res_mode;
  a.ldr = g.ldr;
  a.Hr = g.Hr;
  a.Wr = g.Wr;
  a.res_sy = g.res_sy;
  a.res_sx = g.res_sx;
  a.act = g.act;
  const int th = 8;
  const int iwx = (TW - 1) * g.stride + g.ksize, iwy = (th - 1) * g.stride + g.ksize;
  const size_t stage = (size_t)iwx * iwy * LS + (size_t)g.ksize * g.ksize * 128, red = (size_t)4 * th * TW * 4;
  const size_t lds = (stage > red ? stage : red) * sizeof(float);
  const int tiles = opp_cdiv(g.Hout, th) * opp_cdiv(g.Wout, TW);
  OPP_CHECK_ARG((size_t)g.Hin * g.Win * g.Cin < (1ull << 31), "conv_tail: image too large for 32-bit offsets");
  // algorithmic bytes: the input read once, the 32-column line of every output pixel written (the VALU-bound 4 x K multiply-adds per pixel
  // are 2 % of the layer's work)
  OppProfScope prof(OPP_PROF_CONV_TAIL, stream, (double)g.Bn * ((double)g.Hin * g.Win * g.Cin * 4.0 + (double)g.Hout * g.Wout * 128.0));
  static OppLdsOnce once[4];               // per kernel variant and device (the stride-2 3 x 3 tile stages more than 64 KB)
  auto launch = [&](void (*k)(const TailArgs), OppLdsOnce& o) {
    opp_lds_opt_in(reinterpret_cast<const void*>(k), lds, o);
    hipLaunchKernelGGL(k, dim3(tiles, g.Bn), dim3(256), lds, stream, a);
  };
  if (g.ksize == 3 && g.stride == 1) launch(conv_tail_kernel<3, 1>, once[0]);
  else if (g.ksize == 3) launch(conv_tail_kernel<3, 2>, once[1]);
  else if (g.stride == 1) launch(conv_tail_kernel<1, 1>, once[2]);
  else launch(conv_tail_kernel<1, 2>, once[3]);
  OPP_CHECK_LAUNCH("conv_tail_kernel");
  return OPP_OK;
}
