// Backbone-side helper kernels: weight packing (BN folding, NHWC/tap-major layout, channel
// padding 196 -> 224), the 7x7/s2 stem im2col, and the positional-encoding add.
//
// Reference: ResNetFPN_8_2  src/models/OnePosePlus/backbone/resnet.py:85-164
//            PositionEncodingSine.forward  src/models/OnePosePlus/utils/position_encoding.py:37-42
#include "opp_common.h"

namespace {

// eval BatchNorm as y = x*scale + shift  (resnet.py:25-26; eps 1e-5)
__global__ void fold_bn_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var,
                               float eps, int c, int c_pad, float* __restrict__ scale,
                               float* __restrict__ shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c_pad) return;
  if (i < c) {
    const float s = gamma[i] / sqrtf(var[i] + eps);
    scale[i] = s;
    shift[i] = beta[i] - mean[i] * s;
  } else {
    scale[i] = 0.f;
    shift[i] = 0.f;
  }
}

// w [Cout][Cin][kh][kw] (PyTorch) -> packed [Cout_pad][K], K ordered (32-channel group, tap, channel
// in group): k = ((ci/32) * taps + tap) * 32 + ci % 32 -- the order the implicit-GEMM loader walks;
// optionally * scale[co]
__global__ void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ scale, int cout,
                                 int cin, int ks, int cout_pad, int cin_pad, float* __restrict__ out) {
  const int kpad = opp_conv_k(cin, ks);
  const int tail_grp = opp_conv_tail_grp(cin, ks);
  const size_t total = (size_t)cout_pad * kpad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i / kpad);
    const int r = (int)(i - (size_t)co * kpad);
    const int taps = ks * ks;
    int grp = r / (taps * 32);
    const int rem = r - grp * taps * 32;
    int tap = rem >> 5;
    int ci = grp * 32 + (rem & 31);
    if (tail_grp > 0 && grp >= tail_grp) {
      // K tail: k = tail_grp * taps * 32 + 32 t + 4 q + c  <->  tap 8 t + q, channel 32 tail_grp + c
      const int rt = r - tail_grp * taps * 32;
      tap = (rt >> 5) * 8 + ((rt & 31) >> 2);
      ci = tap < taps ? tail_grp * 32 + (rt & 3) : cin;
      grp = tail_grp;
    }
    float v = 0.f;
    if (co < cout && ci < cin) {
      v = w[((size_t)co * cin + ci) * ks * ks + tap];
      if (scale) v *= scale[co];
    }
    out[i] = v;
  }
}

// stem weights [Cout][1][7][7] -> [Cout][64] (k = ky*7+kx, zero padded), * scale[co]
__global__ void pack_stem_kernel(const float* __restrict__ w, const float* __restrict__ scale, int cout,
                                 float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cout * 64) return;
  const int co = i >> 6, k = i & 63;
  out[i] = k < 49 ? w[co * 49 + k] * (scale ? scale[co] : 1.f) : 0.f;
}

// im2col of the 1-channel image for the 7x7 stride-2 pad-3 stem (resnet.py:101,143):
// col[(b*Ho+oy)*Wo+ox][k] = img[b][2*oy+ky-3][2*ox+kx-3], k = ky*7+kx < 49, else 0.
__global__ void stem_im2col_kernel(const float* __restrict__ img, int B, int H, int W, int Ho, int Wo,
                                   float* __restrict__ col) {
  // one 16-byte store per lane (4 consecutive k), 32-bit index arithmetic (B * Ho * Wo * 16 < 2^31 checked by the launcher)
  const unsigned total = (unsigned)(B * Ho * Wo) * 16u;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k0 = (int)(i & 15u) * 4;
    const unsigned p = i >> 4;
    const int ox = (int)(p % (unsigned)Wo);
    const unsigned t = p / (unsigned)Wo;
    const int oy = (int)(t % (unsigned)Ho);
    const int b = (int)(t / (unsigned)Ho);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + e;
      const int ky = k / 7, kx = k - ky * 7;
      const int iy = 2 * oy + ky - 3, ix = 2 * ox + kx - 3;
      v[e] = (k < 49 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? img[((size_t)b * H + iy) * W + ix] : 0.f;
    }
    reinterpret_cast<float4*>(col)[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// fp16x2 pre-split of a GEMM weight matrix (K-contiguous rows, K % 8 == 0): every 8 consecutive
// fp32 values become 32 bytes [hi x8 | lo x8], hi = fp16_rtz(x), lo = fp16(x - hi)  (same footprint)
// sc[0] = 2^(14 - floor(log2(max|w|))) (so that max|w| * sc[0] is in [2^14, 2^15) and every weight
// above 2^-24 of the largest keeps a normal fp16 lo half), sc[1] = 1 / sc[0]; both exact powers of two
__global__ __launch_bounds__(1024) void h2_scale_kernel(const float* __restrict__ in, size_t n, float* __restrict__ sc) {
  __shared__ float red[16];
  float m = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 1024) {
    const float v = fabsf(in[i]);
    m = (v < INFINITY && v > m) ? v : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    int e = 14;
    if (m > 0.f) e = 14 - ilogbf(m);
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    sc[0] = ldexpf(1.f, e);
    sc[1] = ldexpf(1.f, -e);
  }
}

__global__ void h2_split_kernel(const float* __restrict__ in, uint4* __restrict__ out, size_t n8,
                                const float* __restrict__ sc) {
  const float mul = sc ? sc[0] : 1.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(in)[2 * i], b = reinterpret_cast<const float4*>(in)[2 * i + 1];
    const float v[8] = {a.x * mul, a.y * mul, a.z * mul, a.w * mul, b.x * mul, b.y * mul, b.z * mul, b.w * mul};
    unsigned hi[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const auto h = __builtin_amdgcn_cvt_pkrtz(v[2 * k], v[2 * k + 1]);
      const auto l = __builtin_amdgcn_cvt_pkrtz(v[2 * k] - (float)h[0], v[2 * k + 1] - (float)h[1]);
      hi[k] = __builtin_bit_cast(unsigned, h);
      lo[k] = __builtin_bit_cast(unsigned, l);
    }
    out[2 * i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[2 * i + 1] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// bf16x3 pre-split of a GEMM operand (K-contiguous rows, K % 8 == 0): every 8 consecutive fp32 values become
// 48 bytes [hi x8 | mid x8 | lo x8] with hi = bf16_rne(x), mid = bf16_rne(x - hi), lo = bf16(x - hi - mid):
// x = hi + mid + lo exactly (both residuals are exact fp32 subtractions, the last one has <= 8 significant bits)
__device__ __forceinline__ unsigned b3_level(float a, float b, float& ra, float& rb) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 t = {a, b};
  const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
  ra = a - __uint_as_float(p << 16);
  rb = b - __uint_as_float(p & 0xffff0000u);
  return p;
}
__global__ void b3_split_kernel(const float* __restrict__ in, uint4* __restrict__ out, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(in)[2 * i], b = reinterpret_cast<const float4*>(in)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned hi[4], mid[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float r0, r1, s0, s1, u0, u1;
      hi[k] = b3_level(v[2 * k], v[2 * k + 1], r0, r1);
      mid[k] = b3_level(r0, r1, s0, s1);
      lo[k] = b3_level(s0, s1, u0, u1);
    }
    out[3 * i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[3 * i + 1] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
    out[3 * i + 2] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// out[i] = a[i] + b[i]   (float4 granularity; sizes multiple of 4)
__global__ void add4_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = a[i], y = b[i];
    out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

// out[0 : n4a] = a + b ; out[n4a : n4a + n4c] = c   (token assembly: image tokens + sine encoding, then the cached point tokens)
__global__ void add4_cat_kernel(const float4* __restrict__ a, const float4* __restrict__ b, size_t n4a, const float4* __restrict__ c,
                                size_t n4c, float4* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4a + n4c; i += (size_t)gridDim.x * blockDim.x) {
    if (i < n4a) {
      const float4 x = a[i], y = b[i];
      out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    } else {
      out[i] = c[i - n4a];
    }
  }
}

// generic 2D transpose in[R][Cc] -> out[Cc][R] through an LDS tile (used for layout conversion of
// NCHW <-> NHWC test/interop buffers and the keypoint-MLP weights)
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const size_t zoff = (size_t)blockIdx.z * R * Cc;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    tile[j][tx] = (r < R && c < Cc) ? in[zoff + (size_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;
    if (c < Cc && r < R) out[zoff + (size_t)c * R + r] = tile[tx][j];
  }
}

}  // namespace

int opp_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                int c, int c_pad, float* scale, float* shift, hipStream_t stream) {
  hipLaunchKernelGGL(fold_bn_kernel, dim3(opp_cdiv(c_pad, 256)), dim3(256), 0, stream, gamma, beta, mean, var, eps, c, c_pad, scale, shift);
  OPP_CHECK_LAUNCH("fold_bn_kernel");
  return OPP_OK;
}

int opp_pack_conv(const float* w, const float* scale, int cout, int cin, int ks, int cout_pad, int cin_pad,
                  float* out, hipStream_t stream) {
  OPP_CHECK_ARG(cin_pad == (cin + 31) / 32 * 32, "pack_conv: cin_pad must be cin rounded up to 32");
  const size_t total = (size_t)cout_pad * opp_conv_k(cin, ks);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_conv_kernel, dim3(blocks), dim3(256), 0, stream, w, scale, cout, cin, ks, cout_pad, cin_pad, out);
  OPP_CHECK_LAUNCH("pack_conv_kernel");
  return OPP_OK;
}

int opp_pack_stem(const float* w, const float* scale, int cout, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(pack_stem_kernel, dim3(opp_cdiv(cout * 64, 256)), dim3(256), 0, stream, w, scale, cout, out);
  OPP_CHECK_LAUNCH("pack_stem_kernel");
  return OPP_OK;
}

int opp_stem_im2col(const float* img, int B, int H, int W, float* col, hipStream_t stream) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  OPP_CHECK_ARG((size_t)B * Ho * Wo * 16 < (1ull << 31), "stem im2col: image batch too large for 32-bit indexing");
  const size_t total = (size_t)B * Ho * Wo * 16;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(stem_im2col_kernel, dim3(blocks), dim3(256), 0, stream, img, B, H, W, Ho, Wo, col);
  OPP_CHECK_LAUNCH("stem_im2col_kernel");
  return OPP_OK;
}

int opp_h2_split(const float* in, float* out, size_t n, float* scale2, hipStream_t stream) {
  OPP_CHECK_ARG(n % 8 == 0 && in != out, "h2_split: n %% 8 != 0 or in-place");
  const size_t n8 = n / 8;
  if (scale2) {
    hipLaunchKernelGGL(h2_scale_kernel, dim3(1), dim3(1024), 0, stream, in, n, scale2);
    OPP_CHECK_LAUNCH("h2_scale_kernel");
  }
  const int blocks = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  hipLaunchKernelGGL(h2_split_kernel, dim3(blocks), dim3(256), 0, stream, in, reinterpret_cast<uint4*>(out), n8, scale2);
  OPP_CHECK_LAUNCH("h2_split_kernel");
  return OPP_OK;
}

int opp_b3_split(const float* in, float* out, size_t n, hipStream_t stream) {
  OPP_CHECK_ARG(n % 8 == 0 && in != out, "b3_split: n %% 8 != 0 or in-place");
  const size_t n8 = n / 8;
  const int blocks = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  hipLaunchKernelGGL(b3_split_kernel, dim3(blocks), dim3(256), 0, stream, in, reinterpret_cast<uint4*>(out), n8);
  OPP_CHECK_LAUNCH("b3_split_kernel");
  return OPP_OK;
}

int opp_add(const float* a, const float* b, float* out, size_t n, hipStream_t stream) {
  OPP_CHECK_ARG(n % 4 == 0, "add: n %% 4 != 0");
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(add4_kernel, dim3(blocks), dim3(256), 0, stream, (const float4*)a, (const float4*)b, (float4*)out, n4);
  OPP_CHECK_LAUNCH("add4_kernel");
  return OPP_OK;
}

int opp_add_cat(const float* a, const float* b, size_t na, const float* c, size_t nc, float* out, hipStream_t stream) {
  OPP_CHECK_ARG(na % 4 == 0 && nc % 4 == 0, "add_cat: sizes %% 4 != 0");
  const size_t n4 = (na + nc) / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(add4_cat_kernel, dim3(blocks), dim3(256), 0, stream, (const float4*)a, (const float4*)b, na / 4, (const float4*)c, nc / 4,
                     (float4*)out);
  OPP_CHECK_LAUNCH("add4_cat_kernel");
  return OPP_OK;
}

int opp_transpose(const float* in, float* out, int batch, int R, int Cc, hipStream_t stream) {
  hipLaunchKernelGGL(transpose_kernel, dim3(opp_cdiv(Cc, 32), opp_cdiv(R, 32), batch), dim3(256), 0, stream, in, out, R, Cc);
  OPP_CHECK_LAUNCH("transpose_kernel");
  return OPP_OK;
}
