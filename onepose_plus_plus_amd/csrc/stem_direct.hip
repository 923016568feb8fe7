// Direct 7x7 / stride 2 / pad 3 stem of the backbone (1 -> 128 channels) + folded BatchNorm + ReLU for the bf16x3 arithmetic
// (/root/reference/src/models/OnePosePlus/backbone/resnet.py:101-103, 143: conv1 -> bn1 -> relu).
//
// The launch-per-operator path builds the im2col matrix [H/2 * W/2][64] in memory (16.8 MB written and read back at 512 x 512)
// and runs a K = 64 GEMM over it.  Here a workgroup owns an 8 x 16 block of output pixels: the 21 x 37 input patch goes through
// LDS, the 128 im2col rows are formed there directly as bf16x3 operand rows (k = ky * 7 + kx < 49, zero up to 64), the 48 KB of
// pre-split weights stream L2 -> registers, four waves run the 128 x 128 x 64 block on the MFMA and write bias + ReLU rows.
// Same operand split, same six products per k16-step in the same order, K ascending, bias added to the finished sum: the
// result is bit-identical to opp_stem_im2col + opp_gemm_kernel<bf16x3>.
#include "enc_frag.h"

namespace {

constexpr int TY = 8, TX = 16, ROWS = TY * TX, KP = 64, COUT = 128;
constexpr int SA = a_stride_bytes(KP);                 // 400
constexpr int PH = 2 * TY + 5, PW = 2 * TX + 5, PWS = 40;
constexpr int WROW = KP * 6;                           // bytes per pre-split weight row (opp_pack_b3 layout)
constexpr int CS = COUT + 4;                           // staged output row stride (floats)
static_assert(ROWS * CS * 4 >= ROWS * SA, "the output tile reuses the operand space");

__global__ __launch_bounds__(256) void stem_direct_kernel(const float* __restrict__ img, int H, int W, int Ho, int Wo,
                                                          const char* __restrict__ wsplit, const float* __restrict__ bias,
                                                          float* __restrict__ out, int ldc, char* __restrict__ out3, int ld3) {
  // operand rows [128][400 B]; after the MFMAs the same space stages the output tile [128][CS floats] for 512-byte row stores
  __shared__ __attribute__((aligned(16))) char A[ROWS * CS * 4];
  __shared__ float patch[PH * PWS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int oy0 = blockIdx.y * TY, ox0 = blockIdx.x * TX;

  // weight fragments of this wave's 64 output channels, all four k16-steps: in flight while the operand rows are built
  u32x4 bq[2][4][3];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        bq[j][s][p] = *reinterpret_cast<const u32x4*>(wsplit + (size_t)(wn * 64 + j * 32 + l31) * WROW + s * 96 + half * 48 + p * 16);

  for (int i = tid; i < PH * PW; i += 256) {
    const int py = i / PW, px = i - py * PW;
    const int iy = 2 * oy0 - 3 + py, ix = 2 * ox0 - 3 + px;
    patch[py * PWS + px] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? img[(size_t)iy * W + ix] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < ROWS * (KP / 8) / 256; ++it) {
    const int item = tid + it * 256;
    const int r = item >> 3, g = item & 7;
    const int ly = r / TX, lx = r - ly * TX;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * g + e;
      const int ky = k / 7, kx = k - ky * 7;
      v[e] = k < 49 ? patch[(2 * ly + ky) * PWS + 2 * lx + kx] : 0.f;
    }
    split8_store(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), A + r * SA + g * 48);
  }
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const char* a_lane = A + (wm * 64 + l31) * SA + half * 48;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};   // A part of product pr (0 hi, 1 mid, 2 lo), smallest terms first
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    u32x4 af[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const u32x4*>(a_lane + i * 32 * SA + s * 96 + p * 16);
#pragma unroll
    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i][PA[pr]]), __builtin_bit_cast(bf16x8, bq[j][s][PB[pr]]),
                                                               acc[i][j], 0, 0, 0);
  }

  // bias (BatchNorm folded), ReLU, then through LDS so that a pixel's 128 channels leave as one 512-byte row (16 B per lane)
  __syncthreads();          // every wave is done reading the operand rows
  float* Ct = reinterpret_cast<float*>(A);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wn * 64 + j * 32 + l31;
    const float bv = bias[col];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[i][j][r] + bv;
        v = v < 0.f ? 0.f : v;   // NaN-propagating like torch.relu
        Ct[lr * CS + col] = v;
      }
  }
  __syncthreads();
  // eight channels per lane: 32 B of the fp32 row and, for the convolutions that take their input pre-split (gemm_mfma.hip, ASP), the
  // same eight values as one 48-byte [hi x8 | mid x8 | lo x8] group of the split row (row stride ld3 bytes)
#pragma unroll
  for (int it = 0; it < ROWS * (COUT / 8) / 256; ++it) {
    const int u = tid + it * 256;
    const int lr = u / (COUT / 8), c8 = (u - lr * (COUT / 8)) * 8;
    const int ly = lr / TX, lx = lr - ly * TX;
    const int oy = oy0 + ly, ox = ox0 + lx;
    if (oy < Ho && ox < Wo) {
      const float4 v0 = *reinterpret_cast<const float4*>(Ct + lr * CS + c8), v1 = *reinterpret_cast<const float4*>(Ct + lr * CS + c8 + 4);
      const size_t pix = (size_t)oy * Wo + ox;
      if (out != nullptr) {
        *reinterpret_cast<float4*>(out + pix * ldc + c8) = v0;
        *reinterpret_cast<float4*>(out + pix * ldc + c8 + 4) = v1;
      }
      if (out3 != nullptr) {
        u32x4 hi, mid, lo;
        split8(v0, v1, hi, mid, lo);
        char* d = out3 + pix * (size_t)ld3 + (c8 >> 3) * 48;
        *reinterpret_cast<u32x4*>(d) = hi;
        *reinterpret_cast<u32x4*>(d + 16) = mid;
        *reinterpret_cast<u32x4*>(d + 32) = lo;
      }
    }
  }
}

}  // namespace

bool opp_stem_direct_ok(int cout, int prec) { return cout == COUT && prec == OPP_PREC_BF16X3; }

int opp_stem_direct(const float* img, int H, int W, const float* wsplit, const float* bias, float* out, int ldc, hipStream_t stream, void* out3, int ld3) {
  OPP_CHECK_ARG(img && wsplit && bias && (out || out3) && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && ldc >= COUT, "stem_direct: bad argument");
  OPP_CHECK_ARG((reinterpret_cast<uintptr_t>(wsplit) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && ldc % 4 == 0,
                "stem_direct: weights / output must be 16-byte aligned, ldc a multiple of 4");
  OPP_CHECK_ARG(out3 == nullptr || ((reinterpret_cast<uintptr_t>(out3) & 15) == 0 && ld3 % 16 == 0 && ld3 >= COUT * 6), "stem_direct: bad split output");
  const int Ho = H / 2, Wo = W / 2;
  OPP_CHECK_ARG((size_t)Ho * Wo * ldc < (1ull << 31), "stem_direct: output too large for 32-bit indexing");
  hipLaunchKernelGGL(stem_direct_kernel, dim3(opp_cdiv(Wo, TX), opp_cdiv(Ho, TY)), dim3(256), 0, stream, img, H, W, Ho, Wo,
                     reinterpret_cast<const char*>(wsplit), bias, out, ldc, static_cast<char*>(out3), ld3);
  OPP_CHECK_LAUNCH("stem_direct_kernel");
  return OPP_OK;
}
