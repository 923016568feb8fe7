// Query-image ingest: 8-bit grayscale -> bilinear resize (OpenCV INTER_LINEAR, 8-bit fixed-point
// path) -> fp32 / 255, on the device.  Replaces, per image, cv2.resize + grayscale2tensor of
//   read_grayscale      src/utils/data_io.py:34-69   (cv2.resize(image, (w_new, h_new)).astype('float32'))
//   grayscale2tensor    src/utils/data_io.py:105-106 (image / 255.)
// so that only the decoded 8-bit frame crosses PCIe.
//
// Arithmetic restated from OpenCV's resize (imgproc/resize.cpp; the reference pins opencv-python in
// requirements.txt, the library itself is not vendored -> parity unpinned, see oracle/ingest_oracle.py):
//   fx = (float)((dx + 0.5) * scale_x - 0.5); sx = floor(fx); fx -= sx; clamp to the image (fx = 0 there)
//   alpha = short(round_half_even(fx * 2048)), likewise beta for rows
//   horizontal pass in int (S[sx] * a0 + S[sx+1] * a1), vertical pass
//   dst = ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
//   exact 2x2 decimation (both factors == 2) takes the INTER_AREA fast path: (s00 + s01 + s10 + s11 + 2) >> 2
// HBM-bound byte work: one thread per output pixel, 4 source bytes in, 4 bytes out.
#include "opp_common.h"

namespace {

struct Tap {
  int s0, s1;
  int a0, a1;
};

__device__ __forceinline__ Tap make_tap(int d, double scale, int n_src) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) {
    f = 0.f;
    s = 0;
  }
  if (s >= n_src - 1) {
    f = 0.f;
    s = n_src - 1;
  }
  Tap t;
  t.s0 = s;
  t.s1 = s + 1 < n_src ? s + 1 : n_src - 1;
  t.a0 = (int)(short)__float2int_rn((1.f - f) * 2048.f);   // cvRound = round half to even
  t.a1 = (int)(short)__float2int_rn(f * 2048.f);
  return t;
}

__global__ __launch_bounds__(256) void ingest_u8_kernel(const unsigned char* __restrict__ src, int h, int w, int src_stride,
                                                        int h_new, int w_new, double scale_y, double scale_x, int area2,
                                                        float* __restrict__ dst, int dst_stride, unsigned char* __restrict__ dst_u8) {
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dx >= w_new || dy >= h_new) return;
  int v;
  if (area2) {
    const unsigned char* p = src + (size_t)(2 * dy) * src_stride + 2 * dx;
    v = ((int)p[0] + (int)p[1] + (int)p[src_stride] + (int)p[src_stride + 1] + 2) >> 2;
  } else {
    const Tap tx = make_tap(dx, scale_x, w), ty = make_tap(dy, scale_y, h);
    const unsigned char* r0p = src + (size_t)ty.s0 * src_stride;
    const unsigned char* r1p = src + (size_t)ty.s1 * src_stride;
    const int r0 = (int)r0p[tx.s0] * tx.a0 + (int)r0p[tx.s1] * tx.a1;
    const int r1 = (int)r1p[tx.s0] * tx.a0 + (int)r1p[tx.s1] * tx.a1;
    v = (((ty.a0 * (r0 >> 4)) >> 16) + ((ty.a1 * (r1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
  }
  if (dst) dst[(size_t)dy * dst_stride + dx] = (float)v / 255.f;
  if (dst_u8) dst_u8[(size_t)dy * w_new + dx] = (unsigned char)v;
}

}  // namespace

extern "C" int opp_image_ingest_u8(const unsigned char* src, int h, int w, int src_stride, int h_new, int w_new, float* dst,
                                   int dst_stride, unsigned char* dst_u8, void* stream) {
  OPP_CHECK_ARG(src && (dst || dst_u8), "image_ingest: null argument");
  OPP_CHECK_ARG(h > 0 && w > 0 && h_new > 0 && w_new > 0 && src_stride >= w && (!dst || dst_stride >= w_new),
                "image_ingest: bad sizes %dx%d -> %dx%d", h, w, h_new, w_new);
  // OpenCV: inv_scale = dsize / ssize ; scale = 1. / inv_scale  (double)
  const double inv_x = (double)w_new / (double)w, inv_y = (double)h_new / (double)h;
  const double scale_x = 1.0 / inv_x, scale_y = 1.0 / inv_y;
  const int area2 = (w == 2 * w_new && h == 2 * h_new) ? 1 : 0;
  hipLaunchKernelGGL(ingest_u8_kernel, dim3(opp_cdiv(w_new, 64), opp_cdiv(h_new, 4)), dim3(256), 0, (hipStream_t)stream, src, h, w,
                     src_stride, h_new, w_new, scale_y, scale_x, area2, dst, dst_stride, dst_u8);
  OPP_CHECK_LAUNCH("ingest_u8_kernel");
  return OPP_OK;
}
