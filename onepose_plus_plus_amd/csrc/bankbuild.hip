// Per-object descriptor bank: mean 2D feature of every 3D point over its track.
//
// Reference: mean_descriptors_and_scores  src/sfm_utils/postprocess/feature_process.py:527-541
//   avg_descriptors[i] = np.mean(descriptors[start_i:end_i], axis=0)   (float32 rows, spans from cumsum(idxs))
// numpy reduces axis 0 of a C-contiguous float32 [n][D] array by adding the rows one after the other in float32
// and divides by n in float32; this kernel does exactly that, so the bank is bit-identical to the reference's.
// HBM-bound: every feature row is read once (R * D * 4 bytes), coalesced over the channel index.
#include "opp_internal.h"

namespace {

__global__ __launch_bounds__(256) void segmented_mean_kernel(const float* __restrict__ rows, int D,
                                                             const long long* __restrict__ offsets, int n_seg,
                                                             float* __restrict__ out) {
  const int seg = blockIdx.x;
  if (seg >= n_seg) return;
  const long long r0 = offsets[seg], r1 = offsets[seg + 1];
  const float cnt = (float)(r1 - r0);
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float acc = 0.f;
    for (long long r = r0; r < r1; ++r) acc += rows[(size_t)r * D + c];   // row order = numpy's axis-0 reduction order
    out[(size_t)seg * D + c] = acc / cnt;                                  // empty span: 0 / 0 = NaN like np.mean
  }
}

}  // namespace

extern "C" int opp_segmented_mean(const float* rows, int D, const long long* offsets, int n_seg, float* out, void* stream) {
  OPP_CHECK_ARG(rows && offsets && out && D > 0 && n_seg >= 0, "segmented_mean: bad argument");
  if (n_seg == 0) return OPP_OK;
  const int threads = D >= 256 ? 256 : (D + 63) / 64 * 64;
  hipLaunchKernelGGL(segmented_mean_kernel, dim3(n_seg), dim3(threads), 0, (hipStream_t)stream, rows, D, offsets, n_seg, out);
  OPP_CHECK_LAUNCH("segmented_mean_kernel");
  return OPP_OK;
}
