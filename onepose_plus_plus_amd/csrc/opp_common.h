// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the OnePose++
// 2D-3D matching hot path.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define OPP_OK 0
#define OPP_ERR_INVALID (-1)
#define OPP_ERR_UNSUPPORTED (-2)
#define OPP_ERR_LAUNCH (-3)
#define OPP_ERR_WORKSPACE (-4)

void opp_set_error(const char* fmt, ...);

#define OPP_CHECK_ARG(cond, ...)                  \
  do {                                            \
    if (!(cond)) {                                \
      opp_set_error(__VA_ARGS__);                 \
      return OPP_ERR_INVALID;                     \
    }                                             \
  } while (0)

#define OPP_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      opp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return OPP_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

#define OPP_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != OPP_OK) return rc__; \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize (kernels with more than 64 KB of dynamic LDS) is a PER-DEVICE attribute: one flag per
// (kernel instantiation, device), so that a process driving several GPUs opts in on each of them
struct OppLdsOnce {
  bool done[32] = {};
};
inline void opp_lds_opt_in(const void* kernel, size_t lds_bytes, OppLdsOnce& once) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool tracked = dev >= 0 && dev < 32;
  if (tracked && once.done[dev]) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (tracked) once.done[dev] = true;
}

static inline int opp_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t opp_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------
// GEMM / implicit-GEMM convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//   C[m][n] = epilogue( sum_k A[m][k] * W[n][k] )        ("TN": both operands K-contiguous)
// A is either a dense row-major matrix (optionally the concatenation [A0 | A1] along K) or
// the implicit im2col view of an NHWC activation tensor (3x3 / 1x1, stride 1 / 2).
// ---------------------------------------------------------------------------------------
enum { OPP_ACT_NONE = 0, OPP_ACT_RELU = 1, OPP_ACT_LEAKY = 2, OPP_ACT_QKV = 3 };
// operand arithmetic of a GEMM launch (always fp32 accumulate, fp32 in / fp32 out):
//   FP32    v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain
//   FP16X2  x ~ hi + lo fp16 (22 significant bits, fp16 exponent range), 3 x v_mfma_f32_32x32x16_f16
//   BF16X3  x = hi + mid + lo bf16 EXACTLY (24 bits, fp32 exponent range), 6 x v_mfma_f32_32x32x16_bf16
enum { OPP_PREC_FP32 = 0, OPP_PREC_FP16X2 = 1, OPP_PREC_BF16X3 = 2 };
// automatic tile choice: shortest launch (one forward at a time) or least CU time (several forwards in flight)
enum { OPP_TILES_LATENCY = 0, OPP_TILES_THROUGHPUT = 1 };
enum { OPP_RES_NONE = 0, OPP_RES_DIRECT = 1, OPP_RES_BILINEAR2X = 2 };

// K tail packing of a 3x3 convolution whose input has 32 n + (1..4) channels (the 196-channel stage of the backbone):
// the n full channel groups are walked tap by tap as usual, the last <= 4 channels of ALL taps are packed 8 taps to a
// 32-wide chunk (k = 4 * (tap % 8) + c), so that K is (9 n + 2) * 32 instead of 9 (n + 1) * 32.
// Returns n (the number of full groups) when the packing applies, else 0.
__host__ __device__ inline int opp_conv_tail_grp(int cin, int ks) { return (ks == 3 && cin > 32 && cin % 32 >= 1 && cin % 32 <= 4) ? cin / 32 : 0; }
// length of one packed weight row
__host__ __device__ inline int opp_conv_k(int cin, int ks) {
  const int tg = opp_conv_tail_grp(cin, ks);
  return tg ? (tg * ks * ks + (ks * ks + 7) / 8) * 32 : ks * ks * ((cin + 31) / 32 * 32);
}

struct OppGemm {
  // A operand -- dense mode
  const float* A0 = nullptr;
  const float* A1 = nullptr;
  int ksplit = 0;  // k <  ksplit -> A0[row*lda0 + k] ; k >= ksplit -> A1[row*lda1 + k-ksplit]
  int lda0 = 0, lda1 = 0;
  // A operand -- conv mode (A0 = NHWC input [B][Hin][Win][Cin], Cin % 32 == 0)
  int conv = 0;
  int Bn = 1, Hin = 0, Win = 0, Cin = 0, Hout = 0, Wout = 0, ksize = 1, stride = 1, pad = 0;
  int tail_grp = 0;  // > 0: K tail packing (opp_conv_tail_grp), the first tail_grp channel groups are full
  // B operand: weights [N][K] row-major (row stride ldw)
  const float* W = nullptr;
  int ldw = 0;
  int M = 0, N = 0, K = 0;  // K % 32 == 0 ; rows >= N of W are treated as zero
  // output
  float* C = nullptr;
  int ldc = 0;
  // bf16x3 convolutions (r06): C3 != null -- the epilogue also writes the output PRE-SPLIT (every 8 channels = 48 B [hi x8 | mid x8 | lo x8],
  // row stride ld3 BYTES) for the convolutions that consume it; C may then be null (no fp32 copy).  a_split: A0 is such a buffer.
  void* C3 = nullptr;
  int ld3 = 0;
  int a_split = 0;
  int n_store = 0;  // columns [0, n_store) are written (n_store >= N pads with act(0))
  int n_real = 0;   // > 0: weight rows >= n_real are known to be zero padding (channel counts padded to 32); lets the 128 x 224 tile skip rows >= 208
  const float* bias = nullptr;  // [n_store] or null
  // residual added before the activation
  int res_mode = OPP_RES_NONE;
  const float* R = nullptr;
  int ldr = 0, Hr = 0, Wr = 0;
  float res_sy = 0.f, res_sx = 0.f;
  // activation
  int act = OPP_ACT_NONE;
  int qk_cols = 0;     // OPP_ACT_QKV: columns < qk_cols get elu(x)+1, the rest x / seg_len(row)
  int split_row = 0;   // rows < split_row divide by s0, others by s1
  float s0 = 1.f, s1 = 1.f;
  // query_image_mask support (linear_attention.py:49-53, coarse_matching.py:108-114):
  //   row_mask  OPP_ACT_QKV only: rows < row_mask_rows (the image tokens) get phi(Q), phi(K) and V / S multiplied by
  //             row_mask[row] (0 / 1 floats)
  //   col_mask  score GEMM: -1e9 is added to every column whose col_mask[col] == 0, after the temperature scaling
  const float* row_mask = nullptr;
  int row_mask_rows = 0;
  const float* col_mask = nullptr;
  // generic output scaling: y = acc * out_mul / out_div (applied first; used by the score GEMM)
  float out_mul = 1.f, out_div = 1.f;
  // operand extents in bytes for the buffer descriptors (filled by the launcher)
  unsigned a0_bytes = 0, a1_bytes = 0, w_bytes = 0;
  // OPP_PREC_*.  Split modes: W is pre-split per 8 consecutive k (fp16x2 [hi x8 | lo x8] = 32 B, same footprint as
  // fp32; bf16x3 [hi x8 | mid x8 | lo x8] = 48 B, row stride ldw = 1.5 K floats), A is fp32 and split on the fly
  int prec = OPP_PREC_FP32;
  const float* h2_inv = nullptr;   // device scalar: 1 / (power-of-two scale applied to W before the split), or null
  // fp16x2 range guard: device int OR-ed with 1 when this launch produced a non-finite output (an activation
  // beyond the fp16 range makes its lo half infinite); null = unchecked.  Unused by the other arithmetics.
  int* nonfinite = nullptr;
  unsigned long long* dbg_ts = nullptr;   // -DOPP_TUNING builds (ABL 9): 4 shader-clock stamps per wave
  int tile_policy = OPP_TILES_LATENCY;   // OPP_TILES_*; only consulted when the launcher picks the tile itself
  int xcd_swizzle = 1;
  int vec_epilogue = 0;   // 16 B-per-lane epilogue allowed (alignment / divisibility checked by the launcher)
  // optional softmax statistics of the OUTPUT tile (score GEMM of the coarse matcher): per row
  // and per column (max, sum exp(v - max)) over this tile; [M][tiles_n] and [tiles_m][n_store]
  float* stat_rowmax = nullptr;
  float* stat_rowsum = nullptr;
  float* stat_colmax = nullptr;
  float* stat_colsum = nullptr;
  // optional LayerNorm of the OUTPUT rows fused into the epilogue (tile spans the whole row: n_store == BN):
  //   C = (ln_res ? ln_res : 0) + LN(acc) * ln_gamma + ln_beta      (transformer.py:87-88, :92-94)
  const float* ln_gamma = nullptr;
  const float* ln_beta = nullptr;
  const float* ln_res = nullptr;   // row stride ln_ldres
  int ln_ldres = 0;
  float ln_eps = 1e-5f;
  // algorithmic FLOPs of this launch (unpadded channel counts); 0 -> 2*M*N*K
  double alg_flops = 0.0;
  // dense split-K (weight gradients: K = tokens): k_splits > 1 -> grid.y = k_splits, split s reduces K chunks
  // [s * k_chunks_per_split, ...) and writes its partial product to C + s * split_stride (floats); plain epilogue only
  int k_splits = 1;
  int k_chunks_per_split = 0;
  size_t split_stride = 0;
  // conv mode, automatic tile choice: scratch for split-K partial products (>= 4 * M * ldc floats).  When given, a convolution whose
  // output is at most 64 tiles of 128 x 128 under a K of >= 32 chunks (the 3x3 convolutions of the 1/8-resolution stage at B = 1:
  // 4096 pixels x 256 channels x K = 2304) runs as 4 K slices on 8-wave 128 x 128 tiles -- 256 workgroups instead of 64 .. 256
  // four-wave ones -- and a fixed-order reduction applies bias / residual / activation.  The decision depends on the shape only.
  float* splitk_ws = nullptr;
  size_t splitk_ws_floats = 0;
  // -1: split by shape (above); 0: never; 1: always when the scratch allows.  The match-driven fine branch evaluates a layer on a few
  // patch rows and takes the decision the DENSE convolution of the same layer takes, so that both accumulate in the same order.
  int splitk_force = -1;
};

int opp_gemm_launch(const OppGemm& g, hipStream_t stream);
// picks a tile configuration; exposed for tests / tuning (cfg < 0 = automatic)
int opp_gemm_launch_cfg(const OppGemm& g, int cfg, hipStream_t stream);
// the tile configuration the launcher would pick for g by itself (host only, no launch); + 1000 when a bf16x3 convolution of this shape runs as 4 K slices
int opp_gemm_choose_tile(const OppGemm& g);
bool opp_conv_splitk_by_shape(long long M, int n_store, int K);   // gemm_mfma.hip: the launcher's shape rule for 4 K slices

#ifdef __HIPCC__
// wave64 sum on the DPP path (no LDS round trips): quad butterflies, half-row / row mirrors, then the two
// row broadcasts leave the total in lane 63; fixed order, every lane returns the same value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float opp_dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float opp_wave_sum_dpp(float v) {
  v = opp_dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = opp_dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v = opp_dpp_add<0x141, 0xF>(v);   // row_half_mirror
  v = opp_dpp_add<0x140, 0xF>(v);   // row_mirror: every lane of a 16-lane row holds the row sum
  v = opp_dpp_add<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
  v = opp_dpp_add<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}


// LayerNorm arithmetic shared by every kernel that normalises a token row (gemm_mfma.hip epilogue, attention.hip, enc_chain.hip,
// enc_layer64.hip): the fused multiply-adds are EXPLICIT, so the fused encoder chain and the launch-per-Linear
// path round identically whatever the compiler's contraction / vectorisation choices are in each file.
#ifdef __HIPCC__
__device__ __forceinline__ float opp_ln_sq_acc(float d, float acc) { return __builtin_fmaf(d, d, acc); }
__device__ __forceinline__ float opp_ln_affine(float v, float mean, float rstd, float gm, float bt) {
  return __builtin_fmaf((v - mean) * rstd, gm, bt);
}
#endif

#endif

