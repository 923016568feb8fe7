// Backward of nn.Linear (bias-free, as every Linear of the two LoFTR transformers: loftr_module/transformer.py:26-47) on the
// MFMA GEMM kernel of gemm_mfma.hip -- the training step's parameter and input gradients of the transformer Linears
// (PL_OnePosePlus.training_step, src/lightning_model/OnePosePlus_lightning_model.py:54-81).
//   forward   Y[M][N]  = X[M][K] W[N][K]^T
//   dgrad     dX[M][K] = dY[M][N] W[N][K]          = GEMM(A = dY, "weights" = W^T [K][N])      -- W transposed once (small)
//   wgrad     dW[N][K] = dY^T[N][M] X[M][K]        = GEMM(A = dY^T [N][Mp], "weights" = X^T [K][Mp]), reduction over the TOKENS
// Both GEMM operands must be reduction-contiguous for the kernel ("TN"), hence the transposes; the token dimension is padded
// with zeros to a multiple of 32 x splits.  The weight gradient is a tiny output (N x K <= 768 x 512) under a huge reduction
// (M = 10^4 .. 10^5 tokens): it runs as SPLIT-K -- grid.y slices of the token range write partial products, a fixed-order
// reduction sums them (deterministic) -- so that the launch fills the chip instead of 4 .. 24 workgroups.
// Arithmetic: the model's (bf16x3 by default: exact operand triples, fp32 accumulate; or the exact-fp32 MFMA).
#include "opp_internal.h"

namespace {

// in [R][C] -> out [C][ldo], columns r >= R of out zero-filled up to ldo (ldo >= R): the reduction-contiguous operand
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C, int ldo) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;       // bx: column block of `in`, by: row block of `in`
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    tile[j][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;
    if (c < C && r < ldo) out[(size_t)c * ldo + r] = tile[tx][j];
  }
}

// out[i] = (accumulate ? out[i] : 0) + part[0][i] + part[1][i] + ... in split order (float4 granularity)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float4* __restrict__ part, int splits, size_t stride4, size_t n4,
                                                            float4* __restrict__ out, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = accumulate ? out[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < splits; ++s) {
      const float4 v = part[(size_t)s * stride4 + i];
      a.x += v.x;
      a.y += v.y;
      a.z += v.z;
      a.w += v.w;
    }
    out[i] = a;
  }
}

struct Plan {
  int Mp, splits, chunks_per_split;
  size_t off_wt, off_wts, off_dyt, off_xt, off_xts, off_part, total;   // floats
};

Plan make_plan(int M, int N, int K, int prec) {
  Plan p;
  const int tiles = opp_cdiv(N, 128) * opp_cdiv(K, 128);
  int splits = 256 / (tiles > 0 ? tiles : 1);               // one workgroup per CU (the 8-wave 128 x 128 tile owns a CU): one round,
                                                             // and half the partial-sum traffic of two half-length rounds
  const int chunks = opp_cdiv(M, 32);
  if (splits > chunks / 4) splits = chunks / 4;              // >= 4 chunks of 32 tokens per split
  if (splits < 1) splits = 1;
  p.splits = splits;
  p.chunks_per_split = opp_cdiv(chunks, splits);
  p.Mp = p.chunks_per_split * splits * 32;
  const size_t sf = prec == OPP_PREC_BF16X3 ? 3 : 2;         // split operands take 1.5 x the floats
  size_t o = 0;
  auto take = [&](size_t n) {
    const size_t at = o;
    o += (n + 63) / 64 * 64;
    return at;
  };
  p.off_wt = take((size_t)K * N);
  p.off_wts = take((size_t)K * N * sf / 2);
  if (prec == OPP_PREC_BF16X3) {
    // weight gradient straight from the token-major operands (conv_bwd.hip: register-transposing loader, no transposed / split copies)
    p.off_dyt = p.off_xt = p.off_xts = o;
    p.off_part = take(opp_conv_wgrad_ws_bytes(M, N, K, 1) / sizeof(float));
  } else {
    p.off_dyt = take((size_t)N * p.Mp);
    p.off_xt = take((size_t)K * p.Mp);
    p.off_xts = take((size_t)K * p.Mp * sf / 2);
    p.off_part = take((size_t)p.splits * N * K);
  }
  p.total = o;
  return p;
}

}  // namespace

size_t opp_linear_bwd_ws_bytes(int M, int N, int K, int prec) {
  if (M <= 0 || N <= 0 || K <= 0) return 256;
  return make_plan(M, N, K, prec).total * sizeof(float) + 256;
}

int opp_linear_bwd(const float* dY, const float* X, const float* W, int M, int N, int K, float* dX, float* dW, int accumulate_dw, int prec,
                   void* ws, size_t ws_bytes, hipStream_t stream) {
  OPP_CHECK_ARG(dY && ws && M > 0 && N > 0 && K > 0, "linear_backward: null / empty argument");
  OPP_CHECK_ARG(prec == OPP_PREC_FP32 || prec == OPP_PREC_BF16X3, "linear_backward: arithmetic must be 0 (fp32) or 2 (bf16x3)");
  OPP_CHECK_ARG(N % 32 == 0 && K % 32 == 0, "linear_backward: feature counts must be multiples of 32 (got N %d, K %d)", N, K);
  OPP_CHECK_ARG((!dX || W) && (!dW || X), "linear_backward: dX needs W, dW needs X");
  const Plan p = make_plan(M, N, K, prec);
  OPP_CHECK_ARG(ws_bytes >= p.total * sizeof(float), "linear_backward: workspace too small (%zu < %zu)", ws_bytes, p.total * sizeof(float));
  float* base = static_cast<float*>(ws);
  const bool b3 = prec == OPP_PREC_BF16X3;
  if (dX) {   // dX = dY W : the "weight" operand of the GEMM is W^T [K][N]
    float* wt = base + p.off_wt;
    float* wts = base + p.off_wts;
    OPP_TRY(opp_transpose(W, wt, 1, N, K, stream));
    if (b3) OPP_TRY(opp_b3_split(wt, wts, (size_t)K * N, stream));
    OppGemm g;
    g.prec = prec;
    g.A0 = dY;
    g.lda0 = N;
    g.ksplit = N;
    g.W = b3 ? wts : wt;
    g.ldw = b3 ? N / 2 * 3 : N;
    g.M = M;
    g.N = K;
    g.K = N;
    g.C = dX;
    g.ldc = K;
    g.n_store = K;
    OPP_TRY(opp_gemm_launch(g, stream));
  }
  if (dW && b3) {   // dW = dY^T X on the pixel-major weight-gradient kernel: a 1 x 1 "convolution" over M tokens
    OPP_TRY(opp_conv_wgrad(dY, N, X, K, (size_t)M, nullptr, M, 0, 1, N, K, dW, accumulate_dw, base + p.off_part, opp_conv_wgrad_ws_bytes(M, N, K, 1), stream));
  } else if (dW) {   // dW = dY^T X : reduction over the tokens, split-K
    float* dyt = base + p.off_dyt;
    float* xt = base + p.off_xt;
    float* xts = base + p.off_xts;
    float* part = base + p.off_part;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(opp_cdiv(N, 32), opp_cdiv(p.Mp, 32)), dim3(256), 0, stream, dY, dyt, M, N, p.Mp);
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(opp_cdiv(K, 32), opp_cdiv(p.Mp, 32)), dim3(256), 0, stream, X, xt, M, K, p.Mp);
    OPP_CHECK_LAUNCH("transpose_pad_kernel");
    if (b3) OPP_TRY(opp_b3_split(xt, xts, (size_t)K * p.Mp, stream));
    OppGemm g;
    g.prec = prec;
    g.A0 = dyt;
    g.lda0 = p.Mp;
    g.ksplit = p.Mp;
    g.W = b3 ? xts : xt;
    g.ldw = b3 ? p.Mp / 2 * 3 : p.Mp;
    g.M = N;
    g.N = K;
    g.K = p.Mp;
    g.C = part;
    g.ldc = K;
    g.n_store = K;
    g.k_splits = p.splits;
    g.k_chunks_per_split = p.chunks_per_split;
    g.split_stride = (size_t)N * K;
    g.alg_flops = 2.0 * (double)M * N * K;
    // 128 x 128 tiles (8 waves): the output is at most a few tiles, the parallelism comes from the splits
    OPP_TRY(opp_gemm_launch_cfg(g, b3 ? 25 : 0, stream));
    const size_t n4 = (size_t)N * K / 4;
    const int blocks = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(part), p.splits, (size_t)N * K / 4, n4,
                       reinterpret_cast<float4*>(dW), accumulate_dw);
    OPP_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return OPP_OK;
}
