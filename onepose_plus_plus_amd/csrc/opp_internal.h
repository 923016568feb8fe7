// Internal (non-exported) launchers shared between the .hip translation units.
#pragma once
#include "opp_common.h"

// profile.hip -- HIP-event timing of one armed kernel symbol (opp_profile_start / opp_profile_stop)
// symbol ids: GEMM / conv launches = kind * 256 + tile config (kind 0 dense, 1 implicit-GEMM conv, 2 coarse score
// GEMM with the fused softmax statistics); the bandwidth-bound kernels have fixed ids from 1000 (work = bytes)
enum {
  OPP_PROF_FIRST_NON_GEMM = 1000,
  OPP_PROF_LINATTN_KV = 1000,      // linattn_kv_mfma_kernel: the attention gather  sum_s phi(K_s)^T V_s
  OPP_PROF_LINATTN_APPLY = 1001,   // linattn_apply_pair_kernel
  OPP_PROF_CONF = 1002,            // conf_reg_kernel: dual-softmax product over the N x L score matrix
  OPP_PROF_FINE = 1003,            // fine stage kernel(s)
  OPP_PROF_LINATTN_SMALL = 1004,   // linattn_small_pair_kernel: fine-level attention, one workgroup per match
  OPP_PROF_FINE_GATHER = 1005,     // fine_gather_kernel
  OPP_PROF_FINE_HEAD = 1006,       // fine_head_kernel
  OPP_PROF_FOCAL_FWD = 1007,       // focal_fwd_kernel: coarse focal loss over the B x N x L confidence matrix
  OPP_PROF_FOCAL_BWD = 1008,       // focal_bwd_kernel: its gradient
  OPP_PROF_ENC_CHAIN = 1009,       // enc_chain_kernel: one encoder layer behind the QKV projection (work = FLOPs, MFMA-bound)
  OPP_PROF_SCORE_SWEEP1 = 1010,    // gemm_ss_kernel<STATS>: score tiles -> dual-softmax statistics only (work = FLOPs)
  OPP_PROF_SCORE_SWEEP2 = 1011,
  OPP_PROF_CONV_WGRAD = 1013,      // conv_wgrad_kernel: weight gradient of a convolution / Linear (work = FLOPs)
  OPP_PROF_CONV_SPLITK = 1014,     // opp_gemm_kernel<128,128,conv> over 4 K slices: the 3x3 convolutions of the 1/8-resolution stage (FLOPs)
  OPP_PROF_CONV_TAIL = 1017,       // conv_tail_kernel: columns 192 .. 223 of the 196-channel convolutions on the vector ALU (work = bytes)
  OPP_PROF_LINATTN_REDUCE = 1016,  // linattn_reduce_pair_kernel: fixed-order sum of the KV chunk partials (work = bytes)
  OPP_PROF_SPLITK_EPILOGUE = 1015, // splitk_epilogue_kernel: slices summed + bias / residual / activation (work = bytes)
  OPP_PROF_SCORE_SS = 1012,        // gemm_ss_kernel<STATS_STORE>: score GEMM on pre-split operands, statistics + score matrix written (FLOPs)    // gemm_ss_kernel<CONF>: score tiles recomputed -> confidence matrix written once (work = FLOPs)
};
inline int opp_prof_gemm_symbol(int tile_cfg, int kind) { return kind * 256 + tile_cfg; }
struct OppProfScope {
  OppProfScope(int symbol, hipStream_t stream, double work);
  ~OppProfScope();
  OppProfScope(const OppProfScope&) = delete;
  OppProfScope& operator=(const OppProfScope&) = delete;

 private:
  hipStream_t stream_;
  long long slot_ = -1;
};

// attention.hip
int opp_layernorm(const float* x, int ldx, const float* gamma, const float* beta, const float* res, int ldres,
                  float* out, int ldo, int rows, int C, float eps, hipStream_t stream);
int opp_linattn_chunks(int seg_len);
int opp_linattn_kv(const float* k, const float* v, int ld, int n_seg, int seg_len, int C, int D, float* kv_out,
                   float* ks_out, float* scratch, hipStream_t stream);
int opp_linattn_apply(const float* q, int ldq, const float* kv, const float* ks, float* out, int ldo, int n_seg,
                      int seg_len, int src_len, int C, int D, float eps, hipStream_t stream);
size_t opp_linattn_pair_scratch_floats(int len0, int len1);
int opp_linattn_kv_pair(const float* qkv, int ld, int len0, int len1, float* kv, float* ks, float* scratch,
                        hipStream_t stream);
// coarse focal loss of the training step (loss.hip)
size_t opp_focal_loss_ws_bytes(size_t n);
int opp_focal_loss_fwd(const float* conf, const short* gt, const float* weight, size_t n, float alpha, float gamma, double* sums,
                       void* ws, size_t ws_bytes, hipStream_t stream);
int opp_focal_loss_bwd(const float* conf, const short* gt, const float* weight, size_t n, float alpha, float gamma,
                       const float* scales, float* grad, hipStream_t stream);
// gt_kind 0 int16 / 1 fp32 / 2 uint8; weight: a full array OR the outer product mask0[b][i] * mask1[b][j] (N rows per sample, L columns)
int opp_focal_loss_fwd_ex(const float* conf, const void* gt, int gt_kind, const float* weight, const float* mask0, const float* mask1, int N, int L,
                          size_t n, float alpha, float gamma, double* sums, void* ws, size_t ws_bytes, hipStream_t stream);
int opp_focal_loss_bwd_ex(const float* conf, const void* gt, int gt_kind, const float* weight, const float* mask0, const float* mask1, int N, int L,
                          size_t n, float alpha, float gamma, const float* scales, float* grad, hipStream_t stream);
size_t opp_dual_softmax_bwd_ws_bytes(int B, int N, int L);
int opp_dual_softmax_bwd(const float* g, const float* sim, const float* lse_row, const float* lse_col, int B, int N, int L, float* ds,
                         void* ws, size_t ws_bytes, hipStream_t stream);
// many short segment pairs (fine level: 25 window tokens + 1 point token per match): KV, Ksum and the apply of both
// streams of one segment in one workgroup
bool opp_linattn_small_ok(int len0, int len1, int C, int D);
int opp_linattn_small_pair(const float* qkv, int ld, int n_seg, int len0, int len1, int cross, float* out, int ldo, int C, int D,
                           float eps, hipStream_t stream);
int opp_linattn_apply_pair(const float* qkv, int ld, const float* kv, const float* ks, int cross, float* out, int ldo,
                           int len0, int len1, float eps, hipStream_t stream);
// enc_chain.hip -- one LoFTREncoderLayer behind the Q/K/V projection in ONE launch (bf16x3 arithmetic):
// [attention apply ->] merge -> norm1 -> mlp.0(cat[x, message]) -> ReLU -> mlp.2 -> norm2 -> x + .
struct OppEncChain {
  int C = 256;                   // d_model: 256 (8 waves) or 128 (4 waves)
  const float* X = nullptr;      // tokens [len0 + len1][ldx]: stream 0 rows first
  int ldx = 0;
  float* out = nullptr;          // may alias X (every workgroup reads only the rows it writes)
  int ldo = 0;
  int len0 = 0, len1 = 0;        // rows per stream (apply = 0: only their sum matters)
  // attention message: either given ...
  const float* msg = nullptr;    // [rows][ldm]
  int ldm = 0;
  // ... or computed here from phi(Q) and the reduced KV / Ksum of the two streams (C = 256, 8 heads of 32)
  int apply = 0;
  const float* q = nullptr;      // phi(Q) rows [rows][ldq]
  int ldq = 0;
  const float* kv = nullptr;     // [2][C * 32]
  const float* ks = nullptr;     // [2][C]
  // enc_layer64 only, optional (per-object prefix of the coarse transformer, opp_object_prefix): phi(Q) rows of STREAM 1 from their own
  // buffer ([len1][ldq]) and the reduced KV / Ksum of stream 1 ([C * 32], [C]) instead of q + len0 rows / kv[1] / ks[1]
  const float* q1 = nullptr;
  const float* kv1 = nullptr;
  const float* ks1 = nullptr;
  // enc_layer64 only, optional (r05): the NEXT layer's q | k | v projection folded into this layer's tail -- the finished output tile is
  // in LDS as bf16x3 operands anyway.  wq_next: [3 C][C] fragment-major (opp_pack_frag_b3); qkv_out [len0 + len1][3 C] receives
  // phi(q) | phi(k) | v / S of every row (may be the buffer `q` points into: a workgroup reads and writes only its own rows), stream-1 rows
  // into qkv_out1 [len1][3 C] instead when given; qmask [len0] = query_image_mask of the image tokens (rows of stream 0) or null
  const void* wq_next = nullptr;
  float* qkv_out = nullptr;
  float* qkv_out1 = nullptr;
  const float* qmask = nullptr;
  int cross = 0;
  float eps_attn = 1e-6f;
  // fragment-major bf16x3 weights (opp_pack_frag_b3): merge [C][C], mlp.0 [2C][2C], mlp.2 [C][2C]
  const void* wm = nullptr;
  const void* w1 = nullptr;
  const void* w2 = nullptr;
  const float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;   // norm1 / norm2 affine
  float eps_ln = 1e-5f;
};
size_t opp_frag_b3_bytes(int N, int K);
int opp_pack_frag_b3(const float* w, int N, int K, void* out, hipStream_t stream);
bool opp_enc_chain_ok(int C, int nhead, bool apply);
int opp_enc_chain(const OppEncChain& a, hipStream_t stream);
// enc_layer64.hip -- the same layer on 64-token tiles (C = 256, apply fused): one round of 143 workgroups at 9096 tokens
int opp_enc_layer64(const OppEncChain& a, hipStream_t stream);
// gemm_ss.hip -- GEMM with BOTH operands pre-split (bf16x3, opp_pack_b3 layout) staged by LDS-DMA, 4-wave workgroups on
// 128 x 128 tiles, two workgroups per CU; used by the two-sweep coarse matcher (coarse_match.hip).
// v[m][n] = (sum_k A[m][k] B[n][k]) * out_mul / out_div  (+ -1e9 on the rows whose row_mask is 0)
enum { OPP_SS_STATS = 1, OPP_SS_CONF = 2, OPP_SS_STATS_STORE = 3 };   // 3: statistics + the score tile itself (out[col][row])
struct OppGemmSS {
  const void* A = nullptr;   // [M] rows of K split values, row stride lda BYTES (>= 6 K)
  const void* B = nullptr;   // [N] rows, row stride ldb BYTES
  int lda = 0, ldb = 0;
  int M = 0, N = 0, K = 0;   // K % 32 == 0
  int mode = OPP_SS_STATS;
  float out_mul = 1.f, out_div = 1.f;
  const float* row_mask = nullptr;          // [M] or null
  // OPP_SS_STATS: (max, sum exp(v - max)) partials per tile: rows [M][tiles_n], columns [tiles_m][N]
  float* stat_rowmax = nullptr;
  float* stat_rowsum = nullptr;
  float* stat_colmax = nullptr;
  float* stat_colsum = nullptr;
  // OPP_SS_CONF: merged statistics of the rows (rstat_*, [M]) and columns (cstat_*, [N]) in;
  //   c[m][n] = exp((v - rstat_max[m]) + (v - cstat_max[n])) * (rcp(rstat_sum[m]) * (1 / cstat_sum[n]))
  // stored TRANSPOSED: C[n * ldc + m].  Per tile: for every column its best c, the first row holding it and how many rows
  // hold it ([tiles_m][N]); for every row its max c ([M][tiles_n])
  const float* rstat_max = nullptr;
  const float* rstat_sum = nullptr;
  const float* cstat_max = nullptr;
  const float* cstat_sum = nullptr;
  float* C = nullptr;
  int ldc = 0;
  float* part_best = nullptr;
  int* part_arg = nullptr;
  int* part_ties = nullptr;
  float* part_rowmax = nullptr;
  int a_bytes = 0, b_bytes = 0, vec_store = 0;   // filled by the launcher
  int delay = 0;                                 // filled by the launcher (persistent kernel): cycles the second resident of a CU starts late
  unsigned long long* dbg_ts = nullptr;          // -DOPP_TUNING builds: 4 shader-clock stamps per wave
};
void opp_gemm_ss_debug_timestamps(void* buf, int mode);   // mode 0: both sweeps stamp, else only that OPP_SS_* mode
int opp_gemm_ss_tile_rows();
int opp_gemm_ss_tile_cols();
int opp_gemm_ss(const OppGemmSS& g, hipStream_t stream);
// linear_bwd.hip -- input / weight gradients of a bias-free Linear on the MFMA GEMM (transposed operand roles, split-K)
size_t opp_linear_bwd_ws_bytes(int M, int N, int K, int prec);
int opp_linear_bwd(const float* dY, const float* X, const float* W, int M, int N, int K, float* dX, float* dW, int accumulate_dw, int prec,
                   void* ws, size_t ws_bytes, hipStream_t stream);
// linattn_train.hip -- LinearAttention forward + backward of the training step (raw q, k, v [B][T][H][D], D = 32 / 16)
size_t opp_linattn_train_ws_bytes(int B, int L, int S, int H, int D);
int opp_linattn_train_fwd(const float* q, const float* k, const float* v, const float* q_mask, const float* kv_mask, int B, int L, int S, int H, int D,
                          float eps, float* out, float* kv, float* ks, void* ws, size_t ws_bytes, hipStream_t stream);
int opp_linattn_train_bwd(const float* q, const float* k, const float* v, const float* q_mask, const float* kv_mask, const float* kv, const float* ks,
                          const float* grad_out, int B, int L, int S, int H, int D, float eps, float* gq, float* gk, float* gv, void* ws,
                          size_t ws_bytes, hipStream_t stream);
// stem_direct.hip -- 7x7 / stride 2 stem + bias + ReLU without the im2col matrix (bf16x3, 128 output channels)
bool opp_stem_direct_ok(int cout, int prec);
// out3 != null: the same map once more in the pre-split activation layout (48 B per 8 channels, row stride ld3 bytes; OppGemm::C3)
int opp_stem_direct(const float* img, int H, int W, const float* wsplit, const float* bias, float* out, int ldc, hipStream_t stream, void* out3 = nullptr,
                    int ld3 = 0);
// conv_tail.hip -- the last (cout mod 32 <= 4) output columns of a convolution as fp32 FMA chains (the 196-channel layers: 192 columns on the
// MFMA kernel + 4 here instead of 224 / 256 padded MFMA columns)
size_t opp_conv_tail_weight_floats(int cin_pad, int ks);
int opp_pack_conv_tail(const float* w, const float* scale, int cout, int cin, int ks, int n0, int cin_pad, float* wt, hipStream_t stream);
int opp_conv_tail(const OppGemm& g, const float* wt, int n0, int ncols, hipStream_t stream);
// backbone.hip
int opp_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int c,
                int c_pad, float* scale, float* shift, hipStream_t stream);
int opp_pack_conv(const float* w, const float* scale, int cout, int cin, int ks, int cout_pad, int cin_pad,
                  float* out, hipStream_t stream);
int opp_pack_stem(const float* w, const float* scale, int cout, float* out, hipStream_t stream);
int opp_stem_im2col(const float* img, int B, int H, int W, float* col, hipStream_t stream);
// fp16x2 pre-split; scale2 (device, 2 floats: scale, 1/scale) null = unscaled
int opp_h2_split(const float* in, float* out, size_t n, float* scale2, hipStream_t stream);
// bf16x3 pre-split: out holds 1.5 n floats (48 B per 8 values)
int opp_b3_split(const float* in, float* out, size_t n, hipStream_t stream);
int opp_add(const float* a, const float* b, float* out, size_t n, hipStream_t stream);
int opp_add_cat(const float* a, const float* b, size_t na, const float* c, size_t nc, float* out, hipStream_t stream);
int opp_transpose(const float* in, float* out, int batch, int R, int Cc, hipStream_t stream);
// bn_train.hip: training-mode BatchNorm (batch statistics) over an NHWC tensor [rows][ld] with C real channels:
// out = act((y - mean_batch) * invstd_batch * gamma + beta (+ res)); stat_out [2][C] = batch mean, unbiased variance
size_t opp_bn_train_scratch_bytes(int rows, int ld);
int opp_bn_train(const float* y, int rows, int ld, int C, const float* gamma, const float* beta, float eps, const float* res,
                 int act, float* out, float* stat_out, void* scratch, hipStream_t stream, float* save_mean = nullptr, float* save_invstd = nullptr);
// conv_bwd.hip -- backward of the backbone convolutions / BatchNorm / bilinear upsample for the training step
size_t opp_conv_geo_entries(int P);     // int2 entries of the geometry table of P output pixels
int opp_conv_geo(int B, int Ho, int Wo, int Hin, int Win, int ks, int stride, int pad, void* geo, hipStream_t stream);
size_t opp_conv_wgrad_ws_bytes(int P, int cout_pad, int cin_pad, int ks);
// dW [cout][cin][ks][ks] (+)= sum_p dY[p][co] X[geo(p) + tap][ci]; geo = null: X row p itself (ks = 1: a Linear's weight gradient)
int opp_conv_wgrad(const float* dY, int ldy, const float* X, int ldx, size_t x_pixels, const void* geo, int P, int Win, int ks, int cout, int cin,
                   float* dW, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);
int opp_conv_flip_transpose(const float* w, int cout, int cin, int ks, float* out, hipStream_t stream);
int opp_conv_dilate2(const float* dy, int B, int Ho, int Wo, int ld, float* z, hipStream_t stream);
size_t opp_bn_bwd_scratch_bytes(int rows, int ld);
int opp_bn_backward(const float* dy, const float* y, const float* raw, int rows, int ld, int C, int act, const float* gamma, const float* mean,
                    const float* invstd, float* draw, float* dres, float* dgamma, float* dbeta, int accumulate, void* scratch, hipStream_t stream);
int opp_upsample2x_backward(const float* g, int B, int Hr, int Wr, int ld, float* dr, int accumulate, hipStream_t stream);
// train_misc.hip -- LayerNorm with saved statistics + backward, log-sum-exps of the score matrix, fine windows of a batch
int opp_ln_forward(const float* x, const float* gamma, const float* beta, const float* res, int rows, int C, float eps, float* y, float* mean,
                   float* rstd, hipStream_t stream);
size_t opp_ln_backward_ws_bytes(int rows, int C);
int opp_ln_backward(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, int rows, int C, float* dx,
                    float* dgamma, float* dbeta, void* ws, size_t ws_bytes, hipStream_t stream);
size_t opp_lse_ws_bytes(int B, int N, int L);
int opp_dual_softmax_lse(const float* S, int B, int N, int L, float* lse_row, float* lse_col, float* conf, void* ws, size_t ws_bytes,
                         hipStream_t stream);
int opp_fine_gather_batch(const float* feat, int Hf, int Wf, int C, const long long* b_ids, const long long* j_ids, int M, int wc, int stride, int Wwin,
                          float* win, hipStream_t stream);
int opp_fine_scatter_batch(const float* gwin, int B, int Hf, int Wf, int C, const long long* b_ids, const long long* j_ids, int M, int wc, int stride,
                           int Wwin, float* dfeat, hipStream_t stream);
int opp_assignmatrix(const float* kp2d_coarse, const float* kp2d_fine, int n2d, const long long* assign, int k, int N, int L, int w_c, float scale_x,
                     float scale_y, float coarse_scale, short* conf_gt, float* fine_loc_gt, long long* keys, int* status, hipStream_t stream);
// kpt.hip
int opp_kpt_stats(const float* kpts, int n, float* stats, hipStream_t stream);
int opp_kpt_encode(const float* kpts, const float* stats, const float* bank, int n, const float* const* wt,
                   const float* const* bias, float* tokens, int ldo, hipStream_t stream);
int opp_bank_transpose(const float* bank, int n, int C, float* tokens, int ldo, hipStream_t stream);
// coarse_match.hip
size_t opp_coarse_match_scratch_floats(int N, int L);
size_t opp_coarse_match_stats_floats(int N, int L);
// two-sweep variant (bf16x3): f3s [N][C], f2s [L][C] pre-split (opp_pack_b3); conf [N][L] is written once
int opp_dual_softmax_two_sweep(const void* f3s, const void* f2s, int C, int N, int L, int wc, float out_mul, float out_div,
                               const float* col_mask, float thr, int border, const float* kpts, float base_scale, const float* qscale,
                               float* conf, float* stats, float* scratch, long long* i_ids, long long* j_ids, float* mconf,
                               float* mkpts_c, float* mkpts_3d, int* count, hipStream_t stream);
// single sweep on the split-operand GEMM: score matrix + statistics by gemm_ss, then the in-place passes of opp_dual_softmax_select
int opp_dual_softmax_ss_single(const void* f3s, const void* f2s, int C, int N, int L, int wc, float out_mul, float out_div,
                               const float* col_mask, float thr, int border, const float* kpts, float base_scale, const float* qscale,
                               float* conf, float* stats, float* scratch, long long* i_ids, long long* j_ids, float* mconf,
                               float* mkpts_c, float* mkpts_3d, int* count, hipStream_t stream);
int opp_dual_softmax_select(float* S, int N, int L, int wc, float thr, int border, const float* kpts,
                            float base_scale, const float* qscale, const float* stats, int stats_bm, float* scratch, long long* i_ids, long long* j_ids, float* mconf,
                            float* mkpts_c, float* mkpts_3d, int* count, hipStream_t stream);
// fine.hip
int opp_fine_patch_gather(const float* x1, int Hf, int Wf, int c1, const float* x2o, int c2, const long long* j_ids, int M, int wc, int stride, int org, int P,
                          float* xa, float* up, hipStream_t stream);
int opp_patch_zero_oob(float* buf, int ld, const long long* j_ids, int M, int wc, int stride, int org, int P, int Hf, int Wf, hipStream_t stream);
int opp_fine_points_gather(const float* bank, int n_points, const long long* i_ids, int M, int C, float* f3, int ld3, hipStream_t stream);
int opp_fine_gather(const float* feat, int Hf, int Wf, int ldf, const float* bank, int n_points,
                    const long long* i_ids, const long long* j_ids, int M, int wc, int stride, int Wwin, int C,
                    float* win, int ldw, float* f3, int ld3, hipStream_t stream);
int opp_fine_head_bwd(const float* f3, int ld3, const float* win, int ldw, int M, int Wwin, int C, float temp, const float* gexp, float* gf3, float* gwin,
                      hipStream_t stream);
int opp_fine_head(const float* f3, int ld3, const float* win, int ldw, int M, int Wwin, int C, float temp,
                  const float* mkpts_c, float base_scale, const float* qscale, float* expec, float* mkpts_f,
                  hipStream_t stream);
