/* libopp_hip.so -- C ABI of the MI355X (gfx950) implementation of the OnePose++ 2D-3D
 * matching forward.
 *
 * This is the drop-in boundary UNDER the Python module `onepose_plus_plus_amd.OnePosePlus_model`
 * (which mirrors the reference's `OnePosePlus_model(config)(data)` API,
 * /root/reference/src/models/OnePosePlus/OnePosePlusModel.py:25-201).  The reference is pure
 * Python on PyTorch, so there is no upstream FFI to bind; every entry point below cites the
 * reference code whose GPU work it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - all tensor pointers are DEVICE pointers to fp32 unless noted; int64 = `long long`.
 *  - the library never allocates device memory: the caller passes a workspace and the packed
 *    weight blob (sizes from the *_bytes queries) and owns every buffer.
 *  - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *    without synchronising; 0 = ok, negative = error (message via opp_last_error()).
 *  - activations are NHWC ("pixel-major, channel contiguous"); the 196-channel stages are
 *    padded to 224 channels with zeros.
 *  - batch size is 1 (the reference's inference path, inference_OnePosePlus_worker.py:52-60).
 */
#ifndef OPP_HIP_H
#define OPP_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPP_MAX_LAYERS 16

/* The `model.OnePosePlus` block of configs/experiment/inference_onepose.yaml:26-109, reduced to
 * the values the kernels depend on. */
typedef struct opp_config {
  int initial_dim;         /* loftr_backbone.resnetfpn.initial_dim (128) */
  int block_dims[3];       /* loftr_backbone.resnetfpn.block_dims (128,196,256) */
  int kpt_enc_enable;      /* keypoints_encoding.enable */
  int kpt_enc_dims[3];     /* keypoints_encoding.keypoints_encoder (32,64,128) */
  int pos_enc_enable;      /* positional_encoding.enable */
  int coarse_d_model;      /* loftr_coarse.d_model (256) */
  int coarse_nhead;        /* loftr_coarse.nhead (8) */
  int coarse_n_layers;     /* len(layer_names) * layer_iter_n (6) */
  int coarse_is_cross[OPP_MAX_LAYERS]; /* 0 = self, 1 = cross */
  int fine_d_model;        /* loftr_fine.d_model (128) */
  int fine_nhead;          /* loftr_fine.nhead (8) */
  int fine_n_layers;       /* (2) */
  int fine_is_cross[OPP_MAX_LAYERS];
  int fine_window;         /* loftr_fine.window_size (5) */
  float match_thr;         /* coarse_matching.thr */
  int match_border_rm;     /* coarse_matching.border_rm */
  float match_temperature; /* coarse_matching.dual_softmax.temperature */
  /* Not a reference key: arithmetic of the conv / Linear / score GEMMs (fp32 in, fp32 accumulate, fp32 out in
   * every mode).
   * 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain).
   * 3 = bf16x3 split (the module default): every operand x is carried EXACTLY as hi + mid + lo bf16
   *     (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 24 significant bits, fp32 exponent range, no
   *     range restriction), six bf16 MFMAs per product (the dropped mid*lo, lo*mid, lo*lo terms are <= 2^-26 |a||b|,
   *     below fp32's own rounding), 6/16 of the fp32 MFMA cycles.  Not narrower than fp32.
   * 1 = fp16x2 split (NARROWER than fp32; TUNING LIBRARY ONLY since round 6 -- the product build has no fp16x2 kernels and
   *     opp_create refuses 1 and 2, see opp_supports_precision): x ~ hi + lo fp16 (22-bit mantissa), three fp16 MFMAs
   *     per product, 3/16 of the fp32 MFMA cycles.  Activations must stay inside the fp16 range (|x| < 65504;
   *     the kernels flag non-finite outputs, see opp_forward_coarse `count[1]`), values below 2^-3 keep an
   *     absolute 2^-25 error; weights are pre-scaled per matrix and unrestricted.  The coarse score GEMM stays fp32.
   * 2 = as 1, and the score GEMM runs on the fp16x2 path too (image tokens split once per image). */
  int gemm_precision;
  /* Not a reference key: what the automatic GEMM / conv tile choice minimises.  0 = the duration of each launch
   * (one forward at a time: lowest latency); 1 = the CU time each launch occupies (several forwards in flight on
   * separate streams -- MatcherPool, bench.py --streams > 1 -- where other forwards' kernels fill idle CUs: +3.5...6.5 %
   * images/s with three forwards in flight, -9 % for a forward running alone).  Results are bit-identical. */
  int tile_policy;
  /* Not a reference key: with gemm_precision 3 every LoFTREncoderLayer behind its Q/K/V projection (attention apply, merge,
   * norm1, mlp.0, ReLU, mlp.2, norm2, residual; transformer.py:80-94) runs as ONE kernel whose activations stay in LDS:
   * 2 (the module default) = 64-token tiles at the coarse level (one round of workgroups at 9096 tokens; csrc/enc_layer64.hip),
   * 32-token tiles at the fine level; 1 = 32-token tiles everywhere (csrc/enc_chain.hip); 0 = one launch per Linear.
   * Bit-identical results in all three. */
  int encoder_fusion;
  /* Not a reference key: coarse-matcher variant under gemm_precision 3 (both operands of the score GEMM pre-split once and
   * staged global -> LDS by LDS-DMA, 4-wave workgroups on 128 x 128 tiles, two workgroups per CU; csrc/gemm_ss.hip).
   * 2 (the module default) = ONE sweep: dual-softmax statistics + score matrix, confidences then formed in place;
   * 1 = TWO sweeps (statistics only, then the tiles recomputed and the confidences written once: one N x L write instead of
   *     write + read + write, at twice the MFMA work); 0 = the r02 path (opp_gemm_kernel, image tokens as pre-split weights). */
  int score_two_sweep;
  /* Not a reference key: 1 (the module default) = opp_forward_coarse runs the FPN fine branch of the backbone (layer2 / layer1
   * lateral + output convolutions -> feat_f, needed by the fine stage only) on a side HIP stream next to the coarse level
   * (tokens, transformer, matcher), forking after layer3_outconv and joining before it returns: the short, CU-starved launches
   * of the coarse level and the chip-filling convolutions of the fine branch share the device.  Results are identical (same
   * kernels, same order within each branch); the workspace holds both branches' buffers at once.  0 = one stream. */
  int fpn_overlap;
} opp_config;

typedef struct opp_ctx opp_ctx;

const char* opp_last_error(void);
int opp_version(void);
/* sha256 (hex) of the sources this binary was built from (every file of csrc/ + this header), baked in by the build;
 * the Python binding compares it with the sources next to the library and refuses a stale binary. */
const char* opp_source_hash(void);
/* 1 when this binary can run opp_config.gemm_precision `p` (0 fp32 and 3 bf16x3: always; 1 / 2 fp16x2: only a library built with
 * `python -m onepose_plus_plus_amd.build --tuning`), else 0.  Not a reference interface: the reference has one arithmetic (fp32). */
int opp_supports_precision(int gemm_precision);

/* ---- model handle + weights -------------------------------------------------------------
 * Replaces: OnePosePlus_model.__init__ + load_state_dict (OnePosePlusModel.py:26-94,
 * src/inference/inference_OnePosePlus.py:28-38).  Weight tensors are given in the reference
 * state-dict order without the BatchNorm `num_batches_tracked` counters; opp_weight_name(i)
 * returns the state-dict key expected at index i. */
int opp_create(const opp_config* cfg, opp_ctx** out);
void opp_destroy(opp_ctx* ctx);
/* query_image_mask of the CURRENT sample (OnePosePlusModel.py:158: data['query_image_mask'].flatten(-2)): device
 * array of L = hc*wc floats, 1 = valid image cell, 0 = padding; NULL (default) = no mask.  Applies to the coarse
 * transformer (phi(Q), phi(K), V rows of masked image tokens are zeroed, linear_attention.py:49-53) and to the coarse
 * matcher (-1e9 added to the masked columns of the score matrix, coarse_matching.py:108-114) in every later call made
 * through this ctx: opp_transformer (which = 0, n_seg = 1), opp_coarse_match, opp_forward_coarse. */
int opp_set_query_mask(opp_ctx* ctx, const float* mask);
/* B > 1 (quirk of normalize_3d_keypoints, utils/normalize.py:20-21): the 3D keypoints of EVERY batch element are scaled
 * by the bounding-box extent of batch element 0 and centred on their own mean.  kpts0 [n0][3] = the keypoints of
 * batch element 0 (NULL = each cloud uses its own extent, the B = 1 behaviour).  Applies to opp_encode_points,
 * opp_coarse_tokens and opp_forward_coarse. */
int opp_set_keypoint_extent_ref(opp_ctx* ctx, const float* kpts0, int n0);
/* bf16x3 arithmetic, opt-in (default off; OPP_CONV_TAIL=1 turns it on for every context): the convolutions with 196 output channels
 * (backbone/resnet.py:88-124, block_dims[1]) as a 192-column body on the MFMA kernel (tile 128 x 192, no padded column sub-tile) + their
 * last 4 columns as fp32 FMA chains on the vector ALU (csrc/conv_tail.hip).  Same parity bar; measured slower with one forward in flight
 * and equal with three, hence not the default. */
int opp_set_conv_tail(opp_ctx* ctx, int on);
/* fp16x2 range guard (gemm_precision 1 / 2 only; a no-op otherwise): `flag` is a device int that every stage
 * enqueued through this ctx ORs with 1 when an fp16x2 GEMM produced a non-finite value, i.e. an activation left the
 * fp16 range (|x| >~ 1.3e5) -- or the input itself was not finite.  The caller zeroes it, reads it after the
 * stream work completed, and re-runs in another arithmetic (the Python module switches to bf16x3).  NULL disables. */
int opp_set_status_flag(opp_ctx* ctx, int* flag);
int opp_num_weights(const opp_ctx* ctx);
const char* opp_weight_name(const opp_ctx* ctx, int i);
long long opp_weight_numel(const opp_ctx* ctx, int i);
size_t opp_packed_weights_bytes(const opp_ctx* ctx);
/* folds eval-mode BatchNorm into the convolutions, re-lays conv weights tap-major/NHWC with
 * channel padding, concatenates q/k/v projections, transposes the keypoint MLP. */
int opp_pack_weights(opp_ctx* ctx, const float* const* weights, int n_weights, void* packed,
                     size_t packed_bytes, void* stream);
/* What the NEXT opp_pack_weights lays out: 0 (default) = everything; 1 = a training step: only the backbone convolutions WITHOUT a
 * BatchNorm behind them (the others are packed raw by opp_pack_train_weights; the training graph's transformer / keypoint-encoder
 * nodes, onepose_plus_plus_amd/train_autograd.py, read the parameters directly) -- the per-step repacking is then ~80 launches
 * shorter.  opp_backbone / opp_transformer / opp_coarse_tokens / opp_encode_points / opp_forward_coarse / opp_fine refuse to run
 * until the weights are packed with scope 0 again. */
int opp_set_pack_scope(opp_ctx* ctx, int scope);

/* ---- training-mode forward of the backbone ------------------------------
 * PL_OnePosePlus.training_step runs the same module in train() mode (src/lightning_model/
 * OnePosePlus_lightning_model.py:54-81): every nn.BatchNorm2d of the ResNet-FPN then normalises with the statistics of
 * the current batch over (B, H, W) and updates its running statistics (backbone/resnet.py:25-26, :101-113; plain
 * BatchNorm, no SyncBN upstream).  opp_pack_train_weights packs the backbone convolutions WITHOUT the folded
 * BatchNorm (after opp_pack_weights, same `weights` table; the BatchNorm affine tensors of that table are read in
 * place during the forward and must stay alive).  opp_backbone_train: image [B][H][W]; feat_c [B][H/8][W/8][256],
 * feat_f [B][H/2][W/2][128] (NHWC); bn_stats (may be NULL) receives, for BatchNorm layer i of opp_bn_layer_name(i),
 * 512 floats: [0, C) the batch mean, [C, 2C) the UNBIASED batch variance -- what the caller folds into
 * running_mean / running_var with momentum 0.1 like the BatchNorm2d default.  Forward only (no backward here). */
int opp_num_bn_layers(const opp_ctx* ctx);
const char* opp_bn_layer_name(const opp_ctx* ctx, int i);      /* state-dict prefix, e.g. "backbone.layer1.0.bn1" */
int opp_bn_layer_channels(const opp_ctx* ctx, int i);
size_t opp_packed_train_weights_bytes(const opp_ctx* ctx);
int opp_pack_train_weights(opp_ctx* ctx, const float* const* weights, int n_weights, void* packed,
                           size_t packed_bytes, void* stream);
size_t opp_backbone_train_workspace_bytes(const opp_ctx* ctx, int B, int H, int W);
int opp_backbone_train(opp_ctx* ctx, const float* image, int B, int H, int W, float* feat_c, float* feat_f,
                       float* bn_stats, void* workspace, size_t workspace_bytes, void* stream);

/* ---- stages ---------------------------------------------------------------------------- */
size_t opp_backbone_workspace_bytes(const opp_ctx* ctx, int H, int W);
/* ResNetFPN_8_2.forward (backbone/resnet.py:141-164).  image [H][W] in [0,1];
 * feat_c NHWC [H/8][W/8][256] (x3_out), feat_f NHWC [H/2][W/2][128] (x1_out). */
int opp_backbone(opp_ctx* ctx, const float* image, int H, int W, float* feat_c, float* feat_f,
                 void* workspace, size_t workspace_bytes, void* stream);

/* OnePosePlusModel.py:137-156: tokens[0:L] = feat_c + pe (pe NHWC [L][C], may be NULL),
 * tokens[L:L+N] = bank_c[C][N]^T + MLP(normalize_3d_keypoints(kpts)).  tokens [L+N][C]. */
int opp_coarse_tokens(opp_ctx* ctx, const float* feat_c, const float* pe, int L, const float* kpts,
                      const float* bank_c, int n_points, float* tokens, void* workspace,
                      size_t workspace_bytes, void* stream);

/* The 3D-point half of the tokens above on its own: it depends only on the object (keypoints +
 * coarse bank), not on the query image, so a caller that keeps the bank resident encodes it once
 * per object and passes the result to opp_forward_coarse.  tokens3d [N][C]. */
int opp_encode_points(opp_ctx* ctx, const float* kpts, const float* bank_c, int n_points,
                      float* tokens3d, void* workspace, size_t workspace_bytes, void* stream);

/* Image-independent prefix of the coarse transformer of a RESIDENT object (round 5).  With layer_names = ["self", "cross", ...]
 * (loftr_module/transformer.py:147-160) layer 0 updates the 3D-point stream from itself only, and in layer 1 both streams read the
 * PRE-update tokens of the other one, so the 3D stream after layer 0, its layer-1 q / k / v projections and its layer-1 KV / Ksum
 * depend on (keypoints3d, coarse bank) alone -- OnePosePlusModel.py:160-164 recomputes them per image.  opp_object_prefix evaluates
 * them once from the tokens of opp_encode_points into a caller-owned blob of opp_object_prefix_bytes() bytes (16-byte aligned;
 * 0 = this configuration -- another arithmetic, encoder_fusion != 2, other layer names -- has no prefix and every layer is evaluated
 * per image); opp_set_object_prefix hands the blob to the NEXT opp_forward_coarse calls of this ctx (used when tokens3d_pre is
 * given and n_points matches; NULL = off).  Results are bit-identical with and without it (rows are independent in every kernel
 * of a layer, the KV reduction is chunked per stream). */
size_t opp_object_prefix_bytes(const opp_ctx* ctx, int n_points);
size_t opp_object_prefix_workspace_bytes(const opp_ctx* ctx, int n_points);
int opp_object_prefix(opp_ctx* ctx, const float* tokens3d, int n_points, void* prefix, size_t prefix_bytes, void* workspace,
                      size_t workspace_bytes, void* stream);
int opp_set_object_prefix(opp_ctx* ctx, const void* prefix, int n_points);

/* LocalFeatureTransformer.forward (loftr_module/transformer.py:133-171), in place on
 * tokens = [n_seg*len0 rows of stream "2D" ; n_seg*len1 rows of stream "3D"] x d_model.
 * which = 0: loftr_coarse, 1: loftr_fine. */
size_t opp_transformer_workspace_bytes(const opp_ctx* ctx, int which, int n_seg, int len0, int len1);
int opp_transformer(opp_ctx* ctx, int which, float* tokens, int n_seg, int len0, int len1,
                    void* workspace, size_t workspace_bytes, void* stream);

/* CoarseMatching.forward + get_coarse_match, inference branch (utils/coarse_matching.py:76-242).
 * feat2d [L][C], feat3d [N][C]; conf [N][L] is written (data['conf_matrix']).  Outputs have
 * capacity N; *count (device int) receives M.  base_scale = H/hc; query_scale = device pointer to
 * data['query_image_scale'][0] = (h_scale, w_scale) or NULL. */
size_t opp_coarse_match_workspace_bytes(const opp_ctx* ctx, int n_points, int L);
int opp_coarse_match(opp_ctx* ctx, const float* feat3d, const float* feat2d, int n_points, int hc,
                     int wc, const float* kpts, float base_scale, const float* query_scale, float* conf,
                     long long* i_ids, long long* j_ids, float* mconf, float* mkpts_c,
                     float* mkpts_3d, int* count, void* workspace, size_t workspace_bytes,
                     void* stream);

/* Whole coarse level in one call: backbone -> tokens -> loftr_coarse -> coarse matching
 * (OnePosePlusModel.py:116-167).  feat_f is kept for the fine stage; feat_f = NULL says the caller runs no fine
 * stage (fine_matching.enable = False, OnePosePlusModel.py:169-176): the fine map is then dead -- not an output, read by
 * nothing -- and the FPN branch that produces it is not launched (every output of this call is unchanged).
 * tokens3d_pre: optional result of opp_encode_points for this object (NULL = encode kpts/bank_c here). */
size_t opp_forward_coarse_workspace_bytes(const opp_ctx* ctx, int H, int W, int n_points);
int opp_forward_coarse(opp_ctx* ctx, const float* image, int H, int W, const float* pe,
                       const float* kpts, const float* bank_c, const float* tokens3d_pre, int n_points,
                       float base_scale, const float* query_scale, float* feat_f, float* conf, long long* i_ids,
                       long long* j_ids, float* mconf, float* mkpts_c, float* mkpts_3d, int* count,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Fine level (OnePosePlusModel.py:179-201): FinePreprocess window gather, loftr_fine,
 * FineMatching.  feat_f NHWC [Hf][Wf][d_fine]; bank_f [d_fine][N] (RAW fine bank, quirk q8);
 * ids are the first M entries of the coarse outputs.  base_scale = H/Hf; query_scale as above.
 * run_transformer = config loftr_fine.enable. */
size_t opp_fine_workspace_bytes(const opp_ctx* ctx, int n_matches);
int opp_fine(opp_ctx* ctx, const float* feat_f, int Hf, int Wf, const float* bank_f, int n_points,
             const long long* i_ids, const long long* j_ids, int n_matches, int hc, int wc,
             const float* mkpts_c, float base_scale, const float* query_scale, int run_transformer,
             float* expec_f, float* mkpts_f, void* workspace, size_t workspace_bytes, void* stream);

/* Match-driven fine branch (round 5; eval mode, where the folded BatchNorm makes every backbone operator pointwise in its inputs).
 * The 1/2-resolution half of the FPN fine branch -- layer1_outconv + the bilinear residual, layer1_outconv2 (backbone/resnet.py:154-157):
 * 78 of the 337 GFLOP of a 512 x 512 forward -- is consumed only through the W x W windows around the M coarse matches
 * (loftr_module/fine_preprocess.py:41-55).  opp_set_fine_patch_buffers(x1, x2_out) makes the NEXT opp_forward_coarse calls with
 * feat_f = NULL keep x1 [H/2][W/2][pad32(block_dims[0])] and x2_out [H/4][W/4][pad32(block_dims[1])] (NHWC; element counts from
 * opp_fine_patch_buffer_floats(ctx, H, W, 0 / 1)) in the caller's buffers and stop the backbone there (NULL, NULL = off).  Once the
 * match count is known the caller either
 *   - runs opp_fine_patches: the three convolutions as VALID convolutions over a (W+4)^2 -> (W+2)^2 -> W^2 patch pyramid per match
 *     (49 MFLOP per match) on the same kernel, weights and K order as the dense maps, written straight into the window tokens of
 *     loftr_fine, then the fine transformer and FineMatching exactly as opp_fine; or
 *   - for many matches (break-even ~ 1000 at 512 x 512) completes the dense map with opp_backbone_fine_branch and calls opp_fine.
 * Both give the bits of the one-call path (tests/test_e2e_gpu.py::test_match_driven_fine_branch_*). */
int opp_set_fine_patch_buffers(opp_ctx* ctx, float* x1, float* x2_out);
size_t opp_fine_patch_buffer_floats(const opp_ctx* ctx, int H, int W, int which);
size_t opp_fine_patches_workspace_bytes(const opp_ctx* ctx, int n_matches);
int opp_fine_patches(opp_ctx* ctx, const float* x1, const float* x2_out, int H, int W, const float* bank_f, int n_points,
                     const long long* i_ids, const long long* j_ids, int n_matches, int hc, int wc, const float* mkpts_c,
                     float base_scale, const float* query_scale, int run_transformer, float* expec_f, float* mkpts_f,
                     void* workspace, size_t workspace_bytes, void* stream);
size_t opp_backbone_fine_branch_workspace_bytes(const opp_ctx* ctx, int H, int W);
int opp_backbone_fine_branch(opp_ctx* ctx, const float* x1, const float* x2_out, int H, int W, float* feat_f, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- training step: coarse focal loss (Loss.compute_coarse_loss, src/lightning_model/losses.py:18-55) -------------
 * conf [n] fp32 (the B x N x L confidence matrix, 16-byte aligned), conf_gt [n] int16 (0 / 1; other values are ignored
 * like the reference's `== 1` / `== 0` masks), weight [n] or NULL.
 * forward: sums[4] (device, fp64) = { sum of the positive terms, sum of the negative terms, #positives, #negatives };
 *   term_pos = -alpha (1 - c)^gamma log c, term_neg = -(1 - alpha) c^gamma log(1 - c), c = clamp(conf, 1e-6, 1 - 1e-6);
 *   the caller forms pos_weight * mean_pos + neg_weight * mean_neg (losses.py:44-53) from them without a host sync.
 * backward: grad_conf [n] = d loss / d conf given scales[2] (device) = { g * pos_weight / #pos, g * neg_weight / #neg },
 *   g the incoming gradient of the scalar loss; zero where the clamp saturates. */
size_t opp_focal_loss_workspace_bytes(size_t n);
int opp_focal_loss_forward(const float* conf, const short* conf_gt, const float* weight, size_t n, float alpha,
                           float gamma, double* sums, void* workspace, size_t workspace_bytes, void* stream);
int opp_focal_loss_backward(const float* conf, const short* conf_gt, const float* weight, size_t n, float alpha,
                            float gamma, const float* scales, float* grad_conf, void* stream);

/* The same two passes on the ground truth AS THE CALLER HOLDS IT -- gt_kind 0 = int16, 1 = fp32 (the reference's dataset emits
 * float zeros / ones, OnePosePlus_dataset.py:174-236), 2 = uint8 / bool; values other than 0 / 1 are ignored -- and with the
 * weight of Loss.compute_c_weight (losses.py:103-111) formed on the fly from its two factors: mask0 [B * N] (per 3D point),
 * mask1 [B * L] (per image cell), weight[b][i][j] = mask0[b][i] * mask1[b][j]; n = B * N * L.  Pass `weight` (full array) OR
 * both masks OR neither.  No int16 copy of conf_gt and no B x N x L weight tensor is made. */
int opp_focal_loss_forward_ex(const float* conf, const void* conf_gt, int gt_kind, const float* weight, const float* mask0,
                              const float* mask1, int N, int L, size_t n, float alpha, float gamma, double* sums, void* workspace,
                              size_t workspace_bytes, void* stream);
int opp_focal_loss_backward_ex(const float* conf, const void* conf_gt, int gt_kind, const float* weight, const float* mask0,
                               const float* mask1, int N, int L, size_t n, float alpha, float gamma, const float* scales,
                               float* grad_conf, void* stream);

/* backward of conf = A * B, A = softmax(S, dim = points), B = softmax(S, dim = cells) (utils/coarse_matching.py:115);
 * sim, grad_conf, grad_sim [B][N][L]; lse_row [B][N] = logsumexp_j S_ij, lse_col [B][L] = logsumexp_i S_ij:
 *   grad_sim_ij = 2 conf_ij g_ij - A_ij sum_i' g_i'j conf_i'j - B_ij sum_j' g_ij' conf_ij'. */
size_t opp_dual_softmax_backward_workspace_bytes(int B, int N, int L);
int opp_dual_softmax_backward(const float* grad_conf, const float* sim, const float* lse_row, const float* lse_col,
                              int B, int N, int L, float* grad_sim, void* workspace, size_t workspace_bytes,
                              void* stream);

/* ---- training step: backward of a bias-free nn.Linear (every Linear of loftr_coarse / loftr_fine,
 * loftr_module/transformer.py:26-47) on the MFMA GEMM.  Y [M][N] = X [M][K] W [N][K]^T (PyTorch layouts, fp32):
 *   grad_x [M][K] = grad_out [M][N] W            (NULL to skip; needs W)
 *   grad_w [N][K] = grad_out^T X                 (NULL to skip; needs X; accumulate_grad_w != 0 adds to the existing values):
 *                   a reduction over the M tokens, run as split-K with a fixed-order sum of the partials (deterministic).
 * N, K multiples of 32; prec 0 = exact-fp32 MFMA, 2 = bf16x3 (operands carried exactly, fp32 accumulate).  Both operands
 * are transposed / split into the workspace as needed; nothing is allocated. */
size_t opp_linear_backward_workspace_bytes(int M, int N, int K, int prec);
int opp_linear_backward(const float* grad_out, const float* X, const float* W, int M, int N, int K, float* grad_x, float* grad_w,
                        int accumulate_grad_w, int prec, void* workspace, size_t workspace_bytes, void* stream);

/* ---- training step: LinearAttention.forward with its backward (loftr_module/linear_attention.py:29-61) -------------------
 * Raw projections q [B][L][nhead][D], k, v [B][S][nhead][D] (fp32, contiguous; D = 32 or 16), optional masks q_mask [B][L],
 * kv_mask [B][S] (0 / 1 floats): out = (phi(q) KV) / (phi(q) . Ksum + 1e-6) * S with KV = sum_s phi(k_s)^T v_s / S,
 * phi = elu + 1.  forward also returns KV [B][nhead][D][D] and Ksum [B][nhead][D], which backward takes back together with
 * grad_out [B][L][nhead][D] and produces grad_q, grad_k, grad_v.  Token reductions are chunk partials summed in chunk order
 * (deterministic). */
size_t opp_linear_attention_train_workspace_bytes(int B, int L, int S, int nhead, int D);
int opp_linear_attention_train_forward(const float* q, const float* k, const float* v, const float* q_mask, const float* kv_mask, int B,
                                       int L, int S, int nhead, int D, float* out, float* kv, float* ks, void* workspace,
                                       size_t workspace_bytes, void* stream);
int opp_linear_attention_train_backward(const float* q, const float* k, const float* v, const float* q_mask, const float* kv_mask,
                                        const float* kv, const float* ks, const float* grad_out, int B, int L, int S, int nhead, int D,
                                        float* grad_q, float* grad_k, float* grad_v, void* workspace, size_t workspace_bytes,
                                        void* stream);

/* ---- training step: ResNet-FPN forward that keeps its activations, and its backward ---------------------------------------
 * PL_OnePosePlus.training_step (src/lightning_model/OnePosePlus_lightning_model.py:54-81) differentiates the loss through
 * ResNetFPN_8_2.forward (backbone/resnet.py:141-164): nn.Conv2d (no bias), nn.BatchNorm2d in train(), ReLU / LeakyReLU(0.01),
 * the residual adds and the two bilinear x2 upsamples.  opp_backbone_train_tape is opp_backbone_train writing every tensor the
 * backward needs (raw convolution outputs, block outputs, batch mean / 1/sqrt(var + eps) per BatchNorm) into the caller's
 * `tape` (nothing is recomputed later); opp_backbone_backward takes the gradients of the two outputs, feat_c [B][H/8][W/8][256]
 * and feat_f [B][H/2][W/2][128] (NHWC), and writes the gradient of EVERY backbone parameter: `weights` is the table given to
 * opp_pack_weights (the raw convolution weights are read from it), grads[i] the device pointer that receives the gradient of
 * weights[i] in the PyTorch layout ([cout][cin][kh][kw]; BatchNorm weight / bias [C]); entries of tensors outside the backbone
 * and of the running statistics are ignored and may be NULL.  Convolution input gradients run on the implicit-GEMM kernel with
 * the flipped / transposed weight, weight gradients on a pixel-major split reduction (csrc/conv_bwd.hip), in the arithmetic of
 * opp_config.gemm_precision (3 = bf16x3 or 0 = fp32; the weight-gradient kernel is bf16x3 in both: operands exact, fp32 accumulate). */
size_t opp_backbone_tape_bytes(const opp_ctx* ctx, int B, int H, int W);
size_t opp_backbone_train_tape_workspace_bytes(const opp_ctx* ctx, int B, int H, int W);
int opp_backbone_train_tape(opp_ctx* ctx, const float* image, int B, int H, int W, float* feat_c, float* feat_f, float* bn_stats,
                            void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes, void* stream);
size_t opp_backbone_backward_workspace_bytes(const opp_ctx* ctx, int B, int H, int W);
int opp_backbone_backward(opp_ctx* ctx, const float* image, int B, int H, int W, const void* tape, size_t tape_bytes,
                          const float* const* weights, int n_weights, const float* grad_feat_c, const float* grad_feat_f,
                          float* const* grads, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of one nn.Conv2d (bias-free, kernel 1 or 3, padding k/2, stride 1 or 2; backbone/resnet.py:10-18) over NHWC tensors
 * whose channel counts are padded to multiples of 32 with zeros: x [B][Hin][Win][cin_pad], w [cout][cin][ks][ks] (PyTorch layout),
 * grad_y [B][Ho][Wo][cout_pad].  grad_x [B][Hin][Win][cin_pad] and / or grad_w [cout][cin][ks][ks] (either may be NULL);
 * grad_x_add (optional, same shape as grad_x, MAY ALIAS it) is added to the input gradient -- how the gradients of a tensor
 * with several consumers are accumulated.  prec 0 = fp32 MFMA, 2 = bf16x3. */
size_t opp_conv2d_backward_workspace_bytes(int B, int Hin, int Win, int cin, int cout, int ks, int stride, int prec);
int opp_conv2d_backward_nhwc(const float* x, int B, int Hin, int Win, int cin, const float* w, int cout, int ks, int stride,
                             const float* grad_y, float* grad_x, const float* grad_x_add, float* grad_w, int prec, void* workspace,
                             size_t workspace_bytes, void* stream);
/* Backward of nn.BatchNorm2d in train() followed by (+ residual) -> activation (backbone/resnet.py:25-26, :37-45) over an NHWC
 * tensor [rows][ld] with C real channels: y = act((raw - mean) * invstd * gamma + beta (+ res)); act 0 none, 1 ReLU, 2 LeakyReLU(0.01).
 * grad_y, y (unused for act 0), raw, the forward's batch mean / invstd [ld] in; grad_raw (may alias grad_y), grad_res (optional,
 * the residual branch's gradient; may alias grad_y), grad_gamma / grad_beta [C] (optional) out. */
size_t opp_batchnorm_backward_workspace_bytes(int rows, int ld);
int opp_batchnorm_backward_nhwc(const float* grad_y, const float* y, const float* raw, int rows, int ld, int C, int act,
                                const float* gamma, const float* mean, const float* invstd, float* grad_raw, float* grad_res,
                                float* grad_gamma, float* grad_beta, void* workspace, size_t workspace_bytes, void* stream);
/* Transpose of F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) (backbone/resnet.py:151,155):
 * grad_in [B][Hr][Wr][ld] (+)= sum over the output pixels [B][2 Hr][2 Wr][ld] that read it. */
int opp_upsample2x_backward_nhwc(const float* grad_out, int B, int Hr, int Wr, int ld, float* grad_in, int accumulate, void* stream);
/* nn.LayerNorm (eps 1e-5) over rows of C = 64 / 128 / 256 values with the statistics kept for the backward
 * (loftr_module/transformer.py:87-88, :92-94): y = (residual ? residual : 0) + LN(x) * gamma + beta; mean / rstd [rows] out.
 * backward: grad_x [rows][C], grad_gamma / grad_beta [C] (fixed-order reduction over the rows). */
int opp_layer_norm_train_forward(const float* x, const float* gamma, const float* beta, const float* residual, int rows, int C,
                                 float* y, float* mean, float* rstd, void* stream);
size_t opp_layer_norm_train_backward_workspace_bytes(int rows, int C);
int opp_layer_norm_train_backward(const float* grad_y, const float* x, const float* gamma, const float* mean, const float* rstd,
                                  int rows, int C, float* grad_x, float* grad_gamma, float* grad_beta, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* Dual softmax of the score matrix sim [B][N][L] (utils/coarse_matching.py:115) with what opp_dual_softmax_backward needs:
 * lse_row [B][N] = logsumexp over the cells, lse_col [B][L] = logsumexp over the points; conf (optional, may alias sim) =
 * exp(sim - lse_col) * exp(sim - lse_row). */
size_t opp_dual_softmax_forward_workspace_bytes(int B, int N, int L);
int opp_dual_softmax_forward(const float* sim, int B, int N, int L, float* lse_row, float* lse_col, float* conf, void* workspace,
                             size_t workspace_bytes, void* stream);
/* FinePreprocess on a batch (loftr_module/fine_preprocess.py:41-55: F.unfold + [b_ids, j_ids] indexing): windows [M][W*W][C] from
 * feat_f [B][Hf][Wf][C] around coarse cell j_ids[m] of image b_ids[m] (zeros outside the image); the backward adds the windows'
 * gradients into grad_feat_f (zeroed first; overlapping windows are summed with fp32 atomics). */
int opp_fine_window_gather(const float* feat_f, int B, int Hf, int Wf, int C, const long long* b_ids, const long long* j_ids,
                           int n_matches, int hc, int wc, int window, float* windows, void* stream);
int opp_fine_window_gather_backward(const float* grad_windows, int B, int Hf, int Wf, int C, const long long* b_ids,
                                    const long long* j_ids, int n_matches, int hc, int wc, int window, float* grad_feat_f,
                                    void* stream);

/* FineMatching's expectation head on its own (utils/fine_matching.py:63-94: heatmap = softmax(<f3, win_r> / sqrt(C)), spatial expectation and
 * the summed standard deviation) for the training graph, and its backward.  f3 [M][C] point tokens, win [M][window^2][C] window tokens, expec_f
 * [M][3]; scratch_xy: 4 M floats.  Backward: grad_expec_f [M][3] -> grad_f3 [M][C], grad_win [M][window^2][C] (every element written). */
int opp_fine_head_train_forward(const float* f3, const float* win, int n_matches, int window, int C, float* expec_f, float* scratch_xy,
                                void* stream);
int opp_fine_head_train_backward(const float* f3, const float* win, int n_matches, int window, int C, const float* grad_expec_f,
                                 float* grad_f3, float* grad_win, void* stream);

/* Ground-truth matrices of one training sample, formed on the device from the k assignment pairs
 * (OnePosePlusDataset.build_assignmatrix, src/datasets/OnePosePlus_dataset.py:174-236; the loader builds 229 MB per sample on the host
 * at N = 7000).  kp2d_coarse / kp2d_fine [n2d][2] fp32; assign [2][k] int64 (row 0: 2D keypoint index, row 1: padded 3D point index);
 * scale_x / scale_y = query_img_scale[1] / [0].  conf_gt [N][L] int16 (zeroed here, 1 at the pairs), fine_loc_gt [N][L][2] fp32 (-50
 * filled here, the fine keypoint at the pairs; 16-byte aligned); pairs with a 3D index >= N or j > L are dropped like upstream, of
 * duplicate (i, j) pairs the last one wins (CPU index_put).  keys: k int64 of scratch.  *status (device int): bit 0 = an index upstream
 * raises IndexError on (skipped here), bit 1 = duplicate pairs seen (upstream logs "Keypoints duplicate!"). */
int opp_build_assignmatrix(const float* kp2d_coarse, const float* kp2d_fine, int n2d, const long long* assign, int k, int N, int L,
                           int w_c, float scale_x, float scale_y, float coarse_scale, short* conf_gt, float* fine_loc_gt,
                           long long* keys, int* status, void* stream);

/* ---- building blocks (exported for stage-level parity tests and tuning) ------------------ */
/* NHWC convolution as implicit GEMM on the MFMA.  x [Hin][Win][cin_pad], cin_pad = cin rounded up to 32 (pad
 * channels zero); w_packed [cout_pad][opp_conv_packed_k(cin, ks)] (from opp_pack_conv_weight with the same cin:
 * ks*ks*cin_pad, or the shorter K-tail packing for a 3x3 kernel over 32 n + (1..4) channels); bias [cout_pad] or NULL; residual:
 * res_mode 0 none, 1 same-shape NHWC [Hout][Wout][cout_pad], 2 bilinear x2 (align_corners)
 * upsample of NHWC [Hout/2][Wout/2][cout_pad]; act 0 none, 1 ReLU, 2 LeakyReLU(0.01).
 * tile_cfg < 0 selects automatically (an explicit 27 -- the 128x224 tile -- treats weight rows >= 208 as the zero padding they are for a
 * 196-channel layer: the caller vouches for it).  prec = operand arithmetic: 0 fp32; 1 fp16x2 (w_packed additionally
 * pre-split by opp_pack_h2, h2_scale = the scale2 pointer given to it, or NULL); 2 bf16x3 (w_packed pre-split by
 * opp_pack_b3, 1.5x the floats). */
int opp_conv2d_nhwc(const float* x, int Hin, int Win, int cin, const float* w_packed,
                    const float* bias, int cout_pad, int ks, int stride, const float* residual,
                    int res_mode, int act, float* y, int tile_cfg, int prec, const float* h2_scale,
                    void* stream);
/* The same convolution in the bf16x3 arithmetic with PRE-SPLIT activations (round 6; what the inference backbone chains its layers with --
 * nn.Conv2d + BatchNorm2d + activation of backbone/resnet.py:10-45, :141-164 -- so that a map is split into bf16 triples once, by the
 * epilogue that produces it, instead of in the K loop of every tap and column tile that reads it).  A pre-split map holds, per pixel row,
 * every 8 channels as 48 bytes [hi x8 | mid x8 | lo x8] (bf16; opp_pack_b3 over the fp32 rows gives exactly these bytes), 6 cin_pad bytes
 * per pixel, 16-byte aligned.  x_split != NULL: the input is read from it when the convolution's K walk has no packed tail (any 1x1; 3x3
 * over cin % 32 == 0 real channels), else from x (one of the two must serve).  y_split != NULL: the epilogue also writes the output
 * pre-split (y may then be NULL).  Results are bit-identical to opp_conv2d_nhwc(prec = 2). */
int opp_conv2d_nhwc_split(const float* x, const void* x_split, int Hin, int Win, int cin, const float* w_packed,
                          const float* bias, int cout_pad, int ks, int stride, const float* residual, int res_mode,
                          int act, float* y, void* y_split, int tile_cfg, void* stream);
int opp_pack_conv_weight(const float* w, const float* scale, int cout, int cin, int ks,
                         int cout_pad, int cin_pad, float* out, void* stream);
/* floats per packed weight row of a (cin, ks) convolution */
int opp_conv_packed_k(int cin, int ks);
/* LinearAttention (loftr_module/linear_attention.py:29-61) of the two token streams of one encoder layer, as the
 * transformer runs it.  qkv [n_seg * (len0 + len1)][3 C]: per row phi(Q) | phi(K) | V / S (phi = elu + 1, S = the
 * row's own stream length; the QKV GEMM epilogue produces exactly this), the n_seg segments of stream 0 (len0 rows
 * each) first, then those of stream 1 (len1 rows each).  msg [same rows][C] = phi(Q) (sum_s phi(K_s)^T V_s / S)
 * / (phi(Q) . sum_s phi(K_s) + 1e-6) * S_src; cross = 0: every stream attends to itself, 1: to the other stream's
 * K, V of the same segment.  Supported: (C, nhead) = (256, 8) and (128, 8). */
size_t opp_linear_attention_workspace_bytes(int n_seg, int len0, int len1, int C, int nhead);
int opp_linear_attention(const float* qkv, int n_seg, int len0, int len1, int C, int nhead, int cross,
                         float* msg, void* workspace, size_t workspace_bytes, void* stream);
/* The tile configuration the GEMM / implicit-GEMM convolution launcher picks by itself (tile_cfg < 0) for an M x n_store output over K (a multiple
 * of 32) -- a pure host function of shape, operand arithmetic (prec as in opp_conv2d_nhwc) and tile policy (0 latency, 1 throughput: opp_config.tile_policy),
 * exported so that the policy is testable without a device.  n_real = real output channels when n_store is their padded count (0 = unknown), conv != 0
 * for a convolution.  Returns the tile config (the list under opp_profile_start), + 1000 when a bf16x3 convolution of this shape runs as four K slices
 * on that tile, or a negative OPP_ERR_*.  Replaces nothing in the reference (cuDNN / cuBLAS pick their own kernels: backbone/resnet.py:10-45). */
int opp_gemm_tile_for(int M, int n_real, int n_store, int K, int conv, int prec, int tile_policy);

/* C[M][N] = act(A[M][K] * W[N][K]^T) ; act 0 none, 1 ReLU.  prec as above (1: W pre-split by opp_pack_h2,
 * 2: by opp_pack_b3). */
int opp_linear(const float* A, int M, int K, const float* W, int N, int act, float* C,
               int tile_cfg, int prec, const float* h2_scale, void* stream);
/* fp16x2 pre-split of a K-contiguous weight matrix (n floats, n % 8 == 0, out != in, same size).
 * scale2 (device, 2 floats, may be NULL): receives {s, 1/s}, s the power of two that brings max|w|
 * into [2^14, 2^15) and is applied before the split; pass the same pointer as h2_scale to
 * opp_linear / opp_conv2d_nhwc, which multiply the accumulators by 1/s (exact). */
int opp_pack_h2(const float* in, float* out, size_t n, float* scale2, void* stream);
/* bf16x3 pre-split of a K-contiguous operand (n floats, n % 8 == 0, out != in): out receives 1.5 n floats, every
 * 8 consecutive values as 48 bytes [hi x8 | mid x8 | lo x8] bf16 with hi + mid + lo == x exactly. */
int opp_pack_b3(const float* in, float* out, size_t n, void* stream);
/* C[M][N] = (residual ? residual : 0) + LayerNorm_N(A[M][K] * W[N][K]^T) * gamma + beta, N in {256, 128}:
 * the LayerNorm (eps 1e-5) runs in the GEMM epilogue (merge -> norm1, mlp.2 -> norm2 -> +x of
 * LoFTREncoderLayer.forward, loftr_module/transformer.py:86-94).  residual may alias C. */
int opp_linear_layernorm(const float* A, int M, int K, const float* W, int N, const float* gamma,
                         const float* beta, const float* residual, float* C, int prec,
                         const float* h2_scale, void* stream);
int opp_layer_norm(const float* x, const float* gamma, const float* beta, const float* residual,
                   float* out, int rows, int C, void* stream);

/* ---- query-image ingest ------------------------------------------------------
 * 8-bit grayscale frame [h][src_stride] on the device -> cv2.resize(INTER_LINEAR, 8-bit fixed point)
 * to [h_new][w_new] -> fp32 / 255 (read_grayscale + grayscale2tensor, src/utils/data_io.py:34-69,
 * :105-106).  dst (fp32, row stride dst_stride floats; e.g. a pad_to canvas) and dst_u8 (the resized
 * 8-bit image, packed) are both optional, at least one must be given. */
int opp_image_ingest_u8(const unsigned char* src, int h, int w, int src_stride, int h_new, int w_new,
                        float* dst, int dst_stride, unsigned char* dst_u8, void* stream);

/* ---- per-object descriptor bank (the step before the path) ----------------------
 * mean_descriptors_and_scores (src/sfm_utils/postprocess/feature_process.py:527-541): out[i][:] = mean over rows
 * [offsets[i], offsets[i+1]) of `rows` [R][D] fp32 (the 2D features of the track of 3D point i, concatenated in
 * track order by gather_3d_ann, :255-311); float32 row-by-row accumulation and float32 division like numpy's
 * axis-0 mean, i.e. bit-identical to the reference.  offsets: n_seg + 1 int64 on the device. */
int opp_segmented_mean(const float* rows, int D, const long long* offsets, int n_seg, float* out, void* stream);

/* ---- pose from the matches (next row after the matcher) --------------------
 * PnP-RANSAC on the device: replaces ransac_PnP / cv2.solvePnPRansac(EPNP, 10000 iterations)
 * (src/utils/metric_utils.py:121-204) so the matches never leave the GPU.  pts2d [n][2] (pixels),
 * pts3d [n][3] fp32 on the device; K4 = {fx, fy, cx, cy} on the HOST; `scale` multiplies the 3D
 * points before solving and divides the translation after (reference `point_cloud_rescale`).
 * Outputs on the device: pose_out [12] doubles = row-major [R | t] (3x4), inlier_mask [n] (0/1),
 * n_inliers, ok (0 -> identity pose, the reference's cv2.error branch).  Deterministic for a seed. */
size_t opp_pnp_workspace_bytes(int iterations);
int opp_pnp_ransac(const float* pts2d, const float* pts3d, int n_points, const double* K4,
                   double reproj_error_px, double scale, int iterations, unsigned seed,
                   int refine_iters, double* pose_out, int* inlier_mask, int* n_inliers, int* ok,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Tuning aid (libraries built with -DOPP_TUNING only; otherwise returns an error): device buffer (4 x uint64 per
 * wave) for the phase time stamps written by the timed conv variants (tile_cfg 120 = 256x128 / 8 waves,
 * 121 = 128x128 / 4 waves, 122 = 128x128 / 8 waves); NULL disables. */
int opp_debug_timestamps(void* buf);

/* ---- live kernel timing for bench.py's roofline leg ---------------------------------------
 * Arms HIP-event timing (events recorded on the launch stream) of every launch of ONE kernel symbol:
 *   tile_cfg < 1000: a GEMM tile configuration (0 128x128/4 waves, 1 64x128, 2 64x64, 10/11 deeper prefetch,
 *                    20 256x128/8 waves, 22 128x256/8 waves, 24 128x192/8 waves, 25 128x128/8 waves, 26 64x128/8 waves,
 *                    27 128x224/8 waves on four fragment sets (bf16x3; outputs of <= 208 real / <= 224 stored columns), 30 64x256/8 waves + fused LayerNorm) of
 *                    kind 0 dense GEMM, 1 implicit-GEMM conv, 2 coarse score GEMM with fused softmax statistics;
 *   tile_cfg 1000 linear-attention KV gather, 1001 linear-attention apply, 1002 dual-softmax confidence pass,
 *            1003 the whole fine stage, 1004 fine-level attention (one workgroup per match), 1005 window gather,
 *            1006 fine head, 1007 / 1008 focal loss forward / backward (kind ignored).
 * opp_profile_stop synchronises and returns the summed time, the summed ALGORITHMIC work (FLOPs with unpadded
 * channel counts for GEMMs, bytes for the 1000+ symbols) and the number of launches measured. */
int opp_profile_start(int tile_cfg, int kind, int capacity_launches);
/* What the event pair itself adds to a short launch: an empty kernel timed exactly like an armed symbol (event before, event
 * after, all pairs enqueued back to back, one synchronisation at the end); *mean_us = mean elapsed time of `launches` pairs.
 * bench.py reports it beside the sub-15-us symbols and corrects their roofline fractions by it. */
int opp_profile_event_overhead(int launches, double* mean_us, void* stream);
/* The empty kernel's own duration: `launches` of them back to back between ONE event pair; *mean_us = elapsed / launches. */
int opp_profile_empty_kernel(int launches, double* mean_us, void* stream);
/* What an event pair adds to the reading of a real launch: a spin kernel of ~spin_us timed per launch between its own event pair (*pair_us)
 * and back to back between one pair (*b2b_us = its duration + the inter-kernel gap); pair_us - b2b_us is what bench.py subtracts. */
int opp_profile_event_calibration(int launches, double spin_us, double* pair_us, double* b2b_us, void* stream);
int opp_profile_stop(double* total_ms, double* total_work, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* OPP_HIP_H */
