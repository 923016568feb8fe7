// C ABI (include/opp_hip.h): model handle, weight packing, stage orchestration.
// Mirrors the control flow of OnePosePlus_model.forward
// (/root/reference/src/models/OnePosePlus/OnePosePlusModel.py:96-201) on one HIP stream.
#include <math.h>
#include <stdarg.h>

#include <string>
#include <vector>

#include "../../include/opp_hip.h"
#include "opp_internal.h"

// ----------------------------------------------------------------------------------------
// error state
// ----------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void opp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* opp_last_error(void) { return g_err; }
extern "C" int opp_version(void) { return 1; }

// ----------------------------------------------------------------------------------------
// model description
// ----------------------------------------------------------------------------------------
namespace {

inline int pad32(int c) { return (c + 31) / 32 * 32; }

// opp_config.gemm_precision -> OPP_PREC_* of the conv / Linear GEMMs and of the coarse score GEMM
inline int gemm_prec(const opp_config& cfg) {
  return cfg.gemm_precision == 3 ? OPP_PREC_BF16X3 : cfg.gemm_precision ? OPP_PREC_FP16X2 : OPP_PREC_FP32;
}
inline int score_prec(const opp_config& cfg) {
  return cfg.gemm_precision == 3 ? OPP_PREC_BF16X3 : cfg.gemm_precision == 2 ? OPP_PREC_FP16X2 : OPP_PREC_FP32;
}
// floats occupied by a K-contiguous operand of n values once pre-split for `prec` (bf16x3: 48 B per 8 values)
inline size_t split_floats(size_t n, int prec) { return prec == OPP_PREC_BF16X3 ? n / 2 * 3 : n; }

struct WeightEntry {
  std::string name;
  long long numel;
};

struct ConvDesc {  // one packed convolution
  int w_idx = -1;   // weight index in the state-dict table
  int bn_idx = -1;  // index of <bn>.weight (bias, running_mean, running_var follow) or -1
  int cin = 0, cout = 0, ks = 1;
  float* w = nullptr;     // packed [cout_pad][k_len()]
  float* bias = nullptr;  // [cout_pad] (folded BN shift) or nullptr
  float* h2s = nullptr;   // fp16x2 only: {scale, 1/scale} applied to w before the hi/lo split
  // training mode (opp_pack_train_weights): the same packing WITHOUT the folded BatchNorm, plus the caller-owned
  // affine parameters of the BatchNorm that follows this convolution
  float* w_train = nullptr;
  float* h2s_train = nullptr;
  // bf16x3, cout = 32 n + (1..4) with n >= 2 (the 196-channel layers): the last columns as fp32 FMA chains (conv_tail.hip) beside a body of
  // 32 n columns on the MFMA kernel; [taps][cin_pad][4] fp32, BatchNorm scale folded (w_tail) / raw (w_tail_train)
  float* w_tail = nullptr;
  float* w_tail_train = nullptr;
  int tail_cols() const { return (cout % 32 >= 1 && cout % 32 <= 4 && cout >= 64) ? cout % 32 : 0; }
  int body_cols() const { return cout / 32 * 32; }
  const float* gamma = nullptr;
  const float* beta = nullptr;
  int bn_slot = -1;       // index of that BatchNorm among the backbone's BatchNorm layers (state-dict order)
  int cin_pad() const { return pad32(cin); }
  int cout_pad() const { return pad32(cout); }
  int k_len() const { return opp_conv_k(cin, ks); }   // ks*ks*cin_pad, shorter with K tail packing (opp_common.h)
  size_t w_floats() const { return (size_t)cout_pad() * k_len(); }
};

struct BlockDesc {
  ConvDesc conv1, conv2, down;
  bool has_down = false;
};

struct EncLayerDesc {
  int q_idx = -1;  // q,k,v,merge,mlp0,mlp2,norm1.w,norm1.b,norm2.w,norm2.b follow consecutively
  float *wqkv = nullptr, *wmerge = nullptr, *w1 = nullptr, *w2 = nullptr;
  float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
  float *sqkv = nullptr, *smerge = nullptr, *s1 = nullptr, *s2 = nullptr;   // fp16x2 {scale, 1/scale} per matrix
  // bf16x3 only: merge / mlp.0 / mlp.2 once more in the fragment-major order of the fused layer kernel (enc_chain.hip)
  void *fmerge = nullptr, *f1 = nullptr, *f2 = nullptr;
  void* fqkv = nullptr;   // coarse level: q | k | v [3 C][C] in the same order -- folded into the PREVIOUS layer's kernel (enc_layer64.hip, r05)
};

}  // namespace

struct opp_ctx {
  opp_config cfg;
  std::vector<WeightEntry> table;
  // backbone
  ConvDesc stem;  // packed as [cout][64]
  BlockDesc blocks[6];
  ConvDesc l3_out, l2_out, l2_out2a, l2_out2b, l1_out, l1_out2a, l1_out2b;
  // keypoint MLP
  int kpt_idx = -1;
  float* kpt_wt[4] = {nullptr, nullptr, nullptr, nullptr};
  float* kpt_b[4] = {nullptr, nullptr, nullptr, nullptr};
  // transformers
  std::vector<EncLayerDesc> coarse, fine;
  bool packed = false;
  size_t packed_bytes = 0;
  int* status_flag = nullptr;      // opp_set_status_flag
  const float* query_mask = nullptr;      // opp_set_query_mask: [L] floats (0 / 1) of the CURRENT sample, or null
  const float* kpt_extent_ref = nullptr;  // opp_set_keypoint_extent_ref: keypoints of batch element 0 (quirk q4)
  int kpt_extent_n = 0;
  const float* obj_prefix = nullptr;      // opp_set_object_prefix: result of opp_object_prefix for the CURRENT object, or null
  int obj_prefix_n = 0;
  // opp_set_conv_tail: 196-channel layers as a 192-column MFMA body + a 4-column fp32 tail (conv_tail.hip).  Measured slower for one forward
  // in flight and equal with three, so off unless asked for (OPP_CONV_TAIL=1 turns it on for every context)
  bool conv_tail = getenv("OPP_CONV_TAIL") && getenv("OPP_CONV_TAIL")[0] == '1';
  float* fine_x1 = nullptr;               // opp_set_fine_patch_buffers: caller-owned x1 / x2_out of the CURRENT image (match-driven fine branch)
  float* fine_x2o = nullptr;
  float* scratch_scale = nullptr;  // [256] BN scale temp inside the blob
  float* scratch_h2 = nullptr;     // fp16x2 / bf16x3 pre-split staging (largest weight matrix)
  bool train_packed = false;
  // opp_set_pack_scope: 0 = opp_pack_weights lays out everything; 1 = a training step: only the convolutions without BatchNorm (the
  // others come from opp_pack_train_weights, the training graph reads the transformer /
  // keypoint-encoder parameters directly): `tr_packed` says whether the packed blob holds the transformer + keypoint-MLP weights
  int pack_scope = 0;
  bool tr_packed = false;
  bool bn_packed = false;   // the BatchNorm-folded (eval) packing of the backbone convolutions is present (scope 0 only)
  // fine-branch overlap of opp_forward_coarse (opp_config.fpn_overlap): a side stream and two events, created on first use
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::vector<std::string> bn_names;   // BatchNorm layers of the backbone in state-dict order (prefix, e.g. "backbone.bn1")
  std::vector<int> bn_channels;
};

namespace {

// fp16x2 range guard: the status flag of the ctx whose stage this host thread is enqueuing (every OppGemm
// built below picks it up)
thread_local int* t_status_flag = nullptr;
thread_local int t_tile_policy = OPP_TILES_LATENCY;   // opp_config.tile_policy of that ctx
thread_local bool t_conv_tail = false;                // opp_set_conv_tail of that ctx
struct FlagScope {
  int* prev;
  int prev_policy;
  bool prev_tail;
  explicit FlagScope(const opp_ctx* c) : prev(t_status_flag), prev_policy(t_tile_policy), prev_tail(t_conv_tail) {
    t_status_flag = c ? c->status_flag : nullptr;
    t_tile_policy = c ? c->cfg.tile_policy : OPP_TILES_LATENCY;
    t_conv_tail = c ? c->conv_tail : false;
  }
  ~FlagScope() {
    t_status_flag = prev;
    t_tile_policy = prev_policy;
    t_conv_tail = prev_tail;
  }
};

int add_w(opp_ctx* c, const std::string& name, long long numel) {
  c->table.push_back({name, numel});
  return (int)c->table.size() - 1;
}
int add_bn(opp_ctx* c, const std::string& p, int ch) {
  c->bn_names.push_back(p);
  c->bn_channels.push_back(ch);
  const int i = add_w(c, p + ".weight", ch);
  add_w(c, p + ".bias", ch);
  add_w(c, p + ".running_mean", ch);
  add_w(c, p + ".running_var", ch);
  return i;
}
ConvDesc mk_conv(opp_ctx* c, const std::string& name, int cout, int cin, int ks) {
  ConvDesc d;
  d.cin = cin;
  d.cout = cout;
  d.ks = ks;
  d.w_idx = add_w(c, name, (long long)cout * cin * ks * ks);
  return d;
}
// state-dict order of BasicBlock: conv1, conv2, bn1, bn2, [downsample.0, downsample.1]
BlockDesc mk_block(opp_ctx* c, const std::string& p, int cin, int cout, int stride) {
  BlockDesc b;
  b.conv1 = mk_conv(c, p + ".conv1.weight", cout, cin, 3);
  b.conv2 = mk_conv(c, p + ".conv2.weight", cout, cout, 3);
  b.conv1.bn_idx = add_bn(c, p + ".bn1", cout);
  b.conv1.bn_slot = (int)c->bn_names.size() - 1;
  b.conv2.bn_idx = add_bn(c, p + ".bn2", cout);
  b.conv2.bn_slot = (int)c->bn_names.size() - 1;
  if (stride != 1) {
    b.has_down = true;
    b.down = mk_conv(c, p + ".downsample.0.weight", cout, cin, 1);
    b.down.bn_idx = add_bn(c, p + ".downsample.1", cout);
    b.down.bn_slot = (int)c->bn_names.size() - 1;
  }
  return b;
}
void mk_transformer(opp_ctx* c, const std::string& name, int d, int n_layers, std::vector<EncLayerDesc>& out) {
  for (int i = 0; i < n_layers; ++i) {
    const std::string p = name + ".layers." + std::to_string(i);
    EncLayerDesc e;
    e.q_idx = add_w(c, p + ".q_proj.weight", (long long)d * d);
    add_w(c, p + ".k_proj.weight", (long long)d * d);
    add_w(c, p + ".v_proj.weight", (long long)d * d);
    add_w(c, p + ".merge.weight", (long long)d * d);
    add_w(c, p + ".mlp.0.weight", (long long)4 * d * d);
    add_w(c, p + ".mlp.2.weight", (long long)2 * d * d);
    add_w(c, p + ".norm1.weight", d);
    add_w(c, p + ".norm1.bias", d);
    add_w(c, p + ".norm2.weight", d);
    add_w(c, p + ".norm2.bias", d);
    out.push_back(e);
  }
}

// bump allocator over a caller-provided buffer
struct Arena {
  char* base;
  size_t cap, off = 0;
  bool ok = true;
  Arena(void* b, size_t c) : base((char*)b), cap(c) {}
  float* f(size_t n) { return (float*)raw(n * sizeof(float)); }
  void* raw(size_t bytes) {
    const size_t o = opp_align(off);
    if (base && o + bytes > cap) {
      ok = false;
      return base;
    }
    off = o + bytes;
    return base ? base + o : nullptr;
  }
};

}  // namespace

extern "C" int opp_create(const opp_config* cfg, opp_ctx** out) {
  OPP_CHECK_ARG(cfg && out, "opp_create: null argument");
  OPP_CHECK_ARG(cfg->initial_dim % 32 == 0 && cfg->initial_dim == cfg->block_dims[0],
                "unsupported backbone dims: initial_dim %d block_dims[0] %d", cfg->initial_dim, cfg->block_dims[0]);
  OPP_CHECK_ARG(cfg->block_dims[2] == cfg->coarse_d_model, "block_dims[2] must equal coarse d_model");
  OPP_CHECK_ARG(cfg->block_dims[0] == cfg->fine_d_model, "block_dims[0] must equal fine d_model");
  OPP_CHECK_ARG(cfg->coarse_d_model == 256 && cfg->coarse_nhead == 8, "coarse transformer must be d_model 256 / 8 heads");
  OPP_CHECK_ARG(cfg->fine_d_model == 128 && cfg->fine_nhead == 8, "fine transformer must be d_model 128 / 8 heads");
  OPP_CHECK_ARG(cfg->coarse_n_layers >= 0 && cfg->coarse_n_layers <= OPP_MAX_LAYERS && cfg->fine_n_layers >= 0 &&
                    cfg->fine_n_layers <= OPP_MAX_LAYERS, "too many transformer layers");
  OPP_CHECK_ARG(!cfg->kpt_enc_enable || (cfg->kpt_enc_dims[0] == 32 && cfg->kpt_enc_dims[1] == 64 && cfg->kpt_enc_dims[2] == 128),
                "keypoint encoder must be [32,64,128]");
  OPP_CHECK_ARG(cfg->gemm_precision >= 0 && cfg->gemm_precision <= 3, "gemm_precision must be 0..3");
#ifndef OPP_TUNING
  if (cfg->gemm_precision == 1 || cfg->gemm_precision == 2) {
    opp_set_error("gemm_precision %d (fp16x2, narrower than fp32) exists in the tuning library only (python -m onepose_plus_plus_amd.build --tuning)",
                  cfg->gemm_precision);
    return OPP_ERR_UNSUPPORTED;
  }
#endif
  OPP_CHECK_ARG(cfg->tile_policy == OPP_TILES_LATENCY || cfg->tile_policy == OPP_TILES_THROUGHPUT, "tile_policy must be 0 or 1");
  OPP_CHECK_ARG(cfg->encoder_fusion >= 0 && cfg->encoder_fusion <= 2, "encoder_fusion must be 0, 1 or 2");
  OPP_CHECK_ARG(cfg->fpn_overlap == 0 || cfg->fpn_overlap == 1, "fpn_overlap must be 0 or 1");
  OPP_CHECK_ARG(cfg->score_two_sweep >= 0 && cfg->score_two_sweep <= 2, "score_two_sweep must be 0, 1 or 2");
  OPP_CHECK_ARG(cfg->fine_window >= 1 && cfg->fine_window * cfg->fine_window <= 64 && (cfg->fine_window & 1), "bad fine window");
  opp_ctx* c = new opp_ctx();
  c->cfg = *cfg;
  const int d0 = cfg->initial_dim, d1 = cfg->block_dims[0], d2 = cfg->block_dims[1], d3 = cfg->block_dims[2];
  // --- same order as the reference state dict (backbone/resnet.py:88-124) ---
  c->stem = mk_conv(c, "backbone.conv1.weight", d0, 1, 7);
  c->stem.bn_idx = add_bn(c, "backbone.bn1", d0);
  c->stem.bn_slot = (int)c->bn_names.size() - 1;
  c->blocks[0] = mk_block(c, "backbone.layer1.0", d0, d1, 1);
  c->blocks[1] = mk_block(c, "backbone.layer1.1", d1, d1, 1);
  c->blocks[2] = mk_block(c, "backbone.layer2.0", d1, d2, 2);
  c->blocks[3] = mk_block(c, "backbone.layer2.1", d2, d2, 1);
  c->blocks[4] = mk_block(c, "backbone.layer3.0", d2, d3, 2);
  c->blocks[5] = mk_block(c, "backbone.layer3.1", d3, d3, 1);
  c->l3_out = mk_conv(c, "backbone.layer3_outconv.weight", d3, d3, 1);
  c->l2_out = mk_conv(c, "backbone.layer2_outconv.weight", d3, d2, 1);
  c->l2_out2a = mk_conv(c, "backbone.layer2_outconv2.0.weight", d3, d3, 3);
  c->l2_out2a.bn_idx = add_bn(c, "backbone.layer2_outconv2.1", d3);
  c->l2_out2a.bn_slot = (int)c->bn_names.size() - 1;
  c->l2_out2b = mk_conv(c, "backbone.layer2_outconv2.3.weight", d2, d3, 3);
  c->l1_out = mk_conv(c, "backbone.layer1_outconv.weight", d2, d1, 1);
  c->l1_out2a = mk_conv(c, "backbone.layer1_outconv2.0.weight", d2, d2, 3);
  c->l1_out2a.bn_idx = add_bn(c, "backbone.layer1_outconv2.1", d2);
  c->l1_out2a.bn_slot = (int)c->bn_names.size() - 1;
  c->l1_out2b = mk_conv(c, "backbone.layer1_outconv2.3.weight", d1, d2, 3);
  if (cfg->kpt_enc_enable) {  // utils/position_encoding.py:62-79 -> encoder.{0,3,6,9}
    const int ch[5] = {3, cfg->kpt_enc_dims[0], cfg->kpt_enc_dims[1], cfg->kpt_enc_dims[2], cfg->coarse_d_model};
    for (int i = 0; i < 4; ++i) {
      const std::string p = "kpt_3d_pos_encoding.encoder." + std::to_string(3 * i);
      const int idx = add_w(c, p + ".weight", (long long)ch[i + 1] * ch[i]);
      add_w(c, p + ".bias", ch[i + 1]);
      if (i == 0) c->kpt_idx = idx;
    }
  }
  mk_transformer(c, "loftr_coarse", cfg->coarse_d_model, cfg->coarse_n_layers, c->coarse);
  mk_transformer(c, "loftr_fine", cfg->fine_d_model, cfg->fine_n_layers, c->fine);
  *out = c;
  return OPP_OK;
}

extern "C" void opp_destroy(opp_ctx* ctx) {
  if (!ctx) return;
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  delete ctx;
}
extern "C" int opp_set_conv_tail(opp_ctx* ctx, int on) {
  OPP_CHECK_ARG(ctx && (on == 0 || on == 1), "set_conv_tail: on must be 0 or 1");
  ctx->conv_tail = on != 0;
  return OPP_OK;
}
extern "C" int opp_set_query_mask(opp_ctx* ctx, const float* mask) {
  OPP_CHECK_ARG(ctx, "set_query_mask: null ctx");
  ctx->query_mask = mask;
  return OPP_OK;
}
extern "C" int opp_set_keypoint_extent_ref(opp_ctx* ctx, const float* kpts0, int n0) {
  OPP_CHECK_ARG(ctx && (kpts0 == nullptr || n0 > 0), "set_keypoint_extent_ref: bad argument");
  ctx->kpt_extent_ref = kpts0;
  ctx->kpt_extent_n = kpts0 ? n0 : 0;
  return OPP_OK;
}
extern "C" int opp_set_status_flag(opp_ctx* ctx, int* flag) {
  OPP_CHECK_ARG(ctx, "set_status_flag: null ctx");
  ctx->status_flag = flag;
  return OPP_OK;
}
extern "C" int opp_num_weights(const opp_ctx* ctx) { return ctx ? (int)ctx->table.size() : 0; }
extern "C" const char* opp_weight_name(const opp_ctx* ctx, int i) {
  return (ctx && i >= 0 && i < (int)ctx->table.size()) ? ctx->table[i].name.c_str() : nullptr;
}
extern "C" long long opp_weight_numel(const opp_ctx* ctx, int i) {
  return (ctx && i >= 0 && i < (int)ctx->table.size()) ? ctx->table[i].numel : -1;
}
extern "C" int opp_num_bn_layers(const opp_ctx* ctx) { return ctx ? (int)ctx->bn_names.size() : 0; }
extern "C" const char* opp_bn_layer_name(const opp_ctx* ctx, int i) {
  return (ctx && i >= 0 && i < (int)ctx->bn_names.size()) ? ctx->bn_names[i].c_str() : nullptr;
}
extern "C" int opp_bn_layer_channels(const opp_ctx* ctx, int i) {
  return (ctx && i >= 0 && i < (int)ctx->bn_names.size()) ? ctx->bn_channels[i] : -1;
}

// ----------------------------------------------------------------------------------------
// weight packing
// ----------------------------------------------------------------------------------------
namespace {

std::vector<ConvDesc*> all_convs(opp_ctx* c) {
  std::vector<ConvDesc*> v;
  for (auto& b : c->blocks) {
    v.push_back(&b.conv1);
    v.push_back(&b.conv2);
    if (b.has_down) v.push_back(&b.down);
  }
  for (ConvDesc* d : {&c->l3_out, &c->l2_out, &c->l2_out2a, &c->l2_out2b, &c->l1_out, &c->l1_out2a, &c->l1_out2b}) v.push_back(d);
  return v;
}

// lays out (and, with a real base pointer, assigns) every packed tensor
size_t plan_pack(opp_ctx* c, void* base) {
  Arena a(base, (size_t)-1);
  c->scratch_scale = a.f(256);
  {
    size_t mx = (size_t)c->stem.cout * 64;
    for (ConvDesc* d : all_convs(c)) mx = d->w_floats() > mx ? d->w_floats() : mx;
    const size_t dc = c->cfg.coarse_d_model, df = c->cfg.fine_d_model;
    mx = 4 * dc * dc > mx ? 4 * dc * dc : mx;
    mx = 4 * df * df > mx ? 4 * df * df : mx;
    c->scratch_h2 = c->cfg.gemm_precision ? a.f(split_floats(mx, gemm_prec(c->cfg))) : nullptr;
  }
  const int prec = gemm_prec(c->cfg);
  const bool h2 = prec == OPP_PREC_FP16X2;   // per-matrix power-of-two scales exist for fp16x2 only
  auto wf = [&](size_t n) { return a.f(split_floats(n, prec)); };
  c->stem.w = wf((size_t)c->stem.cout * 64);
  c->stem.bias = a.f(pad32(c->stem.cout));
  c->stem.h2s = h2 ? a.f(2) : nullptr;
  for (ConvDesc* d : all_convs(c)) {
    d->h2s = h2 ? a.f(2) : nullptr;
    d->w = wf(d->w_floats());
    d->bias = d->bn_idx >= 0 ? a.f(d->cout_pad()) : nullptr;
    d->w_tail = (prec == OPP_PREC_BF16X3 && d->tail_cols()) ? a.f(opp_conv_tail_weight_floats(d->cin_pad(), d->ks)) : nullptr;
  }
  if (c->cfg.kpt_enc_enable) {
    const int ch[5] = {3, c->cfg.kpt_enc_dims[0], c->cfg.kpt_enc_dims[1], c->cfg.kpt_enc_dims[2], c->cfg.coarse_d_model};
    for (int i = 0; i < 4; ++i) {
      c->kpt_wt[i] = a.f((size_t)ch[i] * ch[i + 1]);
      c->kpt_b[i] = a.f(ch[i + 1]);
    }
  }
  auto plan_tr = [&](std::vector<EncLayerDesc>& L, int d) {
    for (auto& e : L) {
      e.wqkv = wf((size_t)3 * d * d);
      e.wmerge = wf((size_t)d * d);
      e.w1 = wf((size_t)4 * d * d);
      e.w2 = wf((size_t)2 * d * d);
      e.sqkv = h2 ? a.f(2) : nullptr;
      e.smerge = h2 ? a.f(2) : nullptr;
      e.s1 = h2 ? a.f(2) : nullptr;
      e.s2 = h2 ? a.f(2) : nullptr;
      e.g1 = a.f(d);
      e.b1 = a.f(d);
      e.g2 = a.f(d);
      e.b2 = a.f(d);
      if (prec == OPP_PREC_BF16X3) {
        e.fmerge = a.raw(opp_frag_b3_bytes(d, d));
        e.f1 = a.raw(opp_frag_b3_bytes(2 * d, 2 * d));
        e.f2 = a.raw(opp_frag_b3_bytes(d, 2 * d));
        e.fqkv = d == 256 ? a.raw(opp_frag_b3_bytes(3 * d, d)) : nullptr;
      }
    }
  };
  plan_tr(c->coarse, c->cfg.coarse_d_model);
  plan_tr(c->fine, c->cfg.fine_d_model);
  return opp_align(a.off);
}

int copy_f(float* dst, const float* src, size_t n, hipStream_t s) {
  if (hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
    opp_set_error("pack: device copy failed");
    return OPP_ERR_LAUNCH;
  }
  return OPP_OK;
}

}  // namespace

extern "C" size_t opp_packed_weights_bytes(const opp_ctx* ctx) {
  if (!ctx) return 0;
  opp_ctx tmp = *ctx;  // plan on a copy (null base -> offsets only)
  return plan_pack(&tmp, nullptr);
}

extern "C" int opp_pack_weights(opp_ctx* c, const float* const* w, int n, void* packed, size_t bytes, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  OPP_CHECK_ARG(c && w && packed, "pack: null argument");
  OPP_CHECK_ARG(n == (int)c->table.size(), "pack: expected %d weight tensors, got %d", (int)c->table.size(), n);
  const size_t need = plan_pack(c, packed);
  OPP_CHECK_ARG(bytes >= need, "pack: blob too small (%zu < %zu)", bytes, need);
  for (int i = 0; i < n; ++i) OPP_CHECK_ARG(w[i] != nullptr, "pack: weight %d (%s) is null", i, c->table[i].name.c_str());
  const float eps = 1e-5f;
  auto fold = [&](const ConvDesc& d, float* scale) -> int {
    const int b = d.bn_idx;
    return opp_fold_bn(w[b], w[b + 1], w[b + 2], w[b + 3], eps, d.cout, pad32(d.cout), scale, d.bias, s);
  };
  // A packed matrix goes through the staging buffer when the arithmetic pre-splits it: pack -> staging, split -> its place (no copy back)
  const int prec = gemm_prec(c->cfg);
  auto place = [&](float* dst, size_t nfl, float* sc, auto&& pack_into) -> int {
    if (!c->cfg.gemm_precision) return pack_into(dst);
    OPP_TRY(pack_into(c->scratch_h2));
    return prec == OPP_PREC_BF16X3 ? opp_b3_split(c->scratch_h2, dst, nfl, s) : opp_h2_split(c->scratch_h2, dst, nfl, sc, s);
  };
  // scope 1 (a training step): the BatchNorm-folded packing is not produced -- train() forwards read the raw weights of
  // opp_pack_train_weights, the convolutions without BatchNorm are shared
  const bool with_bn = c->pack_scope == 0;
  if (with_bn) {   // stem (resnet.py:101-103)
    OPP_TRY(fold(c->stem, c->scratch_scale));
    OPP_TRY(place(c->stem.w, (size_t)c->stem.cout * 64, c->stem.h2s,
                  [&](float* dst) { return opp_pack_stem(w[c->stem.w_idx], c->scratch_scale, c->stem.cout, dst, s); }));
  }
  for (ConvDesc* d : all_convs(c)) {
    const float* scale = nullptr;
    if (d->bn_idx >= 0) {
      if (!with_bn) continue;
      OPP_TRY(fold(*d, c->scratch_scale));
      scale = c->scratch_scale;
    }
    OPP_TRY(place(d->w, d->w_floats(), d->h2s, [&](float* dst) {
      return opp_pack_conv(w[d->w_idx], scale, d->cout, d->cin, d->ks, d->cout_pad(), d->cin_pad(), dst, s);
    }));
    if (d->w_tail) OPP_TRY(opp_pack_conv_tail(w[d->w_idx], scale, d->cout, d->cin, d->ks, d->body_cols(), d->cin_pad(), d->w_tail, s));
  }
  const bool with_tr = c->pack_scope == 0;
  if (c->cfg.kpt_enc_enable && with_tr) {
    const int ch[5] = {3, c->cfg.kpt_enc_dims[0], c->cfg.kpt_enc_dims[1], c->cfg.kpt_enc_dims[2], c->cfg.coarse_d_model};
    for (int i = 0; i < 4; ++i) {
      // PyTorch Linear weight [Cout][Cin] -> [Cin][Cout]
      OPP_TRY(opp_transpose(w[c->kpt_idx + 2 * i], c->kpt_wt[i], 1, ch[i + 1], ch[i], s));
      OPP_TRY(copy_f(c->kpt_b[i], w[c->kpt_idx + 2 * i + 1], ch[i + 1], s));
    }
  }
  auto pack_tr = [&](std::vector<EncLayerDesc>& L, int d) -> int {
    for (auto& e : L) {
      const int q = e.q_idx;
      const size_t dd = (size_t)d * d;
      OPP_TRY(copy_f(e.wqkv, w[q], dd, s));
      OPP_TRY(copy_f(e.wqkv + dd, w[q + 1], dd, s));
      OPP_TRY(copy_f(e.wqkv + 2 * dd, w[q + 2], dd, s));
      OPP_TRY(copy_f(e.wmerge, w[q + 3], dd, s));
      OPP_TRY(copy_f(e.w1, w[q + 4], 4 * dd, s));
      OPP_TRY(copy_f(e.w2, w[q + 5], 2 * dd, s));
      OPP_TRY(copy_f(e.g1, w[q + 6], d, s));
      OPP_TRY(copy_f(e.b1, w[q + 7], d, s));
      OPP_TRY(copy_f(e.g2, w[q + 8], d, s));
      OPP_TRY(copy_f(e.b2, w[q + 9], d, s));
      if (e.fqkv) OPP_TRY(opp_pack_frag_b3(e.wqkv, 3 * d, d, e.fqkv, s));     // (e.wqkv is still the fp32 concatenation here)
      if (e.fmerge) {
        OPP_TRY(opp_pack_frag_b3(w[q + 3], d, d, e.fmerge, s));
        OPP_TRY(opp_pack_frag_b3(w[q + 4], 2 * d, 2 * d, e.f1, s));
        OPP_TRY(opp_pack_frag_b3(w[q + 5], d, 2 * d, e.f2, s));
      }
    }
    return OPP_OK;
  };
  if (with_tr) {
    OPP_TRY(pack_tr(c->coarse, c->cfg.coarse_d_model));
    OPP_TRY(pack_tr(c->fine, c->cfg.fine_d_model));
  }
  if (c->cfg.gemm_precision) {   // pre-split the transformer's GEMM weight matrices: fp16 hi/lo (same footprint) or bf16 hi/mid/lo (1.5x)
    auto split = [&](float* wm, size_t n, float* sc) -> int {
      if (prec == OPP_PREC_BF16X3) OPP_TRY(opp_b3_split(wm, c->scratch_h2, n, s));
      else OPP_TRY(opp_h2_split(wm, c->scratch_h2, n, sc, s));
      return copy_f(wm, c->scratch_h2, split_floats(n, prec), s);
    };
    for (auto* L : {&c->coarse, &c->fine}) {
      const size_t d = (L == &c->coarse) ? c->cfg.coarse_d_model : c->cfg.fine_d_model;
      if (!with_tr) break;
      for (auto& e : *L) {
        OPP_TRY(split(e.wqkv, 3 * d * d, e.sqkv));
        OPP_TRY(split(e.wmerge, d * d, e.smerge));
        OPP_TRY(split(e.w1, 4 * d * d, e.s1));
        OPP_TRY(split(e.w2, 2 * d * d, e.s2));
      }
    }
  }
  c->packed = true;
  c->tr_packed = with_tr;
  c->bn_packed = with_bn;
  c->packed_bytes = need;
  return OPP_OK;
}

extern "C" int opp_set_pack_scope(opp_ctx* ctx, int scope) {
  OPP_CHECK_ARG(ctx && (scope == 0 || scope == 1), "set_pack_scope: scope must be 0 (everything) or 1 (training step: raw backbone weights only)");
  ctx->pack_scope = scope;
  return OPP_OK;
}

// ----------------------------------------------------------------------------------------
// training-mode weights: the backbone convolutions WITHOUT the folded BatchNorm
// ----------------------------------------------------------------------------------------
namespace {

size_t plan_pack_train(opp_ctx* c, void* base) {
  Arena a(base, (size_t)-1);
  const int prec = gemm_prec(c->cfg);
  const bool h2 = prec == OPP_PREC_FP16X2;
  c->stem.w_train = a.f(split_floats((size_t)c->stem.cout * 64, prec));
  c->stem.h2s_train = h2 ? a.f(2) : nullptr;
  for (ConvDesc* d : all_convs(c)) {
    if (d->bn_idx < 0) continue;      // no BatchNorm behind it: the eval packing is already the raw weight
    d->w_train = a.f(split_floats(d->w_floats(), prec));
    d->h2s_train = h2 ? a.f(2) : nullptr;
    d->w_tail_train = (prec == OPP_PREC_BF16X3 && d->tail_cols()) ? a.f(opp_conv_tail_weight_floats(d->cin_pad(), d->ks)) : nullptr;
  }
  return opp_align(a.off);
}

}  // namespace

extern "C" size_t opp_packed_train_weights_bytes(const opp_ctx* ctx) {
  if (!ctx) return 0;
  opp_ctx tmp = *ctx;
  return plan_pack_train(&tmp, nullptr);
}

extern "C" int opp_pack_train_weights(opp_ctx* c, const float* const* w, int n, void* packed, size_t bytes, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  OPP_CHECK_ARG(c && w && packed, "pack_train: null argument");
  OPP_CHECK_ARG(c->packed, "pack_train: call opp_pack_weights first (the layers without BatchNorm share its packing)");
  OPP_CHECK_ARG(n == (int)c->table.size(), "pack_train: expected %d weight tensors, got %d", (int)c->table.size(), n);
  const size_t need = plan_pack_train(c, packed);
  OPP_CHECK_ARG(bytes >= need, "pack_train: blob too small (%zu < %zu)", bytes, need);
  const int prec = gemm_prec(c->cfg);
  auto place = [&](float* dst, size_t nfl, float* sc, auto&& pack_into) -> int {   // as in opp_pack_weights: pack -> staging, split -> its place
    if (prec == OPP_PREC_FP32) return pack_into(dst);
    OPP_TRY(pack_into(c->scratch_h2));
    return prec == OPP_PREC_BF16X3 ? opp_b3_split(c->scratch_h2, dst, nfl, s) : opp_h2_split(c->scratch_h2, dst, nfl, sc, s);
  };
  OPP_TRY(place(c->stem.w_train, (size_t)c->stem.cout * 64, c->stem.h2s_train,
                [&](float* dst) { return opp_pack_stem(w[c->stem.w_idx], nullptr, c->stem.cout, dst, s); }));
  c->stem.gamma = w[c->stem.bn_idx];
  c->stem.beta = w[c->stem.bn_idx + 1];
  for (ConvDesc* d : all_convs(c)) {
    if (d->bn_idx < 0) continue;
    OPP_TRY(place(d->w_train, d->w_floats(), d->h2s_train, [&](float* dst) {
      return opp_pack_conv(w[d->w_idx], nullptr, d->cout, d->cin, d->ks, d->cout_pad(), d->cin_pad(), dst, s);
    }));
    if (d->w_tail_train) OPP_TRY(opp_pack_conv_tail(w[d->w_idx], nullptr, d->cout, d->cin, d->ks, d->body_cols(), d->cin_pad(), d->w_tail_train, s));
    d->gamma = w[d->bn_idx];          // caller-owned: must outlive the training forwards
    d->beta = w[d->bn_idx + 1];
  }
  c->train_packed = true;
  return OPP_OK;
}

// ----------------------------------------------------------------------------------------
// backbone
// ----------------------------------------------------------------------------------------
namespace {

// split-K scratch of the convolutions this host thread is enqueuing (backbone_impl sets it per stream branch; null = never split)
constexpr size_t kSplitKScratchFloats = (size_t)4 << 20;      // 4 slices of <= 64 tiles of 128 x 128
thread_local float* t_splitk_ws = nullptr;
thread_local size_t t_splitk_ws_floats = 0;
struct SplitKScope {
  float* prev;
  size_t prev_floats;
  explicit SplitKScope(float* ws, size_t floats = kSplitKScratchFloats) : prev(t_splitk_ws), prev_floats(t_splitk_ws_floats) {
    t_splitk_ws = ws;
    t_splitk_ws_floats = ws ? floats : 0;
  }
  ~SplitKScope() {
    t_splitk_ws = prev;
    t_splitk_ws_floats = prev_floats;
  }
};
// would the DENSE convolution d over `pixels` output pixels run as K slices?  The launcher's own shape rule (opp_conv_splitk_by_shape,
// gemm_mfma.hip) on the dense problem; its remaining conditions (bf16x3, scratch present, plain epilogue, full-width rows) hold for the
// two layers asked about whenever the dense map is evaluated through opp_forward_coarse.
bool conv_splits_by_shape(const ConvDesc& d, size_t pixels, int prec) {
  return prec == OPP_PREC_BF16X3 && opp_conv_splitk_by_shape((long long)pixels, d.cout_pad(), d.k_len());
}

// pad < 0: "same" padding ks / 2 (every convolution of the reference); pad = 0: a VALID convolution over Bn small patches (match-driven
// fine branch: the out-of-image taps are explicit zero rows of the patch)
// Pre-split activations (bf16x3 inference backbone, r06; gemm_mfma.hip "ASP"): a convolution whose K walk has no packed tail (Cin % 32 == 0 real
// channels, or any 1 x 1) can take its input as the [hi x8 | mid x8 | lo x8] rows its PRODUCER's epilogue wrote (x3; 6 bytes per channel,
// padded channels are zeros) and skip the operand split in its K loop; y3 != null asks this convolution's epilogue for such rows of its own
// output (y may then be null).  Same bits as the split done in the K loop: results do not change (tests/test_kernels_gpu.py).
// OPP_ASP=1 switches the chain on (tests / tools; read once per entry call).  OFF by default: measured r06 (profiles/r06_asp_*.txt) a
// convolution reading pre-split rows is 3-5 % faster, one WRITING them 3-12 % slower (6 instead of 4 bytes per value, 8-byte stores), and
// the whole forward loses 3-4 % images/s with the chain on.
inline bool asp_env_on() {
  const char* e = getenv("OPP_ASP");
  return e && e[0] == '1';
}
thread_local bool t_asp_on = false;
struct AspScope {
  bool prev;
  AspScope() : prev(t_asp_on) { t_asp_on = asp_env_on(); }
  explicit AspScope(bool on) : prev(t_asp_on) { t_asp_on = on; }
  ~AspScope() { t_asp_on = prev; }
};
bool conv_takes_split(const ConvDesc& d, int prec) {
  return t_asp_on && prec == OPP_PREC_BF16X3 && opp_conv_tail_grp(d.cin, d.ks) == 0 && !(t_conv_tail && d.w_tail != nullptr);
}
inline size_t split_row_bytes(int c_pad) { return (size_t)c_pad * 6; }

int run_conv(const float* x, int Hin, int Win, const ConvDesc& d, int stride, const float* res, int res_mode, int act,
             float* y, hipStream_t s, int h2, int tile_cfg = -1, int Bn = 1, bool raw = false, int pad = -1, int splitk_force = -1,
             const void* x3 = nullptr, void* y3 = nullptr) {
  OppGemm g;
  const bool a_split = x3 != nullptr && !raw && conv_takes_split(d, h2);
  OPP_CHECK_ARG(a_split || x != nullptr, "conv: no fp32 input for a convolution that cannot take the pre-split one");
  OPP_CHECK_ARG(y3 == nullptr || h2 == OPP_PREC_BF16X3, "conv: pre-split output is a bf16x3 feature");
  static const bool splitk_on = !(getenv("OPP_CONV_SPLITK") && getenv("OPP_CONV_SPLITK")[0] == '0');   // A/B switch of the tools
  g.splitk_ws = splitk_on ? t_splitk_ws : nullptr;
  g.splitk_ws_floats = g.splitk_ws ? t_splitk_ws_floats : 0;
  g.splitk_force = splitk_force;
  g.nonfinite = t_status_flag;
  g.tile_policy = t_tile_policy;
  g.conv = 1;
  g.prec = h2;
  // raw = training mode: unfolded weights, no BatchNorm shift (bn_train applies the batch statistics afterwards)
  const float* h2s = raw ? d.h2s_train : d.h2s;
  g.h2_inv = (h2 == OPP_PREC_FP16X2 && h2s) ? h2s + 1 : nullptr;
  g.A0 = a_split ? static_cast<const float*>(x3) : x;
  g.a_split = a_split ? 1 : 0;
  g.C3 = y3;
  g.ld3 = (int)split_row_bytes(d.cout_pad());
  g.Bn = Bn;
  g.Hin = Hin;
  g.Win = Win;
  g.Cin = d.cin_pad();
  g.ksize = d.ks;
  g.stride = stride;
  g.pad = pad < 0 ? d.ks / 2 : pad;
  g.Hout = (Hin + 2 * g.pad - d.ks) / stride + 1;
  g.Wout = (Win + 2 * g.pad - d.ks) / stride + 1;
  g.W = raw ? d.w_train : d.w;
  g.K = d.k_len();
  g.tail_grp = opp_conv_tail_grp(d.cin, d.ks);
  g.ldw = (int)split_floats((size_t)g.K, h2);
  g.M = Bn * g.Hout * g.Wout;
  g.N = d.cout_pad();
  g.C = y;
  g.ldc = d.cout_pad();
  g.n_store = d.cout_pad();
  g.n_real = d.cout;
  g.bias = raw ? nullptr : d.bias;
  g.res_mode = res_mode;
  g.R = res;
  g.ldr = d.cout_pad();
  if (res_mode == OPP_RES_BILINEAR2X) {
    g.Hr = g.Hout / 2;
    g.Wr = g.Wout / 2;
    // align_corners=True source index scale (in-1)/(out-1), resnet.py:151,155
    g.res_sy = g.Hout > 1 ? (float)(g.Hr - 1) / (float)(g.Hout - 1) : 0.f;
    g.res_sx = g.Wout > 1 ? (float)(g.Wr - 1) / (float)(g.Wout - 1) : 0.f;
  }
  g.act = act;
  g.alg_flops = 2.0 * (double)g.M * (double)d.cout * (double)(d.ks * d.ks * d.cin);
  // 196-channel layers (bf16x3): 192 columns on the MFMA kernel -- no padded sub-tile -- and the last 4 (+ the zero padding channels of
  // the 224-channel row) on the vector ALU.  Shape-only decision: every tile policy, batch size and the match-driven patches take it.
  const float* wt = raw ? d.w_tail_train : d.w_tail;
  if (t_conv_tail && h2 == OPP_PREC_BF16X3 && wt != nullptr && tile_cfg < 0) {
    OppGemm body = g;
    body.N = d.body_cols();
    body.n_store = d.body_cols();
    body.alg_flops = 2.0 * (double)g.M * (double)d.body_cols() * (double)(d.ks * d.ks * d.cin);
    OPP_TRY(opp_gemm_launch_cfg(body, tile_cfg, s));
    return opp_conv_tail(g, wt, d.body_cols(), d.tail_cols(), s);
  }
  return opp_gemm_launch_cfg(g, tile_cfg, s);
}

// BasicBlock.forward (resnet.py:37-45)
// x3 / tmp3 / y3: the pre-split twins of x / tmp / y (null = not kept).  A map is written in fp32 only where something reads it in fp32
// (a shortcut, a convolution with a packed K tail, the caller): y = null with y3 set drops the fp32 copy, and tmp is fp32 only when conv2
// cannot take the split rows.
int run_block(const float* x, int Hin, int Win, const BlockDesc& b, int stride, float* tmp, float* ds, float* y,
              hipStream_t s, int h2, const void* x3 = nullptr, void* tmp3 = nullptr, void* y3 = nullptr) {
  const int Ho = Hin / stride, Wo = Win / stride;
  const bool t_split = tmp3 != nullptr && conv_takes_split(b.conv2, h2);
  OPP_TRY(run_conv(x, Hin, Win, b.conv1, stride, nullptr, OPP_RES_NONE, OPP_ACT_RELU, t_split ? nullptr : tmp, s, h2, -1, 1, false, -1, -1, x3,
                   t_split ? tmp3 : nullptr));
  const float* shortcut = x;
  if (b.has_down) {
    OPP_TRY(run_conv(x, Hin, Win, b.down, stride, nullptr, OPP_RES_NONE, OPP_ACT_NONE, ds, s, h2, -1, 1, false, -1, -1, x3));
    shortcut = ds;
  }
  OPP_CHECK_ARG(shortcut != nullptr, "block: the shortcut needs the fp32 input map");
  return run_conv(t_split ? nullptr : tmp, Ho, Wo, b.conv2, 1, shortcut, OPP_RES_DIRECT, OPP_ACT_RELU, y, s, h2, -1, 1, false, -1, -1,
                  t_split ? tmp3 : nullptr, y3);
}

struct BackboneBufs {
  float *col, *x0, *t1, *x1a, *x1, *t2, *ds2, *x2a, *x2, *t3, *ds3, *x3a, *x3, *l2, *u2, *x2o, *l1, *u1;
  float *sk1 = nullptr, *sk2 = nullptr;   // split-K scratch of the two stream branches (they may run concurrently: opp_config.fpn_overlap)
  // bf16x3: pre-split twins (6 B per channel) of the maps whose consumers can take them (run_conv: conv_takes_split); null otherwise
  void *x0s = nullptr, *t1s = nullptr, *x1as = nullptr, *x1s = nullptr, *x2s = nullptr, *t3s = nullptr, *x3as = nullptr, *x3s = nullptr, *l2s = nullptr,
       *u2s = nullptr;
};

size_t plan_backbone(const opp_ctx* c, int H, int W, Arena& a, BackboneBufs& b) {
  const size_t p2 = (size_t)(H / 2) * (W / 2), p4 = (size_t)(H / 4) * (W / 4), p8 = (size_t)(H / 8) * (W / 8);
  const int c1 = pad32(c->cfg.block_dims[0]), c2 = pad32(c->cfg.block_dims[1]), c3 = pad32(c->cfg.block_dims[2]);
  b.col = a.f(p2 * 64);
  b.x0 = a.f(p2 * c1);
  b.t1 = a.f(p2 * c1);
  b.x1a = a.f(p2 * c1);
  b.x1 = a.f(p2 * c1);
  b.t2 = a.f(p4 * c2);
  b.ds2 = a.f(p4 * c2);
  b.x2a = a.f(p4 * c2);
  b.x2 = a.f(p4 * c2);
  b.t3 = a.f(p8 * c3);
  b.ds3 = a.f(p8 * c3);
  b.x3a = a.f(p8 * c3);
  b.x3 = a.f(p8 * c3);
  b.l2 = a.f(p4 * c3);
  b.u2 = a.f(p4 * c3);
  b.x2o = a.f(p4 * c2);
  b.l1 = a.f(p2 * c2);
  b.u1 = a.f(p2 * c2);
  b.sk1 = a.f(kSplitKScratchFloats);
  b.sk2 = a.f(kSplitKScratchFloats);
  if (gemm_prec(c->cfg) == OPP_PREC_BF16X3 && asp_env_on()) {
    b.x0s = a.raw(p2 * split_row_bytes(c1));
    b.t1s = a.raw(p2 * split_row_bytes(c1));
    b.x1as = a.raw(p2 * split_row_bytes(c1));
    b.x1s = a.raw(p2 * split_row_bytes(c1));
    b.x2s = a.raw(p4 * split_row_bytes(c2));
    b.t3s = a.raw(p8 * split_row_bytes(c3));
    b.x3as = a.raw(p8 * split_row_bytes(c3));
    b.x3s = a.raw(p8 * split_row_bytes(c3));
    b.l2s = a.raw(p4 * split_row_bytes(c3));
    b.u2s = a.raw(p4 * split_row_bytes(c3));
  }
  return a.off;
}

// phase 0: the whole ResNetFPN_8_2.forward; 1: stem .. layer3 + layer3_outconv (-> feat_c, the coarse map);
// 2: the FPN fine branch (-> feat_f), which needs only x1, x2 and feat_c of phase 1 -- the coarse level does not depend on it;
// 3: the 1/4-resolution half of that branch only (-> x2_out); 4: its 1/2-resolution half (x1, x2_out -> feat_f; bufs prepared by the caller).
// x1_ext / x2o_ext: caller-owned buffers that receive x1 / x2_out instead of the workspace (match-driven fine branch).
int backbone_impl(opp_ctx* c, const float* image, int H, int W, float* feat_c, float* feat_f, Arena& a, hipStream_t s, int phase = 0,
                  BackboneBufs* bufs = nullptr, float* x1_ext = nullptr, float* x2o_ext = nullptr) {
  AspScope asp_scope;
  OPP_CHECK_ARG(c && c->packed, "backbone: weights not packed");
  OPP_CHECK_ARG(c->bn_packed, "backbone: weights were packed with scope 1 (training step: no BatchNorm-folded convolutions); repack with opp_set_pack_scope(ctx, 0)");
  OPP_CHECK_ARG(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "backbone: H,W must be multiples of 8 (got %dx%d)", H, W);
  BackboneBufs local;
  BackboneBufs& b = bufs ? *bufs : local;
  if (phase == 0 || phase == 1) {
    plan_backbone(c, H, W, a, b);
    if (!a.ok) {
      opp_set_error("backbone: workspace too small");
      return OPP_ERR_WORKSPACE;
    }
    if (x1_ext) b.x1 = x1_ext;
    if (x2o_ext) b.x2o = x2o_ext;
  }
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
  const int hp = gemm_prec(c->cfg);
  // pre-split twin of a map: kept when one of its consumers takes it (conv_takes_split); the fp32 copy: dropped when all of them do
  auto twin = [&](void* buf, std::initializer_list<const ConvDesc*> consumers) -> void* {
    if (buf)
      for (const ConvDesc* d : consumers)
        if (conv_takes_split(*d, hp)) return buf;
    return nullptr;
  };
  auto fp32_of = [&](float* buf, void* tw, std::initializer_list<const ConvDesc*> consumers) -> float* {
    if (!tw) return buf;
    for (const ConvDesc* d : consumers)
      if (!conv_takes_split(*d, hp)) return buf;
    return nullptr;
  };
  const BlockDesc* B = c->blocks;
  if (phase == 0 || phase == 1) {
    SplitKScope sk_scope(b.sk1);
    // stem: conv7x7/s2 + BN + ReLU (resnet.py:143): one direct kernel (bf16x3), else im2col + GEMM -- bit-identical
    const char* stem_env = getenv("OPP_STEM_DIRECT");          // A/B switch of the tests / tools
    void* x0s = nullptr;
    if (opp_stem_direct_ok(c->stem.cout, hp) && pad32(c->stem.cout) == c->stem.cout && !(stem_env && stem_env[0] == '0') &&
        (size_t)H2 * W2 * pad32(c->stem.cout) < (1ull << 31)) {
      x0s = twin(b.x0s, {&B[0].conv1});
      OPP_TRY(opp_stem_direct(image, H, W, c->stem.w, c->stem.bias, b.x0, pad32(c->stem.cout), s, x0s, (int)split_row_bytes(pad32(c->stem.cout))));
    } else {
      OPP_TRY(opp_stem_im2col(image, 1, H, W, b.col, s));
      OppGemm g;
      g.nonfinite = t_status_flag;
      g.tile_policy = t_tile_policy;
      g.A0 = b.col;
      g.lda0 = 64;
      g.ksplit = 64;
      g.W = c->stem.w;
      g.ldw = (int)split_floats(64, hp);
      g.M = H2 * W2;
      g.N = c->stem.cout;
      g.K = 64;
      g.C = b.x0;
      g.ldc = pad32(c->stem.cout);
      g.n_store = pad32(c->stem.cout);
      g.bias = c->stem.bias;
      g.act = OPP_ACT_RELU;
      g.prec = hp;
      g.h2_inv = hp == OPP_PREC_FP16X2 ? c->stem.h2s + 1 : nullptr;
      OPP_TRY(opp_gemm_launch(g, s));
    }
    // (x0, x1a, x3a are shortcuts and x1 / x2 may leave through x1_ext / feed a convolution with a packed K tail: those keep their fp32 copy)
    void* x1as = twin(b.x1as, {&B[1].conv1});
    OPP_TRY(run_block(b.x0, H2, W2, B[0], 1, b.t1, nullptr, b.x1a, s, hp, x0s, b.t1s, x1as));   // layer1 (:144)
    b.x1s = twin(b.x1s, {&B[2].conv1, &B[2].down, &c->l1_out});
    OPP_TRY(run_block(b.x1a, H2, W2, B[1], 1, b.t1, nullptr, b.x1, s, hp, x1as, b.t1s, b.x1s));
    OPP_TRY(run_block(b.x1, H2, W2, B[2], 2, b.t2, b.ds2, b.x2a, s, hp, b.x1s));                // layer2 (:145)
    b.x2s = twin(b.x2s, {&B[4].conv1, &B[4].down, &c->l2_out});
    OPP_TRY(run_block(b.x2a, H4, W4, B[3], 1, b.t2, nullptr, b.x2, s, hp, nullptr, nullptr, b.x2s));
    void* x3as = twin(b.x3as, {&B[5].conv1});
    OPP_TRY(run_block(b.x2, H4, W4, B[4], 2, b.t3, b.ds3, b.x3a, s, hp, b.x2s, b.t3s, x3as));   // layer3 (:146)
    void* x3s = twin(b.x3s, {&c->l3_out});
    float* x3f = fp32_of(b.x3, x3s, {&c->l3_out});
    OPP_TRY(run_block(b.x3a, H8, W8, B[5], 1, b.t3, nullptr, x3f, s, hp, x3as, b.t3s, x3s));
    // FPN (:149-157)
    OPP_TRY(run_conv(x3f, H8, W8, c->l3_out, 1, nullptr, OPP_RES_NONE, OPP_ACT_NONE, feat_c, s, hp, -1, 1, false, -1, -1, x3s));
  }
  if (phase == 0 || phase == 2 || phase == 3) {
    SplitKScope sk_scope(b.sk2);
    void* l2s = twin(b.l2s, {&c->l2_out2a});
    float* l2f = fp32_of(b.l2, l2s, {&c->l2_out2a});
    OPP_TRY(run_conv(b.x2, H4, W4, c->l2_out, 1, feat_c, OPP_RES_BILINEAR2X, OPP_ACT_NONE, l2f, s, hp, -1, 1, false, -1, -1, b.x2s, l2s));
    void* u2s = twin(b.u2s, {&c->l2_out2b});
    float* u2f = fp32_of(b.u2, u2s, {&c->l2_out2b});
    OPP_TRY(run_conv(l2f, H4, W4, c->l2_out2a, 1, nullptr, OPP_RES_NONE, OPP_ACT_LEAKY, u2f, s, hp, -1, 1, false, -1, -1, l2s, u2s));
    OPP_TRY(run_conv(u2f, H4, W4, c->l2_out2b, 1, nullptr, OPP_RES_NONE, OPP_ACT_NONE, b.x2o, s, hp, -1, 1, false, -1, -1, u2s));
  }
  if (phase == 0 || phase == 2 || phase == 4) {
    SplitKScope sk_scope(b.sk2);
    static const int l1out_cfg = getenv("OPP_L1OUT_CFG") ? atoi(getenv("OPP_L1OUT_CFG")) : -1;   // A/B switch (tools): tile of the K = 128 lateral
    OPP_TRY(run_conv(b.x1, H2, W2, c->l1_out, 1, b.x2o, OPP_RES_BILINEAR2X, OPP_ACT_NONE, b.l1, s, hp, hp == OPP_PREC_BF16X3 ? l1out_cfg : -1, 1, false,
                     -1, -1, b.x1s));
    OPP_TRY(run_conv(b.l1, H2, W2, c->l1_out2a, 1, nullptr, OPP_RES_NONE, OPP_ACT_LEAKY, b.u1, s, hp));
    OPP_TRY(run_conv(b.u1, H2, W2, c->l1_out2b, 1, nullptr, OPP_RES_NONE, OPP_ACT_NONE, feat_f, s, hp));
  }
  return OPP_OK;
}

}  // namespace

namespace {

// ---- training mode: ResNetFPN_8_2.forward with BatchNorm batch statistics, B images per call ----------------
size_t plan_backbone_train(const opp_ctx* c, int B, int H, int W, Arena& a, BackboneBufs& b, float** raw, void** bn_scratch) {
  const size_t p2 = (size_t)B * (H / 2) * (W / 2), p4 = (size_t)B * (H / 4) * (W / 4), p8 = (size_t)B * (H / 8) * (W / 8);
  const int c1 = pad32(c->cfg.block_dims[0]), c2 = pad32(c->cfg.block_dims[1]), c3 = pad32(c->cfg.block_dims[2]);
  b.col = a.f(p2 * 64);
  b.x0 = a.f(p2 * c1);
  b.t1 = a.f(p2 * c1);
  b.x1a = a.f(p2 * c1);
  b.x1 = a.f(p2 * c1);
  b.t2 = a.f(p4 * c2);
  b.ds2 = a.f(p4 * c2);
  b.x2a = a.f(p4 * c2);
  b.x2 = a.f(p4 * c2);
  b.t3 = a.f(p8 * c3);
  b.ds3 = a.f(p8 * c3);
  b.x3a = a.f(p8 * c3);
  b.x3 = a.f(p8 * c3);
  b.l2 = a.f(p4 * c3);
  b.u2 = a.f(p4 * c3);
  b.x2o = a.f(p4 * c2);
  b.l1 = a.f(p2 * c2);
  b.u1 = a.f(p2 * c2);
  size_t mx = p2 * (size_t)(c1 > c2 ? c1 : c2);
  mx = p4 * c3 > mx ? p4 * c3 : mx;
  *raw = a.f(mx);                                        // raw convolution output ahead of each BatchNorm
  *bn_scratch = a.raw(opp_bn_train_scratch_bytes((int)p2, 256));
  return a.off;
}

int backbone_train_impl(opp_ctx* c, const float* image, int B, int H, int W, float* feat_c, float* feat_f, float* bn_stats,
                        Arena& a, hipStream_t s) {
  OPP_CHECK_ARG(c && c->packed && c->train_packed, "backbone_train: training weights not packed (opp_pack_train_weights)");
  OPP_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "backbone_train: bad B / H / W (%d, %dx%d)", B, H, W);
  BackboneBufs b;
  float* raw;
  void* scratch;
  plan_backbone_train(c, B, H, W, a, b, &raw, &scratch);
  if (!a.ok) {
    opp_set_error("backbone_train: workspace too small");
    return OPP_ERR_WORKSPACE;
  }
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
  const int hp = gemm_prec(c->cfg);
  const float eps = 1e-5f;
  // conv -> BatchNorm(batch statistics) -> (+ residual) -> activation   (resnet.py:37-45, :143, :153, :157)
  auto conv_bn = [&](const float* x, int Hin, int Win, const ConvDesc& d, int stride, const float* res, int act, float* y) -> int {
    const int Ho = (Hin + 2 * (d.ks / 2) - d.ks) / stride + 1, Wo = (Win + 2 * (d.ks / 2) - d.ks) / stride + 1;
    OPP_TRY(run_conv(x, Hin, Win, d, stride, nullptr, OPP_RES_NONE, OPP_ACT_NONE, raw, s, hp, -1, B, true));
    return opp_bn_train(raw, B * Ho * Wo, d.cout_pad(), d.cout, d.gamma, d.beta, eps, res, act, y,
                        bn_stats ? bn_stats + (size_t)d.bn_slot * 512 : nullptr, scratch, s);
  };
  auto block = [&](const float* x, int Hin, int Win, const BlockDesc& bd, int stride, float* tmp, float* ds, float* y) -> int {
    const int Ho = Hin / stride, Wo = Win / stride;
    OPP_TRY(conv_bn(x, Hin, Win, bd.conv1, stride, nullptr, OPP_ACT_RELU, tmp));
    const float* shortcut = x;
    if (bd.has_down) {
      OPP_TRY(conv_bn(x, Hin, Win, bd.down, stride, nullptr, OPP_ACT_NONE, ds));
      shortcut = ds;
    }
    return conv_bn(tmp, Ho, Wo, bd.conv2, 1, shortcut, OPP_ACT_RELU, y);
  };
  OPP_TRY(opp_stem_im2col(image, B, H, W, b.col, s));
  {
    OppGemm g;
    g.nonfinite = t_status_flag;
  g.tile_policy = t_tile_policy;
    g.A0 = b.col;
    g.lda0 = 64;
    g.ksplit = 64;
    g.W = c->stem.w_train;
    g.ldw = (int)split_floats(64, hp);
    g.M = B * H2 * W2;
    g.N = c->stem.cout;
    g.K = 64;
    g.C = raw;
    g.ldc = pad32(c->stem.cout);
    g.n_store = pad32(c->stem.cout);
    g.prec = hp;
    g.h2_inv = hp == OPP_PREC_FP16X2 ? c->stem.h2s_train + 1 : nullptr;
    OPP_TRY(opp_gemm_launch(g, s));
    OPP_TRY(opp_bn_train(raw, B * H2 * W2, pad32(c->stem.cout), c->stem.cout, c->stem.gamma, c->stem.beta, eps, nullptr, OPP_ACT_RELU,
                         b.x0, bn_stats ? bn_stats + (size_t)c->stem.bn_slot * 512 : nullptr, scratch, s));
  }
  OPP_TRY(block(b.x0, H2, W2, c->blocks[0], 1, b.t1, nullptr, b.x1a));
  OPP_TRY(block(b.x1a, H2, W2, c->blocks[1], 1, b.t1, nullptr, b.x1));
  OPP_TRY(block(b.x1, H2, W2, c->blocks[2], 2, b.t2, b.ds2, b.x2a));
  OPP_TRY(block(b.x2a, H4, W4, c->blocks[3], 1, b.t2, nullptr, b.x2));
  OPP_TRY(block(b.x2, H4, W4, c->blocks[4], 2, b.t3, b.ds3, b.x3a));
  OPP_TRY(block(b.x3a, H8, W8, c->blocks[5], 1, b.t3, nullptr, b.x3));
  // FPN: the laterals and the last convolutions have no BatchNorm -> same kernels as in eval mode, batched
  OPP_TRY(run_conv(b.x3, H8, W8, c->l3_out, 1, nullptr, OPP_RES_NONE, OPP_ACT_NONE, feat_c, s, hp, -1, B));
  OPP_TRY(run_conv(b.x2, H4, W4, c->l2_out, 1, feat_c, OPP_RES_BILINEAR2X, OPP_ACT_NONE, b.l2, s, hp, -1, B));
  OPP_TRY(conv_bn(b.l2, H4, W4, c->l2_out2a, 1, nullptr, OPP_ACT_LEAKY, b.u2));
  OPP_TRY(run_conv(b.u2, H4, W4, c->l2_out2b, 1, nullptr, OPP_RES_NONE, OPP_ACT_NONE, b.x2o, s, hp, -1, B));
  OPP_TRY(run_conv(b.x1, H2, W2, c->l1_out, 1, b.x2o, OPP_RES_BILINEAR2X, OPP_ACT_NONE, b.l1, s, hp, -1, B));
  OPP_TRY(conv_bn(b.l1, H2, W2, c->l1_out2a, 1, nullptr, OPP_ACT_LEAKY, b.u1));
  OPP_TRY(run_conv(b.u1, H2, W2, c->l1_out2b, 1, nullptr, OPP_RES_NONE, OPP_ACT_NONE, feat_f, s, hp, -1, B));
  return OPP_OK;
}

}  // namespace

extern "C" size_t opp_backbone_train_workspace_bytes(const opp_ctx* ctx, int B, int H, int W) {
  if (!ctx) return 0;
  Arena a(nullptr, 0);
  BackboneBufs b;
  float* raw;
  void* sc;
  return opp_align(plan_backbone_train(ctx, B, H, W, a, b, &raw, &sc)) + 256;
}

extern "C" int opp_backbone_train(opp_ctx* ctx, const float* image, int B, int H, int W, float* feat_c, float* feat_f,
                                  float* bn_stats, void* ws, size_t ws_bytes, void* stream) {
  FlagScope flag_scope(ctx);
  OPP_CHECK_ARG(ctx && image && feat_c && feat_f && ws, "backbone_train: null argument");
  Arena a(ws, ws_bytes);
  return backbone_train_impl(ctx, image, B, H, W, feat_c, feat_f, bn_stats, a, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------
// training step: backbone forward that KEEPS what its backward needs (the "tape"), and that backward
// (PL_OnePosePlus.training_step, lightning_model:54-81, differentiating ResNetFPN_8_2.forward, resnet.py:141-164)
// ----------------------------------------------------------------------------------------
namespace {

struct TapeBlock { float *raw1, *t, *raw2, *rawd, *y; };
struct Tape {
  float *raw0, *x0;                 // stem: raw convolution output, relu(bn(.))
  TapeBlock blk[6];                 // conv1 raw, relu(bn1), conv2 raw, downsample raw (or null), block output
  float *l2, *raw_u2, *u2, *x2o, *l1, *raw_u1, *u1;
  float* stats;                     // [n_bn][512]: batch mean [0, 256) and 1 / sqrt(var + eps) [256, 512) of BatchNorm slot i
};

size_t plan_tape(const opp_ctx* c, int B, int H, int W, Arena& a, Tape& t) {
  const size_t p2 = (size_t)B * (H / 2) * (W / 2), p4 = (size_t)B * (H / 4) * (W / 4), p8 = (size_t)B * (H / 8) * (W / 8);
  const int c1 = pad32(c->cfg.block_dims[0]), c2 = pad32(c->cfg.block_dims[1]), c3 = pad32(c->cfg.block_dims[2]);
  t.raw0 = a.f(p2 * c1);
  t.x0 = a.f(p2 * c1);
  const size_t px[6] = {p2, p2, p4, p4, p8, p8};
  const int ch[6] = {c1, c1, c2, c2, c3, c3};
  for (int i = 0; i < 6; ++i) {
    t.blk[i].raw1 = a.f(px[i] * ch[i]);
    t.blk[i].t = a.f(px[i] * ch[i]);
    t.blk[i].raw2 = a.f(px[i] * ch[i]);
    t.blk[i].rawd = c->blocks[i].has_down ? a.f(px[i] * ch[i]) : nullptr;
    t.blk[i].y = a.f(px[i] * ch[i]);
  }
  t.l2 = a.f(p4 * c3);
  t.raw_u2 = a.f(p4 * c3);
  t.u2 = a.f(p4 * c3);
  t.x2o = a.f(p4 * c2);
  t.l1 = a.f(p2 * c2);
  t.raw_u1 = a.f(p2 * c2);
  t.u1 = a.f(p2 * c2);
  t.stats = a.f(c->bn_names.size() * 512);
  return a.off;
}

size_t plan_tape_ws(const opp_ctx* c, int B, int H, int W, Arena& a, float** col, float** dsbuf, void** scratch) {
  const size_t p2 = (size_t)B * (H / 2) * (W / 2), p4 = (size_t)B * (H / 4) * (W / 4);
  *col = a.f(p2 * 64);
  *dsbuf = a.f(p4 * pad32(c->cfg.block_dims[2] > c->cfg.block_dims[1] ? c->cfg.block_dims[2] : c->cfg.block_dims[1]));
  *scratch = a.raw(opp_bn_train_scratch_bytes((int)p2, 256));
  return a.off;
}

int backbone_tape_impl(opp_ctx* c, const float* image, int B, int H, int W, float* feat_c, float* feat_f, float* bn_stats, Tape& t,
                       Arena& a, hipStream_t s) {
  OPP_CHECK_ARG(c && c->packed && c->train_packed, "backbone_train_tape: training weights not packed (opp_pack_train_weights)");
  OPP_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "backbone_train_tape: bad B / H / W (%d, %dx%d)", B, H, W);
  float *col, *dsbuf;
  void* scratch;
  plan_tape_ws(c, B, H, W, a, &col, &dsbuf, &scratch);
  if (!a.ok) {
    opp_set_error("backbone_train_tape: workspace too small");
    return OPP_ERR_WORKSPACE;
  }
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
  const int hp = gemm_prec(c->cfg);
  const float eps = 1e-5f;
  auto conv_bn = [&](const float* x, int Hin, int Win, const ConvDesc& d, int stride, const float* res, int act, float* raw, float* y) -> int {
    const int Ho = (Hin + 2 * (d.ks / 2) - d.ks) / stride + 1, Wo = (Win + 2 * (d.ks / 2) - d.ks) / stride + 1;
    OPP_TRY(run_conv(x, Hin, Win, d, stride, nullptr, OPP_RES_NONE, OPP_ACT_NONE, raw, s, hp, -1, B, true));
    float* st = t.stats + (size_t)d.bn_slot * 512;
    return opp_bn_train(raw, B * Ho * Wo, d.cout_pad(), d.cout, d.gamma, d.beta, eps, res, act, y,
                        bn_stats ? bn_stats + (size_t)d.bn_slot * 512 : nullptr, scratch, s, st, st + 256);
  };
  auto block = [&](const float* x, int Hin, int Win, const BlockDesc& bd, int stride, TapeBlock& tb) -> int {
    const int Ho = Hin / stride, Wo = Win / stride;
    OPP_TRY(conv_bn(x, Hin, Win, bd.conv1, stride, nullptr, OPP_ACT_RELU, tb.raw1, tb.t));
    const float* shortcut = x;
    if (bd.has_down) {
      OPP_TRY(conv_bn(x, Hin, Win, bd.down, stride, nullptr, OPP_ACT_NONE, tb.rawd, dsbuf));
      shortcut = dsbuf;
    }
    return conv_bn(tb.t, Ho, Wo, bd.conv2, 1, shortcut, OPP_ACT_RELU, tb.raw2, tb.y);
  };
  OPP_TRY(opp_stem_im2col(image, B, H, W, col, s));
  {
    OppGemm g;
    g.tile_policy = t_tile_policy;
    g.A0 = col;
    g.lda0 = 64;
    g.ksplit = 64;
    g.W = c->stem.w_train;
    g.ldw = (int)split_floats(64, hp);
    g.M = B * H2 * W2;
    g.N = c->stem.cout;
    g.K = 64;
    g.C = t.raw0;
    g.ldc = pad32(c->stem.cout);
    g.n_store = pad32(c->stem.cout);
    g.prec = hp;
    OPP_TRY(opp_gemm_launch(g, s));
    float* st = t.stats + (size_t)c->stem.bn_slot * 512;
    OPP_TRY(opp_bn_train(t.raw0, B * H2 * W2, pad32(c->stem.cout), c->stem.cout, c->stem.gamma, c->stem.beta, eps, nullptr, OPP_ACT_RELU,
                         t.x0, bn_stats ? bn_stats + (size_t)c->stem.bn_slot * 512 : nullptr, scratch, s, st, st + 256));
  }
  OPP_TRY(block(t.x0, H2, W2, c->blocks[0], 1, t.blk[0]));
  OPP_TRY(block(t.blk[0].y, H2, W2, c->blocks[1], 1, t.blk[1]));
  OPP_TRY(block(t.blk[1].y, H2, W2, c->blocks[2], 2, t.blk[2]));
  OPP_TRY(block(t.blk[2].y, H4, W4, c->blocks[3], 1, t.blk[3]));
  OPP_TRY(block(t.blk[3].y, H4, W4, c->blocks[4], 2, t.blk[4]));
  OPP_TRY(block(t.blk[4].y, H8, W8, c->blocks[5], 1, t.blk[5]));
  const float *x1 = t.blk[1].y, *x2 = t.blk[3].y, *x3 = t.blk[5].y;
  OPP_TRY(run_conv(x3, H8, W8, c->l3_out, 1, nullptr, OPP_RES_NONE, OPP_ACT_NONE, feat_c, s, hp, -1, B));
  OPP_TRY(run_conv(x2, H4, W4, c->l2_out, 1, feat_c, OPP_RES_BILINEAR2X, OPP_ACT_NONE, t.l2, s, hp, -1, B));
  OPP_TRY(conv_bn(t.l2, H4, W4, c->l2_out2a, 1, nullptr, OPP_ACT_LEAKY, t.raw_u2, t.u2));
  OPP_TRY(run_conv(t.u2, H4, W4, c->l2_out2b, 1, nullptr, OPP_RES_NONE, OPP_ACT_NONE, t.x2o, s, hp, -1, B));
  OPP_TRY(run_conv(x1, H2, W2, c->l1_out, 1, t.x2o, OPP_RES_BILINEAR2X, OPP_ACT_NONE, t.l1, s, hp, -1, B));
  OPP_TRY(conv_bn(t.l1, H2, W2, c->l1_out2a, 1, nullptr, OPP_ACT_LEAKY, t.raw_u1, t.u1));
  OPP_TRY(run_conv(t.u1, H2, W2, c->l1_out2b, 1, nullptr, OPP_RES_NONE, OPP_ACT_NONE, feat_f, s, hp, -1, B));
  return OPP_OK;
}

// ---- backward of ONE convolution over NHWC tensors (channel counts padded to 32) --------------------------------------------
struct ConvBwdWs {
  float *wt_tmp = nullptr, *wt_pack = nullptr, *wt_split = nullptr;   // flipped / transposed weight, packed, pre-split
  float* wt_tail = nullptr;                                            // its last <= 4 output columns for conv_tail.hip (196-channel inputs)
  float* zbuf = nullptr;                                               // zero-inserted dY of a stride-2 convolution
  void* geo = nullptr;
  void* wg = nullptr;
  size_t wg_bytes = 0;
};
struct ConvBwdNeed { size_t wt_tmp = 0, wt_pack = 0, wt_split = 0, wt_tail = 0, zbuf = 0, geo = 0, wg = 0; };

void conv_bwd_need(ConvBwdNeed& n, int B, int Hin, int Win, int cin, int cout, int ks, int stride, bool need_dx, bool need_dw, int prec) {
  const int Ho = (Hin + 2 * (ks / 2) - ks) / stride + 1, Wo = (Win + 2 * (ks / 2) - ks) / stride + 1;
  auto mx = [](size_t& a, size_t b) { a = b > a ? b : a; };
  if (need_dx) {
    mx(n.wt_tmp, (size_t)cout * cin * ks * ks);
    const size_t pk = (size_t)pad32(cin) * opp_conv_k(cout, ks);
    mx(n.wt_pack, pk);
    mx(n.wt_split, split_floats(pk, prec));
    mx(n.wt_tail, opp_conv_tail_weight_floats(pad32(cout), ks));
    if (stride == 2) mx(n.zbuf, (size_t)B * Hin * Win * pad32(cout));
  }
  if (need_dw) {
    if (!(ks == 1 && stride == 1)) mx(n.geo, opp_conv_geo_entries(B * Ho * Wo) * 8);
    mx(n.wg, opp_conv_wgrad_ws_bytes(B * Ho * Wo, pad32(cout), pad32(cin), ks));
  }
}
void conv_bwd_alloc(Arena& a, const ConvBwdNeed& n, ConvBwdWs& w) {
  w.wt_tmp = a.f(n.wt_tmp);
  w.wt_pack = a.f(n.wt_pack);
  w.wt_split = a.f(n.wt_split);
  w.wt_tail = a.f(n.wt_tail);
  w.zbuf = a.f(n.zbuf);
  w.geo = a.raw(n.geo);
  w.wg = a.raw(n.wg);
  w.wg_bytes = n.wg;
}

// x [B][Hin][Win][pad32(cin)], w [cout][cin][ks][ks] (PyTorch), dy [B][Ho][Wo][pad32(cout)] (padded channels zero).
// dx (optional) [B][Hin][Win][pad32(cin)] = conv_transpose(dy, w) (+ dx_add, same shape, may alias dx); dw (optional) in PyTorch layout.
int conv_backward(const float* x, int B, int Hin, int Win, int cin, const float* w, int cout, int ks, int stride, const float* dy, float* dx,
                  const float* dx_add, float* dw, int prec, const ConvBwdWs& ws, hipStream_t s) {
  OPP_CHECK_ARG(stride == 1 || (stride == 2 && Hin % 2 == 0 && Win % 2 == 0), "conv_backward: stride must be 1 or 2 (even input size)");
  OPP_CHECK_ARG(prec == OPP_PREC_FP32 || prec == OPP_PREC_BF16X3, "conv_backward: arithmetic must be fp32 or bf16x3");
  const int pad = ks / 2;
  const int Ho = (Hin + 2 * pad - ks) / stride + 1, Wo = (Win + 2 * pad - ks) / stride + 1;
  const int cin_pad = pad32(cin), cout_pad = pad32(cout);
  if (dw) {
    const void* geo = nullptr;
    if (!(ks == 1 && stride == 1)) {
      OPP_TRY(opp_conv_geo(B, Ho, Wo, Hin, Win, ks, stride, pad, ws.geo, s));
      geo = ws.geo;
    }
    OPP_TRY(opp_conv_wgrad(dy, cout_pad, x, cin_pad, (size_t)B * Hin * Win, geo, B * Ho * Wo, Win, ks, cout, cin, dw, 0, ws.wg, ws.wg_bytes, s));
  }
  if (dx) {
    // the input gradient is a stride-1 convolution of (zero-inserted) dy with the flipped, transposed weight
    OPP_TRY(opp_conv_flip_transpose(w, cout, cin, ks, ws.wt_tmp, s));
    OPP_TRY(opp_pack_conv(ws.wt_tmp, nullptr, cin, cout, ks, cin_pad, cout_pad, ws.wt_pack, s));
    const size_t pk = (size_t)cin_pad * opp_conv_k(cout, ks);
    const float* wp = ws.wt_pack;
    if (prec == OPP_PREC_BF16X3) {
      OPP_TRY(opp_b3_split(ws.wt_pack, ws.wt_split, pk, s));
      wp = ws.wt_split;
    }
    const float* src = dy;
    if (stride == 2) {
      OPP_TRY(opp_conv_dilate2(dy, B, Ho, Wo, cout_pad, ws.zbuf, s));
      src = ws.zbuf;
    }
    OppGemm g;
    g.tile_policy = t_tile_policy;
    g.conv = 1;
    g.prec = prec;
    g.A0 = src;
    g.Bn = B;
    g.Hin = Hin;
    g.Win = Win;
    g.Cin = cout_pad;
    g.ksize = ks;
    g.stride = 1;
    g.pad = pad;
    g.Hout = Hin;
    g.Wout = Win;
    g.W = wp;
    g.K = opp_conv_k(cout, ks);
    g.tail_grp = opp_conv_tail_grp(cout, ks);
    g.ldw = (int)split_floats((size_t)g.K, prec);
    g.M = B * Hin * Win;
    g.N = cin_pad;
    g.C = dx;
    g.ldc = cin_pad;
    g.n_store = cin_pad;
    g.n_real = cin;
    if (dx_add) {
      g.res_mode = OPP_RES_DIRECT;
      g.R = dx_add;
      g.ldr = cin_pad;
    }
    g.alg_flops = 2.0 * (double)B * Ho * Wo * (double)cout * (double)(ks * ks * cin);
    // an input of 196 channels: its gradient is a 196-column output -> 192 columns on the MFMA kernel + a 4-column tail, as in the forward
    const int tail = (cin % 32 >= 1 && cin % 32 <= 4 && cin >= 64) ? cin % 32 : 0;
    if (t_conv_tail && tail && prec == OPP_PREC_BF16X3 && ws.wt_tail != nullptr) {
      const int n0 = cin / 32 * 32;
      OPP_TRY(opp_pack_conv_tail(ws.wt_tmp, nullptr, cin, cout, ks, n0, cout_pad, ws.wt_tail, s));    // wt_tmp = [cin][cout][ks][ks], flipped
      OppGemm body = g;
      body.N = n0;
      body.n_store = n0;
      body.alg_flops = g.alg_flops * (double)n0 / (double)cin;
      OPP_TRY(opp_gemm_launch(body, s));
      OPP_TRY(opp_conv_tail(g, ws.wt_tail, n0, tail, s));
    } else {
      OPP_TRY(opp_gemm_launch(g, s));
    }
  }
  return OPP_OK;
}

struct BwdBufs {
  float *g2a, *g2b, *g2c, *g4a, *g4b, *g4c, *g8a, *g8b, *g8c, *col;
  void* bn_scratch;
  ConvBwdWs cw;
};

size_t plan_backbone_bwd(const opp_ctx* c, int B, int H, int W, Arena& a, BwdBufs& b) {
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
  const size_t p2 = (size_t)B * H2 * W2, p4 = (size_t)B * H4 * W4, p8 = (size_t)B * H8 * W8;
  const int d1 = c->cfg.block_dims[0], d2 = c->cfg.block_dims[1], d3 = c->cfg.block_dims[2];
  const int c1 = pad32(d1), c2 = pad32(d2), c3 = pad32(d3);
  const int c12 = c1 > c2 ? c1 : c2, c23 = c2 > c3 ? c2 : c3;
  b.g2a = a.f(p2 * c12);
  b.g2b = a.f(p2 * c12);
  b.g2c = a.f(p2 * c1);
  b.g4a = a.f(p4 * c23);
  b.g4b = a.f(p4 * c23);
  b.g4c = a.f(p4 * c2);
  b.g8a = a.f(p8 * c3);
  b.g8b = a.f(p8 * c3);
  b.g8c = a.f(p8 * c3);
  b.col = a.f(p2 * 64);
  b.bn_scratch = a.raw(opp_bn_bwd_scratch_bytes((int)p2, 256));
  const int prec = gemm_prec(c->cfg);
  ConvBwdNeed n;
  conv_bwd_need(n, B, H2, W2, 64, c->stem.cout, 1, 1, false, true, prec);                       // stem as a 1x1 over the im2col rows
  const int hin[6] = {H2, H2, H2, H4, H4, H8}, win[6] = {W2, W2, W2, W4, W4, W8}, str[6] = {1, 1, 2, 1, 2, 1};
  for (int i = 0; i < 6; ++i) {
    const BlockDesc& bd = c->blocks[i];
    conv_bwd_need(n, B, hin[i], win[i], bd.conv1.cin, bd.conv1.cout, 3, str[i], true, true, prec);
    conv_bwd_need(n, B, hin[i] / str[i], win[i] / str[i], bd.conv2.cin, bd.conv2.cout, 3, 1, true, true, prec);
    if (bd.has_down) conv_bwd_need(n, B, hin[i], win[i], bd.down.cin, bd.down.cout, 1, str[i], true, true, prec);
  }
  conv_bwd_need(n, B, H8, W8, d3, c->l3_out.cout, 1, 1, true, true, prec);
  conv_bwd_need(n, B, H4, W4, d2, c->l2_out.cout, 1, 1, true, true, prec);
  conv_bwd_need(n, B, H4, W4, c->l2_out2a.cin, c->l2_out2a.cout, 3, 1, true, true, prec);
  conv_bwd_need(n, B, H4, W4, c->l2_out2b.cin, c->l2_out2b.cout, 3, 1, true, true, prec);
  conv_bwd_need(n, B, H2, W2, d1, c->l1_out.cout, 1, 1, true, true, prec);
  conv_bwd_need(n, B, H2, W2, c->l1_out2a.cin, c->l1_out2a.cout, 3, 1, true, true, prec);
  conv_bwd_need(n, B, H2, W2, c->l1_out2b.cin, c->l1_out2b.cout, 3, 1, true, true, prec);
  conv_bwd_alloc(a, n, b.cw);
  return a.off;
}

int backbone_backward_impl(opp_ctx* c, const float* image, int B, int H, int W, const Tape& t, const float* const* w, const float* dfc,
                           const float* dff, float* const* grads, Arena& a, hipStream_t s) {
  BwdBufs b;
  plan_backbone_bwd(c, B, H, W, a, b);
  if (!a.ok) {
    opp_set_error("backbone_backward: workspace too small");
    return OPP_ERR_WORKSPACE;
  }
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
  const int prec = gemm_prec(c->cfg);
  const float *x1 = t.blk[1].y, *x2 = t.blk[3].y, *x3 = t.blk[5].y;
  // gradient of one convolution: weight gradient into the caller's tensor, input gradient into dx (+ dx_add)
  auto conv_b = [&](const ConvDesc& d, const float* x, int Hin, int Win, int stride, const float* dy, float* dx, const float* dx_add) -> int {
    return conv_backward(x, B, Hin, Win, d.cin, w[d.w_idx], d.cout, d.ks, stride, dy, dx, dx_add, grads[d.w_idx], prec, b.cw, s);
  };
  // BatchNorm (+ activation) behind convolution d: dy -> draw (in place unless `draw` is given), dres = dz
  auto bn_b = [&](const ConvDesc& d, int rows, int act, const float* dy, const float* y, const float* raw, float* draw, float* dres) -> int {
    const float* st = t.stats + (size_t)d.bn_slot * 512;
    return opp_bn_backward(dy, y, raw, rows, d.cout_pad(), d.cout, act, d.gamma, st, st + 256, draw, dres, grads[d.bn_idx], grads[d.bn_idx + 1], 0,
                           b.bn_scratch, s);
  };
  // BasicBlock backward (resnet.py:37-45).  dy: gradient of the block output (overwritten); result: gradient of the block input in
  // `out` (same buffer as dy for an identity shortcut; the finer-resolution buffer `out`, which may hold a gradient to add, otherwise)
  auto block_b = [&](const BlockDesc& bd, const TapeBlock& tb, const float* x_in, int Hin, int Win, int stride, float* dy, float* tmp1, float* tmp2,
                     float* out, bool out_has_add) -> int {
    const int Ho = Hin / stride, Wo = Win / stride, rows = B * Ho * Wo;
    OPP_TRY(bn_b(bd.conv2, rows, OPP_ACT_RELU, dy, tb.y, tb.raw2, tmp1, dy));                       // tmp1 = d raw2, dy = dz (shortcut gradient)
    OPP_TRY(conv_b(bd.conv2, tb.t, Ho, Wo, 1, tmp1, tmp2, nullptr));                                  // tmp2 = d t
    OPP_TRY(bn_b(bd.conv1, rows, OPP_ACT_RELU, tmp2, tb.t, tb.raw1, tmp2, nullptr));                  // tmp2 = d raw1
    if (!bd.has_down) return conv_b(bd.conv1, x_in, Hin, Win, 1, tmp2, dy, dy);                       // d x = dgrad + dz, in place
    OPP_TRY(bn_b(bd.down, rows, OPP_ACT_NONE, dy, nullptr, tb.rawd, dy, nullptr));                    // dy = d raw_down
    OPP_TRY(conv_b(bd.down, x_in, Hin, Win, stride, dy, out, out_has_add ? out : nullptr));
    return conv_b(bd.conv1, x_in, Hin, Win, stride, tmp2, out, out);
  };
  auto copy = [&](float* dst, const float* src, size_t n) -> int { return copy_f(dst, src, n, s); };

  // FPN head (resnet.py:149-157), fine branch first
  OPP_TRY(conv_b(c->l1_out2b, t.u1, H2, W2, 1, dff, b.g2a, nullptr));                                 // g2a = d u1
  OPP_TRY(bn_b(c->l1_out2a, B * H2 * W2, OPP_ACT_LEAKY, b.g2a, t.u1, t.raw_u1, b.g2a, nullptr));
  OPP_TRY(conv_b(c->l1_out2a, t.l1, H2, W2, 1, b.g2a, b.g2b, nullptr));                               // g2b = d l1
  OPP_TRY(conv_b(c->l1_out, x1, H2, W2, 1, b.g2b, b.g2c, nullptr));                                   // g2c = d x1 (lateral part)
  OPP_TRY(opp_upsample2x_backward(b.g2b, B, H4, W4, c->l1_out.cout_pad(), b.g4c, 0, s));              // g4c = d x2_out
  OPP_TRY(conv_b(c->l2_out2b, t.u2, H4, W4, 1, b.g4c, b.g4b, nullptr));                               // g4b = d u2
  OPP_TRY(bn_b(c->l2_out2a, B * H4 * W4, OPP_ACT_LEAKY, b.g4b, t.u2, t.raw_u2, b.g4b, nullptr));
  OPP_TRY(conv_b(c->l2_out2a, t.l2, H4, W4, 1, b.g4b, b.g4a, nullptr));                               // g4a = d l2
  OPP_TRY(conv_b(c->l2_out, x2, H4, W4, 1, b.g4a, b.g4c, nullptr));                                   // g4c = d x2 (lateral part)
  OPP_TRY(copy(b.g8a, dfc, (size_t)B * H8 * W8 * c->l3_out.cout_pad()));
  OPP_TRY(opp_upsample2x_backward(b.g4a, B, H8, W8, c->l3_out.cout_pad(), b.g8a, 1, s));              // g8a = d feat_c (both consumers)
  OPP_TRY(conv_b(c->l3_out, x3, H8, W8, 1, b.g8a, b.g8b, nullptr));                                   // g8b = d x3
  // residual stages, last to first
  OPP_TRY(block_b(c->blocks[5], t.blk[5], t.blk[4].y, H8, W8, 1, b.g8b, b.g8a, b.g8c, b.g8b, false));
  OPP_TRY(block_b(c->blocks[4], t.blk[4], x2, H4, W4, 2, b.g8b, b.g8a, b.g8c, b.g4c, true));          // g4c = d x2 (all consumers)
  OPP_TRY(block_b(c->blocks[3], t.blk[3], t.blk[2].y, H4, W4, 1, b.g4c, b.g4a, b.g4b, b.g4c, false));
  OPP_TRY(block_b(c->blocks[2], t.blk[2], x1, H2, W2, 2, b.g4c, b.g4a, b.g4b, b.g2c, true));          // g2c = d x1 (all consumers)
  OPP_TRY(block_b(c->blocks[1], t.blk[1], t.blk[0].y, H2, W2, 1, b.g2c, b.g2a, b.g2b, b.g2c, false));
  OPP_TRY(block_b(c->blocks[0], t.blk[0], t.x0, H2, W2, 1, b.g2c, b.g2a, b.g2b, b.g2c, false));
  // stem (resnet.py:143): BatchNorm + ReLU backward, weight gradient over the im2col rows (k = ky * 7 + kx < 49)
  OPP_TRY(bn_b(c->stem, B * H2 * W2, OPP_ACT_RELU, b.g2c, t.x0, t.raw0, b.g2c, nullptr));
  OPP_TRY(opp_stem_im2col(image, B, H, W, b.col, s));
  return opp_conv_wgrad(b.g2c, pad32(c->stem.cout), b.col, 64, (size_t)B * H2 * W2, nullptr, B * H2 * W2, W2, 1, c->stem.cout, 49,
                        grads[c->stem.w_idx], 0, b.cw.wg, b.cw.wg_bytes, s);
}

}  // namespace

extern "C" size_t opp_backbone_tape_bytes(const opp_ctx* ctx, int B, int H, int W) {
  if (!ctx) return 0;
  Arena a(nullptr, 0);
  Tape t;
  return opp_align(plan_tape(ctx, B, H, W, a, t)) + 256;
}

extern "C" size_t opp_backbone_train_tape_workspace_bytes(const opp_ctx* ctx, int B, int H, int W) {
  if (!ctx) return 0;
  Arena a(nullptr, 0);
  float *col, *ds;
  void* sc;
  return opp_align(plan_tape_ws(ctx, B, H, W, a, &col, &ds, &sc)) + 256;
}

extern "C" int opp_backbone_train_tape(opp_ctx* ctx, const float* image, int B, int H, int W, float* feat_c, float* feat_f, float* bn_stats,
                                       void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, void* stream) {
  FlagScope flag_scope(ctx);
  OPP_CHECK_ARG(ctx && image && feat_c && feat_f && tape && ws, "backbone_train_tape: null argument");
  OPP_CHECK_ARG(gemm_prec(ctx->cfg) != OPP_PREC_FP16X2, "backbone_train_tape: the training step runs in bf16x3 or fp32");
  Arena ta(tape, tape_bytes);
  Tape t;
  plan_tape(ctx, B, H, W, ta, t);
  OPP_CHECK_ARG(ta.ok, "backbone_train_tape: tape buffer too small (%zu bytes)", tape_bytes);
  Arena a(ws, ws_bytes);
  return backbone_tape_impl(ctx, image, B, H, W, feat_c, feat_f, bn_stats, t, a, (hipStream_t)stream);
}

extern "C" size_t opp_backbone_backward_workspace_bytes(const opp_ctx* ctx, int B, int H, int W) {
  if (!ctx) return 0;
  Arena a(nullptr, 0);
  BwdBufs b;
  return opp_align(plan_backbone_bwd(ctx, B, H, W, a, b)) + 256;
}

extern "C" int opp_backbone_backward(opp_ctx* ctx, const float* image, int B, int H, int W, const void* tape, size_t tape_bytes,
                                     const float* const* weights, int n_weights, const float* grad_feat_c, const float* grad_feat_f,
                                     float* const* grads, void* ws, size_t ws_bytes, void* stream) {
  FlagScope flag_scope(ctx);
  OPP_CHECK_ARG(ctx && image && tape && weights && grad_feat_c && grad_feat_f && grads && ws, "backbone_backward: null argument");
  OPP_CHECK_ARG(ctx->train_packed, "backbone_backward: training weights not packed");
  OPP_CHECK_ARG(n_weights == (int)ctx->table.size(), "backbone_backward: expected %d weight tensors, got %d", (int)ctx->table.size(), n_weights);
  Arena ta(const_cast<void*>(tape), tape_bytes);
  Tape t;
  plan_tape(ctx, B, H, W, ta, t);
  OPP_CHECK_ARG(ta.ok, "backbone_backward: tape buffer too small (%zu bytes)", tape_bytes);
  std::vector<const ConvDesc*> cv;
  cv.push_back(&ctx->stem);
  for (ConvDesc* d : all_convs(ctx)) cv.push_back(d);
  for (const ConvDesc* d : cv) {
    OPP_CHECK_ARG(weights[d->w_idx] && grads[d->w_idx], "backbone_backward: weight / gradient pointer %d is null", d->w_idx);
    if (d->bn_idx >= 0) OPP_CHECK_ARG(grads[d->bn_idx] && grads[d->bn_idx + 1], "backbone_backward: BatchNorm gradient pointer %d is null", d->bn_idx);
  }
  Arena a(ws, ws_bytes);
  return backbone_backward_impl(ctx, image, B, H, W, t, weights, grad_feat_c, grad_feat_f, grads, a, (hipStream_t)stream);
}

extern "C" size_t opp_backbone_workspace_bytes(const opp_ctx* ctx, int H, int W) {
  if (!ctx) return 0;
  Arena a(nullptr, 0);
  BackboneBufs b;
  return opp_align(plan_backbone(ctx, H, W, a, b)) + 256;
}

extern "C" int opp_backbone(opp_ctx* ctx, const float* image, int H, int W, float* feat_c, float* feat_f, void* ws,
                            size_t ws_bytes, void* stream) {
  FlagScope flag_scope(ctx);
  OPP_CHECK_ARG(ctx && image && feat_c && feat_f && ws, "backbone: null argument");
  Arena a(ws, ws_bytes);
  return backbone_impl(ctx, image, H, W, feat_c, feat_f, a, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------
// tokens, transformer
// ----------------------------------------------------------------------------------------
namespace {

int encode_points_impl(opp_ctx* c, const float* kpts, const float* bank_c, int n, float* t3, Arena& a, hipStream_t s) {
  const int C = c->cfg.coarse_d_model;
  if (c->cfg.kpt_enc_enable) {
    float* stats = a.f(8);
    float* stats0 = a.f(8);
    if (!a.ok) {
      opp_set_error("encode_points: workspace too small");
      return OPP_ERR_WORKSPACE;
    }
    OPP_TRY(opp_kpt_stats(kpts, n, stats, s));               // utils/normalize.py:16-26
    if (c->kpt_extent_ref && c->kpt_extent_ref != kpts) {
      // B > 1: the scaling comes from the bbox extent of batch element 0, the centre from this sample (quirk q4)
      OPP_TRY(opp_kpt_stats(c->kpt_extent_ref, c->kpt_extent_n, stats0, s));
      if (hipMemcpyAsync(stats + 3, stats0 + 3, sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        opp_set_error("encode_points: copy failed");
        return OPP_ERR_LAUNCH;
      }
    }
    return opp_kpt_encode(kpts, stats, bank_c, n, c->kpt_wt, c->kpt_b, t3, C, s);
  }
  return opp_bank_transpose(bank_c, n, C, t3, C, s);
}

int coarse_tokens_impl(opp_ctx* c, const float* feat_c, const float* pe, int L, const float* kpts, const float* bank_c,
                       int n, const float* tokens3d_pre, float* tokens, Arena& a, hipStream_t s) {
  const int C = c->cfg.coarse_d_model;
  if (pe && tokens3d_pre)      // the serving case: one launch for feat_c + pe and the cached point tokens
    return opp_add_cat(feat_c, pe, (size_t)L * C, tokens3d_pre, (size_t)n * C, tokens, s);
  if (pe) {
    OPP_TRY(opp_add(feat_c, pe, tokens, (size_t)L * C, s));   // OnePosePlusModel.py:137-142
  } else if (hipMemcpyAsync(tokens, feat_c, (size_t)L * C * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
    opp_set_error("coarse_tokens: copy failed");
    return OPP_ERR_LAUNCH;
  }
  float* t3 = tokens + (size_t)L * C;
  if (tokens3d_pre) {   // per-object constant, encoded once (opp_encode_points): just place it
    if (hipMemcpyAsync(t3, tokens3d_pre, (size_t)n * C * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
      opp_set_error("coarse_tokens: copy of cached point tokens failed");
      return OPP_ERR_LAUNCH;
    }
    return OPP_OK;
  }
  return encode_points_impl(c, kpts, bank_c, n, t3, a, s);
}

struct TrBufs {
  float *qkv, *msg, *mrg, *hid, *kv, *ks, *scratch;
};

// scratch floats of the linear attention of (n_seg x (len0 + len1)) tokens: chunk partials of the KV reduction
size_t linattn_scratch_floats(int C, int D, int n_seg, int len0, int len1) {
  const int ch = opp_linattn_chunks(len0) > opp_linattn_chunks(len1) ? opp_linattn_chunks(len0) : opp_linattn_chunks(len1);
  size_t sc = (size_t)n_seg * ch * (C * D + C);
  if (n_seg == 1 && C == 256) {
    const size_t pair = opp_linattn_pair_scratch_floats(len0, len1);
    sc = pair > sc ? pair : sc;
  }
  return sc;
}

// LinearAttention (linear_attention.py:29-61) of both streams of one encoder layer.  qkv [n_seg * (len0 + len1)][3 C]:
// phi(Q) | phi(K) | V / S per row (stream-0 segments first); msg [same rows][C].  kv [2][n_seg][C * D], ks [2][n_seg][C].
// self: each stream attends to itself; cross: to the other stream's (pre-update) K, V (quirk q6).
int run_linattn(const float* qkv, int C, int D, int n_seg, int len0, int len1, bool cross, float* kv, float* ks, float* scratch,
                float* msg, float eps, hipStream_t s) {
  const int T0 = n_seg * len0;
  float* kv0 = kv;
  float* kv1 = kv + (size_t)n_seg * C * D;
  float* ks0 = ks;
  float* ks1 = ks + (size_t)n_seg * C;
  const float* q0 = qkv;
  const float* q1 = qkv + (size_t)T0 * 3 * C;
  if (n_seg == 1 && C == 256 && D == 32) {   // coarse level: MFMA KV reduction and apply, both streams per launch
    OPP_TRY(opp_linattn_kv_pair(qkv, 3 * C, len0, len1, kv0, ks0, scratch, s));
    if (msg == nullptr) return OPP_OK;       // the fused layer kernel applies KV itself (enc_chain.hip)
    return opp_linattn_apply_pair(qkv, 3 * C, kv0, ks0, cross ? 1 : 0, msg, C, len0, len1, eps, s);
  }
  if (opp_linattn_small_ok(len0, len1, C, D))   // fine level: one launch, KV never leaves the CU
    return opp_linattn_small_pair(qkv, 3 * C, n_seg, len0, len1, cross ? 1 : 0, msg, C, C, D, eps, s);
  OPP_TRY(opp_linattn_kv(q0 + C, q0 + 2 * C, 3 * C, n_seg, len0, C, D, kv0, ks0, scratch, s));
  OPP_TRY(opp_linattn_kv(q1 + C, q1 + 2 * C, 3 * C, n_seg, len1, C, D, kv1, ks1, scratch, s));
  OPP_TRY(opp_linattn_apply(q0, 3 * C, cross ? kv1 : kv0, cross ? ks1 : ks0, msg, C, n_seg, len0, cross ? len1 : len0, C, D, eps, s));
  return opp_linattn_apply(q1, 3 * C, cross ? kv0 : kv1, cross ? ks0 : ks1, msg + (size_t)T0 * C, C, n_seg, len1, cross ? len0 : len1, C, D, eps, s);
}

size_t plan_transformer(int C, int D, int n_seg, int len0, int len1, Arena& a, TrBufs& b) {
  const size_t T = (size_t)n_seg * (len0 + len1);
  b.qkv = a.f(T * 3 * C);
  b.msg = a.f(T * C);
  b.mrg = a.f(T * C);
  b.hid = a.f(T * 2 * C);
  b.kv = a.f((size_t)2 * n_seg * C * D);
  b.ks = a.f((size_t)2 * n_seg * C);
  b.scratch = a.f(linattn_scratch_floats(C, D, n_seg, len0, len1));
  return a.off;
}

struct LnArgs {   // LayerNorm fused into the GEMM epilogue (output rows = whole tile rows)
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* res = nullptr;
  int ldres = 0;
  float eps = 1e-5f;
};

int dense_gemm(const float* A0, int lda0, const float* A1, int lda1, int ksplit, const float* W, int M, int N, int K, float* C,
               int act, hipStream_t s, int h2, const float* h2s, const LnArgs* ln = nullptr) {
  OppGemm g;
  g.nonfinite = t_status_flag;
  g.tile_policy = t_tile_policy;
  g.prec = h2;
  g.h2_inv = (h2 == OPP_PREC_FP16X2 && h2s) ? h2s + 1 : nullptr;
  if (ln) {
    g.ln_gamma = ln->gamma;
    g.ln_beta = ln->beta;
    g.ln_res = ln->res;
    g.ln_ldres = ln->ldres;
    g.ln_eps = ln->eps;
  }
  g.A0 = A0;
  g.lda0 = lda0;
  g.A1 = A1;
  g.lda1 = lda1;
  g.ksplit = ksplit;
  g.W = W;
  g.ldw = (int)split_floats((size_t)K, h2);
  g.M = M;
  g.N = N;
  g.K = K;
  g.C = C;
  g.ldc = N;
  g.n_store = N;
  g.act = act;
  return opp_gemm_launch(g, s);
}

// Image-independent prefix of the coarse transformer (layer_names = ["self", "cross", ...], transformer.py:147-160): layer 0 updates
// the 3D-point stream from itself only, and in layer 1 both streams read the PRE-update tokens of the other one (quirk q6), so the
// 3D stream's layer-1 projections phi(Q) | phi(K) | V / S and its KV / Ksum are functions of the object alone.  Layout of the blob:
//   x1 [n][C]   3D tokens after layer 0        qkv1 [n][3 C]   their layer-1 projection rows        kv1 [C * D]   ks1 [C]
struct ObjPrefix {
  float *x1 = nullptr, *qkv1 = nullptr, *kv1 = nullptr, *ks1 = nullptr;
};
size_t obj_prefix_floats(int C, int D, int n) { return (size_t)n * C + (size_t)n * 3 * C + (size_t)C * D + C; }
ObjPrefix obj_prefix_view(float* base, int C, int D, int n) {
  ObjPrefix p;
  p.x1 = base;
  p.qkv1 = p.x1 + (size_t)n * C;
  p.kv1 = p.qkv1 + (size_t)n * 3 * C;
  p.ks1 = p.kv1 + (size_t)C * D;
  return p;
}
// the fused default path only (bf16x3, one 64-token layer kernel per layer): every other configuration computes the whole layer
bool obj_prefix_ok(const opp_ctx* c) {
  return c->cfg.gemm_precision == 3 && c->cfg.encoder_fusion == 2 && c->cfg.coarse_n_layers >= 2 && c->cfg.coarse_is_cross[0] == 0 &&
         c->cfg.coarse_is_cross[1] == 1 && c->cfg.coarse_d_model == 256 && c->cfg.coarse_nhead == 8 && c->tr_packed;
}
enum { OPP_PREFIX_NONE = 0, OPP_PREFIX_USE = 1, OPP_PREFIX_MAKE = 2 };

// LocalFeatureTransformer.forward (transformer.py:133-171) on X = [stream0 ; stream1]
// prefix_mode OPP_PREFIX_USE: X's stream-1 rows already hold ObjPrefix::x1; layer 0 then runs on stream 0 only and layer 1 projects /
// reduces stream 0 only (the 3D stream's share comes from `pre`).  OPP_PREFIX_MAKE (len0 = 0): layer 0 on the 3D stream, then the
// layer-1 projection and KV / Ksum of that stream into `pre`, and stop.  Rows are independent in every kernel of a layer and the
// chunking of the KV reduction is per stream, so both modes reproduce the bits of the full evaluation.
int transformer_impl(const std::vector<EncLayerDesc>& layers, const int* is_cross, int C, int nhead, float* X, int n_seg,
                     int len0, int len1, Arena& a, hipStream_t s, int h2, const float* mask0 = nullptr, int fusion = 1,
                     int prefix_mode = OPP_PREFIX_NONE, const ObjPrefix* pre = nullptr) {
  const int D = C / nhead;
  const int T0 = n_seg * len0, T1 = n_seg * len1, T = T0 + T1;
  if (T == 0 || layers.empty()) return OPP_OK;
  if (prefix_mode != OPP_PREFIX_NONE)
    OPP_CHECK_ARG(pre && n_seg == 1 && C == 256 && D == 32 && fusion == 2 && h2 == OPP_PREC_BF16X3 && layers.size() >= 2 && !is_cross[0] && is_cross[1] &&
                      (prefix_mode == OPP_PREFIX_USE || len0 == 0), "transformer: object prefix on an unsupported configuration");
  TrBufs b;
  plan_transformer(C, D, n_seg, len0, len1, a, b);
  if (!a.ok) {
    opp_set_error("transformer: workspace too small");
    return OPP_ERR_WORKSPACE;
  }
  const float eps_attn = 1e-6f, eps_ln = 1e-5f;
  static const int fuse_env = getenv("OPP_FUSE_LN") ? atoi(getenv("OPP_FUSE_LN")) : 1;   // tuning knob
  const bool fuse_ln = fuse_env && (C == 256 || C == 128);
  for (size_t li = 0; li < layers.size(); ++li) {
    const EncLayerDesc& e = layers[li];
    const bool cross = is_cross[li] != 0;
    // stream-1 rows this layer projects / reduces (l1_qkv) and updates (l1_lay)
    const int l1_qkv = (prefix_mode == OPP_PREFIX_USE && li < 2) ? 0 : len1;
    const int l1_lay = (prefix_mode == OPP_PREFIX_USE && li == 0) ? 0 : len1;
    const bool make_tail = prefix_mode == OPP_PREFIX_MAKE && li == 1;     // projection + KV of the 3D stream into the prefix, then stop
    // the fused coarse path folds layer li + 1's projection into layer li's kernel: only layer 0 launches the GEMM
    static const bool fold_env = !(getenv("OPP_QKV_FOLD") && getenv("OPP_QKV_FOLD")[0] == '0');     // A/B switch of the tools
    const bool fold_path = fold_env && fusion == 2 && h2 == OPP_PREC_BF16X3 && n_seg == 1 && C == 256 && D == 32 && e.fmerge && e.fqkv;
    if (!(fold_path && li > 0)) {  // q/k/v projections of both streams in one GEMM; phi(q), phi(k), v / S fused (transformer.py:76-79)
      OppGemm g;
      g.nonfinite = t_status_flag;
      g.tile_policy = t_tile_policy;
      g.A0 = X;
      g.lda0 = C;
      g.ksplit = C;
      g.W = e.wqkv;
      g.ldw = (int)split_floats((size_t)C, h2);
      g.M = T0 + n_seg * l1_qkv;
      g.N = 3 * C;
      g.K = C;
      g.C = make_tail ? pre->qkv1 : b.qkv;
      g.ldc = 3 * C;
      g.n_store = 3 * C;
      g.act = OPP_ACT_QKV;
      g.qk_cols = 2 * C;
      g.split_row = T0;
      g.s0 = (float)len0;
      g.s1 = (float)len1;
      g.row_mask = mask0;        // query_image_mask over the image tokens (stream 0), or null
      g.row_mask_rows = T0;
      g.prec = h2;
      g.h2_inv = (h2 == OPP_PREC_FP16X2 && e.sqkv) ? e.sqkv + 1 : nullptr;
      OPP_TRY(opp_gemm_launch(g, s));
    }
    if (make_tail) {
      OPP_TRY(opp_linattn_kv_pair(pre->qkv1, 3 * C, 0, len1, b.kv, b.ks, b.scratch, s));
      OPP_TRY(copy_f(pre->kv1, b.kv + (size_t)C * D, (size_t)C * D, s));
      return copy_f(pre->ks1, b.ks + C, C, s);
    }
    // everything behind the projection in ONE launch (bf16x3): [apply ->] merge -> norm1 -> mlp.0 -> ReLU -> mlp.2 -> norm2 -> +x
    const bool chain_apply = n_seg == 1 && C == 256 && D == 32;     // coarse level: the kernel applies KV itself
    if (fusion && h2 == OPP_PREC_BF16X3 && e.fmerge && opp_enc_chain_ok(C, nhead, chain_apply)) {
      OPP_TRY(run_linattn(b.qkv, C, D, n_seg, len0, l1_qkv, cross, b.kv, b.ks, b.scratch, chain_apply ? nullptr : b.msg, eps_attn, s));
      OppEncChain ch;
      ch.C = C;
      ch.X = X;
      ch.ldx = C;
      ch.out = X;
      ch.ldo = C;
      ch.len0 = chain_apply ? len0 : T;
      ch.len1 = chain_apply ? l1_lay : 0;
      if (prefix_mode == OPP_PREFIX_USE && li == 1) {   // the 3D stream's share of this layer comes from the object prefix
        ch.q1 = pre->qkv1;
        ch.kv1 = pre->kv1;
        ch.ks1 = pre->ks1;
      }
      if (fold_path && li + 1 < layers.size() && layers[li + 1].fqkv) {   // this kernel also projects its output rows for layer li + 1
        ch.wq_next = layers[li + 1].fqkv;
        ch.qkv_out = b.qkv;
        ch.qkv_out1 = (prefix_mode == OPP_PREFIX_MAKE && li == 0) ? pre->qkv1 : nullptr;
        ch.qmask = mask0;
      }
      ch.msg = b.msg;
      ch.ldm = C;
      ch.apply = chain_apply ? 1 : 0;
      ch.q = b.qkv;
      ch.ldq = 3 * C;
      ch.kv = b.kv;
      ch.ks = b.ks;
      ch.cross = cross ? 1 : 0;
      ch.eps_attn = eps_attn;
      ch.wm = e.fmerge;
      ch.w1 = e.f1;
      ch.w2 = e.f2;
      ch.g1 = e.g1;
      ch.b1 = e.b1;
      ch.g2 = e.g2;
      ch.b2 = e.b2;
      ch.eps_ln = eps_ln;
      if (chain_apply && fusion == 2) OPP_TRY(opp_enc_layer64(ch, s));   // 64-token tiles: one round at 9096 tokens
      else OPP_TRY(opp_enc_chain(ch, s));
      continue;
    }
    OPP_TRY(run_linattn(b.qkv, C, D, n_seg, len0, len1, cross, b.kv, b.ks, b.scratch, b.msg, eps_attn, s));
    if (fuse_ln) {
      // merge -> norm1 and mlp.2 -> norm2 -> +x in the GEMM epilogues (64-row tiles spanning the whole row)
      LnArgs n1, n2;
      n1.gamma = e.g1;
      n1.beta = e.b1;
      n1.eps = eps_ln;
      n2.gamma = e.g2;
      n2.beta = e.b2;
      n2.res = X;
      n2.ldres = C;
      n2.eps = eps_ln;
      OPP_TRY(dense_gemm(b.msg, C, nullptr, 0, C, e.wmerge, T, C, C, b.mrg, OPP_ACT_NONE, s, h2, e.smerge, &n1));   // merge + norm1 (:86-87)
      OPP_TRY(dense_gemm(X, C, b.mrg, C, C, e.w1, T, 2 * C, 2 * C, b.hid, OPP_ACT_RELU, s, h2, e.s1));               // mlp.0 on cat([x,msg]) (:91)
      OPP_TRY(dense_gemm(b.hid, 2 * C, nullptr, 0, 2 * C, e.w2, T, C, 2 * C, X, OPP_ACT_NONE, s, h2, e.s2, &n2));    // mlp.2 + norm2 + x (:92-94)
    } else {
    OPP_TRY(dense_gemm(b.msg, C, nullptr, 0, C, e.wmerge, T, C, C, b.mrg, OPP_ACT_NONE, s, h2, e.smerge));  // merge (:86)
    OPP_TRY(opp_layernorm(b.mrg, C, e.g1, e.b1, nullptr, 0, b.msg, C, T, C, eps_ln, s));              // norm1 (:87)
    OPP_TRY(dense_gemm(X, C, b.msg, C, C, e.w1, T, 2 * C, 2 * C, b.hid, OPP_ACT_RELU, s, h2, e.s1));  // mlp.0 on cat([x,msg]) (:91)
    OPP_TRY(dense_gemm(b.hid, 2 * C, nullptr, 0, 2 * C, e.w2, T, C, 2 * C, b.mrg, OPP_ACT_NONE, s, h2, e.s2));  // mlp.2
    OPP_TRY(opp_layernorm(b.mrg, C, e.g2, e.b2, X, C, X, C, T, C, eps_ln, s));                        // x + norm2 (:92-94)
    }
  }
  return OPP_OK;
}

}  // namespace

extern "C" int opp_coarse_tokens(opp_ctx* ctx, const float* feat_c, const float* pe, int L, const float* kpts,
                                 const float* bank_c, int n, float* tokens, void* ws, size_t ws_bytes, void* stream) {
  OPP_CHECK_ARG(ctx && ctx->packed && feat_c && kpts && bank_c && tokens && ws, "coarse_tokens: null argument");
  OPP_CHECK_ARG(ctx->tr_packed, "coarse_tokens: weights were packed with scope 1 (backbone only); repack with opp_set_pack_scope(ctx, 0)");
  OPP_CHECK_ARG(L > 0 && n > 0, "coarse_tokens: empty input");
  Arena a(ws, ws_bytes);
  return coarse_tokens_impl(ctx, feat_c, pe, L, kpts, bank_c, n, nullptr, tokens, a, (hipStream_t)stream);
}

extern "C" int opp_encode_points(opp_ctx* ctx, const float* kpts, const float* bank_c, int n, float* tokens3d, void* ws,
                                 size_t ws_bytes, void* stream) {
  OPP_CHECK_ARG(ctx && ctx->packed && kpts && bank_c && tokens3d && ws && n > 0, "encode_points: bad argument");
  OPP_CHECK_ARG(ctx->tr_packed, "encode_points: weights were packed with scope 1 (backbone only); repack with opp_set_pack_scope(ctx, 0)");
  Arena a(ws, ws_bytes);
  return encode_points_impl(ctx, kpts, bank_c, n, tokens3d, a, (hipStream_t)stream);
}

// ---- per-object prefix of the coarse transformer (see ObjPrefix above) ---------------------------------------------------
extern "C" int opp_set_object_prefix(opp_ctx* ctx, const void* prefix, int n_points) {
  OPP_CHECK_ARG(ctx && (prefix == nullptr || n_points > 0), "set_object_prefix: bad argument");
  ctx->obj_prefix = static_cast<const float*>(prefix);
  ctx->obj_prefix_n = prefix ? n_points : 0;
  return OPP_OK;
}

extern "C" size_t opp_object_prefix_bytes(const opp_ctx* ctx, int n_points) {
  if (!ctx || n_points <= 0 || !obj_prefix_ok(ctx)) return 0;     // 0: this configuration evaluates the whole layers per image
  const int C = ctx->cfg.coarse_d_model;
  return opp_align(obj_prefix_floats(C, C / ctx->cfg.coarse_nhead, n_points) * sizeof(float));
}

extern "C" size_t opp_object_prefix_workspace_bytes(const opp_ctx* ctx, int n_points) {
  if (!ctx || n_points <= 0) return 0;
  return opp_transformer_workspace_bytes(ctx, 0, 1, 0, n_points);
}

extern "C" int opp_object_prefix(opp_ctx* ctx, const float* tokens3d, int n_points, void* prefix, size_t prefix_bytes, void* ws, size_t ws_bytes,
                                 void* stream) {
  FlagScope flag_scope(ctx);
  OPP_CHECK_ARG(ctx && ctx->packed && tokens3d && prefix && ws && n_points > 0, "object_prefix: bad argument");
  OPP_CHECK_ARG(obj_prefix_ok(ctx), "object_prefix: this configuration has no image-independent transformer prefix (opp_object_prefix_bytes == 0)");
  const int C = ctx->cfg.coarse_d_model, D = C / ctx->cfg.coarse_nhead;
  OPP_CHECK_ARG(prefix_bytes >= obj_prefix_floats(C, D, n_points) * sizeof(float) && (reinterpret_cast<uintptr_t>(prefix) & 15) == 0,
                "object_prefix: prefix buffer too small or not 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ObjPrefix pre = obj_prefix_view(static_cast<float*>(prefix), C, D, n_points);
  OPP_TRY(copy_f(pre.x1, tokens3d, (size_t)n_points * C, s));
  Arena a(ws, ws_bytes);
  return transformer_impl(ctx->coarse, ctx->cfg.coarse_is_cross, C, ctx->cfg.coarse_nhead, pre.x1, 1, 0, n_points, a, s, gemm_prec(ctx->cfg), nullptr,
                          ctx->cfg.encoder_fusion, OPP_PREFIX_MAKE, &pre);
}

namespace {
size_t plan_linear_attention(int C, int D, int n_seg, int len0, int len1, Arena& a, float** kv, float** ks, float** scratch) {
  *kv = a.f((size_t)2 * n_seg * C * D);
  *ks = a.f((size_t)2 * n_seg * C);
  *scratch = a.f(linattn_scratch_floats(C, D, n_seg, len0, len1));
  return a.off;
}
}  // namespace

extern "C" size_t opp_linear_attention_workspace_bytes(int n_seg, int len0, int len1, int C, int nhead) {
  if (n_seg <= 0 || nhead <= 0 || C % nhead) return 0;
  Arena a(nullptr, 0);
  float *kv, *ks, *sc;
  return opp_align(plan_linear_attention(C, C / nhead, n_seg, len0, len1, a, &kv, &ks, &sc)) + 256;
}

extern "C" int opp_linear_attention(const float* qkv, int n_seg, int len0, int len1, int C, int nhead, int cross, float* msg,
                                    void* ws, size_t ws_bytes, void* stream) {
  OPP_CHECK_ARG(qkv && msg && ws, "linear_attention: null argument");
  OPP_CHECK_ARG(n_seg > 0 && len0 > 0 && len1 > 0 && nhead > 0 && C % nhead == 0, "linear_attention: bad shape");
  Arena a(ws, ws_bytes);
  float *kv, *ks, *sc;
  plan_linear_attention(C, C / nhead, n_seg, len0, len1, a, &kv, &ks, &sc);
  if (!a.ok) {
    opp_set_error("linear_attention: workspace too small");
    return OPP_ERR_WORKSPACE;
  }
  return run_linattn(qkv, C, C / nhead, n_seg, len0, len1, cross != 0, kv, ks, sc, msg, 1e-6f, (hipStream_t)stream);
}

extern "C" size_t opp_transformer_workspace_bytes(const opp_ctx* ctx, int which, int n_seg, int len0, int len1) {
  if (!ctx) return 0;
  const int C = which == 0 ? ctx->cfg.coarse_d_model : ctx->cfg.fine_d_model;
  const int nh = which == 0 ? ctx->cfg.coarse_nhead : ctx->cfg.fine_nhead;
  Arena a(nullptr, 0);
  TrBufs b;
  return opp_align(plan_transformer(C, C / nh, n_seg, len0, len1, a, b)) + 256;
}

extern "C" int opp_transformer(opp_ctx* ctx, int which, float* tokens, int n_seg, int len0, int len1, void* ws,
                               size_t ws_bytes, void* stream) {
  FlagScope flag_scope(ctx);
  OPP_CHECK_ARG(ctx && ctx->packed && tokens && ws, "transformer: null argument");
  OPP_CHECK_ARG(ctx->tr_packed, "transformer: weights were packed with scope 1 (backbone only); repack with opp_set_pack_scope(ctx, 0)");
  OPP_CHECK_ARG(which == 0 || which == 1, "transformer: which must be 0 or 1");
  Arena a(ws, ws_bytes);
  if (which == 0)
    return transformer_impl(ctx->coarse, ctx->cfg.coarse_is_cross, ctx->cfg.coarse_d_model, ctx->cfg.coarse_nhead, tokens, n_seg, len0, len1, a, (hipStream_t)stream, gemm_prec(ctx->cfg),
                            n_seg == 1 ? ctx->query_mask : nullptr, ctx->cfg.encoder_fusion);
  return transformer_impl(ctx->fine, ctx->cfg.fine_is_cross, ctx->cfg.fine_d_model, ctx->cfg.fine_nhead, tokens, n_seg, len0, len1, a, (hipStream_t)stream, gemm_prec(ctx->cfg),
                          nullptr, ctx->cfg.encoder_fusion);
}

// ----------------------------------------------------------------------------------------
// coarse matching
// ----------------------------------------------------------------------------------------
namespace {

int coarse_match_impl(opp_ctx* c, const float* f3, const float* f2, int n, int hc, int wc, const float* kpts, float base_scale,
                      const float* qscale, float* conf, long long* i_ids, long long* j_ids, float* mconf, float* mkpts_c,
                      float* mkpts_3d, int* count, Arena& a, hipStream_t s) {
  const int C = c->cfg.coarse_d_model, L = hc * wc;
  float* scratch = a.f(opp_coarse_match_scratch_floats(n, L));
  float* stats = a.f(opp_coarse_match_stats_floats(n, L));
  const int sprec = score_prec(c->cfg);                // score GEMM on the split-operand path as well
  float* f2_split = sprec != OPP_PREC_FP32 ? a.f(split_floats((size_t)L * C, sprec)) : nullptr;
  // (the split-operand core addresses its operands and output with 32-bit offsets: beyond that the r02 path reports the limit)
  const bool two_sweep = sprec == OPP_PREC_BF16X3 && c->cfg.score_two_sweep && C % 32 == 0 && (size_t)n * L < (1ull << 31) &&
                         (size_t)(n > L ? n : L) * C * 6 < (1ull << 31);
  float* f3_split = two_sweep ? a.f(split_floats((size_t)n * C, sprec)) : nullptr;
  if (!a.ok) {
    opp_set_error("coarse_match: workspace too small");
    return OPP_ERR_WORKSPACE;
  }
  if (two_sweep) {
    // both operands pre-split once (7.7 + 6.3 MB); in the forward they are one token buffer [L + n][C] and the two split buffers are adjacent in
    // the workspace: one launch (r05)
    if (f3 == f2 + (size_t)L * C && f3_split == f2_split + split_floats((size_t)L * C, sprec)) {
      OPP_TRY(opp_b3_split(f2, f2_split, (size_t)(L + n) * C, s));
    } else {
      OPP_TRY(opp_b3_split(f2, f2_split, (size_t)L * C, s));
      OPP_TRY(opp_b3_split(f3, f3_split, (size_t)n * C, s));
    }
    if (c->cfg.score_two_sweep == 2)   // one sweep of the split-operand GEMM (statistics + score matrix), conf formed in place
      return opp_dual_softmax_ss_single(f3_split, f2_split, C, n, L, wc, 1.0f / (float)C, (float)((double)c->cfg.match_temperature + 1e-4),
                                        c->query_mask, c->cfg.match_thr, c->cfg.match_border_rm, kpts, base_scale, qscale, conf, stats, scratch,
                                        i_ids, j_ids, mconf, mkpts_c, mkpts_3d, count, s);
    // the score tiles are computed twice and conf is written once
    return opp_dual_softmax_two_sweep(f3_split, f2_split, C, n, L, wc, 1.0f / (float)C, (float)((double)c->cfg.match_temperature + 1e-4),
                                      c->query_mask, c->cfg.match_thr, c->cfg.match_border_rm, kpts, base_scale, qscale, conf, stats, scratch,
                                      i_ids, j_ids, mconf, mkpts_c, mkpts_3d, count, s);
  }
  // sim = (f3/sqrt(C)) . (f2/sqrt(C)) / (temperature + 1e-4)   (coarse_matching.py:99-107).
  // C = 256: the 1/16 feature scaling is an exact power of two, so it commutes with the sum.
  OppGemm g;
  g.nonfinite = t_status_flag;
  g.tile_policy = t_tile_policy;
  g.A0 = f3;
  g.lda0 = C;
  g.ksplit = C;
  g.W = f2;
  g.ldw = C;
  g.M = n;
  g.N = L;
  g.K = C;
  g.C = conf;
  g.ldc = L;
  g.n_store = L;
  g.out_mul = 1.0f / (float)C;
  g.out_div = (float)((double)c->cfg.match_temperature + 1e-4);
  g.col_mask = c->query_mask;      // masked image cells get -1e9 (coarse_matching.py:108-114), or null
  // tile rows of the score GEMM: 128 (config 0 / 25); bf16x3 with enough rows: the 256x128 8-wave tile (config 20)
  const int score_cfg = sprec == OPP_PREC_BF16X3 ? (n >= 512 ? 20 : 25) : 0;
  const int score_bm = score_cfg == 20 ? 256 : 128;
  {  // dual-softmax (max, sum exp) partials fused into the epilogue: [n][tn] x2, [tm][L] x2
    const size_t tn = opp_cdiv(L, 128), tm = opp_cdiv(n, score_bm);
    g.stat_rowmax = stats;
    g.stat_rowsum = g.stat_rowmax + (size_t)n * tn;
    g.stat_colmax = g.stat_rowsum + (size_t)n * tn;
    g.stat_colsum = g.stat_colmax + tm * (size_t)L;
  }
  if (sprec != OPP_PREC_FP32) {   // the image tokens are the "weight" operand here: split them once per image (4 / 6 MB), unscaled
    if (sprec == OPP_PREC_BF16X3) OPP_TRY(opp_b3_split(f2, f2_split, (size_t)L * C, s));
    else OPP_TRY(opp_h2_split(f2, f2_split, (size_t)L * C, nullptr, s));
    g.W = f2_split;
    g.ldw = (int)split_floats((size_t)C, sprec);
    g.prec = sprec;
  }
  OPP_TRY(opp_gemm_launch_cfg(g, score_cfg, s));   // the partial layout above assumes this tile shape
  return opp_dual_softmax_select(conf, n, L, wc, c->cfg.match_thr, c->cfg.match_border_rm, kpts, base_scale, qscale, stats, score_bm, scratch, i_ids, j_ids,
                                 mconf, mkpts_c, mkpts_3d, count, s);
}

}  // namespace

extern "C" size_t opp_coarse_match_workspace_bytes(const opp_ctx* ctx, int n, int L) {
  (void)ctx;
  return opp_align(opp_coarse_match_scratch_floats(n, L) * sizeof(float)) +
         opp_align(opp_coarse_match_stats_floats(n, L) * sizeof(float)) +
         opp_align((size_t)L * 256 * sizeof(float) * 3 / 2) + opp_align((size_t)n * 256 * sizeof(float) * 3 / 2) + 1024;
}

extern "C" int opp_coarse_match(opp_ctx* ctx, const float* f3, const float* f2, int n, int hc, int wc, const float* kpts,
                                float base_scale, const float* qscale, float* conf, long long* i_ids, long long* j_ids, float* mconf,
                                float* mkpts_c, float* mkpts_3d, int* count, void* ws, size_t ws_bytes, void* stream) {
  FlagScope flag_scope(ctx);
  OPP_CHECK_ARG(ctx && f3 && f2 && kpts && conf && i_ids && j_ids && mconf && mkpts_c && mkpts_3d && count && ws,
                "coarse_match: null argument");
  OPP_CHECK_ARG(n > 0 && hc > 0 && wc > 0, "coarse_match: empty input");
  Arena a(ws, ws_bytes);
  return coarse_match_impl(ctx, f3, f2, n, hc, wc, kpts, base_scale, qscale, conf, i_ids, j_ids, mconf, mkpts_c, mkpts_3d, count, a,
                           (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------
// whole coarse level
// ----------------------------------------------------------------------------------------
namespace {

size_t plan_forward_coarse(const opp_ctx* c, int H, int W, int n, Arena& a, float** feat_c, float** tokens) {
  const int C = c->cfg.coarse_d_model;
  const size_t L = (size_t)(H / 8) * (W / 8);
  *feat_c = a.f(L * C);
  *tokens = a.f((L + n) * C);
  return a.off;
}

}  // namespace

extern "C" size_t opp_forward_coarse_workspace_bytes(const opp_ctx* ctx, int H, int W, int n) {
  if (!ctx) return 0;
  Arena a(nullptr, 0);
  float *fc, *tk;
  size_t base = opp_align(plan_forward_coarse(ctx, H, W, n, a, &fc, &tk));
  const int L = (H / 8) * (W / 8);
  size_t s1 = opp_backbone_workspace_bytes(ctx, H, W);
  size_t s2 = opp_transformer_workspace_bytes(ctx, 0, 1, L, n);
  size_t s3 = opp_coarse_match_workspace_bytes(ctx, n, L);
  if (ctx->cfg.fpn_overlap) return base + s1 + (s2 > s3 ? s2 : s3) + 2048;   // the fine branch runs beside the coarse level
  size_t m = s1 > s2 ? s1 : s2;
  m = m > s3 ? m : s3;
  return base + m + 1024;
}

extern "C" int opp_forward_coarse(opp_ctx* ctx, const float* image, int H, int W, const float* pe, const float* kpts,
                                  const float* bank_c, const float* tokens3d_pre, int n, float base_scale,
                                  const float* qscale, float* feat_f, float* conf,
                                  long long* i_ids, long long* j_ids, float* mconf, float* mkpts_c, float* mkpts_3d,
                                  int* count, void* ws, size_t ws_bytes, void* stream) {
  FlagScope flag_scope(ctx);
  OPP_CHECK_ARG(ctx && ctx->packed && image && kpts && (bank_c || tokens3d_pre) && conf && ws, "forward_coarse: null argument");
  OPP_CHECK_ARG(ctx->tr_packed, "forward_coarse: weights were packed with scope 1 (backbone only); repack with opp_set_pack_scope(ctx, 0)");
  OPP_CHECK_ARG(n > 0, "forward_coarse: empty point cloud");
  OPP_CHECK_ARG(!ctx->cfg.pos_enc_enable || pe, "forward_coarse: positional encoding enabled but pe is null");
  hipStream_t s = (hipStream_t)stream;
  Arena a(ws, ws_bytes);
  float *feat_c, *tokens;
  plan_forward_coarse(ctx, H, W, n, a, &feat_c, &tokens);
  if (!a.ok) {
    opp_set_error("forward_coarse: workspace too small");
    return OPP_ERR_WORKSPACE;
  }
  size_t mark = a.off;
  const int hc = H / 8, wc = W / 8, L = hc * wc, C = ctx->cfg.coarse_d_model;
  bool forked = false;
  // what runs beside the coarse level: the whole FPN fine branch (-> feat_f), or -- match-driven fine branch, opp_set_fine_patch_buffers --
  // only its 1/4-resolution half (-> x2_out; x1 / x2_out land in the caller's buffers), or nothing (feat_f = NULL: the fine map is dead)
  const bool keep_inputs = !feat_f && ctx->fine_x1 && ctx->fine_x2o;
  const int side_phase = feat_f ? 2 : (keep_inputs ? 3 : 0);
  if (side_phase == 0) {
    // the caller runs no fine stage (fine_matching.enable = False): the fine map is not an output of the forward and nothing
    // downstream reads it, so the FPN fine branch (x1_out; ~44 % of the backbone FLOPs) is not launched
    BackboneBufs bufs;
    OPP_TRY(backbone_impl(ctx, image, H, W, feat_c, nullptr, a, s, 1, &bufs));
  } else if (ctx->cfg.fpn_overlap) {
    // The coarse level (tokens, transformer, matcher: many short launches that leave CUs idle) depends only on the
    // coarse map; the FPN fine branch (six chip-filling convolutions, ~40 % of the backbone FLOPs) is needed by the fine
    // stage only.  Run the fine branch on a side stream next to the coarse level: fork after layer3_outconv, join below.
    if (!ctx->side_stream) {
      // all three objects or none: a half-built set must never be seen by a later call
      hipStream_t st = nullptr;
      hipEvent_t ef = nullptr, ej = nullptr;
      const bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                      hipEventCreateWithFlags(&ef, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&ej, hipEventDisableTiming) == hipSuccess;
      if (!ok) {
        if (ej) (void)hipEventDestroy(ej);
        if (ef) (void)hipEventDestroy(ef);
        if (st) (void)hipStreamDestroy(st);
        opp_set_error("forward_coarse: cannot create the side stream of the fine-branch overlap");
        return OPP_ERR_LAUNCH;
      }
      ctx->side_stream = st;
      ctx->ev_fork = ef;
      ctx->ev_join = ej;
    }
    BackboneBufs bufs;
    OPP_TRY(backbone_impl(ctx, image, H, W, feat_c, feat_f, a, s, 1, &bufs, keep_inputs ? ctx->fine_x1 : nullptr, keep_inputs ? ctx->fine_x2o : nullptr));
    mark = a.off;                                    // the backbone buffers stay alive until the join
    // fork: if the dependency cannot be expressed the fine branch runs on the caller's stream (same kernels, no overlap)
    const bool fork_ok = hipEventRecord(ctx->ev_fork, s) == hipSuccess && hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0) == hipSuccess;
    if (!fork_ok) {
      OPP_TRY(backbone_impl(ctx, image, H, W, feat_c, feat_f, a, s, side_phase, &bufs));
    } else {
      const int rc = backbone_impl(ctx, image, H, W, feat_c, feat_f, a, ctx->side_stream, side_phase, &bufs);
      if (hipEventRecord(ctx->ev_join, ctx->side_stream) != hipSuccess) {
        // the join cannot be expressed as an event: wait for the side stream on the host before anything reuses its buffers
        (void)hipStreamSynchronize(ctx->side_stream);
        if (rc != OPP_OK) return rc;
      } else {
        forked = true;
        if (rc != OPP_OK) {
          if (hipStreamWaitEvent(s, ctx->ev_join, 0) != hipSuccess) (void)hipStreamSynchronize(ctx->side_stream);
          return rc;
        }
      }
    }
  } else if (keep_inputs) {
    BackboneBufs bufs;
    OPP_TRY(backbone_impl(ctx, image, H, W, feat_c, nullptr, a, s, 1, &bufs, ctx->fine_x1, ctx->fine_x2o));
    OPP_TRY(backbone_impl(ctx, image, H, W, feat_c, nullptr, a, s, 3, &bufs));
  } else {
    OPP_TRY(backbone_impl(ctx, image, H, W, feat_c, feat_f, a, s));
  }
  struct Join {       // every exit path re-joins the side stream: feat_f and the workspace are the caller's again in stream order
    opp_ctx* c;
    hipStream_t s;
    bool on;
    ~Join() {
      if (on && hipStreamWaitEvent(s, c->ev_join, 0) != hipSuccess) (void)hipStreamSynchronize(c->side_stream);   // never unsynchronised
    }
  } join{ctx, s, forked};
  a.off = mark;
  // resident object with a transformer prefix (opp_set_object_prefix): its 3D tokens enter already past layer 0
  const bool use_prefix = ctx->obj_prefix != nullptr && ctx->obj_prefix_n == n && tokens3d_pre != nullptr && obj_prefix_ok(ctx);
  ObjPrefix pre;
  if (use_prefix) pre = obj_prefix_view(const_cast<float*>(ctx->obj_prefix), C, C / ctx->cfg.coarse_nhead, n);
  OPP_TRY(coarse_tokens_impl(ctx, feat_c, ctx->cfg.pos_enc_enable ? pe : nullptr, L, kpts, bank_c, n, use_prefix ? pre.x1 : tokens3d_pre, tokens, a, s));
  a.off = mark;
  OPP_TRY(transformer_impl(ctx->coarse, ctx->cfg.coarse_is_cross, C, ctx->cfg.coarse_nhead, tokens, 1, L, n, a, s, gemm_prec(ctx->cfg), ctx->query_mask,
                           ctx->cfg.encoder_fusion, use_prefix ? OPP_PREFIX_USE : OPP_PREFIX_NONE, use_prefix ? &pre : nullptr));
  a.off = mark;
  return coarse_match_impl(ctx, tokens + (size_t)L * C, tokens, n, hc, wc, kpts, base_scale, qscale, conf, i_ids, j_ids, mconf, mkpts_c,
                           mkpts_3d, count, a, s);
}

// ----------------------------------------------------------------------------------------
// fine level
// ----------------------------------------------------------------------------------------
namespace {
// loftr_fine + FineMatching on the window tokens X [M * WW][C] and point tokens f3 = X + M * WW * C (OnePosePlusModel.py:187-201)
int fine_tail(opp_ctx* ctx, float* X, int M, const float* mkpts_c, float base_scale, const float* qscale, int run_transformer, float* expec_f,
              float* mkpts_f, Arena& a, hipStream_t s) {
  const int C = ctx->cfg.fine_d_model, Wwin = ctx->cfg.fine_window, WW = Wwin * Wwin;
  float* f3 = X + (size_t)M * WW * C;
  if (run_transformer)
    OPP_TRY(transformer_impl(ctx->fine, ctx->cfg.fine_is_cross, C, ctx->cfg.fine_nhead, X, M, WW, 1, a, s, gemm_prec(ctx->cfg), nullptr, ctx->cfg.encoder_fusion));
  const float temp = (float)(1.0 / sqrt((double)C));   // fine_matching.py:82
  return opp_fine_head(f3, C, X, C, M, Wwin, C, temp, mkpts_c, base_scale, qscale, expec_f, mkpts_f, s);
}

struct PatchBufs {
  float *X, *xa, *l1, *u1, *sk;
  size_t sk_floats;
};
size_t plan_fine_patches(const opp_ctx* c, int M, Arena& a, PatchBufs& b) {
  const int C = c->cfg.fine_d_model, W = c->cfg.fine_window, WW = W * W, P9 = W + 4, P7 = W + 2;
  const int c1 = pad32(c->cfg.block_dims[0]), c2 = pad32(c->cfg.block_dims[1]);
  b.X = a.f((size_t)M * (WW + 1) * C);
  b.xa = a.f((size_t)M * P9 * P9 * c1);
  b.l1 = a.f((size_t)M * P9 * P9 * c2);
  b.u1 = a.f((size_t)M * P7 * P7 * c2);
  // K-slice partials, for the (small) images whose dense convolutions run split: 4 slices of the larger patch output
  const size_t r1 = (size_t)M * P7 * P7 * c2, r2 = (size_t)M * WW * C;
  b.sk_floats = 4 * (r1 > r2 ? r1 : r2);
  b.sk = a.f(b.sk_floats);
  return a.off;
}
}  // namespace

extern "C" int opp_set_fine_patch_buffers(opp_ctx* ctx, float* x1, float* x2_out) {
  OPP_CHECK_ARG(ctx && ((x1 == nullptr) == (x2_out == nullptr)), "set_fine_patch_buffers: give both buffers or neither");
  ctx->fine_x1 = x1;
  ctx->fine_x2o = x2_out;
  return OPP_OK;
}

extern "C" size_t opp_fine_patch_buffer_floats(const opp_ctx* ctx, int H, int W, int which) {
  if (!ctx || H <= 0 || W <= 0) return 0;
  return which == 0 ? (size_t)(H / 2) * (W / 2) * pad32(ctx->cfg.block_dims[0]) : (size_t)(H / 4) * (W / 4) * pad32(ctx->cfg.block_dims[1]);
}

extern "C" size_t opp_fine_patches_workspace_bytes(const opp_ctx* ctx, int M) {
  if (!ctx || M <= 0) return 256;
  Arena a(nullptr, 0);
  PatchBufs b;
  const int WW = ctx->cfg.fine_window * ctx->cfg.fine_window;
  return opp_align(plan_fine_patches(ctx, M, a, b)) + opp_transformer_workspace_bytes(ctx, 1, M, WW, 1) + 2048;
}

// Fine level straight from x1 / x2_out of opp_forward_coarse (opp_set_fine_patch_buffers): the three convolutions behind them
// (layer1_outconv + bilinear residual, layer1_outconv2.0 + folded BatchNorm + LeakyReLU, layer1_outconv2.3; resnet.py:154-157) run as VALID
// convolutions over a (W+4)^2 -> (W+2)^2 -> W^2 patch pyramid per match, on the same implicit-GEMM kernel, weights and K order as the
// dense map -- every window value equals the dense one bit for bit -- and land directly in the window-token buffer of loftr_fine
// (no fine map, no gather).  49 MFLOP per match against 78 GFLOP for the dense half-resolution maps at 512 x 512.
extern "C" int opp_fine_patches(opp_ctx* ctx, const float* x1, const float* x2_out, int H, int W, const float* bank_f, int n, const long long* i_ids,
                                const long long* j_ids, int M, int hc, int wc, const float* mkpts_c, float base_scale, const float* qscale,
                                int run_transformer, float* expec_f, float* mkpts_f, void* ws, size_t ws_bytes, void* stream) {
  FlagScope flag_scope(ctx);
  if (M <= 0) return OPP_OK;
  OPP_CHECK_ARG(ctx && ctx->packed && x1 && x2_out && bank_f && i_ids && j_ids && mkpts_c && expec_f && mkpts_f && ws, "fine_patches: null argument");
  OPP_CHECK_ARG(ctx->bn_packed && (ctx->tr_packed || !run_transformer), "fine_patches: weights were packed with scope 1; repack with opp_set_pack_scope(ctx, 0)");
  OPP_CHECK_ARG(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && hc > 0 && wc > 0 && (H / 2) % hc == 0, "fine_patches: bad image / coarse-map size");
  hipStream_t s = (hipStream_t)stream;
  const int Hf = H / 2, Wf = W / 2, stride = Hf / hc;
  const int C = ctx->cfg.fine_d_model, Wwin = ctx->cfg.fine_window, WW = Wwin * Wwin, P9 = Wwin + 4, P7 = Wwin + 2;
  const int c1 = pad32(ctx->cfg.block_dims[0]), c2 = pad32(ctx->cfg.block_dims[1]);
  OPP_CHECK_ARG(ctx->l1_out2b.cout_pad() == C, "fine_patches: fine map width %d != fine d_model %d", ctx->l1_out2b.cout_pad(), C);
  OPP_CHECK_ARG((size_t)M * P9 * P9 * c2 * 4 < (1ull << 31), "fine_patches: too many matches for 32-bit buffer addressing (%d)", M);
  Arena a(ws, ws_bytes);
  PatchBufs b;
  plan_fine_patches(ctx, M, a, b);
  if (!a.ok) {
    opp_set_error("fine_patches: workspace too small");
    return OPP_ERR_WORKSPACE;
  }
  OppProfScope prof(OPP_PROF_FINE, s, (double)M * ((double)(WW + 1) * C * 4.0 + 5 * 4.0));
  const int hp = gemm_prec(ctx->cfg);
  const int org = -(Wwin / 2);                      // window origin relative to the match's fine-map pixel (fine_preprocess.py:41-47: padding W // 2)
  // same accumulation order as the dense map: a layer runs as K slices here exactly when its dense convolution over Hf x Wf pixels does
  // (small images only; at 512 x 512 neither does)
  SplitKScope sk_scope(b.sk, b.sk_floats);
  const size_t dense_px = (size_t)Hf * Wf;
  const int split_a = conv_splits_by_shape(ctx->l1_out2a, dense_px, hp) ? 1 : 0, split_b = conv_splits_by_shape(ctx->l1_out2b, dense_px, hp) ? 1 : 0;
  // l1 patch = conv1x1(x1 patch) + up2x(x2_out) at the patch pixels (out-of-image pixels: exact zeros)
  OPP_TRY(opp_fine_patch_gather(x1, Hf, Wf, c1, x2_out, c2, j_ids, M, wc, stride, org - 2, P9, b.xa, b.l1, s));
  OPP_TRY(run_conv(b.xa, P9, P9, ctx->l1_out, 1, b.l1, OPP_RES_DIRECT, OPP_ACT_NONE, b.l1, s, hp, -1, M, false, 0, 0));   // (K = 4 chunks: never split)
  // u1 patch = LeakyReLU(BN(conv3x3(l1))) on (W+2)^2 pixels; pixels outside the image are the NEXT convolution's zero padding
  OPP_TRY(run_conv(b.l1, P9, P9, ctx->l1_out2a, 1, nullptr, OPP_RES_NONE, OPP_ACT_LEAKY, b.u1, s, hp, -1, M, false, 0, split_a));
  OPP_TRY(opp_patch_zero_oob(b.u1, c2, j_ids, M, wc, stride, org - 1, P7, Hf, Wf, s));
  // the W x W window of the fine map, written as the match's window tokens; window cells outside the image are the unfold's zero padding
  OPP_TRY(run_conv(b.u1, P7, P7, ctx->l1_out2b, 1, nullptr, OPP_RES_NONE, OPP_ACT_NONE, b.X, s, hp, -1, M, false, 0, split_b));
  OPP_TRY(opp_patch_zero_oob(b.X, C, j_ids, M, wc, stride, org, Wwin, Hf, Wf, s));
  OPP_TRY(opp_fine_points_gather(bank_f, n, i_ids, M, C, b.X + (size_t)M * WW * C, C, s));
  return fine_tail(ctx, b.X, M, mkpts_c, base_scale, qscale, run_transformer, expec_f, mkpts_f, a, s);
}

extern "C" size_t opp_backbone_fine_branch_workspace_bytes(const opp_ctx* ctx, int H, int W) {
  if (!ctx) return 0;
  return 2 * opp_align((size_t)(H / 2) * (W / 2) * pad32(ctx->cfg.block_dims[1]) * sizeof(float)) + opp_align(kSplitKScratchFloats * sizeof(float)) + 1024;
}

// The dense 1/2-resolution half of the FPN fine branch from kept x1 / x2_out (more matches than the patch pyramid pays for): -> feat_f
extern "C" int opp_backbone_fine_branch(opp_ctx* ctx, const float* x1, const float* x2_out, int H, int W, float* feat_f, void* ws, size_t ws_bytes,
                                        void* stream) {
  FlagScope flag_scope(ctx);
  OPP_CHECK_ARG(ctx && ctx->packed && x1 && x2_out && feat_f && ws, "backbone_fine_branch: null argument");
  OPP_CHECK_ARG(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "backbone_fine_branch: H,W must be multiples of 8 (got %dx%d)", H, W);
  Arena a(ws, ws_bytes);
  BackboneBufs b{};
  const size_t p2 = (size_t)(H / 2) * (W / 2);
  b.x1 = const_cast<float*>(x1);
  b.x2o = const_cast<float*>(x2_out);
  b.l1 = a.f(p2 * pad32(ctx->cfg.block_dims[1]));
  b.u1 = a.f(p2 * pad32(ctx->cfg.block_dims[1]));
  b.sk2 = a.f(kSplitKScratchFloats);      // small images: these convolutions run as K slices, as inside the one-call path
  if (!a.ok) {
    opp_set_error("backbone_fine_branch: workspace too small");
    return OPP_ERR_WORKSPACE;
  }
  return backbone_impl(ctx, nullptr, H, W, nullptr, feat_f, a, (hipStream_t)stream, 4, &b);
}

extern "C" size_t opp_fine_workspace_bytes(const opp_ctx* ctx, int M) {
  if (!ctx || M <= 0) return 256;
  const int C = ctx->cfg.fine_d_model, WW = ctx->cfg.fine_window * ctx->cfg.fine_window;
  return opp_align((size_t)M * (WW + 1) * C * sizeof(float)) + opp_transformer_workspace_bytes(ctx, 1, M, WW, 1) + 1024;
}

extern "C" int opp_fine(opp_ctx* ctx, const float* feat_f, int Hf, int Wf, const float* bank_f, int n, const long long* i_ids,
                        const long long* j_ids, int M, int hc, int wc, const float* mkpts_c, float base_scale,
                        const float* qscale, int run_transformer, float* expec_f, float* mkpts_f, void* ws, size_t ws_bytes, void* stream) {
  FlagScope flag_scope(ctx);
  if (M <= 0) return OPP_OK;
  OPP_CHECK_ARG(ctx && ctx->packed && feat_f && bank_f && i_ids && j_ids && mkpts_c && expec_f && mkpts_f && ws,
                "fine: null argument");
  OPP_CHECK_ARG(ctx->tr_packed || !run_transformer, "fine: weights were packed with scope 1 (backbone only); repack with opp_set_pack_scope(ctx, 0)");
  OPP_CHECK_ARG(hc > 0 && Hf % hc == 0, "fine: fine map height %d not a multiple of coarse %d", Hf, hc);
  hipStream_t s = (hipStream_t)stream;
  const int C = ctx->cfg.fine_d_model, Wwin = ctx->cfg.fine_window, WW = Wwin * Wwin;
  Arena a(ws, ws_bytes);
  float* X = a.f((size_t)M * (WW + 1) * C);  // [M*WW window tokens ; M point tokens]
  if (!a.ok) {
    opp_set_error("fine: workspace too small");
    return OPP_ERR_WORKSPACE;
  }
  float* f3 = X + (size_t)M * WW * C;
  // whole fine stage as one profiled span; algorithmic bytes: windows + point descriptors gathered, outputs written
  OppProfScope prof(OPP_PROF_FINE, s, (double)M * ((double)(WW + 1) * C * 4.0 + 5 * 4.0));
  OPP_TRY(opp_fine_gather(feat_f, Hf, Wf, C, bank_f, n, i_ids, j_ids, M, wc, Hf / hc, Wwin, C, X, C, f3, C, s));
  return fine_tail(ctx, X, M, mkpts_c, base_scale, qscale, run_transformer, expec_f, mkpts_f, a, s);
}

// ----------------------------------------------------------------------------------------
// building blocks
// ----------------------------------------------------------------------------------------
extern "C" int opp_conv2d_nhwc(const float* x, int Hin, int Win, int cin, const float* w_packed, const float* bias,
                               int cout_pad, int ks, int stride, const float* residual, int res_mode, int act, float* y,
                               int tile_cfg, int prec, const float* h2_scale, void* stream) {
  OPP_CHECK_ARG(x && w_packed && y, "conv2d: null argument");
  OPP_CHECK_ARG(prec >= OPP_PREC_FP32 && prec <= OPP_PREC_BF16X3, "conv2d: prec must be 0 (fp32), 1 (fp16x2) or 2 (bf16x3)");
  OPP_CHECK_ARG(cin > 0 && cout_pad % 32 == 0, "conv2d: cout_pad must be padded to 32");
  ConvDesc d;
  d.cin = cin;
  d.cout = cout_pad;
  d.ks = ks;
  d.w = const_cast<float*>(w_packed);
  d.bias = const_cast<float*>(bias);
  d.h2s = const_cast<float*>(h2_scale);
  return run_conv(x, Hin, Win, d, stride, residual, res_mode, act, y, (hipStream_t)stream, prec, tile_cfg);
}

extern "C" int opp_conv2d_nhwc_split(const float* x, const void* x_split, int Hin, int Win, int cin, const float* w_packed, const float* bias,
                                     int cout_pad, int ks, int stride, const float* residual, int res_mode, int act, float* y, void* y_split,
                                     int tile_cfg, void* stream) {
  AspScope asp_scope(true);      // the explicit entry: the caller asked for pre-split rows
  OPP_CHECK_ARG((x || x_split) && w_packed && (y || y_split), "conv2d_split: null argument");
  OPP_CHECK_ARG(cin > 0 && cout_pad % 32 == 0, "conv2d_split: cout_pad must be padded to 32");
  ConvDesc d;
  d.cin = cin;
  d.cout = cout_pad;
  d.ks = ks;
  d.w = const_cast<float*>(w_packed);
  d.bias = const_cast<float*>(bias);
  OPP_CHECK_ARG(x || conv_takes_split(d, OPP_PREC_BF16X3), "conv2d_split: a 3x3 convolution over 32 n + (1..4) channels packs its K tail and needs the fp32 input");
  return run_conv(x, Hin, Win, d, stride, residual, res_mode, act, y, (hipStream_t)stream, OPP_PREC_BF16X3, tile_cfg, 1, false, -1, -1, x_split, y_split);
}

extern "C" int opp_pack_conv_weight(const float* w, const float* scale, int cout, int cin, int ks, int cout_pad, int cin_pad,
                                    float* out, void* stream) {
  return opp_pack_conv(w, scale, cout, cin, ks, cout_pad, cin_pad, out, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------
// training loss
// ----------------------------------------------------------------------------------------
extern "C" size_t opp_focal_loss_workspace_bytes(size_t n) { return opp_focal_loss_ws_bytes(n) + 256; }

extern "C" int opp_focal_loss_forward(const float* conf, const short* conf_gt, const float* weight, size_t n, float alpha,
                                      float gamma, double* sums, void* ws, size_t ws_bytes, void* stream) {
  return opp_focal_loss_fwd(conf, conf_gt, weight, n, alpha, gamma, sums, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int opp_focal_loss_backward(const float* conf, const short* conf_gt, const float* weight, size_t n, float alpha,
                                       float gamma, const float* scales, float* grad_conf, void* stream) {
  return opp_focal_loss_bwd(conf, conf_gt, weight, n, alpha, gamma, scales, grad_conf, (hipStream_t)stream);
}

extern "C" int opp_focal_loss_forward_ex(const float* conf, const void* conf_gt, int gt_kind, const float* weight, const float* mask0,
                                         const float* mask1, int N, int L, size_t n, float alpha, float gamma, double* sums, void* ws,
                                         size_t ws_bytes, void* stream) {
  return opp_focal_loss_fwd_ex(conf, conf_gt, gt_kind, weight, mask0, mask1, N, L, n, alpha, gamma, sums, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int opp_focal_loss_backward_ex(const float* conf, const void* conf_gt, int gt_kind, const float* weight, const float* mask0,
                                          const float* mask1, int N, int L, size_t n, float alpha, float gamma, const float* scales,
                                          float* grad_conf, void* stream) {
  return opp_focal_loss_bwd_ex(conf, conf_gt, gt_kind, weight, mask0, mask1, N, L, n, alpha, gamma, scales, grad_conf, (hipStream_t)stream);
}

extern "C" size_t opp_dual_softmax_backward_workspace_bytes(int B, int N, int L) { return opp_dual_softmax_bwd_ws_bytes(B, N, L); }

extern "C" int opp_dual_softmax_backward(const float* grad_conf, const float* sim, const float* lse_row, const float* lse_col, int B,
                                         int N, int L, float* grad_sim, void* ws, size_t ws_bytes, void* stream) {
  return opp_dual_softmax_bwd(grad_conf, sim, lse_row, lse_col, B, N, L, grad_sim, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t opp_linear_backward_workspace_bytes(int M, int N, int K, int prec) { return opp_linear_bwd_ws_bytes(M, N, K, prec); }

extern "C" int opp_linear_backward(const float* grad_out, const float* X, const float* W, int M, int N, int K, float* grad_x, float* grad_w,
                                   int accumulate_grad_w, int prec, void* ws, size_t ws_bytes, void* stream) {
  return opp_linear_bwd(grad_out, X, W, M, N, K, grad_x, grad_w, accumulate_grad_w, prec, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t opp_linear_attention_train_workspace_bytes(int B, int L, int S, int nhead, int D) { return opp_linattn_train_ws_bytes(B, L, S, nhead, D); }

extern "C" int opp_linear_attention_train_forward(const float* q, const float* k, const float* v, const float* q_mask, const float* kv_mask, int B,
                                                  int L, int S, int nhead, int D, float* out, float* kv, float* ks, void* ws, size_t ws_bytes,
                                                  void* stream) {
  return opp_linattn_train_fwd(q, k, v, q_mask, kv_mask, B, L, S, nhead, D, 1e-6f, out, kv, ks, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int opp_linear_attention_train_backward(const float* q, const float* k, const float* v, const float* q_mask, const float* kv_mask,
                                                   const float* kv, const float* ks, const float* grad_out, int B, int L, int S, int nhead, int D,
                                                   float* grad_q, float* grad_k, float* grad_v, void* ws, size_t ws_bytes, void* stream) {
  return opp_linattn_train_bwd(q, k, v, q_mask, kv_mask, kv, ks, grad_out, B, L, S, nhead, D, 1e-6f, grad_q, grad_k, grad_v, ws, ws_bytes,
                               (hipStream_t)stream);
}

extern "C" int opp_conv_packed_k(int cin, int ks) { return opp_conv_k(cin, ks); }

extern "C" int opp_pack_h2(const float* in, float* out, size_t n, float* scale2, void* stream) {
  OPP_CHECK_ARG(in && out, "pack_h2: null argument");
  return opp_h2_split(in, out, n, scale2, (hipStream_t)stream);
}

extern "C" int opp_pack_b3(const float* in, float* out, size_t n, void* stream) {
  OPP_CHECK_ARG(in && out, "pack_b3: null argument");
  return opp_b3_split(in, out, n, (hipStream_t)stream);
}

extern "C" int opp_gemm_tile_for(int M, int n_real, int n_store, int K, int conv, int prec, int tile_policy) {
  if (M <= 0 || n_store <= 0 || K <= 0 || K % 32 != 0 || prec < OPP_PREC_FP32 || prec > OPP_PREC_BF16X3 ||
      (tile_policy != OPP_TILES_LATENCY && tile_policy != OPP_TILES_THROUGHPUT)) {
    opp_set_error("gemm_tile_for: bad shape / arithmetic / policy");
    return OPP_ERR_INVALID;
  }
  OppGemm g;
  g.M = M;
  g.N = n_store;
  g.n_store = n_store;
  g.n_real = n_real;
  g.K = K;
  g.conv = conv != 0;
  g.prec = prec;
  g.tile_policy = tile_policy;
  return opp_gemm_choose_tile(g);
}

extern "C" int opp_linear(const float* A, int M, int K, const float* W, int N, int act, float* C, int tile_cfg, int prec,
                          const float* h2_scale, void* stream) {
  OPP_CHECK_ARG(prec >= OPP_PREC_FP32 && prec <= OPP_PREC_BF16X3, "linear: prec must be 0 (fp32), 1 (fp16x2) or 2 (bf16x3)");
  OppGemm g;
  g.nonfinite = t_status_flag;
  g.tile_policy = t_tile_policy;
  g.prec = prec;
  g.h2_inv = (prec == OPP_PREC_FP16X2 && h2_scale) ? h2_scale + 1 : nullptr;
  g.A0 = A;
  g.lda0 = K;
  g.ksplit = K;
  g.W = W;
  g.ldw = (int)split_floats((size_t)K, prec);
  g.M = M;
  g.N = N;
  g.K = K;
  g.C = C;
  g.ldc = N;
  g.n_store = N;
  g.act = act;
  return opp_gemm_launch_cfg(g, tile_cfg, (hipStream_t)stream);
}

extern "C" int opp_linear_layernorm(const float* A, int M, int K, const float* W, int N, const float* gamma, const float* beta,
                                   const float* residual, float* C, int prec, const float* h2_scale, void* stream) {
  OPP_CHECK_ARG(A && W && gamma && beta && C, "linear_layernorm: null argument");
  OPP_CHECK_ARG(N == 256 || N == 128, "linear_layernorm: N must be 256 or 128");
  OPP_CHECK_ARG(prec >= OPP_PREC_FP32 && prec <= OPP_PREC_BF16X3, "linear_layernorm: prec must be 0, 1 or 2");
  OppGemm g;
  g.nonfinite = t_status_flag;
  g.tile_policy = t_tile_policy;
  g.prec = prec;
  g.h2_inv = (prec == OPP_PREC_FP16X2 && h2_scale) ? h2_scale + 1 : nullptr;
  g.A0 = A;
  g.lda0 = K;
  g.ksplit = K;
  g.W = W;
  g.ldw = (int)split_floats((size_t)K, prec);
  g.M = M;
  g.N = N;
  g.K = K;
  g.C = C;
  g.ldc = N;
  g.n_store = N;
  g.ln_gamma = gamma;
  g.ln_beta = beta;
  g.ln_res = residual;
  g.ln_ldres = N;
  return opp_gemm_launch(g, (hipStream_t)stream);
}

extern "C" int opp_layer_norm(const float* x, const float* gamma, const float* beta, const float* residual, float* out,
                              int rows, int C, void* stream) {
  return opp_layernorm(x, C, gamma, beta, residual, C, out, C, rows, C, 1e-5f, (hipStream_t)stream);
}

// ---- training step: building blocks of the backward, exported for the autograd nodes and the stage-level tests -------------
extern "C" size_t opp_conv2d_backward_workspace_bytes(int B, int Hin, int Win, int cin, int cout, int ks, int stride, int prec) {
  ConvBwdNeed n;
  conv_bwd_need(n, B, Hin, Win, cin, cout, ks, stride, true, true, prec);
  Arena a(nullptr, 0);
  ConvBwdWs w;
  conv_bwd_alloc(a, n, w);
  return opp_align(a.off) + 256;
}

extern "C" int opp_conv2d_backward_nhwc(const float* x, int B, int Hin, int Win, int cin, const float* w, int cout, int ks, int stride,
                                        const float* grad_y, float* grad_x, const float* grad_x_add, float* grad_w, int prec, void* ws,
                                        size_t ws_bytes, void* stream) {
  OPP_CHECK_ARG(x && w && grad_y && ws && (grad_x || grad_w), "conv2d_backward: null argument");
  OPP_CHECK_ARG(ks == 1 || ks == 3, "conv2d_backward: kernel size must be 1 or 3");
  ConvBwdNeed n;
  conv_bwd_need(n, B, Hin, Win, cin, cout, ks, stride, grad_x != nullptr, grad_w != nullptr, prec);
  Arena a(ws, ws_bytes);
  ConvBwdWs cw;
  conv_bwd_alloc(a, n, cw);
  OPP_CHECK_ARG(a.ok, "conv2d_backward: workspace too small");
  return conv_backward(x, B, Hin, Win, cin, w, cout, ks, stride, grad_y, grad_x, grad_x_add, grad_w, prec, cw, (hipStream_t)stream);
}

extern "C" size_t opp_batchnorm_backward_workspace_bytes(int rows, int ld) { return opp_bn_bwd_scratch_bytes(rows, ld) + 256; }

extern "C" int opp_batchnorm_backward_nhwc(const float* grad_y, const float* y, const float* raw, int rows, int ld, int C, int act, const float* gamma,
                                           const float* mean, const float* invstd, float* grad_raw, float* grad_res, float* grad_gamma,
                                           float* grad_beta, void* ws, size_t ws_bytes, void* stream) {
  OPP_CHECK_ARG(ws && ws_bytes >= opp_bn_bwd_scratch_bytes(rows, ld), "batchnorm_backward: workspace too small");
  return opp_bn_backward(grad_y, y, raw, rows, ld, C, act, gamma, mean, invstd, grad_raw, grad_res, grad_gamma, grad_beta, 0, ws, (hipStream_t)stream);
}

extern "C" int opp_upsample2x_backward_nhwc(const float* grad_out, int B, int Hr, int Wr, int ld, float* grad_in, int accumulate, void* stream) {
  return opp_upsample2x_backward(grad_out, B, Hr, Wr, ld, grad_in, accumulate, (hipStream_t)stream);
}

extern "C" int opp_layer_norm_train_forward(const float* x, const float* gamma, const float* beta, const float* residual, int rows, int C, float* y,
                                            float* mean, float* rstd, void* stream) {
  return opp_ln_forward(x, gamma, beta, residual, rows, C, 1e-5f, y, mean, rstd, (hipStream_t)stream);
}

extern "C" size_t opp_layer_norm_train_backward_workspace_bytes(int rows, int C) { return opp_ln_backward_ws_bytes(rows, C) + 256; }

extern "C" int opp_layer_norm_train_backward(const float* grad_y, const float* x, const float* gamma, const float* mean, const float* rstd, int rows,
                                             int C, float* grad_x, float* grad_gamma, float* grad_beta, void* ws, size_t ws_bytes, void* stream) {
  return opp_ln_backward(grad_y, x, gamma, mean, rstd, rows, C, grad_x, grad_gamma, grad_beta, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t opp_dual_softmax_forward_workspace_bytes(int B, int N, int L) { return opp_lse_ws_bytes(B, N, L) + 256; }

extern "C" int opp_dual_softmax_forward(const float* sim, int B, int N, int L, float* lse_row, float* lse_col, float* conf, void* ws, size_t ws_bytes,
                                        void* stream) {
  return opp_dual_softmax_lse(sim, B, N, L, lse_row, lse_col, conf, ws, ws_bytes, (hipStream_t)stream);
}

// FineMatching._s2d_heatmap (utils/fine_matching.py:63-94) on its own, for the training graph: expec_f [M][3] from the point tokens f3 [M][C] and
// the window tokens win [M][W * W][C]; its backward (d loss / d expec_f -> d f3, d win)
extern "C" int opp_fine_head_train_forward(const float* f3, const float* win, int M, int window, int C, float* expec_f, float* scratch_xy, void* stream) {
  OPP_CHECK_ARG(f3 && win && expec_f && scratch_xy && M >= 0 && window > 1, "fine_head_train_forward: bad argument");
  if (M == 0) return OPP_OK;
  // the inference kernel also forms mkpts_f = mkpts_c + offset: scratch_xy [2][M][2] holds a zero mkpts_c and receives the unused sum
  if (hipMemsetAsync(scratch_xy, 0, (size_t)M * 2 * sizeof(float), (hipStream_t)stream) != hipSuccess) {
    opp_set_error("fine_head_train_forward: memset failed");
    return OPP_ERR_LAUNCH;
  }
  return opp_fine_head(f3, C, win, C, M, window, C, (float)(1.0 / sqrt((double)C)), scratch_xy, 1.f, nullptr, expec_f, scratch_xy + (size_t)M * 2, (hipStream_t)stream);
}

extern "C" int opp_fine_head_train_backward(const float* f3, const float* win, int M, int window, int C, const float* grad_expec_f, float* grad_f3,
                                            float* grad_win, void* stream) {
  return opp_fine_head_bwd(f3, C, win, C, M, window, C, (float)(1.0 / sqrt((double)C)), grad_expec_f, grad_f3, grad_win, (hipStream_t)stream);
}

extern "C" int opp_build_assignmatrix(const float* kp2d_coarse, const float* kp2d_fine, int n2d, const long long* assign, int k, int N, int L, int w_c,
                                      float scale_x, float scale_y, float coarse_scale, short* conf_gt, float* fine_loc_gt, long long* keys, int* status,
                                      void* stream) {
  return opp_assignmatrix(kp2d_coarse, kp2d_fine, n2d, assign, k, N, L, w_c, scale_x, scale_y, coarse_scale, conf_gt, fine_loc_gt, keys, status,
                          (hipStream_t)stream);
}

extern "C" int opp_fine_window_gather(const float* feat_f, int B, int Hf, int Wf, int C, const long long* b_ids, const long long* j_ids, int n_matches,
                                      int hc, int wc, int window, float* windows, void* stream) {
  OPP_CHECK_ARG(hc > 0 && Hf % hc == 0 && B > 0, "fine_window_gather: fine map height %d not a multiple of coarse %d", Hf, hc);
  return opp_fine_gather_batch(feat_f, Hf, Wf, C, b_ids, j_ids, n_matches, wc, Hf / hc, window, windows, (hipStream_t)stream);
}

extern "C" int opp_fine_window_gather_backward(const float* grad_windows, int B, int Hf, int Wf, int C, const long long* b_ids, const long long* j_ids,
                                               int n_matches, int hc, int wc, int window, float* grad_feat_f, void* stream) {
  OPP_CHECK_ARG(hc > 0 && Hf % hc == 0 && B > 0, "fine_window_gather_backward: fine map height %d not a multiple of coarse %d", Hf, hc);
  return opp_fine_scatter_batch(grad_windows, B, Hf, Wf, C, b_ids, j_ids, n_matches, wc, Hf / hc, window, grad_feat_f, (hipStream_t)stream);
}
