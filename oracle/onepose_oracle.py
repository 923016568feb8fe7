"""CPU oracle for the OnePose++ 2D-3D matching forward.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  This file is a functional fp32 PyTorch restatement
of the reference algorithm (`OnePosePlus_model.forward`,
/root/reference/src/models/OnePosePlus/OnePosePlusModel.py:96-201), written against a flat
state dict instead of the reference's nn.Module tree.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it; the product package
(onepose_plus_plus_amd/) never does and fails loudly if its HIP library is missing.

Parity pinning: the reference repo ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, generated in the build container
by tests/golden/gen_golden.py (reference imported read-only with the stubs in
oracle/refload.py) and committed under tests/golden/.  tests/test_oracle_golden.py checks
the oracle against those fixtures, and - when /root/reference is present - against the live
reference modules.

Every function cites the reference file:line it follows (paths relative to
/root/reference/src/models/OnePosePlus/).  The quirks q1..q11 of SURVEY.md §7 are
reproduced on purpose and marked.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default (backbone/resnet.py:25-26)
LN_EPS = 1e-5  # nn.LayerNorm / nn.InstanceNorm1d default


# --------------------------------------------------------------------------------------
# backbone: backbone/resnet.py:85-164 (ResNetFPN_8_2) + BasicBlock :20-45
# --------------------------------------------------------------------------------------
_TRAIN = {"on": False}     # set by forward(training=True): BatchNorm2d uses batch statistics and updates `sd` in place


def _bn(sd, p, x):
    # BatchNorm2d (resnet.py:25-26, :102).  eval: running statistics.  train(): statistics of the current batch over
    # (B, H, W), running_mean / running_var updated with momentum 0.1 (unbiased variance), num_batches_tracked += 1 --
    # exactly torch.nn.BatchNorm2d's defaults, which is what the reference constructs.
    if _TRAIN["on"]:
        y = F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                         True, 0.1, BN_EPS)
        sd[p + ".num_batches_tracked"] += 1
        return y
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _basic_block(sd, p, x, stride):
    # resnet.py:37-45: relu(bn1(conv1)) -> bn2(conv2) ; shortcut = downsample(x) if stride!=1
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)))
    y = _bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], None, 1, 1))
    if stride != 1:
        x = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0))
    return F.relu(x + y)


def backbone_forward(sd, img, stages=False):
    """resnet.py:141-164.  Returns (x3_out [B,256,H/8,W/8], x1_out [B,128,H/2,W/2])."""
    b = "backbone."
    x0 = F.relu(_bn(sd, b + "bn1", F.conv2d(img, sd[b + "conv1.weight"], None, 2, 3)))  # :143
    x1 = _basic_block(sd, b + "layer1.1", _basic_block(sd, b + "layer1.0", x0, 1), 1)     # :144
    x2 = _basic_block(sd, b + "layer2.1", _basic_block(sd, b + "layer2.0", x1, 2), 1)     # :145
    x3 = _basic_block(sd, b + "layer3.1", _basic_block(sd, b + "layer3.0", x2, 2), 1)     # :146
    x3_out = F.conv2d(x3, sd[b + "layer3_outconv.weight"])                                # :149
    up3 = F.interpolate(x3_out, scale_factor=2.0, mode="bilinear", align_corners=True)    # :151
    t = F.conv2d(x2, sd[b + "layer2_outconv.weight"]) + up3                               # :152-153
    t = F.conv2d(t, sd[b + "layer2_outconv2.0.weight"], None, 1, 1)
    t = F.leaky_relu(_bn(sd, b + "layer2_outconv2.1", t), 0.01)
    x2_out = F.conv2d(t, sd[b + "layer2_outconv2.3.weight"], None, 1, 1)
    up2 = F.interpolate(x2_out, scale_factor=2.0, mode="bilinear", align_corners=True)    # :155
    t = F.conv2d(x1, sd[b + "layer1_outconv.weight"]) + up2                               # :156-157
    t = F.conv2d(t, sd[b + "layer1_outconv2.0.weight"], None, 1, 1)
    t = F.leaky_relu(_bn(sd, b + "layer1_outconv2.1", t), 0.01)
    x1_out = F.conv2d(t, sd[b + "layer1_outconv2.3.weight"], None, 1, 1)
    if stages:
        return dict(x0=x0, x1=x1, x2=x2, x3=x3, x3_out=x3_out, x2_out=x2_out, x1_out=x1_out)
    return x3_out, x1_out


# --------------------------------------------------------------------------------------
# positional encodings: utils/position_encoding.py
# --------------------------------------------------------------------------------------
def sine_position_table(d_model, max_shape):
    """utils/position_encoding.py:13-35.  Quirk q2: `-log(1e4)/d_model // 2` floors to -1.0,
    so div_term = exp(-arange(0, d_model//2, 2)).  Returns [1, d_model, H, W]."""
    h, w = int(max_shape[0]), int(max_shape[1])
    ypos = torch.arange(1, h + 1, dtype=torch.float32).view(1, h, 1).expand(1, h, w)
    xpos = torch.arange(1, w + 1, dtype=torch.float32).view(1, 1, w).expand(1, h, w)
    rate = (-math.log(10000.0) / d_model) // 2          # == -1.0 for d_model = 256
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * rate).view(-1, 1, 1)
    pe = torch.zeros(d_model, h, w)
    pe[0::4] = torch.sin(xpos * div)
    pe[1::4] = torch.cos(xpos * div)
    pe[2::4] = torch.sin(ypos * div)
    pe[3::4] = torch.cos(ypos * div)
    return pe.unsqueeze(0)


def normalize_3d_keypoints(kpts):
    """utils/normalize.py:16-26.  Quirk q4: extent from batch element 0, mean per batch."""
    extent = kpts[0].max(dim=0).values - kpts[0].min(dim=0).values
    scaling = extent.max() * 0.6
    center = kpts.mean(dim=-2, keepdim=True)
    return (kpts - center) / scaling


def keypoint_encoding(sd, kpts, desc):
    """utils/position_encoding.py:54-79 (KeypointEncoding_linear, norm 'instancenorm').
    Quirk q3: InstanceNorm1d applied to [B,L,C] normalises each point's C-vector over the
    channels (biased variance, eps 1e-5, no affine).  desc is [B,C,L]; returns [B,C,L]."""
    p = "kpt_3d_pos_encoding.encoder."
    idxs = sorted({int(k[len(p):].split(".")[0]) for k in sd if k.startswith(p)})
    x = kpts
    for n, i in enumerate(idxs):
        x = F.linear(x, sd[p + "%d.weight" % i], sd[p + "%d.bias" % i])
        if n < len(idxs) - 1:
            mu = x.mean(dim=-1, keepdim=True)
            var = x.var(dim=-1, unbiased=False, keepdim=True)
            x = F.relu((x - mu) / torch.sqrt(var + LN_EPS))
    return desc + x.transpose(2, 1)


# --------------------------------------------------------------------------------------
# linear-attention transformer: loftr_module/linear_attention.py, loftr_module/transformer.py
# --------------------------------------------------------------------------------------
def linear_attention(q, k, v, q_mask=None, kv_mask=None, eps=1e-6):
    """loftr_module/linear_attention.py:29-61.  q [B,L,H,D], k/v [B,S,H,D].
    Quirk q5: values are divided by S before the KV contraction and multiplied back after."""
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    if q_mask is not None:
        Q = Q * q_mask[:, :, None, None]
    if kv_mask is not None:
        K = K * kv_mask[:, :, None, None]
        v = v * kv_mask[:, :, None, None]
    S = v.size(1)
    v = v / S
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    return (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S).contiguous()


def encoder_layer(sd, p, nhead, x, source, x_mask=None, source_mask=None):
    """loftr_module/transformer.py:65-94 (LoFTREncoderLayer.forward; layernorm, no rezero,
    dropout p=0)."""
    B, _, C = x.shape
    D = C // nhead
    q = F.linear(x, sd[p + ".q_proj.weight"]).view(B, -1, nhead, D)
    k = F.linear(source, sd[p + ".k_proj.weight"]).view(B, -1, nhead, D)
    v = F.linear(source, sd[p + ".v_proj.weight"]).view(B, -1, nhead, D)
    msg = linear_attention(q, k, v, x_mask, source_mask).view(B, -1, C)
    msg = F.linear(msg, sd[p + ".merge.weight"])
    msg = F.layer_norm(msg, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], LN_EPS)
    msg = F.linear(F.relu(F.linear(torch.cat([x, msg], dim=2), sd[p + ".mlp.0.weight"])),
                   sd[p + ".mlp.2.weight"])
    msg = F.layer_norm(msg, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], LN_EPS)
    return x + msg


def local_feature_transformer(sd, name, tcfg, feat3d_cn, feat2d, query_mask=None):
    """loftr_module/transformer.py:133-171.  feat3d_cn is [B,C,N] (transposed first, :145),
    feat2d [B,L,C].  Quirk q6: cross layers update both streams from the pre-update tensors.
    Quirk q7: final_proj is never applied."""
    names = list(tcfg["layer_names"]) * tcfg["layer_iter_n"]
    f3 = feat3d_cn.transpose(1, 2)
    f2 = feat2d
    for i, kind in enumerate(names):
        p = "%s.layers.%d" % (name, i)
        if kind == "self":
            f2, f3 = (encoder_layer(sd, p, tcfg["nhead"], f2, f2, query_mask, query_mask),
                      encoder_layer(sd, p, tcfg["nhead"], f3, f3))
        elif kind == "cross":
            f2, f3 = (encoder_layer(sd, p, tcfg["nhead"], f2, f3, x_mask=query_mask),
                      encoder_layer(sd, p, tcfg["nhead"], f3, f2, source_mask=query_mask))
        else:
            raise NotImplementedError(kind)
    return f3, f2


# --------------------------------------------------------------------------------------
# coarse matching: utils/coarse_matching.py
# --------------------------------------------------------------------------------------
def dual_softmax_conf(feat3d, feat2d, temperature, mask_query=None):
    """utils/coarse_matching.py:99-115 with feat_norm 'sqrt_feat_dim' (:46-48)."""
    C = feat3d.shape[-1]
    a = feat3d / C ** 0.5
    b = feat2d / C ** 0.5
    sim = torch.einsum("nlc,nsc->nls", a, b) / (temperature + 1e-4)
    if mask_query is not None:
        sim = sim + torch.where(mask_query[:, None].bool(), 0.0, -1e9).to(sim.dtype)
    return F.softmax(sim, 1) * F.softmax(sim, 2)


def coarse_match_select(conf, hw_c, thr, border_rm):
    """utils/coarse_matching.py:145-172 (inference branch).
    Quirk q1: mask_border (:10-20) only clears the first `border_rm` rows/cols of the coarse
    grid (the `-b:0` slices are empty).  Quirk q9: matches come out in ascending 3D index;
    a row with several surviving cells takes the first."""
    B, N, L = conf.shape
    h, w = hw_c
    mask = (conf > thr).view(B, N, h, w).clone()
    if border_rm > 0:
        mask[:, :, :border_rm] = False
        mask[:, :, :, :border_rm] = False
    mask = mask.view(B, N, L)
    mask = mask & (conf == conf.max(dim=2, keepdim=True)[0]) & (conf == conf.max(dim=1, keepdim=True)[0])
    has, j_all = mask.max(dim=2)
    b_ids, i_ids = torch.where(has)
    j_ids = j_all[b_ids, i_ids]
    return b_ids, i_ids, j_ids, conf[b_ids, i_ids, j_ids]


def coarse_matching(feat3d, feat2d, data, ccfg, mask_query=None, training=False, randint=torch.randint):
    """utils/coarse_matching.py:76-123 + :219-242; training=True adds the padding branch :177-217 (sub-sample the
    predicted matches, pad with ground-truth pairs of `conf_matrix_gt`; random draws from `randint`)."""
    conf = dual_softmax_conf(feat3d, feat2d, ccfg["dual_softmax"]["temperature"], mask_query)
    data["conf_matrix"] = conf
    hw_c = data["q_hw_c"]
    b_ids, i_ids, j_ids, mconf = coarse_match_select(conf, hw_c, ccfg["thr"], ccfg["border_rm"])
    if training and ccfg["train"]["train_padding"]:
        B, N, L = conf.shape
        max_train = int(B * min(N, L) * ccfg["train"]["train_coarse_percent"])
        pad_min = ccfg["train"]["train_pad_num_gt_min"]
        n_pred = len(b_ids)
        assert pad_min < max_train
        if n_pred <= max_train - pad_min:
            pred_idx = torch.arange(n_pred)
        else:
            pred_idx = randint(n_pred, (max_train - pad_min,))
        spv_b, spv_i, spv_j = torch.where(data["conf_matrix_gt"])
        assert len(spv_b) != 0
        gt_idx = randint(len(spv_b), (max(max_train - n_pred, pad_min),))
        b_ids = torch.cat([b_ids[pred_idx], spv_b[gt_idx]])
        i_ids = torch.cat([i_ids[pred_idx], spv_i[gt_idx]])
        j_ids = torch.cat([j_ids[pred_idx], spv_j[gt_idx]])
        mconf = torch.cat([mconf[pred_idx], torch.zeros(len(gt_idx))])
    scale = data["q_hw_i"][0] / hw_c[0]
    if "query_image_scale" in data:
        scale = scale * data["query_image_scale"][b_ids][:, [1, 0]]
    xy = torch.stack([j_ids % hw_c[1], j_ids // hw_c[1]], dim=1) * scale       # quirk q10
    keep = mconf != 0
    data.update(b_ids=b_ids, i_ids=i_ids, j_ids=j_ids, gt_mask=mconf == 0, m_bids=b_ids[keep],
                mkpts_3d_db=data["keypoints3d"][b_ids, i_ids][keep],
                mkpts_query_c=xy[keep], mconf=mconf[keep])


# --------------------------------------------------------------------------------------
# fine stage: loftr_module/fine_preprocess.py, utils/fine_matching.py
# --------------------------------------------------------------------------------------
def fine_preprocess(data, bank_fine, feat_f, fcfg):
    """loftr_module/fine_preprocess.py:32-55.  Windows are W x W, centred on pixel
    (stride*jy, stride*jx) of the fine map, zero outside, flattened (ky,kx) row-major.
    Quirk q8: the 3D side is the RAW fine bank [B,C,N] gathered at i_ids."""
    W = fcfg["window_size"]
    data["W"] = W
    C = fcfg["d_model"]
    if data["b_ids"].shape[0] == 0:
        return torch.empty(0, C, 1), torch.empty(0, W * W, C)
    stride = data["q_hw_f"][0] // data["q_hw_c"][0]
    B = feat_f.shape[0]
    win = F.unfold(feat_f, kernel_size=(W, W), stride=stride, padding=W // 2)   # [B, C*WW, L]
    win = win.view(B, C, W * W, -1).permute(0, 3, 2, 1)                          # [B, L, WW, C]
    f3 = bank_fine.permute(0, 2, 1)[data["b_ids"], data["i_ids"]].unsqueeze(-1)  # [M, C, 1]
    return f3, win[data["b_ids"], data["j_ids"]]


def fine_matching(feat3d, win, data):
    """utils/fine_matching.py:28-110 ('heatmap' s2d) with kornia 0.4.1 dsnt/meshgrid
    semantics (normalised grid {-1,-.5,0,.5,1}, x fastest)."""
    M, WW, C = win.shape
    W = int(math.sqrt(WW))
    if M == 0:                                                    # :46-55
        data.update(expec_f=torch.empty(0, 3), mkpts_query_f=data["mkpts_query_c"])
        return
    f0 = feat3d[:, feat3d.shape[1] // 2, :]                       # select_left_point :63-68
    sim = torch.einsum("mc,mrc->mr", f0, win)
    heat = torch.softmax((1.0 / C ** 0.5) * sim, dim=1)
    lin = (torch.linspace(0, W - 1, W) / (W - 1) - 0.5) * 2
    gx = lin.view(1, W).expand(W, W).reshape(-1)
    gy = lin.view(W, 1).expand(W, W).reshape(-1)
    grid = torch.stack([gx, gy], dim=-1)                          # [WW, 2]
    coords = torch.stack([(gx * heat).sum(-1), (gy * heat).sum(-1)], dim=-1)     # [M, 2]
    var = torch.sum(grid[None] ** 2 * heat[:, :, None], dim=1) - coords ** 2
    std = torch.sum(torch.sqrt(torch.clamp(var, min=1e-10)), -1)
    data["expec_f"] = torch.cat([coords, std[:, None]], -1)
    scale = data["q_hw_i"][0] / data["q_hw_f"][0]                 # :41
    if "query_image_scale" in data:
        scale = scale * data["query_image_scale"][data["b_ids"]][:, [1, 0]]
    n = len(data["mkpts_query_c"])
    data["mkpts_query_f"] = data["mkpts_query_c"] + (coords * (W // 2) * scale)[:n]  # :104-105


# --------------------------------------------------------------------------------------
# whole forward: OnePosePlusModel.py:96-201
# --------------------------------------------------------------------------------------
_PE_CACHE = {}


def forward(sd, data, cfg, keep_intermediates=False, training=False, randint=torch.randint):
    """Mutates `data` in place like the reference.  Forward only (no_grad).  training=True restates the module in
    train() mode: BatchNorm batch statistics (+ running statistics of `sd` updated IN PLACE) and the training
    branch of get_coarse_match."""
    _TRAIN["on"] = bool(training)
    try:
        return _forward(sd, data, cfg, keep_intermediates, training, randint)
    finally:
        _TRAIN["on"] = False


def _forward(sd, data, cfg, keep_intermediates, training, randint):
    with torch.no_grad():
        img = data["query_image"]
        data.update(bs=img.size(0), q_hw_i=img.shape[2:])
        feat_c, feat_f = backbone_forward(sd, img)
        data.update(q_hw_c=feat_c.shape[2:], q_hw_f=feat_f.shape[2:])
        d_model = cfg["loftr_coarse"]["d_model"]
        if cfg["positional_encoding"]["enable"]:
            key = (d_model, tuple(cfg["positional_encoding"]["pos_emb_shape"]))
            if key not in _PE_CACHE:
                _PE_CACHE[key] = sine_position_table(*key)
            feat_c = feat_c + _PE_CACHE[key][:, :, :feat_c.size(2), :feat_c.size(3)]   # :137-142
        tokens2d = feat_c.flatten(2).transpose(1, 2)
        bank_c = data.get("descriptors3d_coarse_db", data["descriptors3d_db"])          # :145-156
        if cfg["keypoints_encoding"]["enable"]:
            bank_c = keypoint_encoding(sd, normalize_3d_keypoints(data["keypoints3d"]), bank_c)
        qmask = data["query_image_mask"].flatten(-2) if "query_image_mask" in data else None
        f3, f2 = local_feature_transformer(sd, "loftr_coarse", cfg["loftr_coarse"], bank_c,
                                           tokens2d, qmask)                             # :160-164
        if keep_intermediates:
            data.update(_feat_c_tokens=tokens2d, _feat_f=feat_f, _f3=f3, _f2=f2)
        coarse_matching(f3, f2, data, cfg["coarse_matching"], qmask, training, randint)  # :167
        if not cfg["fine_matching"]["enable"]:                                          # :169-176
            data["mkpts_query_f"] = data["mkpts_query_c"]
            return
        g3, win = fine_preprocess(data, data["descriptors3d_db"], feat_f, cfg["loftr_fine"])  # :179-186
        if win.size(0) != 0 and cfg["loftr_fine"]["enable"]:                            # :188-198
            g3, win = local_feature_transformer(sd, "loftr_fine", cfg["loftr_fine"], g3, win)
        else:
            g3 = g3.transpose(1, 2)
        fine_matching(g3, win, data)                                                    # :201
