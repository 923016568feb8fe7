"""Functional restatement of the training graph on plain torch operators (test infrastructure, not product).

`differentiable_forward` evaluates the same graph as `OnePosePlus_model._forward_train(graph=True)` with torch's own conv2d /
batch_norm / einsum, so that torch.autograd gives reference gradients: the CPU tests differentiate it against the gradients the
reference's own autograd produced (tests/test_train_autograd_cpu.py), the GPU tests compare the HIP nodes' gradients with it
(tests/test_train_bwd_gpu.py).  The transformer / keypoint-encoder helpers are the product's own (`train_autograd._transformer`
with hp = None = plain torch ops); the backbone and the head below exist only here.  Moved out of the product package in round 5.
Each function cites the reference code it restates (paths relative to /root/reference/src/models/OnePosePlus/).
"""
import torch
import torch.nn.functional as F

from onepose_plus_plus_amd.train_autograd import DualSoftmax, _kpt_encoding, _transformer

_EPS_BN = 1e-5


def _bn(p, name, x):
    # nn.BatchNorm2d in train(): batch statistics (backbone/resnet.py:25-26); running statistics were already updated
    # by the HIP forward, so none are passed here
    return F.batch_norm(x, None, None, p[name + ".weight"], p[name + ".bias"], True, 0.0, _EPS_BN)


def _block(p, name, x, stride, bn):          # BasicBlock.forward, backbone/resnet.py:37-45
    y = F.relu(bn(p, name + ".bn1", F.conv2d(x, p[name + ".conv1.weight"], None, stride, 1)))
    y = bn(p, name + ".bn2", F.conv2d(y, p[name + ".conv2.weight"], None, 1, 1))
    if stride != 1:
        x = bn(p, name + ".downsample.1", F.conv2d(x, p[name + ".downsample.0.weight"], None, stride, 0))
    return F.relu(x + y)


def _backbone(p, img, bn_eval_stats=None):
    """ResNetFPN_8_2.forward, backbone/resnet.py:141-164.  bn_eval_stats: name -> (running_mean, running_var) for a
    frozen pretrained backbone, which the reference keeps in eval mode (OnePosePlusModel.py:109-113)."""
    bn = _bn if bn_eval_stats is None else (
        lambda pp, n, x: F.batch_norm(x, bn_eval_stats[n][0], bn_eval_stats[n][1], pp[n + ".weight"], pp[n + ".bias"], False, 0.0, _EPS_BN))
    b = "backbone."
    x0 = F.relu(bn(p, b + "bn1", F.conv2d(img, p[b + "conv1.weight"], None, 2, 3)))
    x1 = _block(p, b + "layer1.1", _block(p, b + "layer1.0", x0, 1, bn), 1, bn)
    x2 = _block(p, b + "layer2.1", _block(p, b + "layer2.0", x1, 2, bn), 1, bn)
    x3 = _block(p, b + "layer3.1", _block(p, b + "layer3.0", x2, 2, bn), 1, bn)
    x3_out = F.conv2d(x3, p[b + "layer3_outconv.weight"])
    t = F.conv2d(x2, p[b + "layer2_outconv.weight"]) + F.interpolate(x3_out, scale_factor=2.0, mode="bilinear", align_corners=True)
    t = F.leaky_relu(bn(p, b + "layer2_outconv2.1", F.conv2d(t, p[b + "layer2_outconv2.0.weight"], None, 1, 1)), 0.01)
    x2_out = F.conv2d(t, p[b + "layer2_outconv2.3.weight"], None, 1, 1)
    t = F.conv2d(x1, p[b + "layer1_outconv.weight"]) + F.interpolate(x2_out, scale_factor=2.0, mode="bilinear", align_corners=True)
    t = F.leaky_relu(bn(p, b + "layer1_outconv2.1", F.conv2d(t, p[b + "layer1_outconv2.0.weight"], None, 1, 1)), 0.01)
    return x3_out, F.conv2d(t, p[b + "layer1_outconv2.3.weight"], None, 1, 1)


def differentiable_forward(p, cfg, inputs, matches, pe, bn_eval_stats=None):
    """-> (conf_matrix [B,N,L], expec_f [M',3] or None) as differentiable functions of the parameter dict `p`.
    `inputs`: query_image, keypoints3d, descriptors3d_db, descriptors3d_coarse_db (or None), query_image_mask [B,L] or
    None.  `matches` = (b_ids, i_ids, j_ids) chosen by the HIP forward (constants).  OnePosePlusModel.py:96-201."""
    img = inputs["query_image"]
    feat_c, feat_f = _backbone(p, img, bn_eval_stats)
    if pe is not None:                                        # PositionEncodingSine.forward (position_encoding.py:37-42)
        feat_c = feat_c + pe[:, :, :feat_c.size(2), :feat_c.size(3)]
    tokens2d = feat_c.flatten(2).transpose(1, 2)
    bank_c = inputs["descriptors3d_coarse_db"] if inputs.get("descriptors3d_coarse_db") is not None else inputs["descriptors3d_db"]
    if cfg["keypoints_encoding"]["enable"]:
        bank_c = _kpt_encoding(p, inputs["keypoints3d"], bank_c)
    mask = inputs.get("query_image_mask")
    f3, f2 = _transformer(p, "loftr_coarse", cfg["loftr_coarse"], bank_c.transpose(1, 2), tokens2d, mask)
    C = f3.shape[-1]
    sim = torch.einsum("nlc,nsc->nls", f3 / C ** 0.5, f2 / C ** 0.5) / (cfg["coarse_matching"]["dual_softmax"]["temperature"] + 1e-4)
    if mask is not None:                                      # coarse_matching.py:108-114
        sim = sim + torch.where(mask[:, None].bool(), 0.0, -1e9).to(sim.dtype)
    conf = DualSoftmax.apply(sim)                              # coarse_matching.py:115
    if not cfg["fine_matching"]["enable"]:
        return conf, None
    b_ids, i_ids, j_ids = matches
    if b_ids.numel() == 0:
        return conf, None
    fcfg = cfg["loftr_fine"]
    W, Cf = fcfg["window_size"], fcfg["d_model"]
    stride = feat_f.shape[2] // feat_c.shape[2]
    win = F.unfold(feat_f, kernel_size=(W, W), stride=stride, padding=W // 2)       # fine_preprocess.py:41-55
    win = win.view(feat_f.shape[0], Cf, W * W, -1).permute(0, 3, 2, 1)[b_ids, j_ids]
    g3 = inputs["descriptors3d_db"].permute(0, 2, 1)[b_ids, i_ids].unsqueeze(1)      # [M', 1, C]: the RAW fine bank (quirk q8)
    if fcfg["enable"]:
        g3, win = _transformer(p, "loftr_fine", fcfg, g3, win)
    f0 = g3[:, g3.shape[1] // 2, :]                                                  # fine_matching.py:63-68
    heat = torch.softmax(torch.einsum("mc,mrc->mr", f0, win) / Cf ** 0.5, dim=1)
    lin = (torch.linspace(0, W - 1, W, device=heat.device) / (W - 1) - 0.5) * 2
    gx, gy = lin.view(1, W).expand(W, W).reshape(-1), lin.view(W, 1).expand(W, W).reshape(-1)
    coords = torch.stack([(gx * heat).sum(-1), (gy * heat).sum(-1)], dim=-1)
    grid = torch.stack([gx, gy], dim=-1)
    var = torch.sum(grid[None] ** 2 * heat[:, :, None], dim=1) - coords ** 2
    std = torch.sum(torch.sqrt(torch.clamp(var, min=1e-10)), -1)                      # fine_matching.py:92-94
    return conf, torch.cat([coords, std[:, None]], -1)
