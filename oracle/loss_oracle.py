"""CPU restatement of the training-step callers around the matcher.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and
by nothing on the product path); pinned against the reference's own `Loss` / `fine_supervision` through the
reference-generated fixtures tests/golden/loss_*.npz (tests/golden/gen_golden.py::gen_loss).

Follows
  * Loss.compute_coarse_loss        src/lightning_model/losses.py:18-57
  * Loss._compute_fine_loss_l2_std  src/lightning_model/losses.py:65-99
  * Loss.compute_c_weight / forward src/lightning_model/losses.py:101-142
  * fine_supervision                src/models/OnePosePlus/utils/fine_supervision.py:4-32
in plain differentiable torch ops (fp32 like the reference, or fp64 when the inputs are), so that `autograd` gives the
reference gradients of the restatement as well.
"""
import torch


def coarse_focal_loss(conf, conf_gt, weight, alpha, gamma, pos_w, neg_w):
    """losses.py:25-53.  conf [B,N,L] float, conf_gt [B,N,L] integer 0/1, weight [B,N,L] or None."""
    c = conf.clamp(1e-6, 1 - 1e-6)                                           # :26
    is_pos, is_neg = conf_gt == 1, conf_gt == 0
    pos = -alpha * (1 - c[is_pos]) ** gamma * c[is_pos].log()               # :30-34
    neg = -(1 - alpha) * c[is_neg] ** gamma * (1 - c[is_neg]).log()         # :35-39
    if weight is not None:                                                   # :40-42
        pos = pos * weight[is_pos]
        neg = neg * weight[is_neg]
    if pos.numel() == 0:                                                     # :44-53
        return neg_w * neg.mean()
    if neg.numel() == 0:
        return pos_w * pos.mean()
    return pos_w * pos.mean() + neg_w * neg.mean()


def fine_l2_std_loss(expec_f, expec_f_gt, correct_thr, training):
    """losses.py:65-99.  Returns None when no ground truth lies inside its window and `training` is False."""
    inside = expec_f_gt.abs().max(dim=1).values < correct_thr                # :73-75 (inf-norm)
    inv = 1.0 / expec_f[:, 2].clamp(min=1e-10)                               # :78-79
    w = (inv / inv.mean()).detach()                                          # :80-82
    if int(inside.sum()) == 0:                                               # :85-93
        if not training:
            return None
        inside, w = inside.clone(), w.clone()
        inside[0] = True
        w[0] = 1e-6
    sq = ((expec_f_gt[inside] - expec_f[inside, :2]) ** 2).sum(-1)           # :96-98
    return (sq * w[inside]).mean()


def loss_forward(data, cfg, training=True):
    """losses.py:113-142 -> dict(loss, loss_c, loss_f) (loss_f None when the fine term is skipped)."""
    weight = None
    if "mask0" in data:                                                      # :103-109
        weight = data["mask0"].flatten(-2)[..., None] * data["mask1"].flatten(-2)[:, None]
    loss_c = coarse_focal_loss(data["conf_matrix"], data["conf_matrix_gt"], weight, cfg["focal_alpha"], cfg["focal_gamma"],
                               cfg["pos_weight"], cfg["neg_weight"])
    loss = loss_c * cfg["coarse_weight"]
    loss_f = None
    if "expec_f" in data:
        loss_f = fine_l2_std_loss(data["expec_f"], data["expec_f_gt"], cfg["fine_correct_thr"], training)
        if loss_f is not None:
            loss = loss + loss_f * cfg["fine_weight"]
    return {"loss": loss, "loss_c": loss_c, "loss_f": loss_f}


def fine_supervision(data, resolution, window_size):
    """fine_supervision.py:4-32 -> expec_f_gt [M,2]."""
    coarse_res, fine_res = resolution
    radius = window_size // 2
    b, i, j = data["b_ids"], data["i_ids"], data["j_ids"]
    if "query_image_scale" in data:                                          # :17-18
        s = data["query_image_scale"][b][:, [1, 0]]
        coarse_scale, fine_scale = coarse_res * s, fine_res * s
    else:
        coarse_scale, fine_scale = fine_res, fine_res                        # sic: the reference falls back to fine_scale
    wc = data["q_hw_c"][1]
    cell = torch.stack([j % wc, torch.div(j, wc, rounding_mode="floor")], dim=1) * coarse_scale     # :20-23
    return (data["fine_location_matrix_gt"][b, i, j] - cell) / fine_scale / radius                 # :25-28
