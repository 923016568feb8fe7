"""Training-step callers of the matcher, with the same names, arguments and `data` keys as the reference:

  * `Loss`              - src/lightning_model/losses.py:7-142 (`PL_OnePosePlus.__init__`, lightning_model:26, then
                          `self.loss(batch)` in `training_step`, lightning_model:59-60)
  * `fine_supervision`  - src/models/OnePosePlus/utils/fine_supervision.py:4-32 (lightning_model:57)

The coarse focal loss - the one part of the step that sweeps the whole B x N x L confidence matrix (115 M entries at
B = 4, N = 7000) - runs in `libopp_hip.so` (`csrc/loss.hip`): one pass for the four sums the loss is made of, one
elementwise pass for d loss / d conf.  The fine-level l2-with-std term and the ground-truth offsets work on the few
thousand match rows and stay index / elementwise glue in the module, like the training branch of `get_coarse_match`.

`Loss` needs device tensors (there is no CPU fallback: the HIP library is the product path; the CPU restatement used by
the tests lives in oracle/loss_oracle.py).
"""
import torch
import torch.nn as nn

from . import _lib


_GT_KINDS = {torch.int16: 0, torch.float32: 1, torch.uint8: 2, torch.bool: 2}


def _gt_operand(conf_gt, dev):
    """conf_matrix_gt as the kernels read it: int16, fp32 and 8-bit (bool) tensors go through untouched; other dtypes are cast
    to fp32 (values other than 0 / 1 stay what they are and are ignored, like the reference's `== 1` / `== 0` masks)."""
    g = conf_gt.to(dev)
    if g.dtype not in _GT_KINDS:
        g = g.to(torch.float32)
    g = g.contiguous()
    return g, _GT_KINDS[g.dtype]


class _FocalLoss(torch.autograd.Function):
    """pos_weight * mean(pos terms) + neg_weight * mean(neg terms) of losses.py:25-53 as one differentiable scalar.
    `weight` is either a full tensor broadcastable to conf or a pair (mask0 [B, N], mask1 [B, L]) whose outer product it is
    (Loss.compute_c_weight): the pair is multiplied inside the kernels, nothing of size B x N x L is materialised."""

    @staticmethod
    def forward(ctx, conf, conf_gt, weight, alpha, gamma, pos_w, neg_w):
        if not conf.is_cuda:
            raise RuntimeError("onepose_plus_plus_amd.losses.Loss runs on the HIP device; got a %s tensor" % conf.device)
        lib = _lib.load()
        dev = conf.device
        c = conf.detach().to(torch.float32).contiguous()
        g, kind = _gt_operand(conf_gt, dev)
        if g.shape != c.shape:
            raise ValueError("conf_matrix %s and conf_matrix_gt %s differ in shape" % (tuple(c.shape), tuple(g.shape)))
        w = m0 = m1 = None
        N = L = 0
        if isinstance(weight, (tuple, list)):
            m0 = weight[0].to(device=dev, dtype=torch.float32).contiguous()
            m1 = weight[1].to(device=dev, dtype=torch.float32).contiguous()
            if c.dim() != 3 or tuple(m0.shape) != tuple(c.shape[:2]) or tuple(m1.shape) != (c.shape[0], c.shape[2]):
                raise ValueError("mask pair %s / %s does not match conf_matrix %s" % (tuple(m0.shape), tuple(m1.shape), tuple(c.shape)))
            N, L = int(c.shape[1]), int(c.shape[2])
        elif weight is not None:
            w = weight.to(device=dev, dtype=torch.float32).expand_as(c).contiguous()
        n = c.numel()
        stream = torch.cuda.current_stream(dev).cuda_stream
        sums = torch.empty(4, dtype=torch.float64, device=dev)
        ws = torch.empty(lib.opp_focal_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(dev):
            _lib.check(lib.opp_focal_loss_forward_ex(c.data_ptr(), g.data_ptr(), kind, ptr(w), ptr(m0), ptr(m1), N, L, n, float(alpha),
                                                     float(gamma), sums.data_ptr(), ws.data_ptr(), ws.numel(), stream),
                       "opp_focal_loss_forward_ex")
        pos_sum, neg_sum, n_pos, n_neg = sums.unbind()
        pos_mean = pos_sum / n_pos.clamp(min=1.0)
        neg_mean = neg_sum / n_neg.clamp(min=1.0)
        # losses.py:44-53: an empty positive (negative) set drops that term instead of producing the NaN of an empty mean
        both = pos_w * pos_mean + neg_w * neg_mean
        loss = torch.where(n_pos == 0, neg_w * neg_mean, torch.where(n_neg == 0, pos_w * pos_mean, both))
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(c, g, w if w is not None else empty, m0 if m0 is not None else empty, m1 if m1 is not None else empty, n_pos, n_neg)
        ctx.meta = (float(alpha), float(gamma), float(pos_w), float(neg_w), conf.dtype, kind, w is not None, m0 is not None, N, L)
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        c, g, w, m0, m1, n_pos, n_neg = ctx.saved_tensors
        alpha, gamma, pos_w, neg_w, dtype, kind, has_w, has_m, N, L = ctx.meta
        lib = _lib.load()
        dev = c.device
        go = grad_out.to(torch.float64)
        scales = torch.stack([go * pos_w / n_pos.clamp(min=1.0), go * neg_w / n_neg.clamp(min=1.0)]).to(torch.float32).contiguous()
        grad = torch.empty_like(c)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(lib.opp_focal_loss_backward_ex(c.data_ptr(), g.data_ptr(), kind, w.data_ptr() if has_w else None,
                                                      m0.data_ptr() if has_m else None, m1.data_ptr() if has_m else None, N, L, c.numel(),
                                                      alpha, gamma, scales.data_ptr(), grad.data_ptr(), stream),
                       "opp_focal_loss_backward_ex")
        return grad.to(dtype), None, None, None, None, None, None


class Loss(nn.Module):
    """Same constructor (`config` = the `loss` block of the experiment YAML, train.yaml:129-144) and `forward(data)`
    contract as the reference: reads `conf_matrix`, `conf_matrix_gt`, `expec_f`, `expec_f_gt` (+ `mask0` / `mask1` when
    present), writes `loss` and `loss_scalars`."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.correct_thr = config["fine_correct_thr"]      # losses.py:12 (1 = inside the fine window)
        self.c_pos_w = config["pos_weight"]
        self.c_neg_w = config["neg_weight"]
        self.fine_type = config["fine_type"]

    def compute_coarse_loss(self, conf, conf_gt, weight=None):
        """weight: None, a tensor broadcastable to conf (the reference's argument), or the pair (mask0 [B, N], mask1 [B, L])"""
        if self.config["coarse_type"] != "focal":          # losses.py:56-57
            raise NotImplementedError
        return _FocalLoss.apply(conf, conf_gt, weight, self.config["focal_alpha"], self.config["focal_gamma"],
                                self.c_pos_w, self.c_neg_w)

    def compute_fine_loss(self, expec_f, expec_f_gt):
        if self.fine_type != "l2_with_std":                # losses.py:59-63
            raise NotImplementedError()
        return self._compute_fine_loss_l2_std(expec_f, expec_f_gt)

    def _compute_fine_loss_l2_std(self, expec_f, expec_f_gt):
        """expec_f [M,3] = (x, y, std), expec_f_gt [M,2] (losses.py:65-99): squared offset error of the matches whose
        ground truth falls inside the window, weighted by the (detached, mean-normalised) inverse std."""
        inside = expec_f_gt.abs().amax(dim=1) < self.correct_thr              # inf-norm, :73-75
        inv_std = 1.0 / expec_f[:, 2].clamp(min=1e-10)
        weight = (inv_std / inv_std.mean()).detach()                           # :80-82
        if not bool(inside.any()):                                             # :85-93 (the reference syncs here too)
            if not self.training:
                return None
            inside = inside.clone()
            weight = weight.clone()
            inside[0] = True                                                    # a dummy term keeps DDP ranks in step
            weight[0] = 1e-6
        err = (expec_f_gt[inside] - expec_f[inside, :2]).pow(2).sum(-1)        # :96-99
        return (err * weight[inside]).mean()

    @torch.no_grad()
    def compute_c_weight(self, data):
        """losses.py:103-111 (same values: mask0 [B, N] x mask1 [B, L] -> [B, N, L]).  `forward` does not call it: it hands the
        two factors to the focal kernels, which multiply them on the fly."""
        if "mask0" not in data:
            return None
        return data["mask0"].flatten(-2)[..., None] * data["mask1"].flatten(-2)[:, None]

    @torch.no_grad()
    def _c_weight_factors(self, data):
        if "mask0" not in data:
            return None
        return data["mask0"].flatten(-2).to(torch.float32), data["mask1"].flatten(-2).to(torch.float32)

    def forward(self, data):
        scalars = {}
        loss_c = self.compute_coarse_loss(data["conf_matrix"], data["conf_matrix_gt"], weight=self._c_weight_factors(data))
        loss = loss_c * self.config["coarse_weight"]                           # :125-128
        scalars["loss_c"] = loss_c.clone().detach().cpu()
        if "expec_f" in data:                                                  # :131-138
            loss_f = self.compute_fine_loss(data["expec_f"], data["expec_f_gt"])
            if loss_f is not None:
                loss = loss + loss_f * self.config["fine_weight"]
                scalars["loss_f"] = loss_f.clone().detach().cpu()
            else:
                assert self.training is False
                scalars["loss_f"] = torch.tensor(1.0)                          # the upper bound
        scalars["loss"] = loss.clone().detach().cpu()
        data.update({"loss": loss, "loss_scalars": scalars})


@torch.no_grad()
def fine_supervision(data, config):
    """`expec_f_gt` [M,2]: where inside its 5x5 window (in units of the window radius) the ground-truth 2D location of
    every coarse match lies (fine_supervision.py:4-32).  `config` is the experiment config (`config['OnePosePlus']`)."""
    coarse_res, fine_res = list(config["OnePosePlus"]["loftr_backbone"]["resolution"])
    radius = config["OnePosePlus"]["loftr_fine"]["window_size"] // 2
    b_ids, i_ids, j_ids = data["b_ids"], data["i_ids"], data["j_ids"]
    if "query_image_scale" in data:
        per_match = data["query_image_scale"][b_ids][:, [1, 0]]               # (w, h) scale of the match's image
        coarse_scale, fine_scale = coarse_res * per_match, fine_res * per_match
    else:
        coarse_scale = fine_scale = fine_res                                   # the reference's own fallback (:17-18)
    wc = data["q_hw_c"][1]
    cell_xy = torch.stack([j_ids % wc, j_ids // wc], dim=1) * coarse_scale    # top-left of the coarse cell, image pixels
    target = data["fine_location_matrix_gt"][b_ids, i_ids, j_ids]
    data.update({"expec_f_gt": (target - cell_xy) / fine_scale / radius})
