"""The training step's forward + backward as a graph of hand-written HIP nodes.

`PL_OnePosePlus.training_step` (src/lightning_model/OnePosePlus_lightning_model.py:54-81) runs `self.matcher(batch)`,
then `fine_supervision` and the loss (`losses.py:114-142`), which differentiates two outputs of the matcher:
`conf_matrix` (focal loss over all B x N x L entries) and `expec_f` (fine L2 loss).  With gradients enabled,
`OnePosePlus_model._forward_train(graph=True)` builds the forward ONCE out of `torch.autograd.Function` nodes whose forward AND
backward are kernels of libopp_hip.so -- nothing is re-evaluated in the backward, and no convolution / GEMM / normalisation of
the step runs in a PyTorch (MIOpen / rocBLAS) operator:
  `HipBackbone`         ResNet-FPN with BatchNorm batch statistics; activations kept on a tape; backward = convolution
                        input / weight gradients, BatchNorm + activation backward, upsample transpose (csrc/conv_bwd.hip)
  `HipLinear`           every Linear of the two transformers and of the keypoint encoder (csrc/linear_bwd.hip)
  `HipLinearAttention`  csrc/linattn_train.hip
  `HipLayerNorm`        csrc/train_misc.hip
  `HipCoarseMatch`      score GEMM + dual softmax + mutual-nearest-neighbour selection = the inference kernels (opp_coarse_match);
                        backward = dual-softmax backward (csrc/loss.hip) + the two feature gradients on the Linear-backward GEMMs
  `HipFineGather`       5 x 5 windows of the fine map; backward scatters (csrc/train_misc.hip)
  `HipFineHead`         FineMatching's expectation head (heatmap softmax, spatial expectation, std) forward + backward (csrc/fine.hip, r05)
PyTorch supplies the tape (autograd), elementwise glue (residual adds, ReLU) and index plumbing.

The helpers `_linear` / `_layer_norm` / `_kpt_encoding` / `_linear_attention` / `_encoder_layer` / `_transformer` / `DualSoftmax` fall
back to plain torch ops for CPU tensors (hp = None): that is how tests/torch_graph_ref.py -- the functional restatement of the whole
graph the tests differentiate against the reference's own gradients -- reuses them; the restatement itself (plain-torch backbone,
`differentiable_forward`) is test infrastructure and lives under tests/.

Each function cites the reference code it differentiates (paths relative to src/models/OnePosePlus/).
"""
import os

import torch
import torch.nn.functional as F

_EPS_LN = 1e-5
# `hp` below = arithmetic of the HIP nodes of the graph being built (2 = bf16x3, 0 = fp32, from the module's gemm_precision);
# None = plain torch ops (CPU tensors).  It is passed explicitly: a process-global would be shared by concurrent steps.


class HipLinear(torch.autograd.Function):
    """y = x W^T for a bias-free nn.Linear (loftr_module/transformer.py:26-47) with forward AND backward on the MFMA GEMM of
    libopp_hip.so: grad_x = grad_y W, grad_W = grad_y^T x as a split-K reduction over the tokens (include/opp_hip.h
    `opp_linear_backward`).  x [..., K] fp32 on the device, W [N, K]; N, K multiples of 32."""

    @staticmethod
    def forward(ctx, x, w, prec):
        from . import _lib
        lib = _lib.load()
        dev = x.device
        x2 = x.reshape(-1, x.shape[-1]).to(torch.float32).contiguous()
        wc = w.to(torch.float32).contiguous()
        M, K = x2.shape
        N = wc.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            wop = wc
            if prec == 2:
                wop = torch.empty(N * K // 2 * 3, dtype=torch.float32, device=dev)
                _lib.check(lib.opp_pack_b3(wc.data_ptr(), wop.data_ptr(), N * K, stream), "opp_pack_b3")
            _lib.check(lib.opp_linear(x2.data_ptr(), M, K, wop.data_ptr(), N, 0, y.data_ptr(), -1, prec, None, stream), "opp_linear")
        ctx.save_for_backward(x2, wc)
        ctx.meta = (prec, tuple(x.shape[:-1]))
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        lib = _lib.load()
        x2, wc = ctx.saved_tensors
        prec, lead = ctx.meta
        dev = x2.device
        M, K = x2.shape
        N = wc.shape[0]
        g2 = gy.reshape(M, N).to(torch.float32).contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = torch.empty((M, K), dtype=torch.float32, device=dev) if need_x else None
        dw = torch.empty((N, K), dtype=torch.float32, device=dev) if need_w else None
        nb = lib.opp_linear_backward_workspace_bytes(M, N, K, prec)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.opp_linear_backward(g2.data_ptr(), x2.data_ptr(), wc.data_ptr(), M, N, K, dx.data_ptr() if need_x else None,
                                               dw.data_ptr() if need_w else None, 0, prec, ws.data_ptr(), nb,
                                               torch.cuda.current_stream(dev).cuda_stream), "opp_linear_backward")
        return (dx.view(*lead, K) if need_x else None), dw, None


class HipLinearAttention(torch.autograd.Function):
    """LinearAttention.forward (loftr_module/linear_attention.py:29-61) with forward and backward in libopp_hip.so
    (csrc/linattn_train.hip): q [B, L, H, D], k, v [B, S, H, D] raw projections, optional 0 / 1 masks [B, L], [B, S]."""

    @staticmethod
    def forward(ctx, q, k, v, q_mask, kv_mask):
        from . import _lib
        lib = _lib.load()
        dev = q.device
        qc, kc, vc = (t.to(torch.float32).contiguous() for t in (q, k, v))
        B, L, H, D = qc.shape
        S = kc.shape[1]
        qm = q_mask.to(device=dev, dtype=torch.float32).contiguous() if q_mask is not None else None
        km = kv_mask.to(device=dev, dtype=torch.float32).contiguous() if kv_mask is not None else None
        out = torch.empty_like(qc)
        kv = torch.empty((B, H, D, D), dtype=torch.float32, device=dev)
        ks = torch.empty((B, H, D), dtype=torch.float32, device=dev)
        nb = lib.opp_linear_attention_train_workspace_bytes(B, L, S, H, D)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(dev):
            _lib.check(lib.opp_linear_attention_train_forward(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), ptr(qm), ptr(km), B, L, S, H, D,
                                                              out.data_ptr(), kv.data_ptr(), ks.data_ptr(), ws.data_ptr(), nb,
                                                              torch.cuda.current_stream(dev).cuda_stream), "opp_linear_attention_train_forward")
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(qc, kc, vc, kv, ks, qm if qm is not None else empty, km if km is not None else empty)
        ctx.meta = (qm is not None, km is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        qc, kc, vc, kv, ks, qm, km = ctx.saved_tensors
        has_qm, has_km = ctx.meta
        dev = qc.device
        B, L, H, D = qc.shape
        S = kc.shape[1]
        gc = g.to(torch.float32).contiguous()
        gq, gk, gv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
        nb = lib.opp_linear_attention_train_workspace_bytes(B, L, S, H, D)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.opp_linear_attention_train_backward(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), qm.data_ptr() if has_qm else None,
                                                               km.data_ptr() if has_km else None, kv.data_ptr(), ks.data_ptr(), gc.data_ptr(),
                                                               B, L, S, H, D, gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), ws.data_ptr(), nb,
                                                               torch.cuda.current_stream(dev).cuda_stream), "opp_linear_attention_train_backward")
        return gq, gk, gv, None, None


class HipLayerNorm(torch.autograd.Function):
    """nn.LayerNorm(C) (loftr_module/transformer.py:87-88, :92-94; eps 1e-5) over the last axis, C in (64, 128, 256), forward and
    backward in libopp_hip.so (csrc/train_misc.hip: one wave per row; d gamma / d beta as a fixed-order reduction over the rows)."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        from . import _lib
        lib = _lib.load()
        dev = x.device
        C = x.shape[-1]
        x2 = x.reshape(-1, C).to(torch.float32).contiguous()
        g, b = gamma.to(torch.float32).contiguous(), beta.to(torch.float32).contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=dev)
        rstd = torch.empty(rows, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.opp_layer_norm_train_forward(x2.data_ptr(), g.data_ptr(), b.data_ptr(), None, rows, C, y.data_ptr(), mean.data_ptr(),
                                                        rstd.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "opp_layer_norm_train_forward")
        ctx.save_for_backward(x2, g, mean, rstd)
        ctx.lead = tuple(x.shape[:-1])
        return y.view(*x.shape)

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        lib = _lib.load()
        x2, g, mean, rstd = ctx.saved_tensors
        dev = x2.device
        rows, C = x2.shape
        g2 = gy.reshape(rows, C).to(torch.float32).contiguous()
        dx = torch.empty_like(x2)
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        nb = lib.opp_layer_norm_train_backward_workspace_bytes(rows, C)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.opp_layer_norm_train_backward(g2.data_ptr(), x2.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C,
                                                         dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), nb,
                                                         torch.cuda.current_stream(dev).cuda_stream), "opp_layer_norm_train_backward")
        return dx.view(*ctx.lead, C), dg, db


def _linear(x, w, hp=None):
    """F.linear(x, w) of a bias-free Linear; on the device a HIP node (forward, input and weight gradient on the MFMA GEMM)"""
    if hp is not None and x.is_cuda and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0:
        return HipLinear.apply(x, w, hp)
    return F.linear(x, w)


def _layer_norm(x, w, b, hp=None):
    if hp is not None and x.is_cuda and x.shape[-1] in (64, 128, 256):
        return HipLayerNorm.apply(x, w, b)
    return F.layer_norm(x, (x.shape[-1],), w, b, _EPS_LN)


def _kpt_encoding(p, kpts, desc, hp=None):
    """normalize_3d_keypoints + KeypointEncoding_linear (utils/normalize.py:16-26, utils/position_encoding.py:54-79;
    per-point channel norm = quirk q3, batch-0 extent = quirk q4)."""
    extent = kpts[0].max(dim=0).values - kpts[0].min(dim=0).values
    x = (kpts - kpts.mean(dim=-2, keepdim=True)) / (extent.max() * 0.6)
    pre = "kpt_3d_pos_encoding.encoder."
    idxs = sorted({int(k[len(pre):].split(".")[0]) for k in p if k.startswith(pre)})
    for n, i in enumerate(idxs):
        w, bias = p[pre + "%d.weight" % i], p[pre + "%d.bias" % i]
        if hp is not None and x.is_cuda and w.shape[0] % 32 == 0:
            # the 1x1 Conv1d layers as HIP Linear nodes; the 3-channel input of the first one zero-padded to the GEMM's K granule
            kpad = (-w.shape[1]) % 32
            x = _linear(F.pad(x, (0, kpad)) if kpad else x, F.pad(w, (0, kpad)) if kpad else w, hp) + bias
        else:
            x = F.linear(x, w, bias)
        if n < len(idxs) - 1:
            mu = x.mean(dim=-1, keepdim=True)
            var = x.var(dim=-1, unbiased=False, keepdim=True)
            x = F.relu((x - mu) / torch.sqrt(var + _EPS_LN))
    return desc + x.transpose(2, 1)


def _linear_attention(q, k, v, q_mask=None, kv_mask=None, eps=1e-6, hp=None):      # loftr_module/linear_attention.py:29-61
    if hp is not None and q.is_cuda and q.shape[-1] in (16, 32) and eps == 1e-6 and q.shape[0] <= 65535:   # (grid.z = sample)
        return HipLinearAttention.apply(q, k, v, q_mask, kv_mask)
    Q, K = F.elu(q) + 1, F.elu(k) + 1
    if q_mask is not None:
        Q = Q * q_mask[:, :, None, None]
    if kv_mask is not None:
        K = K * kv_mask[:, :, None, None]
        v = v * kv_mask[:, :, None, None]
    S = v.size(1)
    v = v / S
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    return torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S


def _encoder_layer(p, name, nhead, x, source, x_mask=None, source_mask=None, hp=None):      # loftr_module/transformer.py:65-94
    B, _, C = x.shape
    D = C // nhead
    wq, wk, wv = p[name + ".q_proj.weight"], p[name + ".k_proj.weight"], p[name + ".v_proj.weight"]
    if hp is not None and x.is_cuda:
        # projections that read the same tokens run as ONE Linear node on the stacked weight: its backward transposes / splits
        # the shared input once and reduces one [2C | 3C] x C weight gradient instead of two or three C x C ones
        if source is x:
            q, k, v = _linear(x, torch.cat([wq, wk, wv], 0), hp).split(C, dim=2)
        else:
            q = _linear(x, wq, hp)
            k, v = _linear(source, torch.cat([wk, wv], 0), hp).split(C, dim=2)
        q, k, v = (t.reshape(B, -1, nhead, D) for t in (q, k, v))
    else:
        q = _linear(x, wq).view(B, -1, nhead, D)
        k = _linear(source, wk).view(B, -1, nhead, D)
        v = _linear(source, wv).view(B, -1, nhead, D)
    msg = _linear_attention(q, k, v, x_mask, source_mask, hp=hp).reshape(B, -1, C)
    msg = _layer_norm(_linear(msg, p[name + ".merge.weight"], hp), p[name + ".norm1.weight"], p[name + ".norm1.bias"], hp)
    msg = _linear(F.relu(_linear(torch.cat([x, msg], dim=2), p[name + ".mlp.0.weight"], hp)), p[name + ".mlp.2.weight"], hp)
    return x + _layer_norm(msg, p[name + ".norm2.weight"], p[name + ".norm2.bias"], hp)


def _transformer(p, name, tcfg, f3, f2, mask=None, hp=None):      # loftr_module/transformer.py:133-171 (f3 already [B, N, C])
    for i, kind in enumerate(list(tcfg["layer_names"]) * tcfg["layer_iter_n"]):
        n = "%s.layers.%d" % (name, i)
        if kind == "self":
            f2, f3 = _encoder_layer(p, n, tcfg["nhead"], f2, f2, mask, mask, hp), _encoder_layer(p, n, tcfg["nhead"], f3, f3, hp=hp)
        else:       # cross: both streams from the pre-update tensors (quirk q6)
            f2, f3 = (_encoder_layer(p, n, tcfg["nhead"], f2, f3, x_mask=mask, hp=hp),
                      _encoder_layer(p, n, tcfg["nhead"], f3, f2, source_mask=mask, hp=hp))
    return f3, f2


class DualSoftmax(torch.autograd.Function):
    """conf = softmax(S, dim=1) * softmax(S, dim=2) for S [B, N, L] (utils/coarse_matching.py:115) with a hand-written
    backward: only S and its two log-sum-exps are kept (autograd would keep both softmax outputs and the product - three
    B x N x L tensors), and on the device d loss / dS comes from `opp_dual_softmax_backward` (csrc/loss.hip):
        dS_ij = 2 conf_ij g_ij - A_ij sum_i' g_i'j conf_i'j - B_ij sum_j' g_ij' conf_ij'
    CPU tensors (the CPU tests of this module) evaluate the same formula with torch ops."""

    @staticmethod
    def forward(ctx, sim):
        lse_col = torch.logsumexp(sim, dim=1)              # over the points i, per cell j     [B, L]
        lse_row = torch.logsumexp(sim, dim=2)              # over the cells j, per point i     [B, N]
        ctx.save_for_backward(sim, lse_row, lse_col)
        return torch.exp(sim - lse_col[:, None, :]) * torch.exp(sim - lse_row[:, :, None])

    @staticmethod
    def backward(ctx, g):
        sim, lse_row, lse_col = ctx.saved_tensors
        if sim.is_cuda and sim.dtype == torch.float32:
            from . import _lib
            lib = _lib.load()
            B, N, L = sim.shape
            s, gc = sim.contiguous(), g.to(torch.float32).contiguous()
            lr, lc = lse_row.contiguous(), lse_col.contiguous()
            ds = torch.empty_like(s)
            ws = torch.empty(lib.opp_dual_softmax_backward_workspace_bytes(B, N, L), dtype=torch.uint8, device=s.device)
            with torch.cuda.device(s.device):
                _lib.check(lib.opp_dual_softmax_backward(gc.data_ptr(), s.data_ptr(), lr.data_ptr(), lc.data_ptr(), B, N, L,
                                                         ds.data_ptr(), ws.data_ptr(), ws.numel(),
                                                         torch.cuda.current_stream(s.device).cuda_stream),
                           "opp_dual_softmax_backward")
            return ds
        A = torch.exp(sim - lse_col[:, None, :])
        Bm = torch.exp(sim - lse_row[:, :, None])
        gc = g * A * Bm
        return 2.0 * gc - A * gc.sum(1, keepdim=True) - Bm * gc.sum(2, keepdim=True)


# ---------------------------------------------------------------------------------------------------------------------------
# device graph of the training step: nodes around the C ABI of libopp_hip.so (include/opp_hip.h)
# ---------------------------------------------------------------------------------------------------------------------------
class HipBackbone(torch.autograd.Function):
    """ResNetFPN_8_2.forward in train() mode (backbone/resnet.py:141-164) on the HIP path, keeping its activations on a tape
    (`opp_backbone_train_tape`), with the whole backward in libopp_hip.so (`opp_backbone_backward`: convolution input / weight
    gradients, BatchNorm + activation backward, upsample transpose).  Returns feat_c [B, L, 256] and feat_f [B, Hf * Wf, 128]
    (NHWC = token-major).  `table_idx[k]` = position of params[k] in the C context's weight table."""

    @staticmethod
    def forward(ctx, model, img, table_idx, *params):
        from . import _lib
        dev = img.device
        lib, c = model._ensure_ready(dev, scope=1)
        model._ensure_train_packed(lib, c, dev)
        cfg = model.config
        B, H, W = int(img.shape[0]), int(img.shape[2]), int(img.shape[3])
        dC, dF = cfg["loftr_coarse"]["d_model"], cfg["loftr_fine"]["d_model"]
        feat_c = torch.empty((B, (H // 8) * (W // 8), dC), dtype=torch.float32, device=dev)
        feat_f = torch.empty((B, (H // 2) * (W // 2), dF), dtype=torch.float32, device=dev)
        tape = torch.empty(lib.opp_backbone_tape_bytes(c, B, H, W), dtype=torch.uint8, device=dev)
        ws = model._workspace(lib.opp_backbone_train_tape_workspace_bytes(c, B, H, W), dev)
        n_bn = lib.opp_num_bn_layers(c)
        stats = torch.zeros((n_bn, 512), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.opp_backbone_train_tape(c, img.data_ptr(), B, H, W, feat_c.data_ptr(), feat_f.data_ptr(), stats.data_ptr(),
                                               tape.data_ptr(), tape.numel(), ws.data_ptr(), ws.numel(), stream), "opp_backbone_train_tape")
        model._update_running_stats(lib, c, stats)
        ctx.model, ctx.img, ctx.tape, ctx.table_idx = model, img, tape, table_idx
        ctx.c_ctx = c                                                    # the C context whose packed weights / BatchNorm pointers the tape belongs to
        ctx.ptrs, ctx.keep = model._rt["ptrs"], model._rt["keep"]       # the weight table the tape's forward ran on (kept alive)
        ctx.geom = (B, H, W)
        ctx.shapes = [tuple(p.shape) for p in params]
        # the backward reads the image and the parameters IN PLACE through raw pointers (convolution weights for the input gradients, gamma
        # for the BatchNorm backward): remember their version counters so that an in-place update between forward and backward -- an
        # optimiser step before a delayed backward, a reused input buffer -- raises like PyTorch's saved-tensor check instead of silently
        # producing gradients of another function (r04 advisor finding)
        ctx.versions = [(t, t._version) for t in (img,) + tuple(params)]
        return feat_c, feat_f

    @staticmethod
    def backward(ctx, g_fc, g_ff):
        import ctypes
        from . import _lib
        lib = _lib.load()
        model = ctx.model
        c = model._rt["ctx"]
        if c is not ctx.c_ctx or ctx.tape is None:
            raise RuntimeError("HipBackbone.backward: the module's runtime was re-created (set_gemm_precision / .to()) or the graph was "
                               "already differentiated between this step's forward and backward")
        for t, v in ctx.versions:
            if t._version != v:
                raise RuntimeError("HipBackbone.backward: one of the variables needed for gradient computation has been modified by an inplace "
                                   "operation (the image or a backbone parameter changed between this step's forward and its backward: "
                                   "version %d, expected %d)" % (t._version, v))
        dev = ctx.img.device
        B, H, W = ctx.geom
        ptrs, n = ctx.ptrs
        grads = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ctx.shapes]
        gp = (ctypes.c_void_p * n)()
        for k, i in enumerate(ctx.table_idx):
            gp[i] = grads[k].data_ptr()
        g_fc = g_fc.to(torch.float32).contiguous()
        g_ff = g_ff.to(torch.float32).contiguous()
        nb = lib.opp_backbone_backward_workspace_bytes(c, B, H, W)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.opp_backbone_backward(c, ctx.img.data_ptr(), B, H, W, ctx.tape.data_ptr(), ctx.tape.numel(), ptrs, n, g_fc.data_ptr(),
                                                 g_ff.data_ptr(), gp, ws.data_ptr(), nb, torch.cuda.current_stream(dev).cuda_stream),
                       "opp_backbone_backward")
        ctx.tape = None
        return (None, None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs_input_grad[3:]))


class HipCoarseMatch(torch.autograd.Function):
    """CoarseMatching.forward (utils/coarse_matching.py:99-172) on the transformer outputs f3 [B, N, C] (3D points) and f2 [B, L, C]
    (image cells): forward = the inference kernels, sample by sample (`opp_coarse_match`: score GEMM with the dual-softmax
    statistics, confidences, mutual-nearest-neighbour selection); returns conf_matrix [B, N, L] and leaves the selection (i_all,
    j_all, c_all, counts) in `aux`.  backward: the score matrix is recomputed by one GEMM per sample, `opp_dual_softmax_forward`
    gives its log-sum-exps, `opp_dual_softmax_backward` the gradient of the scores, `opp_linear_backward` those of f3 and f2."""

    @staticmethod
    def forward(ctx, f3, f2, aux):
        from . import _lib
        model, lib, c = aux["model"], aux["lib"], aux["ctx"]
        dev = f3.device
        f3c, f2c = f3.to(torch.float32).contiguous(), f2.to(torch.float32).contiguous()
        B, N, C = f3c.shape
        L = f2c.shape[1]
        hc, wc = aux["hc"], aux["wc"]
        kpts, mask, qscale = aux["kpts"], aux["mask"], aux["qscale"]
        conf = torch.empty((B, N, L), dtype=torch.float32, device=dev)
        i_all = torch.empty((B, N), dtype=torch.int64, device=dev)
        j_all = torch.empty((B, N), dtype=torch.int64, device=dev)
        c_all = torch.empty((B, N), dtype=torch.float32, device=dev)
        mkc = torch.empty((N, 2), dtype=torch.float32, device=dev)
        mk3 = torch.empty((N, 3), dtype=torch.float32, device=dev)
        counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        ws = model._workspace(max(lib.opp_coarse_match_workspace_bytes(c, N, L), 4096), dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        try:
            for b in range(B):
                _lib.check(lib.opp_set_query_mask(c, mask[b].data_ptr() if mask is not None else None), "query_mask")
                _lib.check(lib.opp_coarse_match(c, f3c[b].data_ptr(), f2c[b].data_ptr(), N, hc, wc, kpts[b].data_ptr(), aux["scale_c"],
                                                qscale[b].data_ptr() if qscale is not None else None, conf[b].data_ptr(),
                                                i_all[b].data_ptr(), j_all[b].data_ptr(), c_all[b].data_ptr(), mkc.data_ptr(),
                                                mk3.data_ptr(), counts[b].data_ptr(), ws.data_ptr(), ws.numel(), stream), "opp_coarse_match")
        finally:
            lib.opp_set_query_mask(c, None)
        aux.update({"i_all": i_all, "j_all": j_all, "c_all": c_all, "counts": counts})
        ctx.save_for_backward(f3c, f2c)
        ctx.mask, ctx.hp = mask, aux["hp"]
        ctx.scale = (1.0 / C) / (float(model.config["coarse_matching"]["dual_softmax"]["temperature"]) + 1e-4)
        return conf

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        f3, f2 = ctx.saved_tensors
        dev = f3.device
        B, N, C = f3.shape
        L = f2.shape[1]
        hp = ctx.hp
        stream = torch.cuda.current_stream(dev).cuda_stream
        sim = torch.empty((B, N, L), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            wop = torch.empty(L * C // 2 * 3, dtype=torch.float32, device=dev) if hp == 2 else None
            for b in range(B):
                w = f2[b]
                if hp == 2:
                    _lib.check(lib.opp_pack_b3(f2[b].data_ptr(), wop.data_ptr(), L * C, stream), "opp_pack_b3")
                    w = wop
                _lib.check(lib.opp_linear(f3[b].data_ptr(), N, C, w.data_ptr(), L, 0, sim[b].data_ptr(), -1, hp, None, stream), "opp_linear")
            sim.mul_(ctx.scale)
            if ctx.mask is not None:                                    # coarse_matching.py:108-114
                sim.add_(torch.where(ctx.mask[:, None].bool(), 0.0, -1e9).to(sim.dtype))
            lse_row = torch.empty((B, N), dtype=torch.float32, device=dev)
            lse_col = torch.empty((B, L), dtype=torch.float32, device=dev)
            nb = max(lib.opp_dual_softmax_forward_workspace_bytes(B, N, L), lib.opp_dual_softmax_backward_workspace_bytes(B, N, L),
                     lib.opp_linear_backward_workspace_bytes(N, L, C, hp))
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            _lib.check(lib.opp_dual_softmax_forward(sim.data_ptr(), B, N, L, lse_row.data_ptr(), lse_col.data_ptr(), None, ws.data_ptr(), nb, stream),
                       "opp_dual_softmax_forward")
            gc = g.to(torch.float32).contiguous()
            ds = torch.empty_like(sim)
            _lib.check(lib.opp_dual_softmax_backward(gc.data_ptr(), sim.data_ptr(), lse_row.data_ptr(), lse_col.data_ptr(), B, N, L, ds.data_ptr(),
                                                     ws.data_ptr(), nb, stream), "opp_dual_softmax_backward")
            ds.mul_(ctx.scale)
            # sim_b = f3_b f2_b^T is a Linear with x = f3_b, W = f2_b.  The GEMM wants the cell count in multiples of 32: other image sizes
            # (L = H/8 * W/8) get zero cells appended, which changes neither gradient
            Lp = (L + 31) // 32 * 32
            if Lp != L:
                ds = torch.nn.functional.pad(ds, (0, Lp - L))
                f2 = torch.nn.functional.pad(f2, (0, 0, 0, Lp - L))
                nb2 = lib.opp_linear_backward_workspace_bytes(N, Lp, C, hp)
                if nb2 > nb:
                    nb, ws = nb2, torch.empty(nb2, dtype=torch.uint8, device=dev)
            g3, g2 = torch.empty_like(f3), torch.empty_like(f2)
            for b in range(B):
                _lib.check(lib.opp_linear_backward(ds[b].data_ptr(), f3[b].data_ptr(), f2[b].data_ptr(), N, Lp, C, g3[b].data_ptr(), g2[b].data_ptr(),
                                                   0, hp, ws.data_ptr(), nb, stream), "opp_linear_backward")
        return g3, g2[:, :L], None


class HipFineGather(torch.autograd.Function):
    """FinePreprocess (loftr_module/fine_preprocess.py:41-55): the W x W window of the fine map around every selected coarse cell,
    straight from the NHWC map (no unfold matrix): feat_f [B, Hf * Wf, C] -> windows [M, W * W, C]; the backward adds the window
    gradients back into the map (`opp_fine_window_gather_backward`)."""

    @staticmethod
    def forward(ctx, feat_f, b_ids, j_ids, geom):
        from . import _lib
        lib = _lib.load()
        B, Hf, Wf, hc, wc, W = geom
        dev = feat_f.device
        ff = feat_f.to(torch.float32).contiguous()
        C = ff.shape[-1]
        bi, ji = b_ids.to(torch.int64).contiguous(), j_ids.to(torch.int64).contiguous()
        M = int(bi.numel())
        win = torch.empty((M, W * W, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.opp_fine_window_gather(ff.data_ptr(), B, Hf, Wf, C, bi.data_ptr(), ji.data_ptr(), M, hc, wc, W, win.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream), "opp_fine_window_gather")
        ctx.save_for_backward(bi, ji)
        ctx.geom, ctx.C, ctx.shape = geom, C, tuple(feat_f.shape)
        return win

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        bi, ji = ctx.saved_tensors
        B, Hf, Wf, hc, wc, W = ctx.geom
        dev = g.device
        gc = g.to(torch.float32).contiguous()
        d = torch.empty((B, Hf * Wf, ctx.C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.opp_fine_window_gather_backward(gc.data_ptr(), B, Hf, Wf, ctx.C, bi.data_ptr(), ji.data_ptr(), int(bi.numel()), hc, wc, W,
                                                           d.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "opp_fine_window_gather_backward")
        return d.view(ctx.shape), None, None, None


class HipFineHead(torch.autograd.Function):
    """FineMatching._s2d_heatmap (utils/fine_matching.py:63-94): expec_f [M, 3] = (spatial expectation x, y of softmax(<f0, window> / sqrt(C)), summed
    standard deviation) from the point tokens f0 [M, C] and the window tokens win [M, W * W, C]; forward = the inference kernel, backward =
    `opp_fine_head_train_backward` (csrc/fine.hip) -- the last piece of the fine level that ran as ATen elementwise ops."""

    @staticmethod
    def forward(ctx, f0, win):
        from . import _lib
        lib = _lib.load()
        dev = f0.device
        f0c, wc = f0.to(torch.float32).contiguous(), win.to(torch.float32).contiguous()
        M, WW, C = wc.shape
        W = int(round(WW ** 0.5))
        expec = torch.empty((M, 3), dtype=torch.float32, device=dev)
        scratch = torch.empty(max(4 * M, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.opp_fine_head_train_forward(f0c.data_ptr(), wc.data_ptr(), M, W, C, expec.data_ptr(), scratch.data_ptr(),
                                                       torch.cuda.current_stream(dev).cuda_stream), "opp_fine_head_train_forward")
        ctx.save_for_backward(f0c, wc)
        return expec

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        f0c, wc = ctx.saved_tensors
        dev = f0c.device
        M, WW, C = wc.shape
        W = int(round(WW ** 0.5))
        gc = g.to(torch.float32).contiguous()
        gf, gw = torch.empty_like(f0c), torch.empty_like(wc)
        with torch.cuda.device(dev):
            _lib.check(lib.opp_fine_head_train_backward(f0c.data_ptr(), wc.data_ptr(), M, W, C, gc.data_ptr(), gf.data_ptr(), gw.data_ptr(),
                                                        torch.cuda.current_stream(dev).cuda_stream), "opp_fine_head_train_backward")
        return gf, gw


def graph_precision(model):
    """`hp` of the device graph from the module's gemm_precision"""
    return {"bf16x3": 2, "fp32": 0}[model.gemm_precision]


def backbone_node(model, lib, c, img):
    """feat_c [B, L, dC], feat_f [B, Hf * Wf, dF] with a grad_fn (`HipBackbone`)"""
    names = model._rt["names"]
    params = dict(model.named_parameters())
    idx = [i for i, n in enumerate(names) if n.startswith("backbone.") and n in params]
    return HipBackbone.apply(model, img, tuple(idx), *[params[names[i]] for i in idx])


def coarse_level_graph(model, lib, c, p, feat_c, pe, kpts, bank_c, mask, qscale, hc, wc, scale_c):
    """tokens (OnePosePlusModel.py:137-156) -> loftr_coarse -> CoarseMatching: conf_matrix [B, N, L] with a grad_fn + the selection"""
    cfg = model.config
    hp = graph_precision(model)
    tokens2d = feat_c + pe[None] if pe is not None else feat_c               # PositionEncodingSine on the NHWC tokens
    if cfg["keypoints_encoding"]["enable"]:
        tokens3d = _kpt_encoding(p, kpts, bank_c, hp).transpose(1, 2)
    else:
        tokens3d = bank_c.transpose(1, 2)
    f3, f2 = _transformer(p, "loftr_coarse", cfg["loftr_coarse"], tokens3d.contiguous(), tokens2d, mask, hp)
    aux = {"model": model, "lib": lib, "ctx": c, "hc": hc, "wc": wc, "kpts": kpts, "mask": mask, "qscale": qscale, "scale_c": scale_c, "hp": hp}
    conf = HipCoarseMatch.apply(f3, f2, aux)
    return conf, aux


def fine_level_graph(model, p, feat_f, bank_f, b_ids, i_ids, j_ids, B, hf, wf, hc, wc):
    """FinePreprocess -> loftr_fine -> FineMatching._s2d_heatmap (fine_preprocess.py:41-55, fine_matching.py:63-94): expec_f [M, 3]"""
    cfg = model.config
    hp = graph_precision(model)
    fcfg = cfg["loftr_fine"]
    W, Cf = fcfg["window_size"], fcfg["d_model"]
    win = HipFineGather.apply(feat_f, b_ids, j_ids, (B, hf, wf, hc, wc, W))           # [M, WW, C]
    g3 = bank_f.permute(0, 2, 1)[b_ids, i_ids].unsqueeze(1)                           # [M, 1, C]: the RAW fine bank (quirk q8)
    if fcfg["enable"]:
        g3, win = _transformer(p, "loftr_fine", fcfg, g3.contiguous(), win, None, hp)
    f0 = g3[:, g3.shape[1] // 2, :]                                                   # fine_matching.py:63-68
    if win.is_cuda and Cf % 4 == 0 and W * W <= 64 and W > 1 and win.shape[0] > 0:
        return HipFineHead.apply(f0, win)                                             # heatmap, expectation, std: forward and backward in HIP
    heat = torch.softmax((f0[:, None, :] * win).sum(-1) / Cf ** 0.5, dim=1)           # 'mc,mrc->mr' as multiply + reduce
    lin = (torch.linspace(0, W - 1, W, device=heat.device) / (W - 1) - 0.5) * 2
    gx, gy = lin.view(1, W).expand(W, W).reshape(-1), lin.view(W, 1).expand(W, W).reshape(-1)
    coords = torch.stack([(gx * heat).sum(-1), (gy * heat).sum(-1)], dim=-1)
    grid = torch.stack([gx, gy], dim=-1)
    var = torch.sum(grid[None] ** 2 * heat[:, :, None], dim=1) - coords ** 2
    std = torch.sum(torch.sqrt(torch.clamp(var, min=1e-10)), -1)                      # fine_matching.py:92-94
    return torch.cat([coords, std[:, None]], -1)
