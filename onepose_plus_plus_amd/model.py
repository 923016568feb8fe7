"""`OnePosePlus_model`: host-side mirror of the reference module API on top of libopp_hip.so.

Drop-in for /root/reference/src/models/OnePosePlus/OnePosePlusModel.py:25-201 on the
inference path (`inference.py` -> src/inference/inference_OnePosePlus_worker.py:7-37):

  * same constructor `OnePosePlus_model(config, profiler=None, debug=False)`, same config
    mapping, same state-dict keys/shapes (195 entries; `load_state_dict(strict=True)` of a
    reference checkpoint works, incl. the `matcher.` stripping done by the caller);
  * `model(data) -> None` mutating `data` in place with the same keys / dtypes / shapes;
  * picklable (Ray passes the module to workers, src/inference/inference_OnePosePlus.py:85-95)
    and device-movable with `.cuda()`.

All tensor math runs in hand-written HIP kernels through the C ABI (include/opp_hip.h).
PyTorch is only used for parameter storage, device memory and the current stream.  There is
NO CPU / PyTorch fallback: CPU tensors, training mode, or a missing library raise.
"""
import ctypes
import os
import math
import warnings

import torch
import torch.nn as nn

from . import _lib
from .params import param_spec


class _PassThroughProfiler:
    """Stand-in for src/utils/profiler.py:42-77 (only record_function is reached from the
    hot path: coarse_matching.py:122,167)."""

    class _Ctx:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def record_function(self, name):
        return self._Ctx()

    profile = record_function


class _Node(nn.Module):
    """Anonymous container so that parameters live under the reference's dotted names."""


def _register(root, dotted, tensor, buffer=False):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


def _init_tensor(shape, kind):
    if kind == "conv":      # kaiming_normal_(fan_out, relu): backbone/resnet.py:126-128
        return torch.randn(shape) * math.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))
    if kind == "xavier":    # xavier_uniform_: loftr_module/transformer.py:128-131
        a = math.sqrt(6.0 / (shape[0] + shape[1]))
        return (torch.rand(shape) * 2 - 1) * a
    if kind == "linear_w":  # nn.Linear default
        a = 1.0 / math.sqrt(shape[1])
        return (torch.rand(shape) * 2 - 1) * a
    if kind == "linear_b":
        return torch.zeros(shape)
    if kind in ("bn_weight", "ln_weight", "bn_var"):
        return torch.ones(shape)
    if kind in ("bn_bias", "ln_bias", "bn_mean"):
        return torch.zeros(shape)
    if kind == "bn_count":
        return torch.tensor(0, dtype=torch.long)
    raise ValueError(kind)


def _sine_table(d_model, max_shape):
    """PositionEncodingSine.__init__ (utils/position_encoding.py:13-35) incl. the
    `/ d_model // 2` precedence quirk (q2).  Buffer initialisation, same torch ops as upstream."""
    h, w = int(max_shape[0]), int(max_shape[1])
    pe = torch.zeros((d_model, h, w))
    y_pos = torch.ones((h, w)).cumsum(0).float().unsqueeze(0)
    x_pos = torch.ones((h, w)).cumsum(1).float().unsqueeze(0)
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * ((-math.log(10000.0) / d_model) // 2))[:, None, None]
    pe[0::4] = torch.sin(x_pos * div)
    pe[1::4] = torch.cos(x_pos * div)
    pe[2::4] = torch.sin(y_pos * div)
    pe[3::4] = torch.cos(y_pos * div)
    return pe.unsqueeze(0)


def _validate_config(cfg):
    """Same accept/reject behaviour as the reference constructors for unsupported values."""
    bb = cfg["loftr_backbone"]
    if bb["type"] != "ResNetFPN":                      # backbone/__init__.py:13-14
        raise ValueError("LOFTR_BACKBONE.TYPE and RESOLUTION are not correct")
    if list(bb["resolution"]) != [8, 2]:               # backbone/__init__.py:9-12
        raise NotImplementedError("only ResNetFPN resolution [8, 2] is supported")
    if bb["resnetfpn"]["block_type"] != "BasicBlock":
        raise NotImplementedError("HIP path implements BasicBlock only (SURVEY a16)")
    if list(bb["resnetfpn"]["output_layers"]) != [3, 1]:
        raise NotImplementedError("HIP path implements output_layers [3, 1] only")
    ke = cfg["keypoints_encoding"]
    if ke["enable"]:
        if ke["type"] != "mlp_linear":                 # OnePosePlusModel.py:47-50
            raise NotImplementedError
        if ke["norm_method"] != "instancenorm":
            raise NotImplementedError("HIP path implements keypoint-encoder norm 'instancenorm' only")
    for name in ("loftr_coarse", "loftr_fine"):
        t = cfg[name]
        if t["type"] != "LoFTR":                       # transformer.py:183-198
            raise ValueError()
        if t["norm_method"] != "layernorm":
            raise NotImplementedError("HIP path implements norm_method 'layernorm' only")
        if t["attention"] != "linear":
            raise NotImplementedError("HIP path implements linear attention only (FullAttention unused upstream)")
        if t["kernel_fn"] != "elu + 1":                # linear_attention.py:14-18
            raise ValueError()
        if t["rezero"] is not None:
            raise NotImplementedError("rezero is not supported")
        for n in t["layer_names"]:
            if n not in ("self", "cross"):             # transformer.py:117-120
                raise NotImplementedError
        if t["redraw_interval"] is not None:
            assert t["redraw_interval"] % 2 == 0
    ws = cfg["loftr_fine"]["window_size"]
    if int(ws) != ws or ws < 3 or ws % 2 == 0 or ws * ws > 64:
        # the expectation grid divides by (W - 1) (fine_matching.py:87 via kornia create_meshgrid) and the window
        # centre is W // 2: W = 1 gives NaN offsets upstream, an even W has no centre cell
        raise ValueError("loftr_fine.window_size must be an odd integer in [3, 7], got %r" % (ws,))
    cm = cfg["coarse_matching"]
    if cm["type"] != "dual-softmax":                   # coarse_matching.py:63-66
        raise NotImplementedError()
    if cm["feat_norm_method"] != "sqrt_feat_dim":
        raise NotImplementedError("HIP path implements feat_norm_method 'sqrt_feat_dim' only")
    if cfg["fine_matching"]["enable"] and cfg["fine_matching"]["s2d"]["type"] != "heatmap":
        raise NotImplementedError()


GEMM_PRECISIONS = {"fp32": 0, "bf16x3": 3}        # opp_config.gemm_precision (1 / 2 = fp16x2: tuning library only, not offered here)
DEFAULT_GEMM_PRECISION = "bf16x3"


class OnePosePlus_model(nn.Module):
    def __init__(self, config, profiler=None, debug=False):
        super().__init__()
        self.config = config
        self.profiler = profiler or _PassThroughProfiler()
        self.debug = debug
        _validate_config(config)

        for key, shape, kind in param_spec(config):
            _register(self, key, _init_tensor(shape, kind), buffer=kind in ("bn_mean", "bn_var", "bn_count"))
        if config["positional_encoding"]["enable"]:
            pe = _sine_table(config["loftr_coarse"]["d_model"], config["positional_encoding"]["pos_emb_shape"])
            self.dense_pos_encoding = _Node()
            self.dense_pos_encoding.register_buffer("pe", pe, persistent=False)
        else:
            self.dense_pos_encoding = None

        self.loftr_backbone_pretrained = config["loftr_backbone"]["pretrained"]
        if self.loftr_backbone_pretrained is not None:   # OnePosePlusModel.py:78-94
            ckpt = torch.load(self.loftr_backbone_pretrained, "cpu")["state_dict"]
            for k in list(ckpt.keys()):
                if "backbone" in k:
                    ckpt[k[k.find("backbone") + len("backbone") + 1:]] = ckpt[k]
                ckpt.pop(k)
            self.backbone.load_state_dict(ckpt)
            if config["loftr_backbone"]["pretrained_fix"]:
                for p in self.backbone.parameters():
                    p.requires_grad = False
        self.gemm_precision = os.environ.get("OPP_GEMM_PRECISION", DEFAULT_GEMM_PRECISION)
        self.tile_policy = "latency"
        self.encoder_fusion = int(os.environ.get("OPP_ENCODER_FUSION", "2"))
        self.score_two_sweep = int(os.environ.get("OPP_SCORE_PATH", "2"))
        self.fpn_overlap = os.environ.get("OPP_FPN_OVERLAP", "1") != "0"
        self.skip_unused_fine_map = os.environ.get("OPP_SKIP_UNUSED_FINE_MAP", "0") == "1"
        self.fine_patch_max_matches = int(os.environ.get("OPP_FINE_PATCH_MAX", "1000"))
        self.conv_tail = os.environ.get("OPP_CONV_TAIL", "0") == "1"
        self.fine_patch_pixels_per_match = 64      # patches only while M <= fine-map pixels / this (0 = no such rule; tests force the path)
        self._reset_runtime()

    def set_gemm_precision(self, name):
        """Arithmetic of the conv / Linear / score GEMMs (not a reference option; every mode meets the same
        1e-4 parity bar, tests/test_e2e_gpu.py; fp32 in, fp32 accumulate, fp32 out in every mode):
          "bf16x3"      default (env OPP_GEMM_PRECISION overrides): operands carried EXACTLY as hi + mid + lo
                        bf16 triples (24 significant bits, fp32 exponent range), six bf16 MFMAs per product --
                        not narrower than the reference's fp32, 2.6x the fp32 MFMA peak
          "fp32"        exact fp32 MFMA (bit-for-bit an fmaf chain)
        Arithmetics narrower than fp32 (the fp16x2 modes of rounds 1-5) are not part of the product: they live in the tuning
        library behind the C ABI only.
        See include/opp_hip.h `opp_config.gemm_precision`."""
        if name not in GEMM_PRECISIONS:
            raise ValueError("gemm_precision must be one of %s" % (sorted(GEMM_PRECISIONS),))
        if name != self.gemm_precision:
            self.__del__()
            self.gemm_precision = name
            self._reset_runtime()
        return self

    def set_tile_policy(self, name):
        """What the automatic GEMM / conv tile choice minimises (include/opp_hip.h `opp_config.tile_policy`):
        "latency" (default: one forward at a time) or "throughput" (several forwards in flight on separate streams,
        as `serving.MatcherPool` and `bench.py --streams > 1` run them).  Results are bit-identical either way."""
        if name not in ("latency", "throughput"):
            raise ValueError("tile_policy must be 'latency' or 'throughput'")
        if name != getattr(self, "tile_policy", "latency"):
            self.__del__()
            self.tile_policy = name
            self._reset_runtime()
        return self

    def set_encoder_fusion(self, level):
        """With the bf16x3 arithmetic every encoder layer behind its Q/K/V projection is ONE launch (include/opp_hip.h
        `opp_config.encoder_fusion`): 2 (default) = 64-token tiles at the coarse level, 1 = 32-token tiles, 0 / False =
        one launch per Linear.  Bit-identical results."""
        level = int(level)
        if level not in (0, 1, 2):
            raise ValueError("encoder_fusion must be 0, 1 or 2")
        if level != getattr(self, "encoder_fusion", 2):
            self.__del__()
            self.encoder_fusion = level
            self._reset_runtime()
        return self

    def set_fpn_overlap(self, on):
        """True (default): the fused coarse call runs the FPN fine branch of the backbone on a side HIP stream next to the
        coarse level (include/opp_hip.h `opp_config.fpn_overlap`); False: one stream.  Identical results."""
        on = bool(on)
        if on != getattr(self, "fpn_overlap", True):
            self.__del__()
            self.fpn_overlap = on
            self._reset_runtime()
        return self

    def set_skip_unused_fine_map(self, on):
        """Opt-in (default off = the reference's operator list): with `fine_matching.enable = False` the fine feature map is
        computed by the reference's backbone and then dropped (OnePosePlusModel.py:122-124, 169-176).  True: a B = 1 eval
        forward does not launch the FPN branch that produces it (~44 % of the backbone FLOPs); every entry of `data` is
        unchanged.  No effect while fine matching is enabled."""
        self.skip_unused_fine_map = bool(on)
        return self

    def set_fine_patch_max_matches(self, n):
        """Match-driven fine branch of an eval forward with fine matching enabled (include/opp_hip.h `opp_fine_patches`): the
        1/2-resolution half of the FPN fine branch (23 % of the forward's FLOPs at 512 x 512) is evaluated on a 9x9 -> 7x7 -> 5x5 patch
        pyramid around each coarse match (49 MFLOP per match) when there are at most min(n, fine-map pixels / 64) matches, and as the
        dense map otherwise (default 1000 ~ the measured break-even at 512 x 512; 0 = always the dense map inside the fused coarse
        call, as before round 5).  The match count is only known after the coarse level, so the forward either keeps x1 / x2_out and
        decides then, or -- when the PREVIOUS forward of this module had more matches than the limit (consecutive frames of one object
        look alike) -- runs the dense branch inside the coarse call, beside the coarse level, as before.  Bit-identical results either way."""
        self.fine_patch_max_matches = max(0, int(n))
        return self

    def set_conv_tail(self, on):
        """Opt-in (default off): the 196-channel convolutions as a 192-column MFMA body + a 4-column fp32 tail on the vector ALU
        (include/opp_hip.h `opp_set_conv_tail`) instead of 224 / 256 padded MFMA columns.  Same parity bar; not faster."""
        self.conv_tail = bool(on)
        ctx = self._rt.get("ctx")
        if ctx:
            _lib.check(_lib.load().opp_set_conv_tail(ctx, 1 if self.conv_tail else 0), "opp_set_conv_tail")
        return self

    def set_score_two_sweep(self, mode):
        """Coarse-matcher variant under the bf16x3 arithmetic (include/opp_hip.h `opp_config.score_two_sweep`):
          2 (default) score GEMM on operands pre-split once, staged by LDS-DMA (gemm_ss.hip): statistics + score matrix in one
                      sweep, confidences formed in place;
          1           two sweeps of that GEMM (statistics, then confidences written once; no score matrix in memory);
          0 / False   the r02 path (opp_gemm_kernel with the image tokens as pre-split weights).
        Same indices, confidences equal to rounding."""
        mode = int(mode)
        if mode not in (0, 1, 2):
            raise ValueError("score path must be 0, 1 or 2")
        if mode != getattr(self, "score_two_sweep", 2):
            self.__del__()
            self.score_two_sweep = mode
            self._reset_runtime()
        return self

    # ---- runtime state (never pickled) ---------------------------------------------------
    def _reset_runtime(self):
        self.__dict__["_rt"] = {"ctx": None, "packed": None, "dirty": True, "pe": {}, "ws": None, "names": None}

    def __getstate__(self):
        st = self.__dict__.copy()
        st.pop("_rt", None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._reset_runtime()

    def __del__(self):
        rt = self.__dict__.get("_rt")
        if rt and rt.get("ctx"):
            try:
                _lib.load().opp_destroy(rt["ctx"])
            except Exception:
                pass
            rt["ctx"] = None

    def _apply(self, fn, *a, **k):           # .cuda() / .to() / .float() move the parameters
        out = super()._apply(fn, *a, **k)
        if "_rt" in self.__dict__:
            self._rt["dirty"] = True
            self._rt["pe"] = {}
            self._rt["ws"] = None
            self._rt["obj"] = None
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._rt["dirty"] = True
        self._rt["obj"] = None
        return out

    def repack(self):
        """Call after modifying parameters in place (packed weights are cached)."""
        self._rt["dirty"] = True
        self._rt["obj"] = None

    # ---- C-ABI plumbing ------------------------------------------------------------------
    def _c_config(self):
        cfg = self.config
        c = _lib.OppConfig()
        r = cfg["loftr_backbone"]["resnetfpn"]
        c.initial_dim = int(r["initial_dim"])
        for i in range(3):
            c.block_dims[i] = int(r["block_dims"][i])
        ke = cfg["keypoints_encoding"]
        c.kpt_enc_enable = 1 if ke["enable"] else 0
        dims = list(ke["keypoints_encoder"])
        if ke["enable"] and len(dims) != 3:
            raise NotImplementedError("HIP path implements a 3-hidden-layer keypoint encoder")
        for i in range(3):
            c.kpt_enc_dims[i] = int(dims[i]) if i < len(dims) else 0
        c.pos_enc_enable = 1 if cfg["positional_encoding"]["enable"] else 0
        for name, pre in (("loftr_coarse", "coarse"), ("loftr_fine", "fine")):
            t = cfg[name]
            names = list(t["layer_names"]) * t["layer_iter_n"]
            setattr(c, pre + "_d_model", int(t["d_model"]))
            setattr(c, pre + "_nhead", int(t["nhead"]))
            setattr(c, pre + "_n_layers", len(names))
            arr = getattr(c, pre + "_is_cross")
            for i, n in enumerate(names):
                arr[i] = 1 if n == "cross" else 0
        c.fine_window = int(cfg["loftr_fine"]["window_size"])
        cm = cfg["coarse_matching"]
        c.match_thr = float(cm["thr"])
        c.match_border_rm = int(cm["border_rm"])
        c.match_temperature = float(cm["dual_softmax"]["temperature"])
        if self.gemm_precision not in GEMM_PRECISIONS:
            raise ValueError("gemm_precision must be one of %s" % (sorted(GEMM_PRECISIONS),))
        c.gemm_precision = GEMM_PRECISIONS[self.gemm_precision]
        c.tile_policy = 1 if getattr(self, "tile_policy", "latency") == "throughput" else 0
        c.encoder_fusion = int(getattr(self, "encoder_fusion", 2))
        c.score_two_sweep = int(getattr(self, "score_two_sweep", 2))
        c.fpn_overlap = 1 if getattr(self, "fpn_overlap", True) else 0
        return c

    def _ensure_ready(self, device, scope=0):
        """scope 1 (training graph): only the raw backbone weights are packed (no BatchNorm folding) -- the other stages read the parameters themselves"""
        lib = _lib.load()
        rt = self._rt
        if rt.get("scope", 0) != scope:
            rt["dirty"] = True
        if rt["ctx"] is None:
            ctx = ctypes.c_void_p()
            ccfg = self._c_config()
            _lib.check(lib.opp_create(ctypes.byref(ccfg), ctypes.byref(ctx)), "opp_create")
            rt["ctx"] = ctx
            _lib.check(lib.opp_set_conv_tail(ctx, 1 if getattr(self, "conv_tail", False) else 0), "opp_set_conv_tail")
            rt["names"] = [lib.opp_weight_name(ctx, i).decode() for i in range(lib.opp_num_weights(ctx))]
        if rt["dirty"] or rt["packed"] is None or rt["packed"].device != device:
            sd = dict(self.named_parameters())
            sd.update(dict(self.named_buffers()))
            n = len(rt["names"])
            ptrs = (ctypes.c_void_p * n)()
            keep = []
            for i, name in enumerate(rt["names"]):
                t = sd[name].detach()
                if t.device != device:
                    raise RuntimeError("parameter %s is on %s, input on %s" % (name, t.device, device))
                if t.dtype != torch.float32 or not t.is_contiguous():
                    t = t.float().contiguous()
                if t.numel() != lib.opp_weight_numel(rt["ctx"], i):
                    raise RuntimeError("parameter %s has %d elements, expected %d" %
                                       (name, t.numel(), lib.opp_weight_numel(rt["ctx"], i)))
                keep.append(t)
                ptrs[i] = t.data_ptr()
            nbytes = lib.opp_packed_weights_bytes(rt["ctx"])
            blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.opp_set_pack_scope(rt["ctx"], scope), "opp_set_pack_scope")
            _lib.check(lib.opp_pack_weights(rt["ctx"], ptrs, n, blob.data_ptr(), nbytes, stream), "opp_pack_weights")
            rt["scope"] = scope
            rt["packed"] = blob
            rt["dirty"] = False
            rt["keep"] = keep
            rt["ptrs"] = (ptrs, n)
            rt["packed_train"] = None
        return lib, rt["ctx"]

    def _ensure_train_packed(self, lib, ctx, device):
        """Backbone convolutions without the folded BatchNorm (training mode), packed right after the eval packing of
        the same parameter tensors (include/opp_hip.h `opp_pack_train_weights`)."""
        rt = self._rt
        if rt.get("packed_train") is None:
            ptrs, n = rt["ptrs"]
            nbytes = lib.opp_packed_train_weights_bytes(ctx)
            blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.opp_pack_train_weights(ctx, ptrs, n, blob.data_ptr(), nbytes, stream), "opp_pack_train_weights")
            rt["packed_train"] = blob

    def _pe_tokens(self, hc, wc, device):
        """pe[:, :, :hc, :wc] re-laid as NHWC tokens [hc*wc, C] (OnePosePlusModel.py:137-142)."""
        key = (hc, wc)
        if key not in self._rt["pe"]:
            pe = self.dense_pos_encoding.pe
            if hc > pe.shape[2] or wc > pe.shape[3]:
                raise RuntimeError("feature map %dx%d exceeds pos_emb_shape" % (hc, wc))
            self._rt["pe"][key] = pe[0, :, :hc, :wc].permute(1, 2, 0).reshape(hc * wc, -1).contiguous().to(device)
        return self._rt["pe"][key]

    def _object_tokens(self, lib, ctx, kpts, bank_c, device, stream):
        """-> (encoded 3D-point tokens [N, C] of the current object, transformer prefix blob or None).  They depend only on
        (keypoints3d, coarse bank) -- OnePosePlusModel.py:144-156 recomputes them per image -- so
        they are cached per object: the key is the identity AND version counter of both tensors
        (an in-place edit or a different object re-encodes; writes that bypass the version counter
        need `invalidate_object_cache()`).  The tokens carry the event of the stream that produced
        them, so reuse from another stream is ordered behind the encoding.  Set
        `cache_object_tokens = False` to encode per image like the reference."""
        if not getattr(self, "cache_object_tokens", True):
            return None, None
        src = (kpts, bank_c)
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in src)
        hit = self._rt.get("obj")
        if hit is not None and hit[0] == key and hit[1].device == device:
            if hit[3] is not None:      # encoded on another stream: order this stream behind that work
                torch.cuda.current_stream(device).wait_event(hit[3])
            return hit[1], hit[4]
        n = int(kpts.shape[1])
        tok = torch.empty((n, self.config["loftr_coarse"]["d_model"]), dtype=torch.float32, device=device)
        ws = torch.empty(4096, dtype=torch.uint8, device=device)
        _lib.check(lib.opp_encode_points(ctx, kpts.data_ptr(), bank_c.data_ptr(), n, tok.data_ptr(), ws.data_ptr(),
                                         ws.numel(), stream), "opp_encode_points")
        # ... and, in the default arithmetic, the image-independent prefix of the coarse transformer (layer 0 on the 3D stream, its
        # layer-1 projections and KV sums: include/opp_hip.h `opp_object_prefix`; 0 bytes = this configuration has none)
        prefix = None
        nb = lib.opp_object_prefix_bytes(ctx, n)
        if nb:
            prefix = torch.empty(nb, dtype=torch.uint8, device=device)
            pws = torch.empty(max(lib.opp_object_prefix_workspace_bytes(ctx, n), 4096), dtype=torch.uint8, device=device)
            _lib.check(lib.opp_object_prefix(ctx, tok.data_ptr(), n, prefix.data_ptr(), nb, pws.data_ptr(), pws.numel(), stream), "opp_object_prefix")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._rt["obj"] = (key, tok, src, ev, prefix)     # keep the source tensors alive so the key cannot alias
        return tok, prefix

    def invalidate_object_cache(self):
        """Drops the cached 3D-point tokens.  Needed only after writes the key cannot see: the cache is keyed on
        (data_ptr, tensor._version, shape) of `keypoints3d` and the coarse bank, so edits through `tensor.data`,
        a raw pointer (custom kernels, IPC-shared banks) or anything else that does not bump `_version` must be
        followed by this call (or set `cache_object_tokens = False`)."""
        self._rt["obj"] = None

    def _workspace(self, nbytes, device):
        ws = self._rt["ws"]
        if ws is None or ws.numel() < nbytes or ws.device != device:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._rt["ws"] = ws
        return ws

    # ---- forward ---------------------------------------------------------------------------
    @staticmethod
    def _f32(t, name, device):
        if not torch.is_tensor(t) or not t.is_cuda:
            raise RuntimeError("%s must be a CUDA/ROCm tensor: the HIP path has no CPU fallback" % name)
        if t.device != device:
            raise RuntimeError("%s is on %s, expected %s" % (name, t.device, device))
        if t.dtype != torch.float32:
            t = t.float()
        return t.contiguous()

    def forward(self, data):
        """Same contract as the reference forward (OnePosePlusModel.py:96-201); updates `data`.
        B = 1 without `query_image_mask` (what inference.py runs) takes the fused single-call path; B > 1 and / or a
        `query_image_mask` [B, H/8, W/8] run sample by sample through the same kernels (`_forward_batch`)."""
        img = data["query_image"]
        if not torch.is_tensor(img) or not img.is_cuda:
            raise RuntimeError("query_image must be a CUDA/ROCm tensor: the HIP path has no CPU fallback")
        if img.dim() != 4 or img.size(1) != 1 or img.size(0) < 1:
            raise NotImplementedError("HIP path supports query_image of shape [B,1,H,W] (got %s)" % (tuple(img.shape),))
        # (the reference returns None and reports through `data`; returning the same dict as well costs nothing and lets wrappers
        # that copy their inputs -- DistributedDataParallel rebuilds every dict it is given -- hand the results back: `out = ddp(d)`)
        if self.training:
            # training step (lightning_model:54-81): with gradients enabled the forward is built out of autograd nodes whose forward
            # and backward are HIP kernels (train_autograd.py); `conf_matrix` / `expec_f` then carry a grad_fn
            graph = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
            self._forward_train(data, graph=graph)
            return data
        if img.size(0) > 1 or "query_image_mask" in data:
            self._forward_batch(data)
            return data
        self._forward_single(data)
        return data

    def _forward_batch(self, data):
        """B >= 1 with optional `query_image_mask`: every sample runs the B = 1 path with its own mask
        (linear_attention.py:49-53, coarse_matching.py:108-114) and with the keypoint scaling of batch element 0
        (normalize.py:20-21, quirk q4); the per-sample results are concatenated in batch order, which is the order
        `torch.where` gives the reference (coarse_matching.py:170)."""
        img = data["query_image"]
        device = img.device
        B = int(img.size(0))
        H, W = int(img.shape[2]), int(img.shape[3])
        if H % 8 or W % 8:
            raise RuntimeError("image size must be a multiple of 8, got %dx%d" % (H, W))
        hc, wc = H // 8, W // 8
        mask = None
        if "query_image_mask" in data:
            m = data["query_image_mask"]
            if not torch.is_tensor(m) or m.device != device:
                raise RuntimeError("query_image_mask must be a tensor on %s" % (device,))
            if tuple(m.shape) != (B, hc, wc):
                # the reference flattens the mask and broadcasts it against the hc*wc image tokens
                # (OnePosePlusModel.py:158, linear_attention.py:49-53): any other shape fails there as well
                raise RuntimeError("query_image_mask must have the coarse resolution [B=%d, %d, %d], got %s"
                                   % (B, hc, wc, tuple(m.shape)))
            mask = m.flatten(-2).to(torch.float32).contiguous()
        per_sample = ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db", "query_image", "query_image_scale")
        for k in per_sample:
            if k in data and (not torch.is_tensor(data[k]) or data[k].size(0) != B):
                raise RuntimeError("%s must have batch size %d" % (k, B))
        kpts0 = self._f32(data["keypoints3d"][0:1], "keypoints3d", device)
        outs = []
        for b in range(B):
            d = {k: data[k][b:b + 1] for k in per_sample if k in data}
            sample = (mask[b] if mask is not None else None, kpts0 if b > 0 else None)
            self._forward_single(d, use_token_cache=False, sample=sample)
            outs.append(d)
        first = outs[0]
        data.update({"bs": B, "q_hw_i": img.shape[2:], "q_hw_c": first["q_hw_c"], "q_hw_f": first["q_hw_f"]})
        if "W" in first:
            data["W"] = first["W"]
        cat = lambda k: torch.cat([o[k] for o in outs], 0)
        b_ids = torch.cat([torch.full_like(o["b_ids"], b) for b, o in enumerate(outs)], 0)
        data.update({"conf_matrix": cat("conf_matrix"), "b_ids": b_ids, "i_ids": cat("i_ids"), "j_ids": cat("j_ids"),
                     "gt_mask": cat("gt_mask"), "m_bids": b_ids, "mkpts_3d_db": cat("mkpts_3d_db"),
                     "mkpts_query_c": cat("mkpts_query_c"), "mconf": cat("mconf")})
        if not self.config["fine_matching"]["enable"]:
            data["mkpts_query_f"] = data["mkpts_query_c"]
            return
        if b_ids.numel() == 0:
            data.update({"expec_f": torch.empty(0, 3, device=device), "mkpts_query_f": data["mkpts_query_c"]})
            return
        data.update({"expec_f": torch.cat([o["expec_f"].reshape(-1, 3) for o in outs], 0),
                     "mkpts_query_f": torch.cat([o["mkpts_query_f"].reshape(-1, 2) for o in outs], 0)})

    # injectable source of the random draws of the training branch (coarse_matching.py:192-204 uses torch.randint on
    # the matches' device; tests inject recorded draws so that both implementations select the same indices)
    train_randint = staticmethod(torch.randint)
    bn_momentum = 0.1            # torch.nn.BatchNorm2d default, used by every BatchNorm of resnet.py

    def _update_running_stats(self, lib, ctx, stats):
        """running statistics after a train()-mode forward, like torch.nn.BatchNorm2d: stats [n_bn, 512] = batch mean | unbiased variance"""
        with torch.no_grad():
            m = self.bn_momentum
            for i in range(lib.opp_num_bn_layers(ctx)):
                name = lib.opp_bn_layer_name(ctx, i).decode()
                C = lib.opp_bn_layer_channels(ctx, i)
                self.get_buffer(name + ".running_mean").mul_(1 - m).add_(stats[i, :C], alpha=m)
                self.get_buffer(name + ".running_var").mul_(1 - m).add_(stats[i, C:2 * C], alpha=m)
                self.get_buffer(name + ".num_batches_tracked").add_(1)

    def _forward_train(self, data, graph=False):
        """train()-mode forward (PL_OnePosePlus.training_step, lightning_model:54-81 -> OnePosePlusModel.py:96-201):
        ResNet-FPN with BatchNorm BATCH statistics over the whole batch (+ running-statistics update), per-sample
        coarse level, the training branch of get_coarse_match (coarse_matching.py:177-217: sub-sample the predicted
        matches, pad with ground-truth matches from `conf_matrix_gt`, random draws from `self.train_randint`), fine
        level on the padded match list.  Forward only."""
        cfg = self.config
        img = data["query_image"]
        device = img.device
        B, H, W = int(img.size(0)), int(img.shape[2]), int(img.shape[3])
        if H % 8 or W % 8:
            raise RuntimeError("image size must be a multiple of 8, got %dx%d" % (H, W))
        hc, wc, hf, wf = H // 8, W // 8, H // 2, W // 2
        L = hc * wc
        dC, dF = cfg["loftr_coarse"]["d_model"], cfg["loftr_fine"]["d_model"]
        try:
            with torch.cuda.device(device):
                self._forward_train_impl(data, cfg, img, device, B, H, W, hc, wc, hf, wf, L, dC, dF, graph)
        finally:
            # parameters and running statistics move between training steps: on EVERY exit path the eval packing
            # (folded BatchNorm) and the cached 3D-point tokens (keypoint-MLP weights) of this module are stale
            self._rt["dirty"] = True
            self._rt["obj"] = None

    def _forward_train_impl(self, data, cfg, img, device, B, H, W, hc, wc, hf, wf, L, dC, dF, graph=False):
        """graph = True: gradients are wanted -- the same stages as autograd nodes (train_autograd.py), built once, no re-evaluation"""
        if graph:
            from . import train_autograd as TA
        self._rt["dirty"] = True                      # parameters change between training steps: always repack
        self._rt["obj"] = None
        # a frozen pretrained backbone stays in eval mode (OnePosePlusModel.py:109-113): it runs on the BatchNorm-folded packing,
        # which scope 1 (raw backbone weights only) does not produce
        frozen = bool(self.loftr_backbone_pretrained) and bool(cfg["loftr_backbone"]["pretrained_fix"])
        lib, ctx = self._ensure_ready(device, scope=1 if (graph and not frozen) else 0)
        _lib.check(lib.opp_set_status_flag(ctx, None), "opp_set_status_flag")   # no sticky pointer from an eval forward
        stream = torch.cuda.current_stream(device).cuda_stream
        data.update({"bs": B, "q_hw_i": img.shape[2:], "q_hw_c": torch.Size([hc, wc]), "q_hw_f": torch.Size([hf, wf])})
        img_c = self._f32(img, "query_image", device)
        kpts = self._f32(data["keypoints3d"], "keypoints3d", device)
        bank_f = self._f32(data["descriptors3d_db"], "descriptors3d_db", device)
        bank_c = bank_f if "descriptors3d_coarse_db" not in data else \
            self._f32(data["descriptors3d_coarse_db"], "descriptors3d_coarse_db", device)
        N = int(kpts.shape[1])
        if kpts.shape[0] != B or tuple(bank_c.shape) != (B, dC, N):
            raise RuntimeError("bad point-cloud shapes: keypoints3d %s, coarse bank %s" % (tuple(kpts.shape), tuple(bank_c.shape)))
        qscale = self._f32(data["query_image_scale"], "query_image_scale", device) if "query_image_scale" in data else None
        mask = None
        if "query_image_mask" in data:
            if tuple(data["query_image_mask"].shape) != (B, hc, wc):
                raise RuntimeError("query_image_mask must have the coarse resolution [B, %d, %d]" % (hc, wc))
            mask = data["query_image_mask"].flatten(-2).to(device=device, dtype=torch.float32).contiguous()
        pe = self._pe_tokens(hc, wc, device) if self.dense_pos_encoding is not None else None

        # 1. backbone (OnePosePlusModel.py:109-128); a frozen pretrained backbone stays in eval mode (:109-113)
        feat_c = torch.empty((B, L, dC), dtype=torch.float32, device=device)
        feat_f = torch.empty((B, hf * wf, dF), dtype=torch.float32, device=device)
        if graph and not frozen:
            feat_c, feat_f = TA.backbone_node(self, lib, ctx, img_c)
        elif frozen:
            nb = lib.opp_backbone_workspace_bytes(ctx, H, W)
            ws = self._workspace(nb, device)
            for b in range(B):
                _lib.check(lib.opp_backbone(ctx, img_c[b].data_ptr(), H, W, feat_c[b].data_ptr(), feat_f[b].data_ptr(),
                                            ws.data_ptr(), ws.numel(), stream), "opp_backbone")
        else:
            self._ensure_train_packed(lib, ctx, device)
            n_bn = lib.opp_num_bn_layers(ctx)
            stats = torch.zeros((n_bn, 512), dtype=torch.float32, device=device)
            nb = lib.opp_backbone_train_workspace_bytes(ctx, B, H, W)
            ws = self._workspace(nb, device)
            _lib.check(lib.opp_backbone_train(ctx, img_c.data_ptr(), B, H, W, feat_c.data_ptr(), feat_f.data_ptr(),
                                              stats.data_ptr(), ws.data_ptr(), ws.numel(), stream), "opp_backbone_train")
            self._update_running_stats(lib, ctx, stats)

        # 2./3. coarse level per sample (:131-167)
        scale_c = float(H) / float(hc)
        if graph:
            params = dict(self.named_parameters())
            conf, aux = TA.coarse_level_graph(self, lib, ctx, params, feat_c, pe, kpts, bank_c, mask, qscale, hc, wc, scale_c)
            i_all, j_all, c_all, counts = aux["i_all"], aux["j_all"], aux["c_all"], aux["counts"]
        else:
            conf = torch.empty((B, N, L), dtype=torch.float32, device=device)
            i_all = torch.empty((B, N), dtype=torch.int64, device=device)
            j_all = torch.empty((B, N), dtype=torch.int64, device=device)
            c_all = torch.empty((B, N), dtype=torch.float32, device=device)
            mkc = torch.empty((N, 2), dtype=torch.float32, device=device)
            mk3 = torch.empty((N, 3), dtype=torch.float32, device=device)
            counts = torch.zeros((B, 2), dtype=torch.int32, device=device)
            tokens = torch.empty((L + N, dC), dtype=torch.float32, device=device)
            wsb = max(lib.opp_transformer_workspace_bytes(ctx, 0, 1, L, N), lib.opp_coarse_match_workspace_bytes(ctx, N, L), 4096)
            ws = self._workspace(wsb, device)
        try:
            for b in range(B if not graph else 0):
                _lib.check(lib.opp_set_keypoint_extent_ref(ctx, kpts[0].data_ptr() if b > 0 else None, N if b > 0 else 0), "extent_ref")
                _lib.check(lib.opp_set_query_mask(ctx, mask[b].data_ptr() if mask is not None else None), "query_mask")
                _lib.check(lib.opp_coarse_tokens(ctx, feat_c[b].data_ptr(), pe.data_ptr() if pe is not None else None, L,
                                                 kpts[b].data_ptr(), bank_c[b].data_ptr(), N, tokens.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), stream), "opp_coarse_tokens")
                _lib.check(lib.opp_transformer(ctx, 0, tokens.data_ptr(), 1, L, N, ws.data_ptr(), ws.numel(), stream), "opp_transformer")
                _lib.check(lib.opp_coarse_match(ctx, tokens[L:].data_ptr(), tokens.data_ptr(), N, hc, wc, kpts[b].data_ptr(), scale_c,
                                                qscale[b].data_ptr() if qscale is not None else None, conf[b].data_ptr(),
                                                i_all[b].data_ptr(), j_all[b].data_ptr(), c_all[b].data_ptr(), mkc.data_ptr(),
                                                mk3.data_ptr(), counts[b].data_ptr(), ws.data_ptr(), ws.numel(), stream), "opp_coarse_match")
        finally:
            lib.opp_set_keypoint_extent_ref(ctx, None, 0)
            lib.opp_set_query_mask(ctx, None)
        with self.profiler.record_function("LoFTR/coarse-matching/get_coarse_match/argmax-conf"):
            ms = counts[:, 0].tolist()                                                  # one D2H sync for the batch
        b_ids = torch.cat([torch.full((m_,), b, dtype=torch.int64, device=device) for b, m_ in enumerate(ms)])
        i_ids = torch.cat([i_all[b, :m_] for b, m_ in enumerate(ms)])
        j_ids = torch.cat([j_all[b, :m_] for b, m_ in enumerate(ms)])
        mconf = torch.cat([c_all[b, :m_] for b, m_ in enumerate(ms)])

        # training branch of get_coarse_match (coarse_matching.py:177-217)
        tcfg = cfg["coarse_matching"]["train"]
        if tcfg["train_padding"]:
            max_train = int(B * min(N, L) * tcfg["train_coarse_percent"])
            n_pred = int(b_ids.numel())
            pad_min = tcfg["train_pad_num_gt_min"]
            assert pad_min < max_train, "min-num-gt-pad should be less than num-train-matches"
            if n_pred <= max_train - pad_min:
                pred_idx = torch.arange(n_pred, device=device)
            else:
                pred_idx = self.train_randint(n_pred, (max_train - pad_min,), device=device)
            spv_b, spv_i, spv_j = torch.where(data["conf_matrix_gt"])
            assert len(spv_b) != 0
            gt_idx = self.train_randint(len(spv_b), (max(max_train - n_pred, pad_min),), device=device)
            b_ids = torch.cat([b_ids[pred_idx], spv_b[gt_idx]])
            i_ids = torch.cat([i_ids[pred_idx], spv_i[gt_idx]])
            j_ids = torch.cat([j_ids[pred_idx], spv_j[gt_idx]])
            mconf = torch.cat([mconf[pred_idx], torch.zeros(len(gt_idx), device=device)])   # gt paddings: conf 0
        scale_total = scale_c * qscale[b_ids][:, [1, 0]] if qscale is not None else scale_c   # :222-225
        mk_query = torch.stack([j_ids % wc, j_ids // wc], dim=1) * scale_total
        mk_3d = kpts[b_ids, i_ids]
        # `mconf != 0` of the reference (:226-233) selects exactly the predicted matches, which come first in the padded list
        # (a predicted confidence is > thr >= 0, a padded one is 0): their count is known on the host -> slices, no device sync.
        # With a negative threshold a predicted confidence may itself be 0 (fp32 underflow of the dual-softmax product): then the
        # reference's mask is evaluated as written (one device sync, visualisation outputs only)
        n_keep = int(pred_idx.numel()) if tcfg["train_padding"] else int(b_ids.numel())
        if float(cfg["coarse_matching"]["thr"]) >= 0:
            keep = slice(0, n_keep)
        else:
            keep = mconf != 0
            n_keep = int(keep.sum())
        data.update({"conf_matrix": conf, "b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids, "gt_mask": mconf == 0,
                     "m_bids": b_ids[keep], "mkpts_3d_db": mk_3d[keep], "mkpts_query_c": mk_query[keep], "mconf": mconf[keep]})
        if not cfg["fine_matching"]["enable"]:
            data["mkpts_query_f"] = data["mkpts_query_c"]
            return
        # 4./5. fine level on the padded list, sample by sample (fine_preprocess.py:41-55 indexes [b_ids, j_ids])
        data["W"] = cfg["loftr_fine"]["window_size"]
        Mp = int(b_ids.numel())
        assert Mp > 0, "M is always >0, when training, see coarse_matching.py"          # fine_matching.py:47
        scale_f = float(H) / float(hf)
        if graph:
            expec = TA.fine_level_graph(self, params, feat_f, bank_f, b_ids, i_ids, j_ids, B, hf, wf, hc, wc)
            with torch.no_grad():                                                         # build_mkpts, fine_matching.py:96-110
                qs = scale_f * qscale[b_ids][:, [1, 0]] if qscale is not None else scale_f
                mk_f = data["mkpts_query_c"] + (expec[:, :2].detach() * (data["W"] // 2) * qs)[:n_keep]
            data.update({"expec_f": expec, "mkpts_query_f": mk_f})
            return
        expec = torch.empty((Mp, 3), dtype=torch.float32, device=device)
        mk_f = torch.empty((Mp, 2), dtype=torch.float32, device=device)
        mk_query_f32 = mk_query.to(torch.float32).contiguous()
        for b in range(B):
            sel = torch.nonzero(b_ids == b).flatten()
            mb = int(sel.numel())
            if mb == 0:
                continue
            ii, jj, mc = i_ids[sel].contiguous(), j_ids[sel].contiguous(), mk_query_f32[sel].contiguous()
            ex_b = torch.empty((mb, 3), dtype=torch.float32, device=device)
            mf_b = torch.empty((mb, 2), dtype=torch.float32, device=device)
            fws = self._workspace(lib.opp_fine_workspace_bytes(ctx, mb), device)
            _lib.check(lib.opp_fine(ctx, feat_f[b].data_ptr(), hf, wf, bank_f[b].data_ptr(), N, ii.data_ptr(), jj.data_ptr(), mb,
                                    hc, wc, mc.data_ptr(), scale_f, qscale[b].data_ptr() if qscale is not None else None,
                                    1 if cfg["loftr_fine"]["enable"] else 0, ex_b.data_ptr(), mf_b.data_ptr(), fws.data_ptr(),
                                    fws.numel(), stream), "opp_fine")
            expec[sel] = ex_b
            mk_f[sel] = mf_b           # (the shared workspace is reused by the next sample in stream order)
        # mkpts_query_f = mkpts_query_c + (offsets)[:len(mconf)] (fine_matching.py:104-105): the predicted matches
        # come first in the padded list, and they are the ones with mconf != 0
        data.update({"expec_f": expec, "mkpts_query_f": mk_f[:n_keep]})

    def _forward_single(self, data, use_token_cache=True, sample=None):
        """One sample (B = 1) through the fused coarse call + the fine call.  `sample` = (query mask [L] floats or
        None, keypoints of batch element 0 or None) of a `_forward_batch` sample."""
        try:
            return self._forward_single_impl(data, use_token_cache, sample)
        finally:
            # the C context keeps raw pointers to per-call tensors (query mask, extent reference, object prefix, patch buffers):
            # none of them may outlive the call (the tensors are freed / reused by the caching allocator afterwards)
            ctx = self._rt.get("ctx")
            if ctx:
                lib = _lib.load()
                lib.opp_set_query_mask(ctx, None)
                lib.opp_set_keypoint_extent_ref(ctx, None, 0)
                lib.opp_set_object_prefix(ctx, None, 0)
                lib.opp_set_fine_patch_buffers(ctx, None, None)

    def _forward_single_impl(self, data, use_token_cache=True, sample=None):
        img = data["query_image"]
        device = img.device
        cfg = self.config
        with torch.cuda.device(device):
            lib, ctx = self._ensure_ready(device)
            stream = torch.cuda.current_stream(device).cuda_stream
            smask, sref = sample if sample is not None else (None, None)
            _lib.check(lib.opp_set_query_mask(ctx, smask.data_ptr() if smask is not None else None), "opp_set_query_mask")
            _lib.check(lib.opp_set_keypoint_extent_ref(ctx, sref.data_ptr() if sref is not None else None,
                                                       int(sref.shape[1]) if sref is not None else 0), "opp_set_keypoint_extent_ref")
            H, W = int(img.shape[2]), int(img.shape[3])
            if H % 8 or W % 8:
                raise RuntimeError("image size must be a multiple of 8, got %dx%d" % (H, W))
            hc, wc, hf, wf = H // 8, W // 8, H // 2, W // 2
            L = hc * wc
            data.update({"bs": img.size(0), "q_hw_i": img.shape[2:]})
            data.update({"q_hw_c": torch.Size([hc, wc]), "q_hw_f": torch.Size([hf, wf])})

            img_c = self._f32(img, "query_image", device)
            kpts = self._f32(data["keypoints3d"], "keypoints3d", device)
            bank_f = self._f32(data["descriptors3d_db"], "descriptors3d_db", device)
            bank_c = bank_f if "descriptors3d_coarse_db" not in data else \
                self._f32(data["descriptors3d_coarse_db"], "descriptors3d_coarse_db", device)
            N = int(kpts.shape[1])
            dC, dF = cfg["loftr_coarse"]["d_model"], cfg["loftr_fine"]["d_model"]
            if kpts.shape[0] != 1 or kpts.shape[2] != 3 or tuple(bank_c.shape) != (1, dC, N):
                raise RuntimeError("bad point-cloud shapes: keypoints3d %s, coarse bank %s" %
                                   (tuple(kpts.shape), tuple(bank_c.shape)))
            qscale = None
            if "query_image_scale" in data:
                qscale = self._f32(data["query_image_scale"], "query_image_scale", device)
            pe = self._pe_tokens(hc, wc, device) if self.dense_pos_encoding is not None else None

            feat_f = None                                                                # NHWC fine map
            # match-driven fine branch: the coarse call keeps x1 / x2_out and stops the backbone there; once M is known the windows come
            # from per-match patches (M <= fine_patch_max_matches) or from the dense map completed afterwards
            patch_limit = int(getattr(self, "fine_patch_max_matches", 0))
            if int(getattr(self, "fine_patch_pixels_per_match", 64)) > 0:
                patch_limit = min(patch_limit, (hf * wf) // int(self.fine_patch_pixels_per_match))
            self._rt["fine_path"] = "dense map inside the coarse call"
            last_m = self._rt.get("last_matches")
            patch_mode = bool(cfg["fine_matching"]["enable"]) and patch_limit > 0 and (last_m is None or last_m <= patch_limit)
            x1_keep = x2o_keep = None
            if patch_mode:
                x1_keep = torch.empty(lib.opp_fine_patch_buffer_floats(ctx, H, W, 0), dtype=torch.float32, device=device)
                x2o_keep = torch.empty(lib.opp_fine_patch_buffer_floats(ctx, H, W, 1), dtype=torch.float32, device=device)
            elif cfg["fine_matching"]["enable"] or not getattr(self, "skip_unused_fine_map", False):
                feat_f = torch.empty((hf * wf, dF), dtype=torch.float32, device=device)
            _lib.check(lib.opp_set_fine_patch_buffers(ctx, x1_keep.data_ptr() if patch_mode else None,
                                                      x2o_keep.data_ptr() if patch_mode else None), "opp_set_fine_patch_buffers")
            conf = torch.empty((1, N, L), dtype=torch.float32, device=device)
            i_ids = torch.empty(N, dtype=torch.int64, device=device)
            j_ids = torch.empty(N, dtype=torch.int64, device=device)
            mconf = torch.empty(N, dtype=torch.float32, device=device)
            mk_c = torch.empty((N, 2), dtype=torch.float32, device=device)
            mk_3d = torch.empty((N, 3), dtype=torch.float32, device=device)
            count = torch.zeros(1, dtype=torch.int32, device=device)    # M
            ws_bytes = lib.opp_forward_coarse_workspace_bytes(ctx, H, W, N)
            ws = self._workspace(ws_bytes, device)
            scale_c = float(H) / float(hc)                                               # coarse_matching.py:222
            tok3d, prefix = self._object_tokens(lib, ctx, kpts, bank_c, device, stream) if use_token_cache else (None, None)
            _lib.check(lib.opp_set_object_prefix(ctx, prefix.data_ptr() if prefix is not None else None, N), "opp_set_object_prefix")
            _lib.check(lib.opp_forward_coarse(
                ctx, img_c.data_ptr(), H, W, pe.data_ptr() if pe is not None else None, kpts.data_ptr(),
                bank_c.data_ptr(), tok3d.data_ptr() if tok3d is not None else None, N, scale_c, qscale.data_ptr() if qscale is not None else None,
                feat_f.data_ptr() if feat_f is not None else None, conf.data_ptr(), i_ids.data_ptr(), j_ids.data_ptr(), mconf.data_ptr(),
                mk_c.data_ptr(), mk_3d.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(), stream),
                "opp_forward_coarse")
            with self.profiler.record_function("LoFTR/coarse-matching/get_coarse_match/argmax-conf"):
                M = int(count.item())                                                    # the one D2H sync
            self._rt["last_matches"] = M
            b_ids = torch.zeros(M, dtype=torch.int64, device=device)
            data.update({
                "conf_matrix": conf,
                "b_ids": b_ids, "i_ids": i_ids[:M], "j_ids": j_ids[:M],
                "gt_mask": torch.zeros(M, dtype=torch.bool, device=device),
                "m_bids": b_ids,
                "mkpts_3d_db": mk_3d[:M], "mkpts_query_c": mk_c[:M], "mconf": mconf[:M],
            })
            if not cfg["fine_matching"]["enable"]:                                       # OnePosePlusModel.py:169-176
                data.update({"mkpts_3d_db": data["mkpts_3d_db"], "mkpts_query_f": data["mkpts_query_c"]})
                return
            data.update({"W": cfg["loftr_fine"]["window_size"]})                         # fine_preprocess.py:33
            if M == 0:                                                                   # fine_matching.py:46-55
                warnings.warn("No matches found in coarse-level.")
                data.update({"expec_f": torch.empty(0, 3, device=device), "mkpts_query_f": data["mkpts_query_c"]})
                return
            if tuple(bank_f.shape) != (1, dF, N):
                raise RuntimeError("descriptors3d_db must be [1,%d,%d], got %s" % (dF, N, tuple(bank_f.shape)))
            expec = torch.empty((M, 3), dtype=torch.float32, device=device)
            mk_f = torch.empty((M, 2), dtype=torch.float32, device=device)
            scale_f = float(H) / float(hf)                                               # fine_matching.py:41
            if patch_mode:
                self._rt["fine_path"] = "per-match patches" if M <= patch_limit else "dense map completed after the match count is known"
            if patch_mode and M <= patch_limit:
                fws = self._workspace(lib.opp_fine_patches_workspace_bytes(ctx, M), device)
                _lib.check(lib.opp_fine_patches(
                    ctx, x1_keep.data_ptr(), x2o_keep.data_ptr(), H, W, bank_f.data_ptr(), N, i_ids.data_ptr(), j_ids.data_ptr(), M, hc, wc,
                    mk_c.data_ptr(), scale_f, qscale.data_ptr() if qscale is not None else None,
                    1 if cfg["loftr_fine"]["enable"] else 0, expec.data_ptr(), mk_f.data_ptr(), fws.data_ptr(), fws.numel(), stream),
                    "opp_fine_patches")
                data.update({"expec_f": expec, "mkpts_query_f": mk_f})
                self._rt["last"] = (img_c, kpts, bank_f, bank_c, qscale, x1_keep, x2o_keep)
                return
            if patch_mode:                       # many matches: the dense map from the kept x1 / x2_out, then the window gather
                feat_f = torch.empty((hf * wf, dF), dtype=torch.float32, device=device)
                bws = self._workspace(lib.opp_backbone_fine_branch_workspace_bytes(ctx, H, W), device)
                _lib.check(lib.opp_backbone_fine_branch(ctx, x1_keep.data_ptr(), x2o_keep.data_ptr(), H, W, feat_f.data_ptr(), bws.data_ptr(),
                                                        bws.numel(), stream), "opp_backbone_fine_branch")
            fws_bytes = lib.opp_fine_workspace_bytes(ctx, M)
            fws = self._workspace(fws_bytes, device)
            _lib.check(lib.opp_fine(
                ctx, feat_f.data_ptr(), hf, wf, bank_f.data_ptr(), N, i_ids.data_ptr(), j_ids.data_ptr(), M, hc, wc,
                mk_c.data_ptr(), scale_f, qscale.data_ptr() if qscale is not None else None,
                1 if cfg["loftr_fine"]["enable"] else 0, expec.data_ptr(), mk_f.data_ptr(), fws.data_ptr(),
                fws.numel(), stream), "opp_fine")
            data.update({"expec_f": expec, "mkpts_query_f": mk_f})
            # keep every tensor whose pointer was handed to the stream alive until here
            self._rt["last"] = (img_c, kpts, bank_f, bank_c, qscale, feat_f, x1_keep, x2o_keep)
