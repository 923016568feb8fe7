"""ctypes binding of libopp_hip.so (include/opp_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails the error
is raised to the caller (the product path never routes through PyTorch ops or the oracle).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPP_HIP_LIB: explicit path of another build of the same library (tools/ use the -DOPP_TUNING one)
LIB_PATH = os.environ.get("OPP_HIP_LIB") or os.path.join(_HERE, "libopp_hip.so")
OPP_MAX_LAYERS = 16


class OppConfig(Structure):
    _fields_ = [
        ("initial_dim", c_int),
        ("block_dims", c_int * 3),
        ("kpt_enc_enable", c_int),
        ("kpt_enc_dims", c_int * 3),
        ("pos_enc_enable", c_int),
        ("coarse_d_model", c_int),
        ("coarse_nhead", c_int),
        ("coarse_n_layers", c_int),
        ("coarse_is_cross", c_int * OPP_MAX_LAYERS),
        ("fine_d_model", c_int),
        ("fine_nhead", c_int),
        ("fine_n_layers", c_int),
        ("fine_is_cross", c_int * OPP_MAX_LAYERS),
        ("fine_window", c_int),
        ("match_thr", c_float),
        ("match_border_rm", c_int),
        ("match_temperature", c_float),
        ("gemm_precision", c_int),
        ("tile_policy", c_int),
        ("encoder_fusion", c_int),
        ("score_two_sweep", c_int),
        ("fpn_overlap", c_int),
    ]


# name -> (restype, argtypes).  Must list every symbol declared in include/opp_hip.h
# (tests/test_cabi.py cross-checks this table against the header).
SIGNATURES = {
    "opp_last_error": (c_char_p, []),
    "opp_version": (c_int, []),
    "opp_source_hash": (c_char_p, []),
    "opp_supports_precision": (c_int, [c_int]),
    "opp_profile_event_overhead": (c_int, [c_int, POINTER(ctypes.c_double), c_void_p]),
    "opp_profile_empty_kernel": (c_int, [c_int, POINTER(ctypes.c_double), c_void_p]),
    "opp_profile_event_calibration": (c_int, [c_int, ctypes.c_double, POINTER(ctypes.c_double), POINTER(ctypes.c_double), c_void_p]),
    "opp_create": (c_int, [POINTER(OppConfig), POINTER(c_void_p)]),
    "opp_destroy": (None, [c_void_p]),
    "opp_set_status_flag": (c_int, [c_void_p, c_void_p]),
    "opp_set_query_mask": (c_int, [c_void_p, c_void_p]),
    "opp_set_conv_tail": (c_int, [c_void_p, c_int]),
    "opp_set_keypoint_extent_ref": (c_int, [c_void_p, c_void_p, c_int]),
    "opp_num_weights": (c_int, [c_void_p]),
    "opp_weight_name": (c_char_p, [c_void_p, c_int]),
    "opp_weight_numel": (c_longlong, [c_void_p, c_int]),
    "opp_num_bn_layers": (c_int, [c_void_p]),
    "opp_bn_layer_name": (c_char_p, [c_void_p, c_int]),
    "opp_bn_layer_channels": (c_int, [c_void_p, c_int]),
    "opp_packed_train_weights_bytes": (c_size_t, [c_void_p]),
    "opp_pack_train_weights": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_void_p, c_size_t, c_void_p]),
    "opp_backbone_train_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "opp_backbone_train": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_packed_weights_bytes": (c_size_t, [c_void_p]),
    "opp_set_pack_scope": (c_int, [c_void_p, c_int]),
    "opp_pack_weights": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_void_p, c_size_t, c_void_p]),
    "opp_backbone_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "opp_backbone": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_coarse_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "opp_transformer_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int, c_int]),
    "opp_transformer": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "opp_coarse_match_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "opp_coarse_match": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
    "opp_forward_coarse_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "opp_encode_points": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_object_prefix_bytes": (c_size_t, [c_void_p, c_int]),
    "opp_object_prefix_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "opp_object_prefix": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "opp_set_object_prefix": (c_int, [c_void_p, c_void_p, c_int]),
    "opp_forward_coarse": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_fine_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "opp_fine_head_train_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "opp_fine_head_train_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "opp_build_assignmatrix": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "opp_set_fine_patch_buffers": (c_int, [c_void_p, c_void_p, c_void_p]),
    "opp_fine_patch_buffer_floats": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "opp_fine_patches_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "opp_fine_patches": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_backbone_fine_branch_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "opp_backbone_fine_branch": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_fine": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                         c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_conv2d_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "opp_conv2d_nhwc_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                      c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "opp_conv_packed_k": (c_int, [c_int, c_int]),
    "opp_dual_softmax_backward_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "opp_dual_softmax_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_focal_loss_workspace_bytes": (c_size_t, [c_size_t]),
    "opp_focal_loss_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_focal_loss_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "opp_linear_backward_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "opp_linear_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "opp_focal_loss_forward_ex": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_size_t, c_float, c_float,
                                          c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_focal_loss_backward_ex": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_size_t, c_float, c_float,
                                           c_void_p, c_void_p, c_void_p]),
    "opp_linear_attention_train_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "opp_linear_attention_train_forward": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "opp_linear_attention_train_backward": (c_int, [c_void_p] * 8 + [c_int] * 5 + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "opp_linear_attention_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "opp_linear_attention": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_pack_conv_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "opp_gemm_tile_for": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "opp_linear": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "opp_linear_layernorm": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p]),
    "opp_pack_h2": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "opp_pack_b3": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_layer_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "opp_image_ingest_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "opp_segmented_mean": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "opp_debug_timestamps": (c_int, [c_void_p]),
    "opp_pnp_workspace_bytes": (c_size_t, [c_int]),
    "opp_pnp_ransac": (c_int, [c_void_p, c_void_p, c_int, POINTER(ctypes.c_double), ctypes.c_double, ctypes.c_double,
                               c_int, ctypes.c_uint, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_size_t, c_void_p]),
    "opp_backbone_tape_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "opp_backbone_train_tape_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "opp_backbone_train_tape": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                        c_void_p, c_size_t, c_void_p]),
    "opp_backbone_backward_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "opp_backbone_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, POINTER(c_void_p), c_int, c_void_p,
                                      c_void_p, POINTER(c_void_p), c_void_p, c_size_t, c_void_p]),
    "opp_conv2d_backward_workspace_bytes": (c_size_t, [c_int] * 8),
    "opp_conv2d_backward_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_void_p, c_size_t, c_void_p]),
    "opp_batchnorm_backward_workspace_bytes": (c_size_t, [c_int, c_int]),
    "opp_batchnorm_backward_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_upsample2x_backward_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "opp_layer_norm_train_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "opp_layer_norm_train_backward_workspace_bytes": (c_size_t, [c_int, c_int]),
    "opp_layer_norm_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_size_t, c_void_p]),
    "opp_dual_softmax_forward_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "opp_dual_softmax_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "opp_fine_window_gather": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "opp_fine_window_gather_backward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                c_void_p, c_void_p]),
    "opp_profile_start": (c_int, [c_int, c_int, c_int]),
    "opp_profile_stop": (c_int, [POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int)]),
}

_lib = None


class OppError(RuntimeError):
    pass


def load():
    """Loads libopp_hip.so; raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OppError(
            "libopp_hip.so not found at %s -- build it with `python -m onepose_plus_plus_amd.build` "
            "(there is no CPU/PyTorch fallback for the HIP path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # a binary built from other sources than the ones next to it (the GPU box runs the prebuilt .so that travels with the
    # snapshot) is refused: the library carries the sha256 of its sources (csrc/version.hip, build.py `source_hash`)
    from .build import source_hash
    want = source_hash()
    got = lib.opp_source_hash()
    got = got.decode() if got else "unknown"
    if want is not None and got != want and os.environ.get("OPP_ALLOW_STALE_LIB", "0") != "1":
        raise OppError("%s was built from other sources than the ones in %s (library %s..., sources %s...): rebuild it with "
                       "`python -m onepose_plus_plus_amd.build`" % (LIB_PATH, os.path.join(_HERE, "csrc"), got[:12], want[:12]))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().opp_last_error()
        raise OppError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
