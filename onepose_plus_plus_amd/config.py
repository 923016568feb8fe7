"""Hot-path configuration contract.

`default_config()` restates the `model.OnePosePlus` block of the reference's
configs/experiment/inference_onepose.yaml:26-109 (with `loftr_backbone.pretrained: None`),
the mapping `OnePosePlus_model.__init__` indexes with [] (OnePosePlusModel.py:26-94).
"""
import copy

_DEFAULT = {
    "loftr_backbone": {
        "type": "ResNetFPN",
        "resolution": [8, 2],
        "resnetfpn": {
            "block_type": "BasicBlock",
            "initial_dim": 128,
            "block_dims": [128, 196, 256],
            "output_layers": [3, 1],
        },
        "pretrained": None,
        "pretrained_fix": False,
    },
    "interpol_type": "bilinear",
    "keypoints_encoding": {
        "enable": True,
        "type": "mlp_linear",
        "descriptor_dim": 256,
        "keypoints_encoder": [32, 64, 128],
        "norm_method": "instancenorm",
    },
    "positional_encoding": {"enable": True, "pos_emb_shape": [256, 256]},
    "loftr_coarse": {
        "type": "LoFTR",
        "d_model": 256,
        "d_ffm": 128,
        "nhead": 8,
        "layer_names": ["self", "cross"],
        "layer_iter_n": 3,
        "dropout": 0.0,
        "attention": "linear",
        "norm_method": "layernorm",
        "kernel_fn": "elu + 1",
        "d_kernel": 16,
        "redraw_interval": 2,
        "rezero": None,
        "final_proj": False,
    },
    "coarse_matching": {
        "type": "dual-softmax",
        "thr": 0.1,
        "feat_norm_method": "sqrt_feat_dim",
        "border_rm": 2,
        "dual_softmax": {"temperature": 0.08},
        "train": {
            "train_padding": True,
            "train_coarse_percent": 0.3,
            "train_pad_num_gt_min": 200,
        },
    },
    "loftr_fine": {
        "enable": True,
        "window_size": 5,
        "coarse_layer_norm": False,
        "type": "LoFTR",
        "d_model": 128,
        "nhead": 8,
        "layer_names": ["self", "cross"],
        "layer_iter_n": 1,
        "dropout": 0.0,
        "attention": "linear",
        "norm_method": "layernorm",
        "kernel_fn": "elu + 1",
        "d_kernel": 16,
        "redraw_interval": 2,
        "rezero": None,
        "final_proj": False,
    },
    "fine_matching": {"enable": True, "type": "s2d", "s2d": {"type": "heatmap"}},
}


def default_config(thr=None, fine=True):
    cfg = copy.deepcopy(_DEFAULT)
    if thr is not None:
        cfg["coarse_matching"]["thr"] = thr
    cfg["fine_matching"]["enable"] = bool(fine)
    return cfg
