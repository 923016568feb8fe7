"""Seeded synthetic weights and inputs shared by tests, bench.py and
tests/golden/gen_golden.py.  CPU torch generators only, so the same tensors are produced
in the build container and on the GPU box (same torch build).

Weights are drawn per state-dict key (not by constructing modules) so that the reference
model, the oracle and the HIP module can all be loaded from one `make_state_dict(seed)`
through `load_state_dict(strict=True)`.  Scales follow the reference initialisers:
kaiming-normal fan_out convs (backbone/resnet.py:126-131), xavier-uniform transformer
matrices (loftr_module/transformer.py:128-131), default nn.Linear init for the keypoint MLP
with a zero last bias (utils/position_encoding.py:52); BN statistics/affine are randomised so
that BN folding is exercised.
"""
import math

import torch

from .params import param_spec


def make_state_dict(cfg, seed=0, randomize_norm=True):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    spec = param_spec(cfg)
    last_linear_b = [k for k, _, kind in spec if kind == "linear_b"][-1:] or [None]
    for key, shape, kind in spec:
        if kind == "conv":
            fan_out = shape[0] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        elif kind == "xavier":
            a = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif kind == "linear_w":
            a = 1.0 / math.sqrt(shape[1])
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif kind == "linear_b":
            if key == last_linear_b[0]:
                t = torch.zeros(shape)
            else:
                # fan_in of the matching weight is not in the spec tuple; a fixed small range is fine
                t = (torch.rand(shape, generator=g) * 2 - 1) * 0.2
        elif kind in ("bn_weight", "ln_weight"):
            t = torch.rand(shape, generator=g) + 0.5 if randomize_norm else torch.ones(shape)
        elif kind in ("bn_bias", "ln_bias", "bn_mean"):
            t = torch.randn(shape, generator=g) * 0.1 if randomize_norm else torch.zeros(shape)
        elif kind == "bn_var":
            t = torch.rand(shape, generator=g) + 0.5 if randomize_norm else torch.ones(shape)
        elif kind == "bn_count":
            t = torch.tensor(0, dtype=torch.int64)
        else:
            raise ValueError(kind)
        sd[key] = t
    return sd


def make_inputs(n_points, hw=(512, 512), seed=1, coarse_dim=256, fine_dim=128):
    """SURVEY.md §8d generator order: keypoints3d, descriptors3d_db, descriptors3d_coarse_db,
    query_image; query_image_scale = ones."""
    g = torch.Generator().manual_seed(seed)
    data = {
        "keypoints3d": torch.rand(1, n_points, 3, generator=g) - 0.5,
        "descriptors3d_db": torch.randn(1, fine_dim, n_points, generator=g),
        "descriptors3d_coarse_db": torch.randn(1, coarse_dim, n_points, generator=g),
        "query_image": torch.rand(1, 1, hw[0], hw[1], generator=g),
        "query_image_scale": torch.ones(1, 2),
    }
    return data


def make_planted_matcher_inputs(n_points, n_cells, dim, n_planted, noise=0.1, seed=3):
    """Stage-level coarse-matcher input with thousands of mutual-NN matches:
    f2d ~ N(0,1); f3d[i] = f2d[perm[i]] + noise*N(0,1) for the first n_planted rows."""
    g = torch.Generator().manual_seed(seed)
    f2d = torch.randn(n_cells, dim, generator=g)
    f3d = torch.randn(n_points, dim, generator=g)
    perm = torch.randperm(n_cells, generator=g)[:n_planted]
    f3d[:n_planted] = f2d[perm] + noise * torch.randn(n_planted, dim, generator=g)
    # features entering the matcher are transformer outputs of O(1) magnitude; the planted
    # rows get scaled so that the dual-softmax is sharply peaked
    return f3d[None] * 4.0, f2d[None] * 4.0, perm


def make_fine_ids(n_points, hw_c, m, seed=5):
    """Synthetic coarse matches for the fine stage incl. border cells."""
    g = torch.Generator().manual_seed(seed)
    L = hw_c[0] * hw_c[1]
    i_ids = torch.sort(torch.randperm(n_points, generator=g)[:m]).values
    j_ids = torch.randint(0, L, (m,), generator=g)
    if m >= 4:
        j_ids[0] = 0
        j_ids[1] = hw_c[1] - 1
        j_ids[2] = L - 1
        j_ids[3] = (hw_c[0] - 1) * hw_c[1]
    return i_ids.long(), j_ids.long()
