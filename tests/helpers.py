"""Shared helpers for oracle-side and HIP-side parity tests."""
import os

import numpy as np
import torch

from onepose_plus_plus_amd.config import default_config
from onepose_plus_plus_amd.synthetic import (make_state_dict, make_inputs,
                                             make_planted_matcher_inputs, make_fine_ids)
from tests.golden.cases import E2E_CASES, MATCHER_CASES, FINE_CASES, TRANSFORMER_CASES, HIGHCONF_CASES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerances (north_star): indices bit-exact, confidences / offsets within 1e-4 fp32.
TOL_CONF = 1e-4
TOL_OFFSET = 1e-4          # expec_f (normalised window units)
TOL_PIXEL = 1e-3           # mkpts_query_f in pixels: offsets * 2 * scale(<=2) -> 4e-4 + fp32 ulp at 512


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def e2e_setup(name):
    hw, n, thr, wseed, iseed, fine = E2E_CASES[name]
    cfg = default_config(thr=thr, fine=fine)
    sd = make_state_dict(cfg, wseed)
    data = make_inputs(n, hw, iseed)
    return cfg, sd, data


def highconf_setup(name):
    """-> cfg, sd, data with the optimised coarse bank of the fixture (stored fp16 values) in place."""
    hw, n, n_planted, thr, wseed, iseed = HIGHCONF_CASES[name]
    cfg = default_config(thr=thr)
    sd = make_state_dict(cfg, wseed)
    data = make_inputs(n, hw, iseed)
    data["descriptors3d_coarse_db"] = torch.from_numpy(load_golden(name)["bank_c_f16"]).float()
    return cfg, sd, data


def transformer_inputs(L, n, seed):
    """Seeded O(1) token streams for the coarse transformer stage: tokens2d [1,L,256], bank [1,256,N] (the layout
    LocalFeatureTransformer.forward takes, transformer.py:133-145)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, L, 256, generator=g), torch.randn(1, 256, n, generator=g)


def transformer_digest(f3, f2):
    """f3 [N,C], f2 [L,C] -> every 16th row + reductions over all rows."""
    f3, f2 = f3.float().cpu(), f2.float().cpu()
    return {"f3_rows": f3[::16].contiguous().numpy(), "f2_rows": f2[::16].contiguous().numpy(),
            "f3_rowsum": f3.sum(1).numpy(), "f2_rowsum": f2.sum(1).numpy(),
            "f3_rowabs": f3.abs().sum(1).numpy(), "f2_rowabs": f2.abs().sum(1).numpy(),
            "f3_colsum": f3.sum(0).numpy(), "f2_colsum": f2.sum(0).numpy()}


def matcher_setup(name):
    n, hw_c, n_planted, noise, seed, thr = MATCHER_CASES[name]
    cfg = default_config(thr=thr)
    L = hw_c[0] * hw_c[1]
    f3d, f2d, _ = make_planted_matcher_inputs(n, L, 256, n_planted, noise, seed)
    g = torch.Generator().manual_seed(seed + 100)
    data = {"q_hw_i": torch.Size([hw_c[0] * 8, hw_c[1] * 8]), "q_hw_c": torch.Size(hw_c),
            "keypoints3d": torch.rand(1, n, 3, generator=g) - 0.5,
            "query_image_scale": torch.tensor([[1.25, 0.75]])}
    return cfg, f3d, f2d, data


def fine_setup(name):
    n, hw_i, m, seed = FINE_CASES[name]
    cfg = default_config()
    sd = make_state_dict(cfg, 0)
    hw_c = (hw_i[0] // 8, hw_i[1] // 8)
    hw_f = (hw_i[0] // 2, hw_i[1] // 2)
    g = torch.Generator().manual_seed(seed + 100)
    feat_f = torch.randn(1, 128, hw_f[0], hw_f[1], generator=g)
    bank_f = torch.randn(1, 128, n, generator=g)
    kpts = torch.rand(1, n, 3, generator=g) - 0.5
    i_ids, j_ids = make_fine_ids(n, hw_c, m, seed)
    scale = torch.tensor([[1.25, 0.75]])
    b_ids = torch.zeros(m, dtype=torch.long)
    mk_c = torch.stack([j_ids % hw_c[1], j_ids // hw_c[1]], 1) * (8.0 * scale[b_ids][:, [1, 0]])
    data = {"q_hw_i": torch.Size(hw_i), "q_hw_c": torch.Size(hw_c), "q_hw_f": torch.Size(hw_f),
            "b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids, "mkpts_query_c": mk_c,
            "mkpts_3d_db": kpts[b_ids, i_ids], "query_image_scale": scale}
    return cfg, sd, feat_f, bank_f, data


def conf_digest_t(conf):
    c = conf[0].float().cpu()
    out = {"conf_rowsum": c.sum(1).numpy(), "conf_colsum": c.sum(0).numpy(),
           "conf_rowmax": c.max(1).values.numpy(), "conf_colmax": c.max(0).values.numpy()}
    if c.numel() <= 200000:
        out["conf_matrix"] = c.numpy()
    else:
        out["conf_sample"] = c[::37, ::41].contiguous().numpy()
    return out


def assert_transformer_digest(got, gold, rel, where=""):
    """every stored row / reduction within `rel` of max(1, |gold|_max) (reductions: scaled by the row length)"""
    for k, v in gold.items():
        g = got[k]
        assert g.shape == v.shape, (where, k, g.shape, v.shape)
        lim = rel * max(1.0, float(np.abs(v).max())) * (16.0 if k.endswith(("sum", "abs")) else 1.0)
        err = float(np.abs(g - v).max())
        assert err <= lim, (where, k, err, lim)


def conf_relative_error(got, gold):
    """max |got - gold| / gold over the matched confidences (gold > 0)"""
    g, v = to_np(got).astype(np.float64), to_np(gold).astype(np.float64)
    return float((np.abs(g - v) / np.maximum(v, 1e-30)).max()) if v.size else 0.0


def to_np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def assert_match_outputs(got, gold, tol_conf=TOL_CONF, tol_off=TOL_OFFSET, tol_px=TOL_PIXEL, where=""):
    """got: data dict (tensors) ; gold: dict of numpy arrays from a golden file / oracle."""
    for k in ["b_ids", "i_ids", "j_ids", "m_bids", "gt_mask"]:
        if k in gold:
            g = to_np(got[k])
            assert g.dtype == gold[k].dtype, (where, k, g.dtype, gold[k].dtype)
            assert np.array_equal(g, gold[k]), (where, k, g[:10], gold[k][:10], g.shape, gold[k].shape)
    for k, tol in [("mconf", tol_conf), ("mkpts_query_c", 1e-4), ("mkpts_3d_db", 0.0),
                   ("expec_f", tol_off), ("mkpts_query_f", tol_px)]:
        if k in gold:
            g = to_np(got[k])
            assert g.shape == gold[k].shape, (where, k, g.shape, gold[k].shape)
            assert g.dtype == np.float32, (where, k, g.dtype)
            if g.size:
                err = np.abs(g - gold[k]).max()
                assert err <= tol, (where, k, float(err), tol)
    if "conf_matrix" in got and any(k.startswith("conf_") for k in gold):
        dig = conf_digest_t(got["conf_matrix"])
        for k, v in dig.items():
            if k in gold:
                err = np.abs(v - gold[k]).max()
                lim = tol_conf * (50 if k.endswith("sum") else 1)
                assert err <= lim, (where, k, float(err))
