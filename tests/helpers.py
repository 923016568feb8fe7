"""Shared helpers for oracle-side and HIP-side parity tests."""
import os

import numpy as np
import torch

from onepose_plus_plus_amd.config import default_config
from onepose_plus_plus_amd.synthetic import (make_state_dict, make_inputs,
                                             make_planted_matcher_inputs, make_fine_ids)
from tests.golden.cases import (E2E_CASES, MATCHER_CASES, FINE_CASES, TRANSFORMER_CASES, HIGHCONF_CASES, BATCH_CASES,
                                TRAIN_CASES)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerances (north_star): indices bit-exact, confidences / offsets within 1e-4 fp32.
TOL_CONF = 1e-4
TOL_OFFSET = 1e-4          # expec_f[:, :2]: the fine offsets (normalised window units)
# expec_f[:, 2] is NOT an offset: it is sum_xy sqrt(clamp(var, 1e-10)) (fine_matching.py:92-94), whose derivative
# 1 / (2 sqrt(var)) blows up on sharply peaked heatmaps.  Measured on the high-confidence case (1487 matches): the
# reference's own fp32 result differs from an fp64 evaluation of the same formula by up to 1.9e-4 in that column
# (1.2e-5 in the offsets), so the column gets 5x the offset tolerance.
STD_TOL_FACTOR = 5.0
TOL_PIXEL = 1e-3           # mkpts_query_f in pixels: offsets * 2 * scale(<=2) -> 4e-4 + fp32 ulp at 512


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def e2e_setup(name):
    hw, n, thr, wseed, iseed, fine = E2E_CASES[name]
    cfg = default_config(thr=thr, fine=fine)
    sd = make_state_dict(cfg, wseed)
    data = make_inputs(n, hw, iseed)
    return cfg, sd, data


def batch_setup(name):
    """-> cfg, sd, data of a B > 1 case: per-sample clouds / images, distinct image scales, optional coarse mask"""
    hw, n, thr, wseed, seeds, masked = BATCH_CASES[name]
    cfg = default_config(thr=thr)
    sd = make_state_dict(cfg, wseed)
    parts = [make_inputs(n, hw, sd_) for sd_ in seeds]
    data = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    B = len(seeds)
    data["query_image_scale"] = torch.tensor([[1.0 + 0.25 * b, 1.0 - 0.125 * b] for b in range(B)])
    data["keypoints3d"] = data["keypoints3d"] * torch.tensor([1.0 + 0.5 * b for b in range(B)]).view(B, 1, 1)   # extents differ
    if masked:
        hc, wc = hw[0] // 8, hw[1] // 8
        m = torch.ones(B, hc, wc)
        m[0, :, wc - 3:] = 0            # right padding of sample 0
        m[1, hc - 2:, :] = 0            # bottom padding of sample 1
        data["query_image_mask"] = m
    return cfg, sd, data


def train_setup(name):
    """-> cfg, sd, data of a train()-mode case (incl. `conf_matrix_gt` [B,N,L] int16 with planted ground-truth pairs)"""
    hw, n, thr, wseed, seeds, n_gt, pct, pad_min = TRAIN_CASES[name]
    cfg = default_config(thr=thr)
    cfg["coarse_matching"]["train"] = {"train_padding": True, "train_coarse_percent": pct, "train_pad_num_gt_min": pad_min}
    sd = make_state_dict(cfg, wseed)
    parts = [make_inputs(n, hw, sd_) for sd_ in seeds]
    data = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    B, L = len(seeds), (hw[0] // 8) * (hw[1] // 8)
    data["query_image_scale"] = torch.tensor([[1.0 + 0.25 * (b % 2), 1.0 - 0.125 * (b % 3)] for b in range(B)])
    g = torch.Generator().manual_seed(1000 + wseed)
    gt = torch.zeros(B, n, L, dtype=torch.int16)
    for b in range(B):
        gt[b, torch.randperm(n, generator=g)[:n_gt], torch.randperm(L, generator=g)[:n_gt]] = 1
    data["conf_matrix_gt"] = gt
    return cfg, sd, data


GRAD_TENSORS = ("backbone.conv1.weight", "backbone.bn1.weight", "backbone.layer3.1.bn2.bias", "backbone.layer1_outconv.weight",
                "kpt_3d_pos_encoding.encoder.9.weight", "loftr_coarse.layers.0.q_proj.weight", "loftr_coarse.layers.5.norm2.weight",
                "loftr_fine.layers.1.mlp.2.weight")


def train_loss_weights(conf_shape, expec_shape):
    """fixed scalar L = sum(conf_matrix * wc) + sum(expec_f * we) whose parameter gradients the train fixtures store"""
    g = torch.Generator().manual_seed(5)
    return torch.rand(tuple(conf_shape), generator=g), torch.randn(tuple(expec_shape), generator=g)


def assert_train_grads(named_grads, gold, rel=2e-3, where=""):
    """named_grads: name -> gradient tensor of the same scalar; digest of every parameter + the stored tensors"""
    names = [str(n) for n in gold["grad_names"]]
    dig = gold["grad_digest"]
    assert len(names) > 100
    for n, (gs, gn) in zip(names, dig):
        g = named_grads[n].double().cpu()
        assert abs(float(g.norm()) - gn) <= rel * max(gn, 1e-12) + 1e-9, (where, n, float(g.norm()), gn)
        assert abs(float(g.sum()) - gs) <= rel * gn * (g.numel() ** 0.5) + 1e-9, (where, n, float(g.sum()), gs)
    for n in GRAD_TENSORS:
        g, v = to_np(named_grads[n]), gold["grad/" + n]
        err = np.abs(g - v).max()
        assert err <= rel * np.abs(v).max() + 1e-12, (where, n, float(err), float(np.abs(v).max()))


class RecordedRandint:
    """stands in for torch.randint in the training branch of get_coarse_match: replays the draws recorded when the
    reference produced the fixture (the draws themselves are device / generator specific, coarse_matching.py:192-204)"""

    def __init__(self, draws):
        self.draws = [torch.as_tensor(d) for d in draws]

    def __call__(self, high, size, device=None, **kw):
        d = self.draws.pop(0)
        assert tuple(d.shape) == tuple(size) and (d.numel() == 0 or int(d.max()) < high), (d.shape, size, high)
        return d.to(device) if device is not None else d


def bankbuild_inputs(n_points, dim, seed):
    """Synthetic SfM tracks for the bank builder: every filtered 3D point merges 1-3 COLMAP points, each observed in
    1-12 images -> (kp3d_id_feature {old id: [n_obs, dim] float32}, kp3d_id_score, xyzs [n_points, 3], points_idxs)."""
    rng = np.random.default_rng(seed)
    feat, score, points_idxs, nxt = {}, {}, {}, 7
    for i in range(n_points):
        olds = []
        for _ in range(int(rng.integers(1, 4))):
            n_obs = int(rng.integers(1, 13))
            feat[nxt] = (rng.standard_normal((n_obs, dim)) * np.exp(rng.uniform(-3, 3))).astype(np.float32)
            score[nxt] = rng.random(n_obs).astype(np.float32)
            olds.append(nxt)
            nxt += int(rng.integers(1, 5))
        points_idxs[i] = olds
    return feat, score, rng.standard_normal((n_points, 3)), points_idxs


BANKBUILD_CASES = {"bankbuild_n500_d128": (500, 128, 21), "bankbuild_n300_d256": (300, 256, 22)}


def highconf_geometry(name):
    """The synthetic object of a high-confidence case: a pinhole camera K, a ground-truth object pose [R | t] and
    3D keypoints such that planted point i projects exactly onto the coarse-grid coordinate (8 jx, 8 jy) of its
    planted cell -- so the matches of that case are geometrically consistent and PnP on them must recover the pose.
    -> K [3,3], pose [3,4] (float64 numpy), keypoints3d [1,n,3] float32 tensor, cells [n_planted] int64 tensor."""
    hw, n, n_planted, thr, wseed, iseed = HIGHCONF_CASES[name]
    hc, wc = hw[0] // 8, hw[1] // 8
    g = torch.Generator().manual_seed(1234)
    interior = torch.tensor([y * wc + x for y in range(2, hc) for x in range(2, wc)])
    cells = interior[torch.randperm(len(interior), generator=g)[:n_planted]]
    K = np.array([[600.0, 0.0, hw[1] / 2.0], [0.0, 600.0, hw[0] / 2.0], [0.0, 0.0, 1.0]])
    ax = np.array([0.3, -0.5, 0.2])
    ax = ax / np.linalg.norm(ax)
    th = np.deg2rad(25.0)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)
    t = np.array([0.02, -0.03, 0.6])
    z = (torch.rand(n, generator=g, dtype=torch.float64) * 0.2 + 0.5).numpy()
    uv = np.zeros((n, 2))
    uv[:n_planted, 0] = (cells % wc).numpy() * 8.0
    uv[:n_planted, 1] = (cells // wc).numpy() * 8.0
    uv[n_planted:] = (torch.rand(n - n_planted, 2, generator=g, dtype=torch.float64) * torch.tensor([hw[1] - 1.0, hw[0] - 1.0])).numpy()
    xc = np.stack([(uv[:, 0] - K[0, 2]) / K[0, 0] * z, (uv[:, 1] - K[1, 2]) / K[1, 1] * z, z], 1)
    xo = (xc - t) @ R          # R^T (x_cam - t), row-vector form
    return K, np.concatenate([R, t[:, None]], 1), torch.from_numpy(xo.astype(np.float32))[None], cells


def pose_errors(pose_pred, pose_gt):
    """query_pose_error of the reference (src/utils/metric_utils.py:88-104): rotation error in degrees, translation
    error in cm (poses [3,4] / [4,4], translation in metres)."""
    Rp, tp, Rg, tg = pose_pred[:3, :3], pose_pred[:3, 3], pose_gt[:3, :3], pose_gt[:3, 3]
    cos = np.clip((np.trace(Rp.T @ Rg) - 1.0) / 2.0, -1.0, 1.0)
    return float(np.rad2deg(np.abs(np.arccos(cos)))), float(np.linalg.norm(tp - tg) * 100.0)


def highconf_setup(name):
    """-> cfg, sd, data with the fixture's object (keypoints of `highconf_geometry`, optimised coarse bank stored as
    fp16 values) in place."""
    hw, n, n_planted, thr, wseed, iseed = HIGHCONF_CASES[name]
    cfg = default_config(thr=thr)
    sd = make_state_dict(cfg, wseed)
    data = make_inputs(n, hw, iseed)
    gold = load_golden(name)
    data["keypoints3d"] = torch.from_numpy(gold["keypoints3d"])
    data["descriptors3d_coarse_db"] = torch.from_numpy(gold["bank_c_f16"]).float()
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


def assert_batched_outputs(got, gold, tol_conf=TOL_CONF, tol_off=TOL_OFFSET, tol_px=TOL_PIXEL, where=""):
    """B > 1: same checks as assert_match_outputs with a [B,N,L] confidence matrix"""
    conf = got["conf_matrix"].detach().float().cpu()
    dig = {"conf_rowsum": conf.sum(2).numpy(), "conf_colsum": conf.sum(1).numpy(),
           "conf_rowmax": conf.max(2).values.numpy(), "conf_colmax": conf.max(1).values.numpy(), "conf_matrix": conf.numpy()}
    for k, v in dig.items():
        if k in gold:
            err = np.abs(v - gold[k]).max()
            assert err <= tol_conf * (50 if k.endswith("sum") else 1), (where, k, float(err))
    rest = {k: v for k, v in gold.items() if not k.startswith("conf_")}
    assert_match_outputs({k: v for k, v in got.items() if k != "conf_matrix"}, rest, tol_conf, tol_off, tol_px, where)
    meta = gold["meta"]
    assert got["bs"] == meta[0] and tuple(got["q_hw_i"]) == tuple(meta[1:3]) and tuple(got["q_hw_c"]) == tuple(meta[3:5])


def assert_train_outputs(got, state, gold, tol_conf=TOL_CONF, tol_off=TOL_OFFSET, tol_px=TOL_PIXEL, tol_bn=1e-4, where=""):
    """train()-mode forward: matches / confidences / fine offsets like the batched case (M' padded rows, M predicted
    ones) + the BatchNorm running statistics after the forward (`state`: name -> tensor)."""
    assert_batched_outputs(got, {k: v for k, v in gold.items() if not k.startswith(("bn/", "randint_", "n_randint", "grad"))},
                           tol_conf, tol_off, tol_px, where)
    assert len(gold["b_ids"]) > len(gold["mconf"]) > 0          # ground-truth padding present
    n = 0
    for k, v in gold.items():
        if k.startswith("bn/"):
            t = to_np(state[k[3:]])
            if k.endswith("num_batches_tracked"):
                assert int(t) == int(v), (where, k)
            else:
                err = np.abs(t - v).max()
                assert err <= tol_bn * max(1.0, float(np.abs(v).max())), (where, k, float(err))
            n += 1
    assert n == 3 * 17          # 17 BatchNorm layers: stem, 2 + 2 + 3 + 2 + 3 + 2 in the blocks, 2 in the FPN heads


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
            if g.size and k == "expec_f":
                err = np.abs(g[:, :2] - gold[k][:, :2]).max()
                assert err <= tol, (where, "expec_f offsets", float(err), tol)
                err = np.abs(g[:, 2] - gold[k][:, 2]).max()
                assert err <= STD_TOL_FACTOR * tol, (where, "expec_f std", float(err), STD_TOL_FACTOR * tol)
            elif g.size:
                err = np.abs(g - gold[k]).max()
                assert err <= tol, (where, k, float(err), tol)
    if "conf_matrix" in got and any(k.startswith("conf_") for k in gold):
        dig = conf_digest_t(got["conf_matrix"])
        for k, v in dig.items():
            if k in gold:
                err = np.abs(v - gold[k]).max()
                lim = tol_conf * (50 if k.endswith("sum") else 1)
                assert err <= lim, (where, k, float(err))


def loss_inputs(name):
    """-> (data dict of CPU tensors for Loss / fine_supervision, experiment-config stub).  conf_matrix carries entries
    that sit on and beyond the clamp bounds (exact 0 / 1, 1e-8, 1 - 1e-8) on both positive and negative cells."""
    from tests.golden.cases import LOSS_CASES
    B, N, hw_c, n_pos, M, seed, masked, no_inside = LOSS_CASES[name]
    L = hw_c[0] * hw_c[1]
    g = torch.Generator().manual_seed(seed)
    conf = torch.rand(B, N, L, generator=g)
    gt = torch.zeros(B, N, L, dtype=torch.int16)
    for b in range(B):
        if n_pos:
            gt[b, torch.randperm(N, generator=g)[:n_pos], torch.randperm(L, generator=g)[:n_pos]] = 1
    flat = conf.view(-1)
    special = torch.tensor([0.0, 1.0, 1e-8, 1.0 - 1e-8, 1e-6, 1.0 - 1e-6, 5e-7])
    idx_neg = torch.nonzero(gt.view(-1) == 0)[:, 0]
    flat[idx_neg[torch.randperm(len(idx_neg), generator=g)[:len(special)]]] = special
    idx_pos = torch.nonzero(gt.view(-1) == 1)[:, 0]
    if len(idx_pos) >= len(special):
        flat[idx_pos[torch.randperm(len(idx_pos), generator=g)[:len(special)]]] = special
    b_ids = torch.randint(0, B, (M,), generator=g).sort().values
    i_ids = torch.randint(0, N, (M,), generator=g)
    j_ids = torch.randint(0, L, (M,), generator=g)
    scale = torch.tensor([[1.0 + 0.25 * (b % 2), 1.0 - 0.125 * (b % 3)] for b in range(B)])
    cell = torch.stack([j_ids % hw_c[1], j_ids // hw_c[1]], 1) * (8.0 * scale[b_ids][:, [1, 0]])
    loc = torch.full((B, N, L, 2), -50.0)
    # ground-truth 2D locations: most inside the 5x5 window of the matched cell, some far outside
    off = (torch.rand(M, 2, generator=g) - 0.5) * 8.0 * scale[b_ids][:, [1, 0]]
    far = torch.rand(M, generator=g) < 0.25
    off[far] = off[far] * 6.0 + 20.0
    if no_inside:
        off = off.abs() + 30.0
    loc[b_ids, i_ids, j_ids] = cell + off
    expec = torch.cat([torch.rand(M, 2, generator=g) * 2 - 1, torch.rand(M, 1, generator=g) * 1.7 + 0.08], 1)
    expec[0, 2] = 1e-12                                # below the std clamp (losses.py:79)
    data = {"conf_matrix": conf, "conf_matrix_gt": gt, "b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids,
            "q_hw_c": torch.Size(hw_c), "query_image_scale": scale, "fine_location_matrix_gt": loc, "expec_f": expec}
    if masked:
        data["mask0"] = (torch.rand(B, 1, N, generator=g) > 0.2).float()          # flatten(-2) -> [B, N]
        data["mask1"] = (torch.rand(B, hw_c[0], hw_c[1], generator=g) > 0.2).float()
    hparams = {"OnePosePlus": {"loftr_backbone": {"resolution": [8, 2]}, "loftr_fine": {"window_size": 5}}}
    return data, hparams
