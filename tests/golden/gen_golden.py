"""Generate the committed golden vectors by running the UPSTREAM reference
(/root/reference, imported read-only through oracle/refload.py stubs) on seeded synthetic
inputs.  Runs only in the build container (the GPU box has no /root/reference):

    python tests/golden/gen_golden.py

Outputs tests/golden/*.npz (reference outputs only; inputs are regenerated from seeds).
Large tensors (conf_matrix at full size) are stored as strided samples + reductions.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.refload import load_reference_model_class, load_reference_modules  # noqa: E402
from onepose_plus_plus_amd.config import default_config  # noqa: E402
from onepose_plus_plus_amd.synthetic import (make_state_dict, make_inputs,  # noqa: E402
                                             make_planted_matcher_inputs, make_fine_ids)
from tests.golden.cases import E2E_CASES, MATCHER_CASES, FINE_CASES  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def conf_digest(conf):
    """Compact description of a [1,N,L] confidence matrix."""
    c = conf[0]
    out = {
        "conf_rowsum": c.sum(1).numpy(), "conf_colsum": c.sum(0).numpy(),
        "conf_rowmax": c.max(1).values.numpy(), "conf_colmax": c.max(0).values.numpy(),
    }
    if c.numel() <= 200000:
        out["conf_matrix"] = c.numpy()
    else:
        out["conf_sample"] = c[::37, ::41].contiguous().numpy()
    return out


def gen_e2e():
    cls = load_reference_model_class()
    for name, (hw, n, thr, wseed, iseed, fine) in E2E_CASES.items():
        cfg = default_config(thr=thr, fine=fine)
        model = cls(cfg).eval()
        model.load_state_dict(make_state_dict(cfg, wseed), strict=True)
        data = make_inputs(n, hw, iseed)
        with torch.no_grad():
            model(data)
        out = {}
        for k in ["b_ids", "i_ids", "j_ids", "gt_mask", "m_bids", "mkpts_3d_db",
                  "mkpts_query_c", "mconf", "expec_f", "mkpts_query_f"]:
            if k in data:
                out[k] = data[k].numpy()
        out.update(conf_digest(data["conf_matrix"]))
        out["meta"] = np.array([data["bs"], *data["q_hw_i"], *data["q_hw_c"], *data["q_hw_f"],
                                data.get("W", -1)], dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "M =", len(data["mconf"]))


def gen_stage_features():
    """Backbone / transformer intermediates of one small case, for stage-level parity."""
    cls = load_reference_model_class()
    hw, n, wseed, iseed = (128, 128), 300, 0, 1
    cfg = default_config(thr=0.0)
    model = cls(cfg).eval()
    model.load_state_dict(make_state_dict(cfg, wseed), strict=True)
    data = make_inputs(n, hw, iseed)
    with torch.no_grad():
        feat_c, feat_f = model.backbone(data["query_image"])
        tokens2d = model.dense_pos_encoding(feat_c).flatten(2).transpose(1, 2)
        from src.models.OnePosePlus.utils.normalize import normalize_3d_keypoints
        bank = model.kpt_3d_pos_encoding(normalize_3d_keypoints(data["keypoints3d"]),
                                         data["descriptors3d_coarse_db"])
        f3, f2 = model.loftr_coarse(bank, tokens2d)
    np.savez_compressed(os.path.join(HERE, "stages_128x128_n300.npz"),
                        feat_c=feat_c.numpy(), feat_f=feat_f[:, :, ::4, ::4].contiguous().numpy(),
                        feat_f_sum=feat_f.sum((0, 1)).numpy(),
                        tokens2d=tokens2d.numpy(), bank_enc=bank.numpy(),
                        f3=f3.numpy(), f2=f2.numpy())
    print("stages done")


def gen_matcher():
    mods = load_reference_modules()
    for name, (n, hw_c, n_planted, noise, seed, thr) in MATCHER_CASES.items():
        cfg = default_config(thr=thr)
        cm = mods["CoarseMatching"](cfg["coarse_matching"], profiler=mods["PassThroughProfiler"]()).eval()
        L = hw_c[0] * hw_c[1]
        f3d, f2d, _ = make_planted_matcher_inputs(n, L, 256, n_planted, noise, seed)
        g = torch.Generator().manual_seed(seed + 100)
        data = {"q_hw_i": torch.Size([hw_c[0] * 8, hw_c[1] * 8]), "q_hw_c": torch.Size(hw_c),
                "keypoints3d": torch.rand(1, n, 3, generator=g) - 0.5,
                "query_image_scale": torch.tensor([[1.25, 0.75]])}
        with torch.no_grad():
            cm(f3d, f2d, data)
        out = {k: data[k].numpy() for k in ["b_ids", "i_ids", "j_ids", "mconf", "mkpts_query_c",
                                            "mkpts_3d_db", "m_bids", "gt_mask"]}
        out.update(conf_digest(data["conf_matrix"]))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "M =", len(data["mconf"]))


def gen_fine():
    cls = load_reference_model_class()
    for name, (n, hw_i, m, seed) in FINE_CASES.items():
        cfg = default_config()
        model = cls(cfg).eval()
        model.load_state_dict(make_state_dict(cfg, 0), strict=True)
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
        with torch.no_grad():
            f3, win = model.fine_preprocess(data, bank_f, feat_f)
            win0 = win.clone()
            f3, win = model.loftr_fine(f3, win)
            model.fine_matching(f3, win, data)
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            expec_f=data["expec_f"].numpy(), mkpts_query_f=data["mkpts_query_f"].numpy(),
                            win_in_sum=win0.sum(-1).numpy(), f3_out=f3.numpy(),
                            win_out_sum=win.sum(-1).numpy())
        print(name, "M =", m)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    gen_stage_features()
    gen_matcher()
    gen_fine()
    gen_e2e()
