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
from tests.golden.cases import (E2E_CASES, MATCHER_CASES, FINE_CASES, TRANSFORMER_CASES, HIGHCONF_CASES,  # noqa: E402
                                BATCH_CASES, TRAIN_CASES)

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


def e2e_outputs(data):
    out = {}
    for k in ["b_ids", "i_ids", "j_ids", "gt_mask", "m_bids", "mkpts_3d_db",
              "mkpts_query_c", "mconf", "expec_f", "mkpts_query_f"]:
        if k in data:
            out[k] = data[k].numpy()
    out.update(conf_digest(data["conf_matrix"]))
    out["meta"] = np.array([data["bs"], *data["q_hw_i"], *data["q_hw_c"], *data["q_hw_f"],
                            data.get("W", -1)], dtype=np.int64)
    return out


def gen_transformer():
    """loftr_coarse alone at L = 4096 image tokens x N = 5000 points on seeded O(1) token streams; the 9.3 MB of
    outputs are stored as every 16th row plus per-row / per-channel reductions of all rows."""
    from tests.helpers import transformer_inputs, transformer_digest
    cls = load_reference_model_class()
    for name, (L, n, seed) in TRANSFORMER_CASES.items():
        cfg = default_config()
        model = cls(cfg).eval()
        model.load_state_dict(make_state_dict(cfg, 0), strict=True)
        tokens2d, bank = transformer_inputs(L, n, seed)
        with torch.no_grad():
            f3, f2 = model.loftr_coarse(bank, tokens2d)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **transformer_digest(f3[0], f2[0]))
        print(name, "done")


def optimise_highconf_bank(name, steps=160, lr=0.05):
    """A coarse descriptor bank for which backbone + positional encodings + the 6-layer transformer + dual softmax
    give hundreds of confident mutual matches (random banks give conf <= 0.08 with random weights): gradient ascent
    on log conf[i, cell(i)] of `n_planted` point -> interior-cell pairs THROUGH THE ORACLE (autograd on its
    functional forward), image / weights / keypoints fixed.  The result is rounded to fp16-representable values
    and becomes a stored fixture INPUT; the reference is then run on it like on any other input."""
    from oracle import onepose_oracle as O
    from tests.helpers import highconf_geometry
    hw, n, n_planted, thr, wseed, iseed = HIGHCONF_CASES[name]
    cfg = default_config(thr=thr)
    sd = make_state_dict(cfg, wseed)
    data = make_inputs(n, hw, iseed)
    _, _, kpts, cells = highconf_geometry(name)      # planted point i projects onto its cell under a known pose
    data["keypoints3d"] = kpts
    with torch.no_grad():
        feat_c, _ = O.backbone_forward(sd, data["query_image"])
        pe = O.sine_position_table(256, (256, 256))[:, :, :feat_c.size(2), :feat_c.size(3)]
        tokens2d = (feat_c + pe).flatten(2).transpose(1, 2)
        nk = O.normalize_3d_keypoints(data["keypoints3d"])
    rows = torch.arange(n_planted)
    bank = data["descriptors3d_coarse_db"].clone().requires_grad_(True)
    opt = torch.optim.Adam([bank], lr=lr)
    temp = cfg["coarse_matching"]["dual_softmax"]["temperature"]
    for it in range(steps):
        opt.zero_grad()
        enc = O.keypoint_encoding(sd, nk, bank)
        f3, f2 = O.local_feature_transformer(sd, "loftr_coarse", cfg["loftr_coarse"], enc, tokens2d)
        sim = torch.einsum("nlc,nsc->nls", f3 / 16.0, f2 / 16.0) / (temp + 1e-4)
        logc = torch.log_softmax(sim, 1) + torch.log_softmax(sim, 2)
        loss = -logc[0, rows, cells].mean()
        loss.backward()
        opt.step()
        if it % 10 == 0 or it == steps - 1:
            with torch.no_grad():
                c = logc[0, rows, cells].exp()
            print("  step %3d loss %.4f  conf(planted) median %.3f  >0.5: %d" % (it, loss.item(), c.median().item(), int((c > 0.5).sum())), flush=True)
    return bank.detach().half()


def gen_highconf():
    cls = load_reference_model_class()
    for name, (hw, n, n_planted, thr, wseed, iseed) in HIGHCONF_CASES.items():
        path = os.path.join(HERE, name + ".npz")
        from tests.helpers import highconf_geometry
        if os.path.exists(path) and "--reoptimise" not in sys.argv:
            bank16 = torch.from_numpy(np.load(path)["bank_c_f16"])       # keep the committed fixture input
        else:
            bank16 = optimise_highconf_bank(name)
        cfg = default_config(thr=thr)
        model = cls(cfg).eval()
        model.load_state_dict(make_state_dict(cfg, wseed), strict=True)
        data = make_inputs(n, hw, iseed)
        data["keypoints3d"] = highconf_geometry(name)[2]
        data["descriptors3d_coarse_db"] = bank16.float()
        with torch.no_grad():
            model(data)
        out = e2e_outputs(data)
        out["bank_c_f16"] = bank16.numpy()
        out["keypoints3d"] = data["keypoints3d"].numpy()
        np.savez_compressed(path, **out)
        c = data["mconf"]
        print(name, "M =", len(c), " conf > 0.5:", int((c > 0.5).sum()), " max %.4f" % c.max().item())


def conf_digest_batched(conf):
    """[B,N,L] -> per-sample reductions (stacked) + the whole matrix when small"""
    out = {"conf_rowsum": conf.sum(2).numpy(), "conf_colsum": conf.sum(1).numpy(),
           "conf_rowmax": conf.max(2).values.numpy(), "conf_colmax": conf.max(1).values.numpy()}
    if conf.numel() <= 400000:
        out["conf_matrix"] = conf.numpy()
    return out


def gen_batch():
    from tests.helpers import batch_setup
    cls = load_reference_model_class()
    for name in BATCH_CASES:
        cfg, sd, data = batch_setup(name)
        model = cls(cfg).eval()
        model.load_state_dict(sd, strict=True)
        with torch.no_grad():
            model(data)
        out = {}
        for k in ["b_ids", "i_ids", "j_ids", "gt_mask", "m_bids", "mkpts_3d_db", "mkpts_query_c", "mconf", "expec_f",
                  "mkpts_query_f"]:
            out[k] = data[k].numpy()
        out.update(conf_digest_batched(data["conf_matrix"]))
        out["meta"] = np.array([data["bs"], *data["q_hw_i"], *data["q_hw_c"], *data["q_hw_f"], data.get("W", -1)], dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "M =", len(data["mconf"]), "per sample:", torch.bincount(data["b_ids"], minlength=int(data["bs"])).tolist())


def gen_train():
    """The reference module in train() mode (what PL_OnePosePlus.training_step runs, lightning_model:54-81), forward
    only: BatchNorm batch statistics + running-statistics update, training branch of get_coarse_match.  The
    torch.randint draws of that branch are recorded and stored, so that other implementations can replay them."""
    from tests.helpers import train_setup
    cls = load_reference_model_class()
    for name in TRAIN_CASES:
        cfg, sd, data = train_setup(name)
        model = cls(cfg)
        model.load_state_dict(sd, strict=True)
        model.train()
        draws = []
        real = torch.randint

        def rec(*a, **kw):
            d = real(*a, **kw)
            draws.append(d.clone())
            return d
        torch.manual_seed(123)
        torch.randint = rec
        try:
            with torch.no_grad():
                model(data)
        finally:
            torch.randint = real
        out = {}
        for k in ["b_ids", "i_ids", "j_ids", "gt_mask", "m_bids", "mkpts_3d_db", "mkpts_query_c", "mconf", "expec_f",
                  "mkpts_query_f"]:
            out[k] = data[k].numpy()
        out.update(conf_digest_batched(data["conf_matrix"]))
        out["meta"] = np.array([data["bs"], *data["q_hw_i"], *data["q_hw_c"], *data["q_hw_f"], data.get("W", -1)], dtype=np.int64)
        for i, d in enumerate(draws):
            out["randint_%d" % i] = d.numpy()
        out["n_randint"] = np.array(len(draws))
        # gradients of a fixed scalar of the two outputs the loss differentiates (losses.py:125-133), by the
        # reference's own autograd: digest (sum, L2 norm) for every parameter + a few whole tensors
        from tests.helpers import train_loss_weights, GRAD_TENSORS
        d2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in train_setup(name)[2].items()}
        model2 = cls(cfg)
        model2.load_state_dict(sd, strict=True)
        model2.train()
        replay = [d.clone() for d in draws]
        torch.randint = lambda *a, **kw: replay.pop(0)
        try:
            model2(d2)
        finally:
            torch.randint = real
        wc, we = train_loss_weights(d2["conf_matrix"].shape, d2["expec_f"].shape)
        ((d2["conf_matrix"] * wc).sum() + (d2["expec_f"] * we).sum()).backward()
        gnames = [n for n, p_ in model2.named_parameters() if p_.grad is not None]
        out["grad_names"] = np.array(gnames)
        out["grad_digest"] = np.array([[float(model2.get_parameter(n).grad.double().sum()),
                                        float(model2.get_parameter(n).grad.double().norm())] for n in gnames])
        for n in GRAD_TENSORS:
            out["grad/" + n] = model2.get_parameter(n).grad.numpy()
        after = model.state_dict()
        for k, v in after.items():          # running statistics after ONE training forward
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                out["bn/" + k] = v.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "M' =", len(data["b_ids"]), "M =", len(data["mconf"]), "draws", [tuple(d.shape) for d in draws])


def gen_bankbuild():
    """The reference's own gather_3d_ann + mean_descriptors_and_scores (feature_process.py:255-311, :527-541) on
    synthetic tracks; the module's imports of packages that are not installed here are stubbed."""
    import types
    from tests.helpers import bankbuild_inputs, BANKBUILD_CASES
    for name in ("h5py", "ray", "ray.actor", "loguru", "tqdm"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["ray"].remote = lambda *a, **k: (lambda f: f)
    sys.modules["ray.actor"].ActorHandle = object
    sys.modules["loguru"].logger = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None, error=lambda *a, **k: None)
    sys.modules["tqdm"].tqdm = lambda x, *a, **k: x
    # load the file itself (its packages' __init__ pull in cv2 / COLMAP tooling that this step does not use)
    import importlib.util
    saved = {k: sys.modules.get(k) for k in ("src", "src.utils", "src.utils.colmap", "src.utils.ray_utils")}
    for k in saved:
        sys.modules[k] = types.ModuleType(k)
    sys.modules["src.utils.colmap"].read_write_model = None
    for fn in ("ProgressBar", "chunks", "chunk_index", "split_dict"):
        setattr(sys.modules["src.utils.ray_utils"], fn, None)
    try:
        spec = importlib.util.spec_from_file_location("ref_feature_process", "/root/reference/src/sfm_utils/postprocess/feature_process.py")
        FP = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(FP)
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v
    for name, (n, dim, seed) in BANKBUILD_CASES.items():
        feat, score, xyzs, points_idxs = bankbuild_inputs(n, dim, seed)
        pos, desc, scores, idxs = FP.gather_3d_ann(feat, score, xyzs, points_idxs, verbose=False)
        avg, avg_scores, idxs2 = FP.mean_descriptors_and_scores(desc, scores, idxs)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kp3d_position=pos, idxs=idxs, avg_descriptors=avg,
                            avg_scores=avg_scores, desc_rowsum=desc.sum(1), n_rows=np.array(desc.shape[0]))
        print(name, "points", len(idxs), "rows", desc.shape[0], avg.dtype)


def gen_loss():
    """The reference's own Loss (src/lightning_model/losses.py) and fine_supervision
    (src/models/OnePosePlus/utils/fine_supervision.py) on seeded inputs: loss values, expec_f_gt and the gradients of the
    total loss w.r.t. conf_matrix / expec_f by the reference's autograd.  The two files are loaded by path (their
    packages import pytorch_lightning / kornia), loguru is stubbed."""
    import importlib.util
    import types
    from tests.helpers import loss_inputs
    from tests.golden.cases import LOSS_CASES, LOSS_CONFIG
    if "loguru" not in sys.modules:
        m = types.ModuleType("loguru")
        m.logger = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None, error=lambda *a, **k: None)
        sys.modules["loguru"] = m

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, os.path.join("/root/reference", path))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    L = load("src/lightning_model/losses.py", "ref_losses")
    FS = load("src/models/OnePosePlus/utils/fine_supervision.py", "ref_fine_supervision")
    for name in LOSS_CASES:
        data, hparams = loss_inputs(name)
        data["conf_matrix"].requires_grad_(True)
        data["expec_f"].requires_grad_(True)
        FS.fine_supervision(data, hparams)
        loss_mod = L.Loss(dict(LOSS_CONFIG))
        loss_mod.train()
        loss_mod(data)
        data["loss"].backward()
        gc = data["conf_matrix"].grad
        out = {"expec_f_gt": data["expec_f_gt"].numpy(), "loss": data["loss"].detach().numpy(),
               "loss_c": data["loss_scalars"]["loss_c"].numpy(),
               "loss_f": data["loss_scalars"]["loss_f"].numpy() if "loss_f" in data["loss_scalars"] else np.array(np.nan),
               "grad_conf": gc.numpy(), "grad_expec": data["expec_f"].grad.numpy()}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "loss", float(data["loss"]), "loss_c", float(out["loss_c"]), "loss_f", float(out["loss_f"]),
              "|grad_conf|", float(gc.abs().sum()))


def gen_e2e():
    cls = load_reference_model_class()
    for name, (hw, n, thr, wseed, iseed, fine) in E2E_CASES.items():
        if os.path.exists(os.path.join(HERE, name + ".npz")) and "--missing-only" in sys.argv:
            continue
        cfg = default_config(thr=thr, fine=fine)
        model = cls(cfg).eval()
        model.load_state_dict(make_state_dict(cfg, wseed), strict=True)
        data = make_inputs(n, hw, iseed)
        with torch.no_grad():
            model(data)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **e2e_outputs(data))
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
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = {"stages": gen_stage_features, "matcher": gen_matcher, "fine": gen_fine, "e2e": gen_e2e,
             "transformer": gen_transformer, "highconf": gen_highconf, "batch": gen_batch,
             "train": gen_train, "bankbuild": gen_bankbuild, "loss": gen_loss}
    for k, fn in steps.items():
        if not only or k in only:
            fn()
