"""GPU: whole `OnePosePlus_model(data)` forward through the HIP path against (a) the golden
vectors produced by the upstream reference and (b) the CPU oracle, plus size-independent
properties at BASELINE sizes (512x512 x 5k / 15k points)."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden.cases import E2E_CASES, HIGHCONF_CASES, BATCH_CASES, TRAIN_CASES

pytestmark = pytest.mark.gpu


PRECISIONS = ["bf16x3", "fp32"]     # every GEMM arithmetic of the product meets the same parity bar


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", list(E2E_CASES))
def test_e2e_vs_golden(name, precision):
    from tests import hip_ops as ops
    cfg, sd, data = H.e2e_setup(name)
    model = ops.make_model(cfg, sd, precision)
    out = ops.run_model(model, data)
    gold = H.load_golden(name)
    H.assert_match_outputs(out, gold, where=name)
    meta = gold["meta"]
    assert out["bs"] == meta[0] and tuple(out["q_hw_i"]) == tuple(meta[1:3])
    assert tuple(out["q_hw_c"]) == tuple(meta[3:5]) and tuple(out["q_hw_f"]) == tuple(meta[5:7])
    for k in ("i_ids", "j_ids", "b_ids", "m_bids"):
        assert out[k].dtype == torch.int64 and out[k].is_cuda
    assert out["gt_mask"].dtype == torch.bool
    assert out["conf_matrix"].shape == (1, data["keypoints3d"].shape[1], meta[3] * meta[4])
    if cfg["fine_matching"]["enable"]:
        assert out["W"] == meta[7]
        assert out["expec_f"].shape == (len(gold["mconf"]), 3)
    assert out["mkpts_query_f"].shape == (len(gold["mconf"]), 2)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", list(HIGHCONF_CASES))
def test_e2e_high_confidence_vs_golden(name, precision):
    """Full coarse-to-fine forward whose coarse bank was optimised (through the oracle) so that ~1500 pairs reach
    conf > 0.5 (up to 0.999) AFTER backbone + positional encodings + 6 transformer layers: every GEMM arithmetic must
    reproduce the reference's indices bit-exactly and its confidences / fine offsets within 1e-4 where the bar
    actually bites; the relative error is reported too."""
    from tests import hip_ops as ops
    cfg, sd, data = H.highconf_setup(name)
    out = ops.run_model(ops.make_model(cfg, sd, precision), data)
    gold = H.load_golden(name)
    assert len(gold["mconf"]) > 1000 and (gold["mconf"] > 0.5).sum() > 1000
    H.assert_match_outputs(out, gold, where=name)
    rel = H.conf_relative_error(out["mconf"], gold["mconf"])
    absd = float(np.abs(H.to_np(out["mconf"]) - gold["mconf"]).max())
    de = np.abs(H.to_np(out["expec_f"]) - gold["expec_f"])
    print("%s [%s]: M %d, mconf max abs err %.2e, max rel err %.2e, fine offsets max abs err %.2e, std column %.2e" %
          (name, precision, len(gold["mconf"]), absd, rel, float(de[:, :2].max()), float(de[:, 2].max())))
    assert rel < 1e-3


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("name", list(BATCH_CASES))
def test_batched_masked_vs_golden(name, precision):
    """B > 1 with per-sample point clouds / image scales and a coarse-resolution `query_image_mask` against the
    reference: the mask path of linear attention and of the coarse matcher, the batch-0 keypoint extent (quirk q4)
    and the (b, i) ordering of the concatenated matches."""
    from tests import hip_ops as ops
    cfg, sd, data = H.batch_setup(name)
    out = ops.run_model(ops.make_model(cfg, sd, precision), data)
    gold = H.load_golden(name)
    assert len(gold["mconf"]) > 0 and len(np.unique(gold["b_ids"])) == len(BATCH_CASES[name][4])
    H.assert_batched_outputs(out, gold, where=name)
    if "query_image_mask" in data:      # masked cells carry exactly zero confidence and are never matched
        m = data["query_image_mask"].flatten(-2).bool()
        conf = out["conf_matrix"].cpu()
        assert (conf.transpose(1, 2)[~m] == 0).all()
        assert m[out["b_ids"].cpu(), out["j_ids"].cpu()].all()


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("name", list(TRAIN_CASES))
def test_train_mode_forward_vs_golden(name, precision):
    """`model.train(); model(batch)` (forward of PL_OnePosePlus.training_step, lightning_model:54-81) on the HIP path
    against the reference module in train() mode: BatchNorm batch statistics over the B images, running statistics
    after the step, materialised conf_matrix, training branch of get_coarse_match (predictions + ground-truth
    padding, the reference's recorded torch.randint draws replayed), fine level on the padded list."""
    from tests import hip_ops as ops
    cfg, sd, data = H.train_setup(name)
    gold = H.load_golden(name)
    model = ops.make_model(cfg, sd, precision)
    model.train()
    model.train_randint = H.RecordedRandint([gold["randint_%d" % i] for i in range(int(gold["n_randint"]))])
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    with torch.no_grad():
        model(d)
    torch.cuda.synchronize()
    H.assert_train_outputs(d, model.state_dict(), gold, tol_bn=1e-4, where=name)
    assert d["gt_mask"].sum().item() == len(gold["b_ids"]) - len(gold["mconf"])
    # the step moved the running statistics: the same module in eval() now differs from a fresh one
    model.eval()
    e1 = ops.run_model(model, {k: v[:1] for k, v in data.items() if k != "conf_matrix_gt"})
    e0 = ops.run_model(ops.make_model(cfg, sd, precision), {k: v[:1] for k, v in data.items() if k != "conf_matrix_gt"})
    assert not torch.equal(e0["conf_matrix"], e1["conf_matrix"])


def test_train_step_invalidates_resident_object_tokens():
    """eval on a RESIDENT object (same keypoint / bank tensors -> the per-object token cache is hit), one training step
    plus a parameter update that moves the keypoint-MLP weights, eval again on the SAME tensors: the second eval must
    equal a freshly built module that loaded the updated state dict (the token cache and the folded packing follow
    the parameters on every exit path of the train-mode forward, fine level on and off)."""
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.config import default_config
    name = "train_b2_128x128_n300"
    cfg, sd, data = H.train_setup(name)
    gold = H.load_golden(name)
    for fine in (True, False):
        cfg_f = default_config(thr=cfg["coarse_matching"]["thr"], fine=fine)
        cfg_f["coarse_matching"]["train"] = cfg["coarse_matching"]["train"]
        model = ops.make_model(cfg_f, sd)
        resident = {k: v[:1].cuda() for k, v in data.items() if k != "conf_matrix_gt"}
        with torch.no_grad():
            model.eval()
            d0 = dict(resident)
            model(d0)                                                       # fills the object-token cache
            model.train()
            model.train_randint = H.RecordedRandint([gold["randint_%d" % i] for i in range(int(gold["n_randint"]))])
            model({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()})
            for n_, p in model.named_parameters():                          # "optimiser" update, keypoint MLP included
                p.mul_(1.03)
            model.eval()
            d1 = dict(resident)
            model(d1)                                                       # same tensors: must NOT reuse stale tokens
            fresh = ops.make_model(cfg_f, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
            d2 = {k: v.clone() for k, v in resident.items()}
            fresh.eval()
            fresh(d2)
        torch.cuda.synchronize()
        assert not torch.equal(d0["conf_matrix"], d1["conf_matrix"])
        assert torch.equal(d1["conf_matrix"], d2["conf_matrix"]), "fine=%s" % fine
        assert torch.equal(d1["i_ids"], d2["i_ids"]) and torch.equal(d1["mconf"], d2["mconf"])


@pytest.mark.parametrize("name", ["train_b2_128x128_n300", "train_b4_64x96_n150"])
def test_training_step_gradients(name):
    """One training step as PL_OnePosePlus.training_step runs it (lightning_model:54-81): `matcher(batch)` in train()
    mode with gradients enabled -- forward values from the HIP path, `conf_matrix` / `expec_f` carrying a grad_fn --
    then a scalar of those two outputs and `.backward()`.  Every parameter gradient is compared with the gradient the
    reference's own autograd produced for the same scalar (stored in the fixture)."""
    from tests import hip_ops as ops
    cfg, sd, data = H.train_setup(name)
    gold = H.load_golden(name)
    model = ops.make_model(cfg, sd)
    model.train()
    model.train_randint = H.RecordedRandint([gold["randint_%d" % i] for i in range(int(gold["n_randint"]))])
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    model(d)                                                        # gradients enabled
    assert d["conf_matrix"].requires_grad and d["expec_f"].requires_grad and not d["mconf"].requires_grad
    H.assert_train_outputs(d, model.state_dict(), gold, tol_bn=1e-4, where=name)     # values are the HIP forward's
    wc, we = H.train_loss_weights(d["conf_matrix"].shape, d["expec_f"].shape)
    loss = (d["conf_matrix"] * wc.cuda()).sum() + (d["expec_f"] * we.cuda()).sum()
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    # forward and backward = HIP nodes.  The bound is NOT kernel accuracy (every backward kernel is held to 5e-6, the whole backbone
    # backward to 3e-5 of each gradient's largest entry, in tests/test_train_bwd_gpu.py): the fixture's gradients come from the
    # reference's fp32 CPU run, and a pre-activation within fp32 rounding of zero takes different sides of a ReLU kink in two
    # fp32 evaluations -- one such element moves the affected gradients by up to ~1e-2 of their size (tools/bwd_diag.py; the
    # backbone has ~10^6 pre-activations, the smallest |z| is ~1e-7)
    H.assert_train_grads(grads, gold, rel=1e-2, where=name)
    # an optimiser step changes the parameters, and the next forward (eval) sees them: the packed weights follow
    with torch.no_grad():
        for p in model.parameters():
            p.add_(p.grad, alpha=-1e-3)
    model.eval()
    e1 = ops.run_model(model, {k: v[:1] for k, v in data.items() if k != "conf_matrix_gt"})
    e0 = ops.run_model(ops.make_model(cfg, sd), {k: v[:1] for k, v in data.items() if k != "conf_matrix_gt"})
    assert not torch.equal(e0["conf_matrix"], e1["conf_matrix"])


@pytest.mark.parametrize("N", [7000, 15000])
def test_training_step_at_baseline_config5_size(N):
    """BASELINE configs[4] shape on one GPU: B = 4 (train.yaml:185), 512x512 images, N = 7000 points (shape3d_train,
    train.yaml:194) and N = 15000 (BASELINE.json configs[4]: "15k-point clouds" = max_num_kp3d): train()-mode forward on the HIP path +
    backward, one optimiser-style update; properties only (the fixtures above pin the numbers at small sizes)."""
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.config import default_config
    from onepose_plus_plus_amd.synthetic import make_state_dict, make_inputs
    B, hw = 4, (512, 512)
    cfg = default_config(thr=0.2)
    model = ops.make_model(cfg, make_state_dict(cfg, 0))
    model.train()
    parts = [make_inputs(N, hw, 30 + b) for b in range(B)]
    d = {k: torch.cat([p[k] for p in parts], 0).cuda() for k in parts[0]}
    g = torch.Generator().manual_seed(9)
    gt = torch.zeros(B, N, 4096, dtype=torch.int16)
    for b in range(B):
        gt[b, torch.randperm(N, generator=g)[:1500], torch.randperm(4096, generator=g)[:1500]] = 1
    d["conf_matrix_gt"] = gt.cuda()
    model(d)
    max_train = int(B * 4096 * 0.3)
    assert d["conf_matrix"].shape == (B, N, 4096) and d["conf_matrix"].requires_grad
    assert len(d["b_ids"]) == max(max_train, len(d["mconf"]) + 200) and d["expec_f"].shape == (len(d["b_ids"]), 3)
    assert d["gt_mask"].sum().item() == len(d["b_ids"]) - len(d["mconf"])
    loss = -(torch.log(d["conf_matrix"][d["conf_matrix_gt"] == 1].clamp(1e-6))).mean() + (d["expec_f"][:, :2] ** 2).sum(-1).mean()
    loss.backward()
    n = 0
    for name, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        n += 1
    assert n == 144 and torch.isfinite(loss)


def test_training_step_flow_with_hip_loss():
    """`PL_OnePosePlus.training_step` (lightning_model:54-60) with this package's pieces: model(batch) in train() mode,
    `fine_supervision(batch, hparams)`, `Loss(hparams['loss'])(batch)`, backward, optimiser step.  The loss equals the
    oracle's restatement of the reference Loss on the same outputs; every parameter receives a finite gradient."""
    from oracle import loss_oracle as LO
    from tests import hip_ops as ops
    from tests.golden.cases import LOSS_CONFIG
    from onepose_plus_plus_amd.losses import Loss, fine_supervision
    name = "train_b4_64x96_n150"
    cfg, sd, data = H.train_setup(name)
    model = ops.make_model(cfg, sd)
    model.train()
    B, N = data["keypoints3d"].shape[:2]
    L = data["conf_matrix_gt"].shape[2]
    g = torch.Generator().manual_seed(77)
    data["fine_location_matrix_gt"] = torch.full((B, N, L, 2), -50.0)
    pos = torch.nonzero(data["conf_matrix_gt"] == 1)
    wc = 96 // 8
    scale = data["query_image_scale"][pos[:, 0]][:, [1, 0]]
    cell = torch.stack([pos[:, 2] % wc, pos[:, 2] // wc], 1) * 8.0 * scale
    data["fine_location_matrix_gt"][pos[:, 0], pos[:, 1], pos[:, 2]] = cell + (torch.rand(len(pos), 2, generator=g) - 0.5) * 6.0 * scale
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    hparams = {"OnePosePlus": cfg, "loss": dict(LOSS_CONFIG)}
    loss_mod = Loss(hparams["loss"]).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    model(d)
    fine_supervision(d, hparams)
    loss_mod(d)
    assert d["expec_f_gt"].shape == (len(d["b_ids"]), 2)
    ref = LO.loss_forward({k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()
                           if k in ("conf_matrix", "conf_matrix_gt", "expec_f", "expec_f_gt")}, LOSS_CONFIG, training=True)
    assert abs(float(d["loss"].detach()) - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
    assert set(d["loss_scalars"]) == {"loss_c", "loss_f", "loss"}
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt.zero_grad()
    d["loss"].backward()
    n = 0
    for k, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        n += 1
    assert n == 144
    opt.step()
    assert any(not torch.equal(before[k], p.detach()) for k, p in model.named_parameters())


def test_worker_flow_recovers_pose():
    """The data flow of the reference's worker (`extract_matches`, inference_OnePosePlus_worker.py:7-37: model(data)
    then compute_query_pose_errors -> ransac_PnP on `mkpts_query_f` / `mkpts_3d_db`, metric_utils.py:221-270) through
    the drop-in aliases (`dropin.install`), on the synthetic object of the high-confidence case whose planted
    points project onto their cells under a known pose: the pose comes back from the ~1490 HIP matches."""
    import sys
    from onepose_plus_plus_amd import dropin
    from onepose_plus_plus_amd.pose import ransac_PnP
    name = "highconf_512x512_n3000"
    cfg, sd, data = H.highconf_setup(name)
    K, pose_gt, kpts, cells = H.highconf_geometry(name)
    assert torch.equal(kpts, data["keypoints3d"])
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "src" or k.startswith("src.")}
    try:
        dropin.install(pnp=False)
        from src.models.OnePosePlus.OnePosePlusModel import OnePosePlus_model as RefPathModel   # inference_OnePosePlus.py:11
        match_model = RefPathModel(cfg)                              # build_model: constructor, strict load, eval
        match_model.load_state_dict({k: v for k, v in sd.items()}, strict=True)
        match_model.eval()
        match_model.cuda()                                           # worker: inference_OnePosePlus_worker.py:43
        data_c = {k: v.cuda() if isinstance(v, torch.Tensor) else v for k, v in data.items()}
        data_c["query_intrinsic"] = torch.from_numpy(K)[None].cuda()
        data_c["query_pose_gt"] = torch.from_numpy(np.concatenate([pose_gt, [[0, 0, 0, 1.0]]]))[None].cuda()
        with torch.no_grad():
            match_model(data_c)
    finally:
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
            if k not in saved:
                del sys.modules[k]
    # compute_query_pose_errors (metric_utils.py:221-270) with the on-device solver and the reference's settings
    m_bids = data_c["m_bids"]
    assert (m_bids == 0).all() and len(m_bids) > 1400
    pose, homo, inliers, state = ransac_PnP(data_c["query_intrinsic"][0].cpu().numpy(), data_c["mkpts_query_f"],
                                            data_c["mkpts_3d_db"], scale=1000, pnp_reprojection_error=5,
                                            img_hw=[512, 512], use_pycolmap_ransac=False)
    assert state and pose.shape == (3, 4) and homo.shape == (4, 4)
    r_err, t_err = H.pose_errors(pose, data_c["query_pose_gt"][0].cpu().numpy())
    print("worker flow: %d matches, %d inliers, R err %.3f deg, t err %.3f cm" % (len(m_bids), len(inliers), r_err, t_err))
    assert len(inliers) > 0.8 * len(m_bids)
    assert r_err < 1.0 and t_err < 1.0                                # the 1 cm / 1 deg bar of the reference's metrics


def _scaled_backbone(sd, S):
    """The same network with every activation of the ResNet stages multiplied by the power of two S: stem BN affine
    x S, running means / shifts of the stage BNs x S, the three FPN laterals x 1/S.  All of it commutes exactly with
    fp32 arithmetic (and with the bf16x3 split), so the reference's outputs are bit-identical to the unscaled ones."""
    import re
    big = {k: v.clone() for k, v in sd.items()}
    big["backbone.bn1.weight"] = sd["backbone.bn1.weight"] * S
    big["backbone.bn1.bias"] = sd["backbone.bn1.bias"] * S
    for k in sd:
        if re.match(r"backbone\.layer[123]\.\d\.(bn\d|downsample\.1)\.(bias|running_mean)$", k):
            big[k] = sd[k] * S
    for k in ("backbone.layer3_outconv.weight", "backbone.layer2_outconv.weight", "backbone.layer1_outconv.weight"):
        big[k] = sd[k] / S
    return big


@pytest.mark.parametrize("scale_log2", [18, -20])
def test_activation_range_1e5_and_1e_minus_6(scale_log2):
    """Backbone activations of ~1.5e5..1.2e6 (S = 2^18, beyond the fp16 range) and of ~1e-6 (S = 2^-20): the default
    bf16x3 arithmetic and fp32 reproduce the reference's golden outputs unchanged (the scaled network is exactly
    equivalent: neither arithmetic has a range restriction)."""
    from tests import hip_ops as ops
    name = "e2e_128x128_n300_thr0"
    cfg, sd, data = H.e2e_setup(name)
    big = _scaled_backbone(sd, 2.0 ** scale_log2)
    gold = H.load_golden(name)
    outs = {}
    for precision in ("bf16x3", "fp32"):
        outs[precision] = ops.run_model(ops.make_model(cfg, big, precision), data)
        H.assert_match_outputs(outs[precision], gold, where="%s at 2^%d" % (precision, scale_log2))
    plain = ops.run_model(ops.make_model(cfg, sd, "bf16x3"), data)
    assert torch.equal(plain["conf_matrix"], outs["bf16x3"]["conf_matrix"])        # the split commutes with 2^k


@pytest.mark.parametrize("precision", PRECISIONS)
def test_e2e_vs_oracle_and_determinism(precision):
    from oracle import onepose_oracle as O
    from tests import hip_ops as ops
    cfg, sd, data = H.e2e_setup("e2e_128x128_n300_thr0")
    model = ops.make_model(cfg, sd, precision)
    out1 = ops.run_model(model, data)
    out2 = ops.run_model(model, data)
    for k in ("conf_matrix", "i_ids", "j_ids", "mconf", "expec_f", "mkpts_query_f"):
        assert torch.equal(out1[k], out2[k]), k            # bit-reproducible run to run
    ref = {k: v.clone() for k, v in data.items()}
    O.forward(sd, ref, cfg)
    gold = {k: H.to_np(ref[k]) for k in ("i_ids", "j_ids", "mconf", "expec_f", "mkpts_query_f", "mkpts_query_c")}
    gold.update(H.conf_digest_t(ref["conf_matrix"]))
    H.assert_match_outputs(out1, gold, where="oracle")
    # input tensors are not modified (ownership contract)
    fresh = H.e2e_setup("e2e_128x128_n300_thr0")[2]
    for k, v in fresh.items():
        assert torch.equal(v, data[k]), k


RANDOM_SHAPES = [(s, 8 * (4 + (s * 7) % 17), 8 * (4 + (s * 11) % 15), 3 + (s * 37) % 401, (0.0, 0.02, 0.05)[s % 3]) for s in range(12)]


@pytest.mark.parametrize("seed,h,w,n,thr", RANDOM_SHAPES)
def test_random_shapes_vs_oracle(seed, h, w, n, thr):
    """Ragged shapes nobody tuned for (H, W any multiple of 8 in 32..160, N = 3..403, anisotropic image scales):
    full coarse-to-fine forward against the pinned oracle.  The whole confidence matrix within 1e-4; the match set
    identical, where a pair may differ only if the ORACLE's own decision is marginal (confidence within 1e-4 of the
    threshold, or of the runner-up in its row / column)."""
    from oracle import onepose_oracle as O
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.config import default_config
    from onepose_plus_plus_amd.synthetic import make_state_dict, make_inputs
    cfg = default_config(thr=thr)
    sd = make_state_dict(cfg, 100 + seed)
    data = make_inputs(n, (h, w), 200 + seed)
    data["query_image_scale"] = torch.tensor([[1.0 + 0.125 * (seed % 4), 1.0 - 0.0625 * (seed % 5)]])
    out = ops.run_model(ops.make_model(cfg, sd), data)
    ref = {k: v.clone() for k, v in data.items()}
    O.forward(sd, ref, cfg)
    conf_o = ref["conf_matrix"][0]
    conf_h = out["conf_matrix"][0].cpu()
    assert conf_h.shape == conf_o.shape == (n, (h // 8) * (w // 8))
    assert (conf_h - conf_o).abs().max() <= H.TOL_CONF
    pairs_o = set(zip(ref["i_ids"].tolist(), ref["j_ids"].tolist()))
    pairs_h = set(zip(out["i_ids"].tolist(), out["j_ids"].tolist()))
    top2_r = conf_o.topk(min(2, conf_o.shape[1]), dim=1).values
    top2_c = conf_o.topk(min(2, conf_o.shape[0]), dim=0).values
    for (i, j) in pairs_o ^ pairs_h:
        c = float(conf_o[i, j])
        marginal = abs(c - thr) <= 1e-4
        marginal |= top2_r.shape[1] > 1 and float(top2_r[i, 0] - top2_r[i, 1]) <= 1e-4
        marginal |= top2_c.shape[0] > 1 and float(top2_c[0, j] - top2_c[1, j]) <= 1e-4
        assert marginal, (seed, i, j, c)
    common = sorted(pairs_o & pairs_h)
    assert len(common) >= len(pairs_o) - 2
    if common:
        io = {p: k for k, p in enumerate(zip(ref["i_ids"].tolist(), ref["j_ids"].tolist()))}
        ih = {p: k for k, p in enumerate(zip(out["i_ids"].tolist(), out["j_ids"].tolist()))}
        ko, kh = [io[p] for p in common], [ih[p] for p in common]
        assert (out["mconf"].cpu()[kh] - ref["mconf"][ko]).abs().max() <= H.TOL_CONF
        assert (out["expec_f"].cpu()[kh, :2] - ref["expec_f"][ko, :2]).abs().max() <= H.TOL_OFFSET
        assert (out["expec_f"].cpu()[kh, 2] - ref["expec_f"][ko, 2]).abs().max() <= H.TOL_OFFSET * H.STD_TOL_FACTOR
        assert (out["mkpts_query_f"].cpu()[kh] - ref["mkpts_query_f"][ko]).abs().max() <= H.TOL_PIXEL
        assert torch.equal(out["mkpts_3d_db"].cpu()[kh], ref["mkpts_3d_db"][ko])
    assert (out["i_ids"][1:] >= out["i_ids"][:-1]).all()


@pytest.mark.parametrize("n,precision", [(5000, "bf16x3"), (5000, "fp32"), (15000, "bf16x3"), (15000, "fp32")])
def test_full_size_properties(n, precision):
    """BASELINE sizes: properties that hold for any input (no oracle needed)."""
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.synthetic import make_state_dict, make_inputs
    from onepose_plus_plus_amd.config import default_config
    cfg = default_config(thr=0.0)
    model = ops.make_model(cfg, make_state_dict(cfg, 0), precision)
    out = ops.run_model(model, make_inputs(n, (512, 512), 1))
    conf = out["conf_matrix"][0]
    assert torch.isfinite(conf).all() and (conf >= 0).all()
    assert conf.sum(1).max() <= 1 + 1e-4 and conf.sum(0).max() <= 1 + 1e-4      # product of two softmaxes
    i, j, c = out["i_ids"], out["j_ids"], out["mconf"]
    assert (i[1:] > i[:-1]).all()                                                   # ascending, unique 3D ids
    assert len(torch.unique(j)) == len(j)                                           # mutual NN: unique cells
    assert torch.equal(c, conf[i, j])
    assert torch.equal(c, conf.max(1).values[i]) and torch.equal(c, conf.max(0).values[j])
    jy, jx = j // 64, j % 64
    assert (jy >= 2).all() and (jx >= 2).all()                                      # border quirk q1
    # every mutual-NN pair above thr outside the border is reported (completeness)
    rmax, rarg = conf.max(1)
    cmax = conf.max(0).values
    ok = (rmax > 0.0) & (rmax == cmax[rarg]) & (rarg // 64 >= 2) & (rarg % 64 >= 2)
    assert int(ok.sum()) == len(i)
    ex = out["expec_f"]
    assert ex.shape == (len(i), 3) and (ex[:, :2].abs() <= 1).all() and (ex[:, 2] > 0).all()
    assert (out["mkpts_query_f"] - out["mkpts_query_c"]).abs().max() <= 2.0 * 2 + 1e-3


def test_no_cpu_fallback():
    from onepose_plus_plus_amd import OnePosePlus_model, default_config
    m = OnePosePlus_model(default_config()).eval()
    with pytest.raises(RuntimeError):
        m({"query_image": torch.zeros(1, 1, 64, 64)})


def test_object_token_cache_is_transparent():
    """The per-object cache of encoded 3D tokens must never change results: new object, same
    object again, and an in-place edit of the bank all give what an uncached model gives."""
    from tests import hip_ops as ops
    cfg, sd, data_a = H.e2e_setup("e2e_128x128_n300_thr0")
    data_b = H.e2e_setup("e2e_128x128_n300_thr0")[2]
    g = torch.Generator().manual_seed(99)
    data_b["descriptors3d_coarse_db"] = torch.randn(1, 256, 300, generator=g)
    data_b["keypoints3d"] = torch.rand(1, 300, 3, generator=g) - 0.5
    cached = ops.make_model(cfg, sd)
    plain = ops.make_model(cfg, sd)
    plain.cache_object_tokens = False
    da = {k: v.cuda() for k, v in data_a.items()}
    db = {k: v.cuda() for k, v in data_b.items()}

    def run(m, d):
        d = dict(d)
        with torch.no_grad():
            m(d)
        torch.cuda.synchronize()
        return d

    for d in (da, db, da, da):
        o1, o2 = run(cached, d), run(plain, d)
        for k in ("conf_matrix", "i_ids", "j_ids", "mconf", "expec_f"):
            assert torch.equal(o1[k], o2[k]), k
    da["descriptors3d_coarse_db"].mul_(0.5)          # in-place edit -> version bump -> re-encode
    o1, o2 = run(cached, da), run(plain, da)
    assert torch.equal(o1["conf_matrix"], o2["conf_matrix"])


def test_matcher_pool_matches_sequential():
    """n forwards in flight on separate HIP streams give exactly the sequential results, in order."""
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.serving import MatcherPool
    from onepose_plus_plus_amd.synthetic import make_inputs
    cfg, sd, _ = H.e2e_setup("e2e_128x128_n300_thr0")
    seq_model = ops.make_model(cfg, sd)
    datas = [make_inputs(300, (128, 128), 50 + i) for i in range(7)]
    ref = [ops.run_model(seq_model, d) for d in datas]
    pool = MatcherPool(cfg, sd, n_streams=3)
    got = pool.map([{k: v.cuda() for k, v in d.items()} for d in datas])
    torch.cuda.synchronize()
    for r, g in zip(ref, got):
        for k in ("conf_matrix", "i_ids", "j_ids", "mconf", "expec_f", "mkpts_query_f"):
            assert torch.equal(r[k], g[k]), k


def test_matcher_pool_waits_for_the_producer_stream():
    """Inputs produced ASYNCHRONOUSLY on the caller's stream (what ingest.read_grayscale_u8 does: upload + resize
    kernels) must be complete before a pool worker's stream reads them: the image buffer is overwritten behind a long
    queue of work on the current stream, then `map` is called at once."""
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.serving import MatcherPool
    from onepose_plus_plus_amd.synthetic import make_inputs
    cfg, sd, _ = H.e2e_setup("e2e_128x128_n300_thr0")
    datas = [make_inputs(300, (128, 128), 70 + i) for i in range(4)]
    ref = [ops.run_model(ops.make_model(cfg, sd), d) for d in datas]
    pool = MatcherPool(cfg, sd, n_streams=2)
    dev = [{k: v.cuda() for k, v in d.items()} for d in datas]
    real = [d["query_image"].clone() for d in dev]
    for d in dev:
        d["query_image"].zero_()                     # stale content
    torch.cuda.synchronize()
    busy = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):                              # ~tens of ms of queued work on the current stream ...
        busy = busy @ busy * 1e-3
    for d, r in zip(dev, real):
        d["query_image"].copy_(r)                    # ... behind which the real images are written
    got = pool.map(dev)
    torch.cuda.synchronize()
    for r, g in zip(ref, got):
        assert torch.equal(r["conf_matrix"], g["conf_matrix"]) and torch.equal(r["i_ids"], g["i_ids"])


def test_tile_policy_does_not_change_results():
    """`set_tile_policy("throughput")` (MatcherPool, bench --streams > 1) picks other tile shapes: bit-identical outputs."""
    from tests import hip_ops as ops
    cfg, sd, data = H.e2e_setup("e2e_512x512_n2000_thr0")
    a = ops.run_model(ops.make_model(cfg, sd), data)
    m = ops.make_model(cfg, sd)
    m.set_tile_policy("throughput")
    b = ops.run_model(m.cuda(), data)
    for k in ("conf_matrix", "i_ids", "j_ids", "mconf", "expec_f", "mkpts_query_f"):
        assert torch.equal(a[k], b[k]), k


def test_precisions_agree_at_full_size():
    """512x512 x 5k points, thr 0: the bf16x3-split GEMMs select exactly the matches of the fp32 GEMMs and
    agree on confidences / fine offsets far inside the 1e-4 bar."""
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.synthetic import make_state_dict, make_inputs
    from onepose_plus_plus_amd.config import default_config
    cfg = default_config(thr=0.0)
    sd = make_state_dict(cfg, 0)
    outs = [ops.run_model(ops.make_model(cfg, sd, p), make_inputs(5000, (512, 512), 1)) for p in PRECISIONS]
    a = outs[0]
    assert len(a["i_ids"]) > 0
    for b in outs[1:]:
        assert torch.equal(a["i_ids"], b["i_ids"]) and torch.equal(a["j_ids"], b["j_ids"])
        assert (a["conf_matrix"] - b["conf_matrix"]).abs().max() < 2e-5
        assert (a["mconf"] - b["mconf"]).abs().max() < 2e-5
        assert (a["expec_f"] - b["expec_f"]).abs().max() < 1e-4


@pytest.mark.gpu
def test_fine_branch_overlap_is_transparent():
    """`opp_config.fpn_overlap`: the FPN fine branch of the backbone on a side HIP stream next to the coarse level (fork after
    layer3_outconv, join before opp_forward_coarse returns; backbone/resnet.py:149-157 vs OnePosePlusModel.py:131-167).  Same
    kernels in the same order within each branch: every output, the fine-stage ones that consume feat_f included, is
    bit-identical to the single-stream run, repeatedly and with two forwards alternating on one module."""
    from tests import hip_ops as ops
    name = "highconf_512x512_n3000"
    cfg, sd, data = H.highconf_setup(name)
    on = ops.make_model(cfg, sd, "bf16x3")
    off = ops.make_model(cfg, sd, "bf16x3").set_fpn_overlap(False).cuda()
    ref = ops.run_model(off, data)
    assert len(ref["mconf"]) > 1000
    for _ in range(3):
        got = ops.run_model(on, data)
        for k in ("conf_matrix", "i_ids", "j_ids", "mconf", "expec_f", "mkpts_query_f", "mkpts_query_c"):
            assert torch.equal(got[k], ref[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [True, False])
def test_skipping_the_unused_fine_map_changes_no_output(overlap):
    """Coarse-only matching (`fine_matching.enable = False`, OnePosePlusModel.py:169-176: the reference computes the fine
    map in its backbone and never reads it).  `set_skip_unused_fine_map(True)` hands `opp_forward_coarse` a NULL feat_f and
    the FPN fine branch is not launched: every entry of `data` equals the run that computes the dead map, bit for bit; with
    fine matching enabled the switch does nothing."""
    import copy
    from tests import hip_ops as ops
    name = "highconf_512x512_n3000"
    cfg, sd, data = H.highconf_setup(name)
    coarse_cfg = copy.deepcopy(cfg)
    coarse_cfg["fine_matching"]["enable"] = False
    full = ops.make_model(coarse_cfg, sd, "bf16x3").set_fpn_overlap(overlap).cuda()
    lean = ops.make_model(coarse_cfg, sd, "bf16x3").set_fpn_overlap(overlap).set_skip_unused_fine_map(True).cuda()
    ref = ops.run_model(full, data)
    assert len(ref["mconf"]) > 1000 and "expec_f" not in ref
    for _ in range(2):
        got = ops.run_model(lean, data)
        assert set(got) == set(ref)
        for k in ("conf_matrix", "i_ids", "j_ids", "mconf", "mkpts_query_c", "mkpts_query_f", "mkpts_3d_db"):
            assert torch.equal(got[k], ref[k]), k
    fine_on = ops.make_model(cfg, sd, "bf16x3").set_skip_unused_fine_map(True).cuda()
    a, b = ops.run_model(fine_on, data), ops.run_model(ops.make_model(cfg, sd, "bf16x3"), data)
    assert torch.equal(a["expec_f"], b["expec_f"]) and torch.equal(a["mkpts_query_f"], b["mkpts_query_f"])


@pytest.mark.gpu
@pytest.mark.parametrize("hw,n", [((512, 512), 5000), ((96, 136), 777)])
def test_object_prefix_is_bit_identical(hw, n):
    """The per-object transformer prefix (include/opp_hip.h `opp_object_prefix`: layer 0 on the 3D stream, its layer-1 projections and
    KV sums, evaluated once per resident object) changes no bit of any output: BASELINE size and a ragged one (L = 12 x 17 = 204 cells,
    N = 777: partial 64-token tiles / KV chunks in both streams), coarse-to-fine, against a module that redoes everything per image; the
    library reports no prefix for the other arithmetics (every layer is then evaluated per image)."""
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.config import default_config
    from onepose_plus_plus_amd.synthetic import make_inputs, make_state_dict
    cfg = default_config(thr=0.0)
    sd = make_state_dict(cfg, 3)
    d0 = {k: v.cuda() for k, v in make_inputs(n, hw, 21).items()}
    cached, plain = ops.make_model(cfg, sd), ops.make_model(cfg, sd)
    plain.cache_object_tokens = False
    outs = []
    for m in (cached, plain, cached):                # third run: the prefix is reused, not recomputed
        d = dict(d0)
        with torch.no_grad():
            m(d)
        torch.cuda.synchronize()
        outs.append(d)
    assert cached._rt["obj"][4] is not None, "the default arithmetic must have a prefix"
    for o in outs[1:]:
        for k in ("conf_matrix", "i_ids", "j_ids", "mconf", "expec_f", "mkpts_query_f"):
            assert torch.equal(outs[0][k], o[k]), k
    lib, ctx = ops.ctx_of(ops.make_model(cfg, sd, "fp32"))
    assert lib.opp_object_prefix_bytes(ctx, n) == 0


def _run_fine_variant(cfg, sd, data, precision, patch_max):
    from tests import hip_ops as ops
    m = ops.make_model(cfg, sd, precision).set_fine_patch_max_matches(patch_max)
    m.fine_patch_pixels_per_match = 0                # force the path under test whatever the match count
    d = {k: v.cuda() for k, v in data.items()}
    with torch.no_grad():
        m(d)
    torch.cuda.synchronize()
    return d


@pytest.mark.parametrize("precision", PRECISIONS)
def test_match_driven_fine_branch_is_bit_identical_on_the_highconf_case(precision):
    """The match-driven fine branch (include/opp_hip.h `opp_fine_patches`: layer1_outconv / layer1_outconv2 as VALID convolutions over a
    9x9 -> 7x7 -> 5x5 patch pyramid per match, resnet.py:154-157 + fine_preprocess.py:41-55) on the fixture with 1487 confident matches:
    per-match patches, the dense map completed from the kept x1 / x2_out, and the dense map inside the fused coarse call (the path the
    reference-generated goldens pin, test_e2e_high_confidence_vs_golden) give the same bits, in every arithmetic."""
    name = "highconf_512x512_n3000"
    cfg, sd, data = H.highconf_setup(name)
    fused = _run_fine_variant(cfg, sd, data, precision, 0)
    patches = _run_fine_variant(cfg, sd, data, precision, 1 << 20)
    kept = _run_fine_variant(cfg, sd, data, precision, 1)
    assert fused["expec_f"].shape[0] > 1000
    for other in (patches, kept):
        for k in ("conf_matrix", "i_ids", "j_ids", "mconf", "expec_f", "mkpts_query_f"):
            assert torch.equal(fused[k], other[k]), (precision, k)


@pytest.mark.parametrize("window", [3, 5, 7])
@pytest.mark.parametrize("hw", [(64, 96), (136, 104)])
def test_match_driven_fine_branch_at_the_image_border(hw, window):
    """border_rm = 0 and thr = 0: matches on the first / last coarse rows and columns, whose (W+4)^2 / (W+2)^2 patches and W^2 windows
    reach outside the image -- the out-of-image pixels must be the zero padding of the dense convolutions / of the window unfold
    (resnet.py:17-19, fine_preprocess.py:41-47).  Bit-identical to the dense path for window sizes 3 / 5 / 7."""
    from onepose_plus_plus_amd.config import default_config
    from onepose_plus_plus_amd.synthetic import make_inputs, make_state_dict
    cfg = default_config(thr=0.0)
    cfg["coarse_matching"]["border_rm"] = 0
    cfg["loftr_fine"]["window_size"] = window
    sd = make_state_dict(cfg, 11)
    n = (hw[0] // 8) * (hw[1] // 8) + 40              # more points than cells: every cell can be somebody's mutual nearest neighbour
    data = make_inputs(n, hw, 5)
    fused = _run_fine_variant(cfg, sd, data, None, 0)
    patches = _run_fine_variant(cfg, sd, data, None, 1 << 20)
    wc = hw[1] // 8
    jy, jx = fused["j_ids"] // wc, fused["j_ids"] % wc
    assert int(((jy == 0) | (jx == 0) | (jy == hw[0] // 8 - 1) | (jx == wc - 1)).sum()) > 0, "no border match: the case tests nothing"
    for k in ("i_ids", "j_ids", "mconf", "expec_f", "mkpts_query_f"):
        assert torch.equal(fused[k], patches[k]), k


def test_conv_tail_opt_in_meets_the_same_bar():
    """`model.set_conv_tail(True)` (include/opp_hip.h `opp_set_conv_tail`; off by default): the eight 196-channel convolutions as a
    192-column body on the 128 x 192 MFMA tile + their last 4 columns as fp32 FMA chains (csrc/conv_tail.hip: 1 x 1 / 3 x 3, stride 1 / 2,
    same-tensor and bilinear residuals, the VALID patch convolutions of the match-driven fine branch).  Reference-generated goldens at the
    bar of the default path: the high-confidence case end to end, a small full forward, and a train()-mode batch (raw weights, B = 2)."""
    from tests import hip_ops as ops
    name = "highconf_512x512_n3000"
    cfg, sd, data = H.highconf_setup(name)
    for patch_max in (0, 1 << 20):                   # dense fine map / per-match patches
        m = ops.make_model(cfg, sd).set_conv_tail(True).set_fine_patch_max_matches(patch_max)
        m.fine_patch_pixels_per_match = 0
        out = ops.run_model(m, data)
        H.assert_match_outputs(out, H.load_golden(name), where=name + " + conv tail")
    plain = ops.run_model(ops.make_model(cfg, sd), data)
    assert torch.equal(plain["i_ids"], out["i_ids"]) and not torch.equal(plain["conf_matrix"], out["conf_matrix"])   # (it is a different evaluation)
    name = "e2e_96x64_n77_thr0"
    cfg, sd, data = H.e2e_setup(name)
    H.assert_match_outputs(ops.run_model(ops.make_model(cfg, sd).set_conv_tail(True), data), H.load_golden(name), where=name + " + conv tail")
    name = "train_b2_128x128_n300"
    cfg, sd, data = H.train_setup(name)
    gold = H.load_golden(name)
    model = ops.make_model(cfg, sd).set_conv_tail(True)
    model.train()
    model.train_randint = H.RecordedRandint([gold["randint_%d" % i] for i in range(int(gold["n_randint"]))])
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    with torch.no_grad():
        model(d)
    H.assert_train_outputs(d, model.state_dict(), gold, tol_bn=1e-4, where=name + " + conv tail")
