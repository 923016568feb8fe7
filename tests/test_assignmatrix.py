"""Device-side ground-truth builder (onepose_plus_plus_amd/assignmatrix.py) against fixtures produced by the reference's
own `OnePosePlusDataset.build_assignmatrix` (tests/golden/gen_assignmatrix_golden.py) and against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import assignmatrix_oracle as AO
from tests import helpers as H
from tests.golden.cases import ASSIGN_CASES, make_assign_inputs


def _dense_from_golden(gold):
    N, L = [int(v) for v in gold["conf_shape"]]
    conf = torch.zeros(N, L, dtype=torch.int16)
    floc = torch.full((N, L, 2), -50.0)
    pos, fpos = torch.from_numpy(gold["conf_pos"]), torch.from_numpy(gold["floc_pos"])
    conf[pos[:, 0], pos[:, 1]] = 1
    floc[fpos[:, 0], fpos[:, 1]] = torch.from_numpy(gold["floc_val"])
    return conf, floc


@pytest.mark.parametrize("name", list(ASSIGN_CASES))
def test_oracle_matches_reference_fixture(name):
    kc, kf, am, meta = make_assign_inputs(ASSIGN_CASES[name])
    conf, floc = AO.build_assignmatrix(kc, kf, am, meta["shape3d"], meta["L"], meta["w_c"], meta["scale"], meta["coarse_scale"])
    gconf, gfloc = _dense_from_golden(H.load_golden(name))
    assert torch.equal(conf, gconf) and torch.equal(floc, gfloc)
    # the case does exercise what it claims: dropped padded indices, duplicate cells whose candidates differ
    a = am.long()
    assert int((a[1] >= meta["shape3d"]).sum()) > 0 and int(conf.sum()) < int((a[1] < meta["shape3d"]).sum())


def test_reference_method_live():
    """when /root/reference is mounted: the reference's own method (compiled from its source) reproduces the committed fixtures"""
    import os
    if not os.path.isdir("/root/reference/src/datasets"):
        pytest.skip("reference tree not mounted")
    import types
    from tests.golden.gen_assignmatrix_golden import reference_method
    fn = reference_method()
    for name, case in ASSIGN_CASES.items():
        kc, kf, am, meta = make_assign_inputs(case)
        self = types.SimpleNamespace(shape3d=meta["shape3d"], n_query_coarse_grid=meta["L"], w_c=meta["w_c"], query_img_scale=meta["scale"],
                                     coarse_scale=meta["coarse_scale"])
        conf, floc = fn(self, kc, kf, am)
        gconf, gfloc = _dense_from_golden(H.load_golden(name))
        assert torch.equal(conf, gconf) and torch.equal(floc, gfloc), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ASSIGN_CASES))
def test_hip_builder_matches_reference_fixture(name):
    from onepose_plus_plus_amd.assignmatrix import build_assignmatrix
    kc, kf, am, meta = make_assign_inputs(ASSIGN_CASES[name])
    conf, floc = build_assignmatrix(kc, kf, am, meta["shape3d"], meta["L"], meta["w_c"], meta["scale"], meta["coarse_scale"])
    gconf, gfloc = _dense_from_golden(H.load_golden(name))
    assert conf.dtype == torch.int16 and floc.dtype == torch.float32
    assert torch.equal(conf.cpu(), gconf) and torch.equal(floc.cpu(), gfloc)        # bit-exact: integer / copy work
    # two runs: identical (duplicates resolve by pair order, not by scheduling)
    conf2, floc2 = build_assignmatrix(kc, kf, am, meta["shape3d"], meta["L"], meta["w_c"], meta["scale"], meta["coarse_scale"])
    assert torch.equal(conf, conf2) and torch.equal(floc, floc2)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_hip_builder_matches_oracle_on_random_cases(seed):
    from onepose_plus_plus_amd.assignmatrix import build_assignmatrix
    g = torch.Generator().manual_seed(100 + seed)
    N = int(torch.randint(50, 900, (1,), generator=g))
    hc, wc = int(torch.randint(3, 20, (1,), generator=g)), int(torch.randint(3, 20, (1,), generator=g))
    case = (N, hc, wc, int(torch.randint(20, 400, (1,), generator=g)), int(torch.randint(1, 1500, (1,), generator=g)),
            (float(0.5 + torch.rand(1, generator=g)), float(0.5 + torch.rand(1, generator=g))), 1000 + seed)
    kc, kf, am, meta = make_assign_inputs(case)
    ref = AO.build_assignmatrix(kc, kf, am, meta["shape3d"], meta["L"], meta["w_c"], meta["scale"], meta["coarse_scale"])
    got = build_assignmatrix(kc, kf, am, meta["shape3d"], meta["L"], meta["w_c"], meta["scale"], meta["coarse_scale"])
    assert torch.equal(got[0].cpu(), ref[0]) and torch.equal(got[1].cpu(), ref[1])


@pytest.mark.gpu
def test_hip_builder_empty_and_out_of_range():
    from onepose_plus_plus_amd.assignmatrix import build_assignmatrix
    kc = torch.tensor([[4.0, 4.0], [100.0, 60.0]])
    conf, floc = build_assignmatrix(kc, kc, torch.zeros(2, 0), 10, 8 * 8, 8, torch.ones(2))       # no pairs at all
    assert int(conf.sum()) == 0 and bool((floc == -50).all())
    # (x = 12.5 -> 12 (half to even), y = 7.5 -> 8): j = 8 * 8 + 12 = 76 > L: dropped by the reference's mask, no error
    conf, floc = build_assignmatrix(kc, kc, torch.tensor([[1.0], [3.0]]), 10, 64, 8, torch.ones(2))
    assert int(conf.sum()) == 0
    # j == L exactly passes `j_ids > L` upstream and then raises IndexError: so does the builder (x = 0, y = 8 -> j = 64)
    with pytest.raises(IndexError):
        build_assignmatrix(torch.tensor([[0.0, 64.0]]), torch.zeros(1, 2), torch.tensor([[0.0], [3.0]]), 10, 64, 8, torch.ones(2))


@pytest.mark.gpu
def test_training_step_on_device_built_ground_truth():
    """The consumer side: `conf_matrix_gt` / `fine_location_matrix_gt` of a batch formed on the device from the assignment pairs feed
    `model(batch)` (training branch of get_coarse_match pads with them), `fine_supervision` and `Loss` exactly like the loader's host-built
    matrices (lightning_model:54-60) -- the same loss value as with the oracle-built matrices uploaded from the host."""
    from onepose_plus_plus_amd.assignmatrix import build_assignmatrix
    from onepose_plus_plus_amd.config import default_config
    from onepose_plus_plus_amd.losses import Loss, fine_supervision
    from onepose_plus_plus_amd.synthetic import make_inputs, make_state_dict
    from tests import hip_ops as ops
    from tests.golden.cases import LOSS_CONFIG
    hw, N, B = (64, 96), 150, 2
    hc, wc = hw[0] // 8, hw[1] // 8
    cfg = default_config(thr=0.2)
    cfg["coarse_matching"]["train"] = {"train_padding": True, "train_coarse_percent": 0.3, "train_pad_num_gt_min": 5}
    sd = make_state_dict(cfg, 2)
    parts = [make_inputs(N, hw, 60 + b) for b in range(B)]
    base = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    dev_gt, host_gt = [], []
    for b in range(B):
        kc, kf, am, meta = make_assign_inputs((N, hc, wc, 90, 120, (1.0, 1.0), 200 + b))
        dev_gt.append(build_assignmatrix(kc, kf, am, N, hc * wc, wc, meta["scale"], meta["coarse_scale"]))
        host_gt.append(AO.build_assignmatrix(kc, kf, am, N, hc * wc, wc, meta["scale"], meta["coarse_scale"]))
    hparams = {"OnePosePlus": cfg, "loss": dict(LOSS_CONFIG)}
    losses = []
    for gts, to_dev in ((dev_gt, False), (host_gt, True)):
        model = ops.make_model(cfg, sd)
        model.train()
        model.train_randint = lambda high, size, device=None, **kw: (torch.arange(size[0]) * 7 % high).to(device)
        d = {k: v.cuda() for k, v in base.items()}
        d["conf_matrix_gt"] = torch.stack([g[0].cuda() if to_dev else g[0] for g in gts])
        d["fine_location_matrix_gt"] = torch.stack([g[1].cuda() if to_dev else g[1] for g in gts])
        model(d)
        fine_supervision(d, hparams)
        Loss(hparams["loss"]).train()(d)
        d["loss"].backward()
        assert torch.isfinite(d["loss"]) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        losses.append(float(d["loss"].detach()))
    assert losses[0] == losses[1]
