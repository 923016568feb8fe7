"""Training-step callers: the CPU restatement (oracle/loss_oracle.py) against the reference-generated fixtures
(tests/golden/loss_*.npz, made by gen_golden.py::gen_loss from the reference's own Loss / fine_supervision), and the
module's index glue (`fine_supervision`, the fine-level term) which is plain torch and runs anywhere."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as LO
from tests import helpers as H
from tests.golden.cases import LOSS_CASES, LOSS_CONFIG

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name", list(LOSS_CASES))
def test_oracle_matches_reference_loss(name):
    gold = _gold(name)
    data, hp = H.loss_inputs(name)
    gt = LO.fine_supervision(data, hp["OnePosePlus"]["loftr_backbone"]["resolution"], hp["OnePosePlus"]["loftr_fine"]["window_size"])
    assert np.array_equal(gt.numpy(), gold["expec_f_gt"])
    data["expec_f_gt"] = gt
    data["conf_matrix"].requires_grad_(True)
    data["expec_f"].requires_grad_(True)
    out = LO.loss_forward(data, LOSS_CONFIG, training=True)
    assert abs(float(out["loss_c"].detach()) - float(gold["loss_c"])) <= 1e-6 * abs(float(gold["loss_c"]))
    assert abs(float(out["loss_f"].detach()) - float(gold["loss_f"])) <= 1e-6 * abs(float(gold["loss_f"])) + 1e-12
    assert abs(float(out["loss"].detach()) - float(gold["loss"])) <= 1e-6 * abs(float(gold["loss"]))
    out["loss"].backward()
    gc, ge = data["conf_matrix"].grad.numpy(), data["expec_f"].grad.numpy()
    assert np.abs(gc - gold["grad_conf"]).max() <= 1e-6 * np.abs(gold["grad_conf"]).max()
    assert np.abs(ge - gold["grad_expec"]).max() <= 1e-6 * max(np.abs(gold["grad_expec"]).max(), 1e-12)
    # the clamp blocks the gradient outside [1e-6, 1 - 1e-6] and passes it on the bounds
    conf = data["conf_matrix"].detach().numpy()
    assert (gold["grad_conf"][(conf < 1e-6) | (conf > np.float32(1 - 1e-6))] == 0).all()


@pytest.mark.parametrize("name", list(LOSS_CASES))
def test_module_fine_supervision_and_fine_term(name):
    """`onepose_plus_plus_amd.losses.fine_supervision` and the fine-level term of `Loss` (index-sized torch glue)."""
    from onepose_plus_plus_amd.losses import Loss, fine_supervision
    gold = _gold(name)
    data, hp = H.loss_inputs(name)
    fine_supervision(data, hp)
    assert np.array_equal(data["expec_f_gt"].numpy(), gold["expec_f_gt"])
    mod = Loss(dict(LOSS_CONFIG)).train()
    e = data["expec_f"].clone().requires_grad_(True)
    lf = mod.compute_fine_loss(e, data["expec_f_gt"])
    assert abs(float(lf.detach()) - float(gold["loss_f"])) <= 1e-6 * abs(float(gold["loss_f"])) + 1e-12
    (lf * LOSS_CONFIG["fine_weight"]).backward()
    assert np.abs(e.grad.numpy() - gold["grad_expec"]).max() <= 1e-6 * max(np.abs(gold["grad_expec"]).max(), 1e-12)
    if name.endswith("no_inside"):
        assert mod.eval().compute_fine_loss(data["expec_f"], data["expec_f_gt"]) is None      # losses.py:92-93


def test_loss_needs_the_device():
    """no CPU fallback: the focal loss is the HIP kernel or an error"""
    from onepose_plus_plus_amd.losses import Loss
    data, hp = H.loss_inputs("loss_b1_n77_nopos")
    with pytest.raises(RuntimeError):
        Loss(dict(LOSS_CONFIG)).compute_coarse_loss(data["conf_matrix"], data["conf_matrix_gt"])


def test_unsupported_loss_types_raise_like_the_reference():
    from onepose_plus_plus_amd.losses import Loss
    cfg = dict(LOSS_CONFIG, coarse_type="cross_entropy")
    with pytest.raises(NotImplementedError):
        Loss(cfg).compute_coarse_loss(torch.rand(1, 2, 2), torch.zeros(1, 2, 2, dtype=torch.int16))
    cfg = dict(LOSS_CONFIG, fine_type="l2")
    with pytest.raises(NotImplementedError):
        Loss(cfg).compute_fine_loss(torch.rand(3, 3), torch.rand(3, 2))
