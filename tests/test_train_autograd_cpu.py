"""CPU: the differentiable restatement used for the backward of the train()-mode forward
(onepose_plus_plus_amd/train_autograd.py) against gradients produced by the reference's own autograd (stored in the
train fixtures by tests/golden/gen_golden.py): forward values and every parameter gradient."""
import numpy as np
import pytest
import torch

from onepose_plus_plus_amd import train_autograd as TA
from onepose_plus_plus_amd.model import _sine_table
from tests import helpers as H
from tests.golden.cases import TRAIN_CASES


@pytest.mark.parametrize("name", ["train_b2_128x128_n300_sub", "train_b4_64x96_n150"])
def test_differentiable_forward_matches_reference_gradients(name):
    cfg, sd, data = H.train_setup(name)
    gold = H.load_golden(name)
    p = {k: v.clone().requires_grad_(not k.endswith(("running_mean", "running_var"))) for k, v in sd.items() if v.is_floating_point()}
    inputs = {k: data[k] for k in ("query_image", "keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    inputs["query_image_mask"] = None
    matches = tuple(torch.from_numpy(gold[k]) for k in ("b_ids", "i_ids", "j_ids"))
    pe = _sine_table(256, cfg["positional_encoding"]["pos_emb_shape"])
    conf, expec = TA.differentiable_forward(p, cfg, inputs, matches, pe)
    assert np.abs(conf.detach().numpy() - gold["conf_matrix"]).max() < 2e-5
    assert np.abs(expec.detach().numpy() - gold["expec_f"])[:, :2].max() < 5e-5
    wc, we = H.train_loss_weights(conf.shape, expec.shape)
    names = [str(n) for n in gold["grad_names"]]
    grads = torch.autograd.grad((conf * wc).sum() + (expec * we).sum(), [p[n] for n in names])
    H.assert_train_grads(dict(zip(names, grads)), gold, rel=5e-4, where=name)


def test_autograd_function_is_wired_into_training_mode():
    """train() + gradients enabled routes through TrainForward (no GPU here: the HIP forward itself is covered by
    tests/test_e2e_gpu.py::test_training_step_gradients)."""
    from onepose_plus_plus_amd import OnePosePlus_model, default_config
    m = OnePosePlus_model(default_config()).train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m({"query_image": torch.zeros(2, 1, 64, 64)})
    assert hasattr(TA.TrainForward, "apply")
