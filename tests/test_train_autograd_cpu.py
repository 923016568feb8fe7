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
    from tests.torch_graph_ref import differentiable_forward
    conf, expec = differentiable_forward(p, cfg, inputs, matches, pe)
    assert np.abs(conf.detach().numpy() - gold["conf_matrix"]).max() < 2e-5
    assert np.abs(expec.detach().numpy() - gold["expec_f"])[:, :2].max() < 5e-5
    wc, we = H.train_loss_weights(conf.shape, expec.shape)
    names = [str(n) for n in gold["grad_names"]]
    grads = torch.autograd.grad((conf * wc).sum() + (expec * we).sum(), [p[n] for n in names])
    H.assert_train_grads(dict(zip(names, grads)), gold, rel=5e-4, where=name)


def test_autograd_function_is_wired_into_training_mode():
    """train() + gradients enabled builds the graph of HIP nodes (no GPU here: the graph itself is covered by
    tests/test_e2e_gpu.py::test_training_step_gradients and tests/test_train_bwd_gpu.py); CPU tensors are refused, and the
    re-evaluating TrainForward of earlier rounds is gone."""
    from onepose_plus_plus_amd import OnePosePlus_model, default_config
    m = OnePosePlus_model(default_config()).train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m({"query_image": torch.zeros(2, 1, 64, 64)})
    for node in ("HipBackbone", "HipLinear", "HipLinearAttention", "HipLayerNorm", "HipCoarseMatch", "HipFineGather"):
        assert hasattr(getattr(TA, node), "apply")
    assert not hasattr(TA, "TrainForward")


def test_graph_helpers_fall_back_to_torch_ops_on_cpu_tensors_only_in_the_restatement():
    """`hp = None` (the CPU restatement) never touches the HIP library; with `hp` set, CPU tensors still take the torch branch of
    each helper (the device graph itself only ever passes device tensors)."""
    x = torch.randn(2, 5, 64)
    w = torch.randn(32, 64)
    assert torch.equal(TA._linear(x, w), torch.nn.functional.linear(x, w))
    assert torch.equal(TA._linear(x, w, 2), torch.nn.functional.linear(x, w))
    g, b = torch.rand(64), torch.rand(64)
    assert torch.equal(TA._layer_norm(x, g, b, 2), torch.nn.functional.layer_norm(x, (64,), g, b, 1e-5))


def test_dual_softmax_formula_matches_autograd():
    """`DualSoftmax` (the hand-written backward formula; on CPU tensors evaluated with torch ops, on the device by
    opp_dual_softmax_backward) against torch.autograd of softmax(S, 1) * softmax(S, 2) in fp64, masked cells included."""
    import torch.nn.functional as F
    from onepose_plus_plus_amd.train_autograd import DualSoftmax
    g = torch.Generator().manual_seed(0)
    for shape in ((2, 7, 5), (1, 30, 12), (3, 4, 25)):
        S = (torch.randn(shape, generator=g, dtype=torch.float64) * 3).requires_grad_(True)
        go = torch.randn(shape, generator=g, dtype=torch.float64)
        S.data[0, :, :2] -= 1e9
        c1 = F.softmax(S, 1) * F.softmax(S, 2)
        (c1 * go).sum().backward()
        g1 = S.grad.clone()
        S.grad = None
        c2 = DualSoftmax.apply(S)
        (c2 * go).sum().backward()
        assert (c1 - c2).abs().max() <= 1e-14
        assert (g1 - S.grad).abs().max() <= 1e-14
        assert (S.grad[0, :, :2] == 0).all()
