"""CPU (build container, /root/reference mounted read-only): the reference's OWN entry points accept the HIP
module through `onepose_plus_plus_amd.dropin.install()` -- `build_model` (src/inference/inference_OnePosePlus.py:
28-38: constructor, checkpoint with `matcher.`-prefixed keys, strict load, eval) and the worker module
(src/inference/inference_OnePosePlus_worker.py), with stubs for the packages that are not installed here (ray, cv2,
loguru, ...).  The forward itself needs a GPU: the worker's data flow is exercised on the device by
tests/test_e2e_gpu.py::test_worker_flow_recovers_pose."""
import os
import sys
import types

import pytest
import torch

from oracle.refload import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted")


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


@pytest.fixture()
def reference_env():
    saved = dict(sys.modules)
    saved_path = list(sys.path)

    class _Logger:
        def __getattr__(self, k):
            return lambda *a, **kw: None

    def _remote(*a, **kw):
        def deco(f):
            f.remote = f
            return f
        return deco if not (len(a) == 1 and callable(a[0])) else a[0]

    _stub("loguru", logger=_Logger())
    _stub("ray", remote=_remote, init=lambda *a, **k: None, get=lambda x: x, put=lambda x: x)
    _stub("ray.actor", ActorHandle=object)
    _stub("cv2")
    _stub("natsort", natsorted=sorted)
    _stub("h5py")
    _stub("pycolmap")
    _stub("open3d")
    _stub("kornia")
    _stub("pytorch_lightning")
    _stub("plyfile", PlyData=object, PlyElement=object)
    sys.path.insert(0, "/root/reference")
    yield
    sys.path[:] = saved_path
    for k in list(sys.modules):
        if k not in saved and not k.startswith("onepose_plus_plus_amd"):   # keep this package's modules (one identity)
            del sys.modules[k]


def test_reference_build_model_constructs_the_hip_module(reference_env, tmp_path):
    from onepose_plus_plus_amd import OnePosePlus_model, default_config, dropin
    from onepose_plus_plus_amd.synthetic import make_state_dict
    done = dropin.install(pnp=False)
    assert set(done) >= {"src.models.OnePosePlus.OnePosePlusModel", "src.models.OnePosePlus"}
    from src.models.OnePosePlus.OnePosePlusModel import OnePosePlus_model as A     # both reference import paths
    from src.models.OnePosePlus import OnePosePlus_model as B
    assert A is OnePosePlus_model and B is OnePosePlus_model
    import src.inference.inference_OnePosePlus as I          # the reference's own file, unmodified
    assert I.OnePosePlus_model is OnePosePlus_model
    cfg = default_config()
    sd = make_state_dict(cfg, 5)
    ckpt = tmp_path / "ckpt.ckpt"
    torch.save({"state_dict": {"matcher." + k: v for k, v in sd.items()}}, ckpt)   # PL_OnePosePlus checkpoint layout
    model = I.build_model(cfg, str(ckpt))
    assert isinstance(model, OnePosePlus_model) and not model.training
    got = model.state_dict()
    assert set(got) == set(sd)
    for k in ("backbone.layer2.0.conv1.weight", "loftr_coarse.layers.3.mlp.0.weight", "backbone.bn1.running_var"):
        assert torch.equal(got[k], sd[k]), k
    import src.inference.inference_OnePosePlus_worker as W       # extract_matches / worker loop: import-clean
    assert callable(W.extract_matches) and callable(W.inference_onepose_plus_worker)
    # pose step: compute_query_pose_errors (the function extract_matches calls right after the forward) now resolves
    # `ransac_PnP` to the on-device solver, which keeps the reference's signature
    import inspect
    import src.utils.metric_utils as MU
    ref_params = list(inspect.signature(MU.ransac_PnP).parameters)
    done = dropin.install(pnp=True)
    from onepose_plus_plus_amd import pose
    assert done["src.utils.metric_utils"] == "ransac_PnP" and MU.ransac_PnP is pose.ransac_PnP
    assert list(inspect.signature(pose.ransac_PnP).parameters)[:len(ref_params)] == ref_params
    assert W.compute_query_pose_errors is MU.compute_query_pose_errors
    fn = getattr(MU.compute_query_pose_errors, "__wrapped__", MU.compute_query_pose_errors)   # under @torch.no_grad()
    assert fn.__globals__["ransac_PnP"] is pose.ransac_PnP


def test_reference_lightning_module_builds_on_the_hip_matcher_and_loss(reference_env):
    """The reference's own training wrapper (src/lightning_model/OnePosePlus_lightning_model.py, unmodified; a stand-in
    for pytorch_lightning.LightningModule, which is not installed here): after `dropin.install()` its constructor builds
    the HIP matcher and the HIP-backed Loss, and `training_step` calls this package's `fine_supervision`."""
    from onepose_plus_plus_amd import OnePosePlus_model, default_config, dropin, losses

    class _LightningModule(torch.nn.Module):
        def save_hyperparameters(self):
            pass                                    # the test sets .hparams itself (Lightning collects the ctor kwargs)

    hp = {"OnePosePlus": default_config(), "trainer": {"n_val_pairs_to_plot": 100, "world_size": 1}, "pretrained_ckpt": None,
          "loss": {"coarse_type": "focal", "coarse_weight": 1.0, "fine_type": "l2_with_std", "fine_weight": 0.81,
                   "focal_alpha": 0.5, "focal_gamma": 2.0, "pos_weight": 1.0, "neg_weight": 1.0, "fine_correct_thr": 1.0}}
    _LightningModule.hparams = hp
    sys.modules["pytorch_lightning"].LightningModule = _LightningModule
    _stub("matplotlib")
    _stub("matplotlib.pyplot")
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    import src.utils                                                        # the reference's package (empty __init__)
    _stub("src.utils.plot_utils", draw_reprojection_pair=None)              # plotting (matplotlib colour maps at import)
    done = dropin.install(pnp=True, loss=True)
    assert done["src.lightning_model.losses"] == "Loss"
    assert done["src.models.OnePosePlus.utils.fine_supervision"] == "fine_supervision"
    try:
        import src.lightning_model.OnePosePlus_lightning_model as LM       # the reference's own file
    except ImportError as e:                                               # a plotting / comm helper needs more stubs
        pytest.skip("reference training wrapper not importable here: %s" % e)
    assert LM.Loss is losses.Loss and LM.fine_supervision is losses.fine_supervision
    assert LM.OnePosePlus_model is OnePosePlus_model
    pl_model = LM.PL_OnePosePlus()
    assert isinstance(pl_model.matcher, OnePosePlus_model) and isinstance(pl_model.loss, losses.Loss)
    assert pl_model.loss.c_pos_w == 1.0 and pl_model.loss.fine_type == "l2_with_std"
    # checkpoint keys of the wrapper = `matcher.` + the module's keys (what build_model strips, inference_OnePosePlus.py:32-36)
    keys = set(pl_model.state_dict())
    assert {"matcher." + k for k in pl_model.matcher.state_dict()} == keys


def test_install_without_reference_modules_registers_both_paths():
    """On a box without the reference checkout the aliases are still importable (what the GPU tests use)."""
    from onepose_plus_plus_amd import OnePosePlus_model, dropin
    saved = {k: v for k, v in sys.modules.items() if k == "src" or k.startswith("src.")}
    saved_path = list(sys.path)
    try:
        for k in saved:
            sys.modules.pop(k, None)
        path = [p for p in sys.path if p != "/root/reference"]
        old, sys.path[:] = list(sys.path), path
        dropin.install(pnp=False)
        from src.models.OnePosePlus.OnePosePlusModel import OnePosePlus_model as A
        from src.models.OnePosePlus import OnePosePlus_model as B
        assert A is OnePosePlus_model and B is OnePosePlus_model
        sys.path[:] = old
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:     # every stand-in install() registered
            del sys.modules[k]
        sys.modules.update(saved)
