"""One-call swap of the MI355X-native module into a checkout of the reference.

    import onepose_plus_plus_amd.dropin as dropin
    dropin.install()                      # before (or after) the reference's own imports
    from src.inference.inference_OnePosePlus import build_model      # now builds the HIP module

`install()` makes both import paths the reference uses resolve to `onepose_plus_plus_amd.OnePosePlus_model`:
  * `from src.models.OnePosePlus.OnePosePlusModel import OnePosePlus_model`
        (src/inference/inference_OnePosePlus.py:11, demo.py:17)
  * `from src.models.OnePosePlus import OnePosePlus_model`
        (src/lightning_model/OnePosePlus_lightning_model.py:8 via src/models/OnePosePlus/__init__.py:1)
and, with `pnp=True`, routes `src.utils.metric_utils.ransac_PnP` (called by `compute_query_pose_errors`,
metric_utils.py:262-270, i.e. by every caller of the matcher: inference worker, demo, validation step) to the
on-device `onepose_plus_plus_amd.pose.ransac_PnP`, which keeps the reference's signature and return tuple; with
`loss=True` the training step's `Loss` (src/lightning_model/losses.py) and `fine_supervision`
(src/models/OnePosePlus/utils/fine_supervision.py) resolve to `onepose_plus_plus_amd.losses`.

If the reference modules were imported already their attributes are patched in place; otherwise light-weight module
objects are registered under those names, so the reference's own model files (and their kornia / timm imports) are
never loaded.  Nothing here touches the matching arithmetic: it is import plumbing only.
"""
import importlib
import sys
import types

from .model import OnePosePlus_model

_MODEL_PATHS = ("src.models.OnePosePlus.OnePosePlusModel", "src.models.OnePosePlus")


def _real_dirs(name):
    """directories of package `name` found on sys.path (empty on a box without the reference checkout)"""
    import os
    rel = os.path.join(*name.split("."))
    return [os.path.join(p, rel) for p in sys.path if p and os.path.isdir(os.path.join(p, rel))]


def _ensure_package(name):
    """registers empty parent packages (src, src.models) only when the real ones cannot be imported"""
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except Exception:
        mod = types.ModuleType(name)
        mod.__path__ = _real_dirs(name)       # sub-modules that do import cleanly still come from the checkout
        sys.modules[name] = mod
        parent, _, leaf = name.rpartition(".")
        if parent:
            setattr(_ensure_package(parent), leaf, mod)
        return mod


def _patch_attr(path, name, obj, done):
    """sets `path.name = obj` on the imported module, or on a light-weight stand-in registered under that path when the
    reference file was not imported yet (so that a later `from path import name` resolves to `obj`)"""
    mod = sys.modules.get(path)
    if mod is None:
        parent, _, leaf = path.rpartition(".")
        pkg = _ensure_package(parent)
        mod = types.ModuleType(path)
        sys.modules[path] = mod
        setattr(pkg, leaf, mod)
    setattr(mod, name, obj)
    done[path] = name


def install(pnp=True, loss=True):
    """-> dict of what was patched (for logging / tests)"""
    done = {}
    for path in _MODEL_PATHS:
        mod = sys.modules.get(path)
        if mod is None:
            parent, _, leaf = path.rpartition(".")
            pkg = _ensure_package(parent)
            mod = types.ModuleType(path)
            if path == "src.models.OnePosePlus":
                # a stand-in for the package (its real __init__ would load the reference model and kornia / timm); the
                # reference's other sub-packages (optimizers, utils, ...) stay importable from the checkout, if there is one
                mod.__path__ = _real_dirs(path)
            sys.modules[path] = mod
            setattr(pkg, leaf, mod)
        mod.OnePosePlus_model = OnePosePlus_model
        done[path] = "OnePosePlus_model"
    # keep the sub-module reachable as an attribute of the package (import machinery convention)
    setattr(sys.modules["src.models.OnePosePlus"], "OnePosePlusModel", sys.modules["src.models.OnePosePlus.OnePosePlusModel"])
    if pnp:
        from .pose import ransac_PnP
        try:
            mu = importlib.import_module("src.utils.metric_utils")
            mu.ransac_PnP = ransac_PnP
            done["src.utils.metric_utils"] = "ransac_PnP"
        except Exception as e:      # the reference (or one of its dependencies) is not importable here
            done["src.utils.metric_utils"] = "not patched: %s" % (e,)
    if loss:
        # training step (src/lightning_model/OnePosePlus_lightning_model.py:9,14): `fine_supervision(batch, hparams)` and
        # `Loss(hparams["loss"])` -> the module's own (the focal loss over the B x N x L matrix runs in libopp_hip.so)
        from .losses import Loss, fine_supervision
        _patch_attr("src.lightning_model.losses", "Loss", Loss, done)
        _patch_attr("src.models.OnePosePlus.utils.fine_supervision", "fine_supervision", fine_supervision, done)
        lm = sys.modules.get("src.lightning_model.OnePosePlus_lightning_model")
        if lm is not None:          # imported before install(): its module-level names were bound already
            lm.Loss, lm.fine_supervision = Loss, fine_supervision
            done["src.lightning_model.OnePosePlus_lightning_model"] = "Loss, fine_supervision"
    return done
