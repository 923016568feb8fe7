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
on-device `onepose_plus_plus_amd.pose.ransac_PnP`, which keeps the reference's signature and return tuple.

If the reference modules were imported already their attributes are patched in place; otherwise light-weight module
objects are registered under those names, so the reference's own model files (and their kornia / timm imports) are
never loaded.  Nothing here touches the matching arithmetic: it is import plumbing only.
"""
import importlib
import sys
import types

from .model import OnePosePlus_model

_MODEL_PATHS = ("src.models.OnePosePlus.OnePosePlusModel", "src.models.OnePosePlus")


def _ensure_package(name):
    """registers empty parent packages (src, src.models) only when the real ones cannot be imported"""
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except Exception:
        mod = types.ModuleType(name)
        mod.__path__ = []
        sys.modules[name] = mod
        parent, _, leaf = name.rpartition(".")
        if parent:
            setattr(_ensure_package(parent), leaf, mod)
        return mod


def install(pnp=True):
    """-> dict of what was patched (for logging / tests)"""
    done = {}
    for path in _MODEL_PATHS:
        mod = sys.modules.get(path)
        if mod is None:
            parent, _, leaf = path.rpartition(".")
            pkg = _ensure_package(parent)
            mod = types.ModuleType(path)
            if path == "src.models.OnePosePlus":
                mod.__path__ = []
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
    return done
