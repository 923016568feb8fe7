"""Import the upstream reference model (read-only, /root/reference) with stubs for the
packages that are absent in this image (loguru, timm, kornia, pytorch_lightning).

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/gen_golden.py (to produce the committed
golden vectors) and by CPU tests that cross-check the oracle against the live reference
when /root/reference is present.  Nothing on the product path may import this.

The kornia functions are restated from the kornia==0.4.1 semantics the reference pins
(requirements.txt:13), used at src/models/OnePosePlus/utils/fine_matching.py:86-87.
"""
import contextlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("OPP_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models", "OnePosePlus"))


def _install_stubs():
    import torch

    if "loguru" not in sys.modules:
        m = types.ModuleType("loguru")

        class _Logger:
            def __getattr__(self, name):
                return lambda *a, **k: None

        m.logger = _Logger()
        sys.modules["loguru"] = m

    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        registry = types.ModuleType("timm.models.registry")
        registry.register_model = lambda f: f
        timm.models = models
        models.registry = registry
        sys.modules["timm"] = timm
        sys.modules["timm.models"] = models
        sys.modules["timm.models.registry"] = registry

    if "kornia" not in sys.modules:
        kornia = types.ModuleType("kornia")
        geometry = types.ModuleType("kornia.geometry")
        subpix = types.ModuleType("kornia.geometry.subpix")
        dsnt = types.ModuleType("kornia.geometry.subpix.dsnt")
        utils = types.ModuleType("kornia.utils")
        grid = types.ModuleType("kornia.utils.grid")

        def create_meshgrid(height, width, normalized_coordinates=True, device=None):
            # kornia 0.4.1: linspace(0, n-1, n); normalized -> (x/(n-1) - 0.5) * 2 ; stacked (x, y)
            xs = torch.linspace(0, width - 1, width, device=device, dtype=torch.float32)
            ys = torch.linspace(0, height - 1, height, device=device, dtype=torch.float32)
            if normalized_coordinates:
                xs = (xs / (width - 1) - 0.5) * 2
                ys = (ys / (height - 1) - 0.5) * 2
            base = torch.stack(torch.meshgrid([xs, ys], indexing="ij")).transpose(1, 2)  # 2xHxW
            return base.unsqueeze(0).permute(0, 2, 3, 1)  # 1xHxWx2

        def spatial_expectation2d(inp, normalized_coordinates=True):
            b, c, h, w = inp.shape
            g = create_meshgrid(h, w, normalized_coordinates, inp.device).to(inp.dtype)
            pos_x = g[..., 0].reshape(-1)
            pos_y = g[..., 1].reshape(-1)
            flat = inp.view(b, c, -1)
            ex = torch.sum(pos_x * flat, -1, keepdim=True)
            ey = torch.sum(pos_y * flat, -1, keepdim=True)
            return torch.cat([ex, ey], -1).view(b, c, 2)

        dsnt.spatial_expectation2d = spatial_expectation2d
        grid.create_meshgrid = create_meshgrid
        kornia.geometry = geometry
        geometry.subpix = subpix
        subpix.dsnt = dsnt
        kornia.utils = utils
        utils.grid = grid
        for name, mod in [("kornia", kornia), ("kornia.geometry", geometry),
                          ("kornia.geometry.subpix", subpix),
                          ("kornia.geometry.subpix.dsnt", dsnt),
                          ("kornia.utils", utils), ("kornia.utils.grid", grid)]:
            sys.modules[name] = mod

    # src.utils.profiler pulls in pytorch_lightning; provide only the pass-through profiler
    if "src.utils.profiler" not in sys.modules:
        prof = types.ModuleType("src.utils.profiler")

        class PassThroughProfiler:
            @contextlib.contextmanager
            def record_function(self, name):
                yield

            profile = record_function

        prof.PassThroughProfiler = PassThroughProfiler
        sys.modules["src.utils.profiler"] = prof


def load_reference_model_class():
    """Returns the upstream OnePosePlus_model class (OnePosePlusModel.py:25)."""
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # make `src` and `src.utils` importable as namespace packages without running their __init__
    from src.models.OnePosePlus.OnePosePlusModel import OnePosePlus_model
    return OnePosePlus_model


def load_reference_modules():
    """Stage-level upstream modules for stage-wise golden vectors."""
    load_reference_model_class()
    from src.models.OnePosePlus.utils.coarse_matching import CoarseMatching
    from src.models.OnePosePlus.utils.fine_matching import FineMatching
    from src.models.OnePosePlus.loftr_module import LocalFeatureTransformer, FinePreprocess
    from src.utils.profiler import PassThroughProfiler
    return dict(CoarseMatching=CoarseMatching, FineMatching=FineMatching,
                LocalFeatureTransformer=LocalFeatureTransformer,
                FinePreprocess=FinePreprocess, PassThroughProfiler=PassThroughProfiler)
