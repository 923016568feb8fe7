"""CPU: the ingest oracle (restated cv2.resize 8-bit INTER_LINEAR + /255; parity unpinned, see its header),
the host-side resize rule, and the object-bank reader for the reference's .npz format."""
import os
import numpy as np
import pytest
import torch

from oracle import ingest_oracle as IO
from onepose_plus_plus_amd import ingest as PI
from onepose_plus_plus_amd.bank import ObjectBank


def test_resize_hand_computed_and_identities():
    a = np.array([[0, 255]], dtype=np.uint8)
    # 2 -> 4 columns: taps (0,0) (.25) (.75) (1,0) in 11-bit fixed point: 0, 64, 191, 255
    assert IO.resize_u8_linear(a, 4, 1).tolist() == [[0, 64, 191, 255]]
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    assert np.array_equal(IO.resize_u8_linear(img, 64, 48), img)                       # same size = copy
    dec = IO.resize_u8_linear(img, 32, 24).astype(np.int64)                             # exact 2x2 -> INTER_AREA fast path
    s = img.astype(np.int64)
    assert np.array_equal(dec, (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2)
    const = np.full((17, 23), 201, dtype=np.uint8)
    assert (IO.resize_u8_linear(const, 40, 31) == 201).all()                            # partition of unity
    ramp = np.tile(np.arange(0, 256, dtype=np.uint8), (4, 1))
    up = IO.resize_u8_linear(ramp, 700, 9).astype(np.int64)
    assert (np.diff(up, axis=1) >= 0).all() and up.min() == 0 and up.max() == 255       # monotone, range kept


@pytest.mark.parametrize("w,h,resize,df", [(640, 480, [512, 512], 8), (640, 480, [512], 8), (1440, 1920, [512], 8),
                                             (640, 480, [-1], 8), (333, 517, None, 8), (333, 517, None, None),
                                             (640, 480, [500, 300], 16)])
def test_process_resize_matches_oracle(w, h, resize, df):
    assert PI.process_resize(w, h, resize, df) == IO.process_resize(w, h, resize, df)


def test_read_grayscale_oracle_shapes():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    out, scales, mask = IO.read_grayscale_u8(img, resize=[512, 512], df=8)
    assert out.shape == (1, 512, 512) and out.dtype == np.float32 and mask is None
    assert np.allclose(scales, [480 / 512, 640 / 512])
    assert out.min() >= 0 and out.max() <= 1
    out, scales, mask = IO.read_grayscale_u8(img, resize=[256], df=8, pad_to=256)
    assert out.shape == (1, 256, 256) and mask.sum() == 192 * 256 and (out[0, 192:] == 0).all()


def test_object_bank_npz_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    n = 300
    kp = torch.rand(n, 3, generator=g, dtype=torch.float64).numpy()
    df = torch.randn(128, n, generator=g).numpy()
    dc = torch.randn(256, n, generator=g).numpy()
    sc = torch.rand(n, 1, generator=g).numpy()
    path = str(tmp_path / "anno_3d_average.npz")
    ObjectBank.save_npz(path, kp, df, sc, dc)
    bank = ObjectBank.from_npz(path, shape3d=15000, device="cpu")
    assert bank.keypoints3d.shape == (1, n, 3) and bank.keypoints3d.dtype == torch.float32
    assert torch.equal(bank.keypoints3d[0], torch.Tensor(kp))
    assert torch.equal(bank.descriptors3d_db[0], torch.from_numpy(df)) and bank.descriptors3d_db.is_contiguous()
    assert torch.equal(bank.descriptors3d_coarse_db[0], torch.from_numpy(dc))
    d = bank.data(torch.zeros(1, 64, 64), [1.0, 2.0], query_image_path="x.png")
    assert d["query_image"].shape == (1, 1, 64, 64) and d["query_image_scale"].shape == (1, 2)
    assert d["descriptors3d_coarse_db"] is bank.descriptors3d_coarse_db and d["query_image_path"] == "x.png"
    # more points than shape3d: drawn with replacement by torch.randint, same indices for every array
    g1 = torch.Generator().manual_seed(5)
    small = ObjectBank.from_npz(path, shape3d=100, device="cpu", generator=g1)
    idx = torch.randint(n, (100,), generator=torch.Generator().manual_seed(5))
    assert small.num_3d_orig == n and small.keypoints3d.shape == (1, 100, 3)
    assert torch.equal(small.keypoints3d[0], torch.Tensor(kp)[idx])
    assert torch.equal(small.descriptors3d_db[0], torch.from_numpy(df)[:, idx])
    assert torch.equal(small.descriptors3d_coarse_db[0], torch.from_numpy(dc)[:, idx])
    no_coarse = ObjectBank.from_npz(path, load_3d_coarse=False, device="cpu")
    assert no_coarse.descriptors3d_coarse_db is None and "descriptors3d_coarse_db" not in no_coarse.data(torch.zeros(1, 8, 8))


def test_ingest_refuses_cpu():
    with pytest.raises((RuntimeError, OSError)):
        PI.read_grayscale_u8(np.zeros((8, 8), dtype=np.uint8), device="cpu")


def _load_reference_dataset_class():
    """The reference's own dataset file (src/datasets/OnePosePlus_inference_dataset.py) and its data_utils, loaded by
    path with stubs for the packages that are not installed here (cv2, loguru) -- neither is used by read_anno3d."""
    import importlib.util
    import sys
    import types
    saved = {k: sys.modules.get(k) for k in ("cv2", "loguru", "src", "src.utils", "src.utils.data_io", "src.utils.data_utils")}
    try:
        for k in ("cv2", "src", "src.utils", "src.utils.data_io"):
            sys.modules[k] = types.ModuleType(k)
        lg = types.ModuleType("loguru")
        lg.logger = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None)
        sys.modules["loguru"] = lg
        sys.modules["src.utils.data_io"].read_grayscale = None
        spec = importlib.util.spec_from_file_location("src.utils.data_utils", "/root/reference/src/utils/data_utils.py")
        du = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(du)
        sys.modules["src.utils.data_utils"] = du
        sys.modules["src.utils"].data_utils = du
        spec = importlib.util.spec_from_file_location("ref_inference_dataset", "/root/reference/src/datasets/OnePosePlus_inference_dataset.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.OnePosePlusInferenceDataset
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="/root/reference not mounted")
@pytest.mark.parametrize("n,shape3d", [(300, 7000), (900, 500)])
def test_object_bank_matches_reference_read_anno3d(tmp_path, n, shape3d):
    """`ObjectBank.from_npz` against the reference's own `OnePosePlusInferenceDataset.read_anno3d`
    (src/datasets/OnePosePlus_inference_dataset.py:109-160) on the same files: identical tensors, including the
    with-replacement sub-sampling above `shape3d` (same torch RNG state -> same draw)."""
    import types
    cls = _load_reference_dataset_class()
    rng = np.random.default_rng(n)
    path = str(tmp_path / "anno_3d_average.npz")
    ObjectBank.save_npz(path, rng.standard_normal((n, 3)), rng.standard_normal((128, n)).astype(np.float32),
                        np.ones((n, 1)), rng.standard_normal((256, n)).astype(np.float32))
    stub = types.SimpleNamespace(shape3d=shape3d)
    torch.manual_seed(77)
    kp, df, dc, sc, n_orig = cls.read_anno3d(stub, path, pad=True, load_3d_coarse=True)
    torch.manual_seed(77)
    bank = ObjectBank.from_npz(path, shape3d=shape3d, device="cpu")
    assert bank.num_3d_orig == n_orig == n
    assert torch.equal(bank.keypoints3d[0], kp) and torch.equal(bank.descriptors3d_db[0], df)
    assert torch.equal(bank.descriptors3d_coarse_db[0], dc) and torch.equal(bank.scores3d, sc)
    assert kp.shape[0] == min(n, shape3d)
