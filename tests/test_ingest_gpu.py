"""GPU: `opp_image_ingest_u8` (8-bit frame -> resized fp32 / 255) bit-exact against the ingest oracle, and the
resident object bank + ingest feeding the model."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # (h, w), resize, df, pad_to
    ((480, 640), [512, 512], 8, None),
    ((480, 640), [512], 8, None),
    ((1920, 1440), [512], 8, None),
    ((1024, 1024), [512, 512], 8, None),      # exact 2x decimation (area fast path)
    ((96, 128), [512, 512], 8, None),         # up-sampling
    ((333, 517), None, 8, None),              # df crop only
    ((480, 640), [256], 8, 256),              # padded canvas + mask
    ((5, 7), [3, 2], None, None),
    ((1, 1), [4, 4], None, None),
]


@pytest.mark.parametrize("hw,resize,df,pad_to", CASES)
def test_ingest_bit_exact_vs_oracle(hw, resize, df, pad_to):
    from oracle import ingest_oracle as IO
    from onepose_plus_plus_amd import ingest as PI
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    img = rng.integers(0, 256, hw, dtype=np.uint8)
    ref, ref_scales, ref_mask = IO.read_grayscale_u8(img, resize=resize, df=df, pad_to=pad_to)
    out, scales, mask = PI.read_grayscale_u8(img, resize=resize, df=df, pad_to=pad_to, ret_scales=True, ret_pad_mask=True)
    torch.cuda.synchronize()
    assert out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == ref.shape
    assert np.array_equal(out.cpu().numpy(), ref)                     # bit-exact, including the / 255
    assert np.array_equal(scales.numpy(), ref_scales)
    if pad_to is None:
        assert mask is None
    else:
        assert np.array_equal(mask.cpu().numpy(), ref_mask)


def test_ingest_strided_source_and_u8_output():
    from oracle import ingest_oracle as IO
    from onepose_plus_plus_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    full = torch.from_numpy(rng.integers(0, 256, (200, 300), dtype=np.uint8)).cuda()
    view = full[10:170, 20:260]                                       # row stride 300, 160 x 240 window
    dst = torch.empty(128, 128, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.opp_image_ingest_u8(view.data_ptr(), 160, 240, view.stride(0), 128, 128, None, 0, dst.data_ptr(), s), "ingest")
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), IO.resize_u8_linear(view.cpu().numpy(), 128, 128))
    assert lib.opp_image_ingest_u8(view.data_ptr(), 160, 240, 100, 128, 128, None, 0, dst.data_ptr(), s) != 0   # stride < w


def test_bank_and_ingest_feed_the_model(tmp_path):
    """Frame -> ingest -> ObjectBank.data -> model == the same forward on oracle-prepared CPU inputs."""
    from oracle import ingest_oracle as IO
    from onepose_plus_plus_amd import ingest as PI
    from onepose_plus_plus_amd.bank import ObjectBank
    from onepose_plus_plus_amd.config import default_config
    from onepose_plus_plus_amd.synthetic import make_state_dict, make_inputs
    from tests import hip_ops as ops
    cfg = default_config(thr=0.0)
    sd = make_state_dict(cfg, 0)
    base = make_inputs(400, (128, 128), 3)
    path = str(tmp_path / "anno_3d_average.npz")
    ObjectBank.save_npz(path, base["keypoints3d"][0].double().numpy(), base["descriptors3d_db"][0].numpy(),
                        np.ones((400, 1), np.float32), base["descriptors3d_coarse_db"][0].numpy())
    bank = ObjectBank.from_npz(path, shape3d=15000)
    frame = np.random.default_rng(9).integers(0, 256, (150, 200), dtype=np.uint8)
    img, scales = PI.read_grayscale_u8(frame, resize=[128, 128], df=8, ret_scales=True)
    model = ops.make_model(cfg, sd)
    d1 = bank.data(img, scales)
    d2 = bank.data(img, scales)
    with torch.no_grad():
        model(d1)
        model(d2)                                                      # second image: cached object tokens
    ref_img, ref_scales, _ = IO.read_grayscale_u8(frame, resize=[128, 128], df=8)
    ref = dict(base)
    ref["query_image"] = torch.from_numpy(ref_img)[None]
    ref["query_image_scale"] = torch.from_numpy(ref_scales)[None]
    out = ops.run_model(ops.make_model(cfg, sd), ref)
    for k in ("i_ids", "j_ids", "mconf", "mkpts_query_f", "mkpts_3d_db"):
        assert torch.equal(d1[k], out[k]) and torch.equal(d2[k], out[k]), k
