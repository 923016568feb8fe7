"""CPU: host-side mirror of the reference module API (construction, state-dict contract,
pickling, error behaviour) -- no device work."""
import pickle

import pytest
import torch

from onepose_plus_plus_amd import OnePosePlus_model, default_config
from onepose_plus_plus_amd.synthetic import make_state_dict
from oracle.refload import reference_available, load_reference_model_class


def test_state_dict_contract():
    cfg = default_config()
    m = OnePosePlus_model(cfg)
    sd = m.state_dict()
    assert len(sd) == 195
    assert sum(v.numel() for k, v in sd.items() if "num_batches_tracked" not in k) == 10233184 - 0
    assert "dense_pos_encoding.pe" not in sd               # non-persistent buffer
    assert m.dense_pos_encoding.pe.shape == (1, 256, 256, 256)
    gold = make_state_dict(cfg, 0)
    assert list(sd.keys()) == list(gold.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(gold[k].shape), k
    m.load_state_dict(gold, strict=True)
    # checkpoint convention of the caller: strip 'matcher.' (inference_OnePosePlus.py:32-36)
    ckpt = {"matcher." + k: v for k, v in gold.items()}
    stripped = {k.replace("matcher.", ""): v for k, v in ckpt.items()}
    OnePosePlus_model(cfg).load_state_dict(stripped, strict=True)
    with pytest.raises(RuntimeError):
        OnePosePlus_model(cfg).load_state_dict({k: v for k, v in list(gold.items())[:-1]}, strict=True)


@pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted")
def test_state_dict_matches_live_reference():
    cfg = default_config()
    ref = load_reference_model_class()(cfg)
    ours = OnePosePlus_model(cfg)
    rs, os_ = ref.state_dict(), ours.state_dict()
    assert list(rs.keys()) == list(os_.keys())
    assert all(rs[k].shape == os_[k].shape and rs[k].dtype == os_[k].dtype for k in rs)
    ours.load_state_dict(rs, strict=True)
    ref.load_state_dict(os_, strict=True)
    assert torch.equal(ref.dense_pos_encoding.pe, ours.dense_pos_encoding.pe)


def test_pickle_roundtrip():
    """Ray pickles the module into its workers (inference_OnePosePlus.py:85-95)."""
    cfg = default_config()
    m = OnePosePlus_model(cfg).eval()
    m.load_state_dict(make_state_dict(cfg, 0))
    m2 = pickle.loads(pickle.dumps(m))
    assert not m2.training
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert m2._rt["ctx"] is None and m2._rt["dirty"]


def test_pretrained_backbone_loading(tmp_path):
    """OnePosePlusModel.py:78-94: keys containing 'backbone' are stripped and strict-loaded."""
    cfg = default_config()
    gold = make_state_dict(cfg, 5)
    ck = {"matcher.backbone." + k[len("backbone."):]: v for k, v in gold.items() if k.startswith("backbone.")}
    ck["matcher.loftr_coarse.layers.0.q_proj.weight"] = torch.zeros(1)    # must be ignored
    path = tmp_path / "loftr.ckpt"
    torch.save({"state_dict": ck}, path)
    cfg["loftr_backbone"]["pretrained"] = str(path)
    cfg["loftr_backbone"]["pretrained_fix"] = True
    m = OnePosePlus_model(cfg)
    assert torch.equal(m.backbone.conv1.weight, gold["backbone.conv1.weight"])
    assert not any(p.requires_grad for p in m.backbone.parameters())
    assert all(p.requires_grad for p in m.loftr_coarse.parameters())


def test_unsupported_config_values_raise_like_reference():
    cfg = default_config()
    cfg["loftr_backbone"]["resolution"] = [16, 4]
    with pytest.raises(NotImplementedError):          # backbone/__init__.py:9-12
        OnePosePlus_model(cfg)
    cfg = default_config()
    cfg["loftr_backbone"]["type"] = "VGG"
    with pytest.raises(ValueError):                   # backbone/__init__.py:13-14
        OnePosePlus_model(cfg)
    cfg = default_config()
    cfg["keypoints_encoding"]["type"] = "mlp_conv"
    with pytest.raises(NotImplementedError):          # OnePosePlusModel.py:47-50
        OnePosePlus_model(cfg)
    cfg = default_config()
    cfg["coarse_matching"]["type"] = "sinkhorn"
    with pytest.raises(NotImplementedError):          # coarse_matching.py:63-66
        OnePosePlus_model(cfg)
    cfg = default_config()
    cfg["loftr_coarse"]["layer_names"] = ["self", "foo"]
    with pytest.raises(NotImplementedError):          # transformer.py:117-120
        OnePosePlus_model(cfg)


def test_inference_only_and_no_cpu_fallback():
    m = OnePosePlus_model(default_config())
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # train() mode included
        m({"query_image": torch.zeros(1, 1, 64, 64)})
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m({"query_image": torch.zeros(1, 1, 64, 64)})
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # masks / batches are supported, CPU tensors are not
        m({"query_image": torch.zeros(2, 1, 64, 64), "query_image_mask": torch.ones(2, 8, 8)})



def test_product_never_imports_oracle():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "onepose_plus_plus_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_gemm_precision_selector(monkeypatch):
    """Host side of `opp_config.gemm_precision`: names, default, env override, pickling, bad values."""
    import pickle
    from onepose_plus_plus_amd import model as M
    m = OnePosePlus_model(default_config())
    assert m.gemm_precision == M.DEFAULT_GEMM_PRECISION == "bf16x3"        # the default is not narrower than fp32
    assert [m.set_gemm_precision(p)._c_config().gemm_precision for p in ("fp32", "bf16x3")] == [0, 3]
    for narrower in ("fp16x2", "fp16x2_all", "bf16"):                      # nothing narrower than fp32 is offered
        with pytest.raises(ValueError):
            m.set_gemm_precision(narrower)
    assert pickle.loads(pickle.dumps(m)).gemm_precision == "bf16x3"              # travels to Ray-style workers
    with pytest.raises(ValueError):
        m.set_gemm_precision("bf16")
    monkeypatch.setenv("OPP_GEMM_PRECISION", "fp32")
    assert OnePosePlus_model(default_config()).gemm_precision == "fp32"
    monkeypatch.setenv("OPP_GEMM_PRECISION", "tf32")
    with pytest.raises(ValueError):
        OnePosePlus_model(default_config())._c_config()


def test_scheduling_switches(monkeypatch):
    """Host side of the scheduling switches (results are bit-identical for every value; GPU tests check that): defaults, setters,
    `opp_config` fields, env overrides, pickling, bad values."""
    import pickle
    for k in ("OPP_ENCODER_FUSION", "OPP_SCORE_PATH", "OPP_FPN_OVERLAP", "OPP_SKIP_UNUSED_FINE_MAP"):
        monkeypatch.delenv(k, raising=False)
    m = OnePosePlus_model(default_config())
    c = m._c_config()
    assert (c.encoder_fusion, c.score_two_sweep, c.fpn_overlap, c.tile_policy) == (2, 2, 1, 0)
    assert m.skip_unused_fine_map is False                       # the default launches every operator of the reference
    m.set_encoder_fusion(1).set_score_two_sweep(0).set_fpn_overlap(False).set_tile_policy("throughput").set_skip_unused_fine_map(True)
    c = m._c_config()
    assert (c.encoder_fusion, c.score_two_sweep, c.fpn_overlap, c.tile_policy) == (1, 0, 0, 1)
    m2 = pickle.loads(pickle.dumps(m))                           # the switches travel with the module
    c2 = m2._c_config()
    assert (c2.encoder_fusion, c2.score_two_sweep, c2.fpn_overlap, c2.tile_policy, m2.skip_unused_fine_map) == (1, 0, 0, 1, True)
    for bad in (lambda: m.set_encoder_fusion(3), lambda: m.set_score_two_sweep(5), lambda: m.set_tile_policy("fastest")):
        with pytest.raises(ValueError):
            bad()
    monkeypatch.setenv("OPP_ENCODER_FUSION", "0")
    monkeypatch.setenv("OPP_SCORE_PATH", "1")
    monkeypatch.setenv("OPP_FPN_OVERLAP", "0")
    monkeypatch.setenv("OPP_SKIP_UNUSED_FINE_MAP", "1")
    e = OnePosePlus_model(default_config())
    c = e._c_config()
    assert (c.encoder_fusion, c.score_two_sweep, c.fpn_overlap, e.skip_unused_fine_map) == (0, 1, 0, True)



def test_smi_trace_summary_reads_amd_smi_samples():
    """tools/smi_trace.py (clock / power evidence beside a bench run): the amd-smi JSON samples are reduced to activity, socket
    power and per-XCD gfx clocks over the busy part of the run; malformed samples are skipped."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("smi_trace", os.path.join(root, "tools", "smi_trace.py"))
    smi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(smi)

    def sample(t, act, watts, clocks):
        gpu = {"gpu": 0, "usage": {"gfx_activity": {"value": act, "unit": "%"}}, "power": {"socket_power": {"value": watts, "unit": "W"}},
               "clock": {"gfx_%d" % i: {"clk": {"value": c, "unit": "MHz"}, "max_clk": {"value": 2400, "unit": "MHz"}} for i, c in enumerate(clocks)}}
        gpu["clock"]["mem_0"] = {"clk": {"value": 2000, "unit": "MHz"}}
        return json.dumps({"t": t, "tool": "amd-smi", "out": json.dumps({"gpu_data": [gpu]})})

    lines = [sample(0.1, 0, 250, [150] * 8), sample(1.0, 100, 1380, [2000, 2100] * 4), sample(2.0, 98, 1360, [2050] * 8), "not json",
             json.dumps({"t": 3.0, "tool": "amd-smi", "out": "{}"})]
    rows = smi.smi_rows(lines)
    assert len(rows) == 3 and rows[1]["xcds"] == 8 and rows[1]["gfx_clk_mhz_mean"] == 2050.0 and rows[1]["gfx_clk_mhz_limit"] == 2400.0
    s = smi.smi_summary(rows)
    assert s["samples"] == 3 and s["busy_samples (gfx_activity >= 50 %)"] == 2
    assert s["socket_power_w"] == {"min": 1360.0, "mean": 1370.0, "max": 1380.0}
    assert s["gfx_clk_mhz_mean"]["mean"] == 2050.0 and s["gfx_clk_mhz_min"]["min"] == 2000.0


def _k3(cin):
    """packed K of a 3x3 convolution (opp_conv_packed_k): 32 n + (1..4) channels pack their last channels 8 taps to a chunk"""
    g, t = divmod(cin, 32)
    return (g * 9 + 2) * 32 if 1 <= t <= 4 else ((cin + 31) // 32) * 32 * 9


# layer, M = output pixels, real / stored output channels, K, conv -> (latency policy, throughput policy) tile of the default arithmetic; + 1000 = four K slices.
# Cross-checked against the kernel traces of both conditions (profiles/r05_kernel_stats_bench_streams1.csv: 5 launches of 256x128, 2 of 128x224, 6 + 4 of
# 128x128, 2 of 64x128, 2 of 64x64 per forward; ..._default_streams4.csv: 10 of 128x256, 5 of 256x128, 4 K-sliced).
TILE_POLICY_CASES = [
    ("layer1 3x3 128->128 @256^2", 65536, 128, 128, 1152, 1, 20, 20),
    ("layer1_outconv2.1 3x3 196->128 @256^2", 65536, 128, 128, _k3(196), 1, 20, 20),
    ("layer1_outconv2.0 3x3 196->196 @256^2", 65536, 196, 224, _k3(196), 1, 27, 22),      # the 224-column ring tile: latency policy, full rounds only
    ("layer1_outconv 1x1 128->196 @256^2", 65536, 196, 224, 128, 1, 27, 22),
    ("layer2 3x3 196->196 @128^2", 16384, 196, 224, _k3(196), 1, 25, 22),                 # half a round: never the ring tile
    ("layer2_outconv2.0 3x3 256->256 @128^2", 16384, 256, 256, 2304, 1, 25, 22),
    ("layer2_outconv 1x1 196->256 @128^2", 16384, 256, 256, 224, 1, 26, 22),
    ("layer3 3x3 256->256 @64^2", 4096, 256, 256, 2304, 1, 1025, 1025),                    # split-K is a shape decision, the same under both policies
    ("layer3_outconv 1x1 256->256 @64^2", 4096, 256, 256, 256, 1, 2, 26),
    ("batch of 4: layer2 3x3 196->196", 65536, 196, 224, _k3(196), 1, 27, 22),
]


@pytest.mark.parametrize("case", TILE_POLICY_CASES, ids=[c[0] for c in TILE_POLICY_CASES])
def test_tile_policy_of_the_launcher(case):
    """The launcher's own tile choice is a pure host function (opp_gemm_tile_for): pinned here per backbone layer shape and tile policy.
    The 128 x 224 ring tile only where it measured faster: latency policy, 224 stored / <= 208 real columns, a grid of >= 256 tiles."""
    from onepose_plus_plus_amd import _lib
    lib = _lib.load()
    _, M, n_real, n_store, K, conv, lat, thr = case
    assert lib.opp_gemm_tile_for(M, n_real, n_store, K, conv, 2, 0) == lat
    assert lib.opp_gemm_tile_for(M, n_real, n_store, K, conv, 2, 1) == thr


def test_tile_policy_guards():
    from onepose_plus_plus_amd import _lib
    lib = _lib.load()
    # unknown or too many real columns: the ring tile's "rows >= 208 are zero padding" premise does not hold -> 128 x 256
    assert lib.opp_gemm_tile_for(65536, 0, 224, _k3(196), 1, 2, 0) == 22
    assert lib.opp_gemm_tile_for(65536, 224, 224, _k3(196), 1, 2, 0) == 22
    assert lib.opp_gemm_tile_for(65536, 196, 224, _k3(196), 1, 0, 0) != 27       # exact-fp32 arithmetic has no such tile
    assert lib.opp_gemm_tile_for(65536, 196, 224, _k3(196), 1, 1, 0) != 27       # nor fp16x2
    for bad in [(0, 1, 128, 128, 1, 2, 0), (128, 1, 128, 100, 1, 2, 0), (128, 1, 128, 128, 1, 5, 0), (128, 1, 128, 128, 1, 2, 7)]:
        assert lib.opp_gemm_tile_for(*bad) < 0
