"""GPU: each stage of the hot path (C ABI) against the CPU oracle / the reference golden
vectors on the same seeded inputs."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden.cases import MATCHER_CASES, FINE_CASES, TRANSFORMER_CASES

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


@pytest.fixture(scope="module", params=["bf16x3", "fp32"])
def small(request):
    from oracle import onepose_oracle as O
    from tests import hip_ops as ops
    cfg, sd, data = H.e2e_setup("e2e_128x128_n300_thr0")
    model = ops.make_model(cfg, sd, request.param)
    with torch.no_grad():
        st = O.backbone_forward(sd, data["query_image"], stages=True)
    return cfg, sd, data, model, st


def test_backbone_small(small):
    from tests import hip_ops as ops
    cfg, sd, data, model, st = small
    fc, ff = ops.backbone(model, data["query_image"])
    assert torch.isfinite(fc).all() and torch.isfinite(ff).all()
    assert _rel(fc, st["x3_out"]) < 2e-5, _rel(fc, st["x3_out"])
    assert _rel(ff, st["x1_out"]) < 2e-5, _rel(ff, st["x1_out"])
    gold = H.load_golden("stages_128x128_n300")
    assert np.abs(fc.numpy() - gold["feat_c"]).max() < 5e-5 * max(1.0, np.abs(gold["feat_c"]).max())
    assert np.abs(ff[:, :, ::4, ::4].numpy() - gold["feat_f"]).max() < 5e-5 * max(1.0, np.abs(gold["feat_f"]).max())


@pytest.mark.parametrize("hw", [(64, 96), (96, 64), (8, 8), (512, 512)])
def test_backbone_shapes(small, hw):
    from oracle import onepose_oracle as O
    from tests import hip_ops as ops
    cfg, sd, _, model, _ = small
    g = torch.Generator().manual_seed(hw[0] * 31 + hw[1])
    img = torch.rand(1, 1, hw[0], hw[1], generator=g)
    with torch.no_grad():
        rc, rf = O.backbone_forward(sd, img)
    fc, ff = ops.backbone(model, img)
    assert _rel(fc, rc) < 2e-5 and _rel(ff, rf) < 2e-5, (_rel(fc, rc), _rel(ff, rf))


@pytest.mark.parametrize("hw", [(8, 8), (40, 72), (200, 328), (512, 512)])
def test_direct_stem_is_bit_identical_to_im2col_gemm(hw):
    """csrc/stem_direct.hip (7x7 / stride 2 stem + folded BatchNorm + ReLU in one kernel, no im2col matrix in memory;
    backbone/resnet.py:101-103,143) against the im2col + GEMM path of the same arithmetic (OPP_STEM_DIRECT=0): the same
    operand split and product order, so both feature maps agree bit for bit -- partial 8 x 16 pixel tiles included."""
    import os
    from tests import hip_ops as ops
    cfg, sd, _ = H.e2e_setup("e2e_128x128_n300_thr0")
    model = ops.make_model(cfg, sd, "bf16x3")
    img = torch.rand(1, 1, hw[0], hw[1], generator=torch.Generator().manual_seed(hw[0] + 7 * hw[1]))
    fc, ff = ops.backbone(model, img)
    os.environ["OPP_STEM_DIRECT"] = "0"
    try:
        rc, rf = ops.backbone(model, img)
    finally:
        del os.environ["OPP_STEM_DIRECT"]
    assert torch.equal(fc, rc) and torch.equal(ff, rf)


@pytest.mark.parametrize("hw", [(8, 8), (40, 72), (128, 128), (512, 512)])
def test_presplit_activation_chain_is_bit_identical(hw):
    """Round 6: the bf16x3 backbone can hand its maps from layer to layer as pre-split bf16 triples (written once by the producing epilogue, read by
    every convolution whose K walk has no packed tail; csrc/api.hip backbone_impl, gemm_mfma.hip ASP) instead of splitting fp32 rows in every K
    loop -- with OPP_ASP=1 (opt-in: measured slower overall).  Same triples, same product order -> both feature maps agree bit for
    bit with the default fp32-row chain (K slices of the 1/8-resolution layers and the bilinear laterals included)."""
    import os
    from tests import hip_ops as ops
    cfg, sd, _ = H.e2e_setup("e2e_128x128_n300_thr0")
    model = ops.make_model(cfg, sd, "bf16x3")
    img = torch.rand(1, 1, hw[0], hw[1], generator=torch.Generator().manual_seed(3 * hw[0] + hw[1]))
    rc, rf = ops.backbone(model, img)
    os.environ["OPP_ASP"] = "1"
    try:
        fc, ff = ops.backbone(model, img)
    finally:
        del os.environ["OPP_ASP"]
    assert torch.isfinite(fc).all() and torch.isfinite(ff).all()
    assert torch.equal(fc, rc) and torch.equal(ff, rf)


def test_tokens_and_transformer(small):
    from oracle import onepose_oracle as O
    from tests import hip_ops as ops
    cfg, sd, data, model, st = small
    gold = H.load_golden("stages_128x128_n300")
    feat_c = torch.from_numpy(gold["feat_c"])
    hc, wc = feat_c.shape[2:]
    pe = O.sine_position_table(256, (256, 256))[0, :, :hc, :wc].permute(1, 2, 0).reshape(hc * wc, 256)
    tok = ops.coarse_tokens(model, feat_c, pe, data["keypoints3d"], data["descriptors3d_coarse_db"])
    L = hc * wc
    assert np.abs(tok[:L].numpy() - gold["tokens2d"][0]).max() < 1e-5
    ref3 = torch.from_numpy(gold["bank_enc"])[0].T
    assert (tok[L:] - ref3).abs().max() < 3e-5, (tok[L:] - ref3).abs().max()
    # transformer on the reference's own tokens
    X = torch.cat([torch.from_numpy(gold["tokens2d"][0]), ref3], 0)
    out = ops.transformer(model, 0, X, 1, L, ref3.shape[0])
    assert _rel(out[:L], torch.from_numpy(gold["f2"][0])) < 5e-5
    assert _rel(out[L:], torch.from_numpy(gold["f3"][0])) < 5e-5


@pytest.mark.parametrize("name", list(MATCHER_CASES))
def test_matcher_vs_golden(small, name):
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.synthetic import make_state_dict
    cfg, f3d, f2d, data = H.matcher_setup(name)
    model0 = ops.make_model(cfg, make_state_dict(cfg, 0), small[3].gemm_precision)      # thr / border_rm come from cfg
    got = ops.coarse_match(model0, f3d[0], f2d[0], tuple(data["q_hw_c"]), data["keypoints3d"][0], 8.0,
                           data["query_image_scale"][0])
    gold = H.load_golden(name)
    H.assert_match_outputs(got, gold, where=name)
    c = got["conf_matrix"][0]
    assert (c >= 0).all() and c.sum(1).max() <= 1 + 1e-4 and c.sum(0).max() <= 1 + 1e-4


@pytest.mark.parametrize("name", list(MATCHER_CASES))
def test_three_resident_score_gemm_against_the_two_resident_kernel(name):
    """Round 6: the single-sweep score GEMM runs three workgroups per CU on two LDS stages (gemm_ss_res3_kernel); OPP_SS_RES3=0 selects the
    two-resident kernel on three stages.  Same tile, same accumulation sequence: the SCORE tiles are bit-identical, so the column statistics are;
    the row statistics are merged online over two 64-column halves instead of two parts against a common maximum, i.e. they agree to rounding --
    match indices identical, confidences within 1e-6 of each other (both within the golden bar, test_matcher_vs_golden)."""
    import os
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.synthetic import make_state_dict
    cfg, f3d, f2d, data = H.matcher_setup(name)
    model0 = ops.make_model(cfg, make_state_dict(cfg, 0), "bf16x3")
    args = (model0, f3d[0], f2d[0], tuple(data["q_hw_c"]), data["keypoints3d"][0], 8.0, data["query_image_scale"][0])
    got = ops.coarse_match(*args)
    os.environ["OPP_SS_RES3"] = "0"
    try:
        ref = ops.coarse_match(*args)
    finally:
        del os.environ["OPP_SS_RES3"]
    assert torch.equal(got["i_ids"], ref["i_ids"]) and torch.equal(got["j_ids"], ref["j_ids"])
    assert (got["conf_matrix"] - ref["conf_matrix"]).abs().max() <= 1e-6
    assert (got["mconf"] - ref["mconf"]).abs().max() <= 1e-6


@pytest.mark.parametrize("name", list(FINE_CASES))
def test_fine_vs_golden(small, name):
    from tests import hip_ops as ops
    _, _, _, model0, _ = small
    cfg, sd, feat_f, bank_f, data = H.fine_setup(name)
    ex, mf = ops.fine(model0, feat_f, bank_f, data["i_ids"], data["j_ids"], tuple(data["q_hw_c"]),
                      data["mkpts_query_c"], 2.0, data["query_image_scale"][0])
    gold = H.load_golden(name)
    H.assert_match_outputs({"expec_f": ex, "mkpts_query_f": mf}, {k: gold[k] for k in ("expec_f", "mkpts_query_f")},
                           where=name)


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
def test_matcher_exact_ties_follow_reference_rules(precision):
    """Duplicated image cells (tie inside a row -> first column wins) and duplicated 3D points
    (both rows are reported with the same cell), coarse_matching.py:158-172 / quirk q9."""
    from oracle import onepose_oracle as O
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.config import default_config
    from onepose_plus_plus_amd.synthetic import make_state_dict
    cfg = default_config(thr=0.05)
    g = torch.Generator().manual_seed(42)
    hw_c = (16, 24)
    L, N = hw_c[0] * hw_c[1], 300
    f2d = torch.randn(L, 256, generator=g) * 4
    f3d = torch.randn(N, 256, generator=g) * 4
    a, b = 2 * 24 + 5, 9 * 24 + 7          # two interior cells made identical: point 0 matches both
    perm = [c for c in torch.randperm(L, generator=g).tolist() if c not in (a, b)]
    cells = torch.tensor(perm[:200])
    f3d[:200] = f2d[cells] + 0.4 * torch.randn(200, 256, generator=g)
    f2d[b] = f2d[a]
    f3d[0] = f2d[a] + 0.4 * torch.randn(256, generator=g)
    f3d[150] = f3d[149]                     # duplicated 3D point (same tile)
    f3d[290] = f3d[10]                      # duplicated 3D point (different 128-row tile)
    kpts = torch.rand(1, N, 3, generator=g) - 0.5
    data = {"q_hw_i": torch.Size([128, 192]), "q_hw_c": torch.Size(hw_c), "keypoints3d": kpts}
    O.coarse_matching(f3d[None], f2d[None], data, cfg["coarse_matching"])
    model = ops.make_model(cfg, make_state_dict(cfg, 0), precision)
    got = ops.coarse_match(model, f3d, f2d, hw_c, kpts[0], 8.0, None)
    ref_i, ref_j = data["i_ids"].tolist(), data["j_ids"].tolist()
    assert 0 in ref_i and ref_j[ref_i.index(0)] == min(a, b)          # the tie resolves to the first cell
    assert got["i_ids"].tolist() == ref_i and got["j_ids"].tolist() == ref_j
    assert (got["mconf"] - data["mconf"]).abs().max() < 1e-4
    c = got["conf_matrix"][0]
    assert torch.equal(c[:, a], c[:, b]) and torch.equal(c[150], c[149]) and torch.equal(c[290], c[10])


@pytest.mark.parametrize("hw,n", [((16, 24), 5), ((8, 8), 3), ((24, 16), 1), ((64, 64), 129)])
def test_tiny_shapes_vs_oracle(hw, n):
    """Degenerate sizes: a 1-cell coarse grid, a single 3D point, N just over one tile."""
    from oracle import onepose_oracle as O
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.config import default_config
    from onepose_plus_plus_amd.synthetic import make_state_dict, make_inputs
    cfg = default_config(thr=0.0)
    cfg["coarse_matching"]["border_rm"] = 0
    sd = make_state_dict(cfg, 3)
    data = make_inputs(n, hw, 21)
    ref = dict(data)
    O.forward(sd, ref, cfg)
    out = ops.run_model(ops.make_model(cfg, sd), data)
    assert out["i_ids"].tolist() == ref["i_ids"].tolist() and out["j_ids"].tolist() == ref["j_ids"].tolist()
    if n == 1:   # zero bbox extent: normalize_3d_keypoints divides 0/0 upstream -> NaN everywhere, no match
        assert torch.isnan(ref["conf_matrix"]).all() and torch.isnan(out["conf_matrix"]).all()
        return
    assert (out["conf_matrix"].cpu() - ref["conf_matrix"]).abs().max() < 1e-4
    if len(ref["mconf"]):
        assert (out["expec_f"].cpu() - ref["expec_f"]).abs().max() < 1e-4
        assert (out["mkpts_query_f"].cpu() - ref["mkpts_query_f"]).abs().max() < 1e-3


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("name", list(TRANSFORMER_CASES))
def test_transformer_stage_full_size_vs_golden(name, precision):
    """loftr_coarse alone at L = 4096 image tokens x N = 5000 points against the reference's outputs."""
    from tests import hip_ops as ops
    from onepose_plus_plus_amd.config import default_config
    from onepose_plus_plus_amd.synthetic import make_state_dict
    L, n, seed = TRANSFORMER_CASES[name]
    cfg = default_config()
    model = ops.make_model(cfg, make_state_dict(cfg, 0), precision)
    tokens2d, bank = H.transformer_inputs(L, n, seed)
    x = torch.cat([tokens2d[0], bank[0].t().contiguous()], 0)           # [L + N, C]: image tokens first
    y = ops.transformer(model, 0, x, 1, L, n)
    H.assert_transformer_digest(H.transformer_digest(y[L:], y[:L]), H.load_golden(name), rel=5e-5, where=name)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [2, 1])
@pytest.mark.parametrize("n,hw_c,planted", [(5000, (64, 64), 3000), (77, (12, 8), 40), (300, (16, 24), 200), (1000, (33, 20), 500)])
def test_two_sweep_matcher_equals_the_materialised_path(n, hw_c, planted, mode):
    """Coarse matcher as two sweeps of the split-operand score GEMM (gemm_ss.hip: statistics, then confidences written
    once; coarse_matching.py:99-115, :145-172) against the r02 path that materialises the score matrix and sweeps it in
    place: same match indices, confidences equal to rounding (the score tiles are bit-identical, only the summation order
    of the softmax statistics differs), ragged N / L (tiles that end inside a 256 x 128 block) included."""
    from tests import hip_ops as ops
    from onepose_plus_plus_amd import default_config
    from onepose_plus_plus_amd.synthetic import make_state_dict
    cfg = default_config(thr=0.1)
    sd = make_state_dict(cfg, 0)
    g = torch.Generator().manual_seed(5 + n)
    L = hw_c[0] * hw_c[1]
    f2d = torch.randn(L, 256, generator=g) * 4
    f3d = torch.randn(n, 256, generator=g) * 4
    m = min(planted, L, n)
    cells = torch.randperm(L, generator=g)[:m]
    f3d[:m] = f2d[cells] + 0.4 * torch.randn(m, 256, generator=g)
    kpts = torch.rand(n, 3, generator=g) - 0.5
    two = ops.make_model(cfg, sd, "bf16x3").set_score_two_sweep(mode).cuda()
    one = ops.make_model(cfg, sd, "bf16x3").set_score_two_sweep(0).cuda()
    a = ops.coarse_match(two, f3d, f2d, hw_c, kpts, 8.0, None)
    b = ops.coarse_match(one, f3d, f2d, hw_c, kpts, 8.0, None)
    assert len(b["i_ids"]) > m // 4
    assert torch.equal(a["i_ids"], b["i_ids"]) and torch.equal(a["j_ids"], b["j_ids"])
    ca, cb = a["conf_matrix"][0], b["conf_matrix"][0]
    assert torch.isfinite(ca).all()
    # the softmax statistics are sums of up to 5000 exponentials accumulated in a different (fixed) order
    assert (ca - cb).abs().max() <= 1e-5 * max(1.0, cb.abs().max().item()), (ca - cb).abs().max()
    assert (a["mconf"] - b["mconf"]).abs().max() <= 1e-5, (a["mconf"] - b["mconf"]).abs().max()
    assert torch.equal(a["mkpts_query_c"], b["mkpts_query_c"]) and torch.equal(a["mkpts_3d_db"], b["mkpts_3d_db"])


def test_persistent_score_gemm_meets_the_same_bar():
    """OPP_SS_PERSIST=1 (gemm_ss.hip: persistent single-sweep score GEMM, opt-in since it measured equal to the one-tile kernel): the same
    matcher fixtures, tie rules and two-sweep comparison in a process that runs it (the switch is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_stages_gpu.py"), "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu",
                        "-k", "matcher_vs_golden or exact_ties or two_sweep_matcher"],
                       env=dict(os.environ, OPP_SS_PERSIST="1"), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]
