"""CPU: pin the oracle (oracle/onepose_oracle.py) against the golden vectors produced by the
upstream reference (tests/golden/gen_golden.py) and, where /root/reference is mounted,
against the live reference modules."""
import numpy as np
import pytest
import torch

from oracle import onepose_oracle as O
from oracle.refload import reference_available, load_reference_model_class
from tests import helpers as H
from tests.golden.cases import (E2E_CASES, MATCHER_CASES, FINE_CASES, TRANSFORMER_CASES, HIGHCONF_CASES, BATCH_CASES,
                                TRAIN_CASES)

SMALL_E2E = [n for n in E2E_CASES if "512" not in n]


@pytest.mark.parametrize("name", SMALL_E2E + ["e2e_512x512_n2000_thr0", "e2e_512x512_n7000_thr0", "e2e_512x512_n15000_thr0"])
def test_oracle_e2e_vs_golden(name):
    cfg, sd, data = H.e2e_setup(name)
    O.forward(sd, data, cfg)
    gold = H.load_golden(name)
    H.assert_match_outputs(data, gold, tol_conf=2e-5, tol_off=5e-5, tol_px=2e-4, where=name)
    meta = gold["meta"]
    assert data["bs"] == meta[0] and tuple(data["q_hw_i"]) == tuple(meta[1:3])
    assert tuple(data["q_hw_c"]) == tuple(meta[3:5]) and tuple(data["q_hw_f"]) == tuple(meta[5:7])
    if cfg["fine_matching"]["enable"]:
        assert data["W"] == meta[7]


def test_oracle_stages_vs_golden():
    gold = H.load_golden("stages_128x128_n300")
    cfg, sd, data = H.e2e_setup("e2e_128x128_n300_thr0")
    with torch.no_grad():
        feat_c, feat_f = O.backbone_forward(sd, data["query_image"])
        assert np.abs(feat_c.numpy() - gold["feat_c"]).max() < 2e-5
        assert np.abs(feat_f[:, :, ::4, ::4].numpy() - gold["feat_f"]).max() < 2e-5
        pe = O.sine_position_table(256, (256, 256))
        tok = (feat_c + pe[:, :, :16, :16]).flatten(2).transpose(1, 2)
        assert np.abs(tok.numpy() - gold["tokens2d"]).max() < 2e-5
        bank = O.keypoint_encoding(sd, O.normalize_3d_keypoints(data["keypoints3d"]),
                                   data["descriptors3d_coarse_db"])
        assert np.abs(bank.numpy() - gold["bank_enc"]).max() < 2e-5
        f3, f2 = O.local_feature_transformer(sd, "loftr_coarse", cfg["loftr_coarse"], bank, tok)
        assert np.abs(f3.numpy() - gold["f3"]).max() < 5e-5
        assert np.abs(f2.numpy() - gold["f2"]).max() < 5e-5


@pytest.mark.parametrize("name", list(TRANSFORMER_CASES))
def test_oracle_transformer_vs_golden(name):
    """loftr_coarse at the headline size (L = 4096, N = 5000) on seeded token streams."""
    L, n, seed = TRANSFORMER_CASES[name]
    cfg = H.default_config()
    sd = H.make_state_dict(cfg, 0)
    tokens2d, bank = H.transformer_inputs(L, n, seed)
    with torch.no_grad():
        f3, f2 = O.local_feature_transformer(sd, "loftr_coarse", cfg["loftr_coarse"], bank, tokens2d)
    H.assert_transformer_digest(H.transformer_digest(f3[0], f2[0]), H.load_golden(name), rel=2e-5, where=name)


@pytest.mark.parametrize("name", list(HIGHCONF_CASES))
def test_oracle_highconf_vs_golden(name):
    """Whole forward with > 1400 matches of confidence > 0.5 (up to 0.999) after backbone + transformer: here the
    1e-4 bar on confidences bites end to end; relative error reported as well."""
    cfg, sd, data = H.highconf_setup(name)
    O.forward(sd, data, cfg)
    gold = H.load_golden(name)
    assert len(gold["mconf"]) > 1000 and (gold["mconf"] > 0.5).sum() > 1000 and gold["mconf"].max() > 0.99
    H.assert_match_outputs(data, gold, tol_conf=2e-5, tol_off=5e-5, tol_px=2e-4, where=name)
    rel = H.conf_relative_error(data["mconf"], gold["mconf"])
    print("%s: oracle vs reference mconf max rel err %.2e" % (name, rel))
    assert rel < 1e-4


@pytest.mark.parametrize("name", list(BATCH_CASES))
def test_oracle_batched_masked_vs_golden(name):
    """B > 1, per-sample clouds and scales, coarse-resolution `query_image_mask`."""
    cfg, sd, data = H.batch_setup(name)
    O.forward(sd, data, cfg)
    H.assert_batched_outputs(data, H.load_golden(name), tol_conf=2e-5, tol_off=5e-5, tol_px=2e-4, where=name)


@pytest.mark.parametrize("name", list(TRAIN_CASES))
def test_oracle_train_mode_vs_golden(name):
    """train()-mode forward: BatchNorm batch statistics + running-statistics update, training branch of
    get_coarse_match with the reference's recorded random draws."""
    cfg, sd, data = H.train_setup(name)
    gold = H.load_golden(name)
    draws = [gold["randint_%d" % i] for i in range(int(gold["n_randint"]))]
    sd = {k: v.clone() for k, v in sd.items()}
    O.forward(sd, data, cfg, training=True, randint=H.RecordedRandint(draws))
    H.assert_train_outputs(data, sd, gold, tol_conf=2e-5, tol_off=5e-5, tol_px=2e-4, tol_bn=1e-5, where=name)


@pytest.mark.parametrize("name", list(MATCHER_CASES))
def test_oracle_matcher_vs_golden(name):
    cfg, f3d, f2d, data = H.matcher_setup(name)
    O.coarse_matching(f3d, f2d, data, cfg["coarse_matching"])
    gold = H.load_golden(name)
    assert len(gold["i_ids"]) > 200
    assert np.all(np.diff(gold["i_ids"]) > 0)          # quirk q9: ascending 3D index
    H.assert_match_outputs(data, gold, tol_conf=2e-5, where=name)


@pytest.mark.parametrize("name", list(FINE_CASES))
def test_oracle_fine_vs_golden(name):
    cfg, sd, feat_f, bank_f, data = H.fine_setup(name)
    gold = H.load_golden(name)
    with torch.no_grad():
        f3, win = O.fine_preprocess(data, bank_f, feat_f, cfg["loftr_fine"])
        assert np.abs(win.sum(-1).numpy() - gold["win_in_sum"]).max() < 1e-4
        f3, win = O.local_feature_transformer(sd, "loftr_fine", cfg["loftr_fine"], f3, win)
        assert np.abs(f3.numpy() - gold["f3_out"]).max() < 5e-5
        O.fine_matching(f3, win, data)
    H.assert_match_outputs(data, gold, tol_off=5e-5, tol_px=2e-4, where=name)


def test_mask_border_quirk():
    """q1: only the first border_rm rows / cols are cleared, the far border survives."""
    conf = torch.zeros(1, 4, 36)
    conf[0, 0, 0 * 6 + 3] = 0.9    # row 0 -> cleared
    conf[0, 1, 3 * 6 + 1] = 0.9    # col 1 -> cleared
    conf[0, 2, 5 * 6 + 5] = 0.9    # far corner survives
    conf[0, 3, 2 * 6 + 2] = 0.9    # interior survives
    b, i, j, c = O.coarse_match_select(conf, (6, 6), 0.1, 2)
    assert i.tolist() == [2, 3] and j.tolist() == [35, 14]


def test_position_table_quirk():
    """q2: div_term = exp(-k), k = 0,2,4,... (floor division quirk)."""
    pe = O.sine_position_table(256, (8, 8))
    assert torch.allclose(pe[0, 0, 0, :3], torch.sin(torch.tensor([1.0, 2.0, 3.0])))
    assert torch.allclose(pe[0, 4, 0, :3], torch.sin(torch.tensor([1.0, 2.0, 3.0]) * float(np.exp(-2.0))))


@pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted")
def test_oracle_vs_live_reference():
    name = "e2e_96x64_n77_thr0"
    cfg, sd, data = H.e2e_setup(name)
    ref = load_reference_model_class()(cfg).eval()
    ref.load_state_dict(sd, strict=True)
    d_ref = {k: v.clone() for k, v in data.items()}
    with torch.no_grad():
        ref(d_ref)
    O.forward(sd, data, cfg)
    gold = {k: H.to_np(d_ref[k]) for k in ["b_ids", "i_ids", "j_ids", "mconf", "expec_f",
                                           "mkpts_query_f", "mkpts_query_c", "mkpts_3d_db"]}
    gold.update(H.conf_digest_t(d_ref["conf_matrix"]))
    H.assert_match_outputs(data, gold, tol_conf=2e-5, tol_off=5e-5, tol_px=2e-4, where="live")
