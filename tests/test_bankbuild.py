"""Per-object descriptor bank builder: the numpy restatement (oracle/bankbuild_oracle.py) and the
device path (onepose_plus_plus_amd/bankbuild.py -> opp_segmented_mean) against outputs of the reference's own
gather_3d_ann / mean_descriptors_and_scores (tests/golden/bankbuild_*.npz) -- bit for bit."""
import numpy as np
import pytest
import torch

from tests import helpers as H


@pytest.mark.parametrize("name", list(H.BANKBUILD_CASES))
def test_oracle_bankbuild_vs_golden(name):
    from oracle import bankbuild_oracle as B
    n, dim, seed = H.BANKBUILD_CASES[name]
    gold = H.load_golden(name)
    feat, score, xyzs, points_idxs = H.bankbuild_inputs(n, dim, seed)
    pos, desc, scores, idxs = B.gather_3d_ann(feat, score, xyzs, points_idxs)
    assert np.array_equal(pos, gold["kp3d_position"]) and np.array_equal(idxs, gold["idxs"])
    assert desc.shape[0] == int(gold["n_rows"]) and np.array_equal(desc.sum(1), gold["desc_rowsum"])
    avg, avg_scores, _ = B.mean_descriptors_and_scores(desc, scores, idxs)
    assert avg.dtype == gold["avg_descriptors"].dtype == np.float32
    assert np.array_equal(avg, gold["avg_descriptors"])                 # numpy's axis-0 mean, reproduced exactly
    assert np.array_equal(avg_scores, gold["avg_scores"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(H.BANKBUILD_CASES))
def test_hip_bankbuild_vs_golden(name, tmp_path):
    from onepose_plus_plus_amd import bankbuild as BB
    from onepose_plus_plus_amd.bank import ObjectBank
    n, dim, seed = H.BANKBUILD_CASES[name]
    gold = H.load_golden(name)
    feat, score, xyzs, points_idxs = H.bankbuild_inputs(n, dim, seed)
    pos, desc, scores, idxs = BB.gather_3d_ann(feat, score, xyzs, points_idxs, verbose=False)
    assert np.array_equal(pos, gold["kp3d_position"]) and np.array_equal(idxs, gold["idxs"])
    assert np.array_equal(desc.sum(1), gold["desc_rowsum"])
    avg, avg_scores, _ = BB.mean_descriptors_and_scores(desc, scores, idxs)
    assert avg.dtype == np.float32 and np.array_equal(avg, gold["avg_descriptors"])      # bit-exact on the device
    assert np.array_equal(avg_scores, gold["avg_scores"])
    # the written file is what the matching path reads (save_3d_anno layout: descriptors3d [D, N])
    out = tmp_path / "anno_3d_average.npz"
    BB.build_object_bank(feat, score, xyzs, points_idxs, str(out))
    z = np.load(out)
    assert z["descriptors3d"].shape == (dim, n) and np.array_equal(z["descriptors3d"].T, gold["avg_descriptors"])
    assert z["keypoints3d"].shape == (n, 3) and z["scores3d"].shape == (n, 1)


@pytest.mark.gpu
def test_segmented_mean_edge_cases():
    """single-row spans, an empty span (NaN like np.mean of an empty slice), D not a multiple of 64"""
    from onepose_plus_plus_amd import bankbuild as BB
    rng = np.random.default_rng(0)
    desc = rng.standard_normal((10, 96)).astype(np.float32)
    avg, _, _ = BB.mean_descriptors_and_scores(desc, np.ones((10, 1)), np.array([1, 0, 4, 5]))
    assert np.array_equal(avg[0], desc[0]) and np.isnan(avg[1]).all()
    with np.errstate(all="ignore"):
        assert np.array_equal(avg[2], np.mean(desc[1:5], axis=0)) and np.array_equal(avg[3], np.mean(desc[5:10], axis=0))
