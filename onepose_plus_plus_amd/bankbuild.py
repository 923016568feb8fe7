"""Per-object descriptor bank builder: the step BEFORE the matching path.

Mirrors the computational core of `get_kpt_ann` (/root/reference/src/sfm_utils/postprocess/feature_process.py:544-649):

    gather_3d_ann                  :255-311   concatenate, per filtered 3D point, the 2D features of its merged tracks
    mean_descriptors_and_scores    :527-541   mean feature per 3D point (scores are the constant 1 upstream)
    save_3d_anno                   :316-319   anno_3d_average[_coarse].npz = {keypoints3d, descriptors3d [D,N], scores3d}

with the same signatures and return values, so `get_kpt_ann` can call them in place of its own.  The reference grows
its arrays with `np.append` inside the loops (quadratic copying); here the gather is one concatenation and the
segmented mean runs on the device (`opp_segmented_mean`, csrc/bankbuild.hip), bit-identical to numpy's float32
axis-0 mean.  The COLMAP / h5py readers around these functions (`count_features`, `read_model`) are I/O glue of the
SfM pipeline and stay the reference's.  The written file is what `bank.ObjectBank.from_npz` and the reference's
`OnePosePlusInferenceDataset` read.
"""
import numpy as np
import torch

from . import _lib


def gather_3d_ann(kp3d_id_feature, kp3d_id_score, xyzs, points_idxs, pba=None, verbose=True):
    """-> (kp3d_position [N,3], kp3d_descriptors [R,D], kp3d_scores [R,1], idxs [N]) exactly like
    feature_process.py:255-311 (`pba` / `verbose` accepted for signature compatibility)."""
    desc, scores, pos, idxs = [], [], [], []
    for new_point_idx, old_points_idxs in points_idxs.items():
        n = 0
        for old in old_points_idxs:
            f = kp3d_id_feature[old]
            desc.append(f)
            scores.append(np.asarray(kp3d_id_score[old]).reshape(-1, 1))
            n += f.shape[0]
        pos.append(np.asarray(xyzs[new_point_idx]).reshape(1, 3))
        idxs.append(n)
    kp3d_position = np.concatenate(pos, 0).astype(np.float64) if pos else np.empty((0, 3))
    kp3d_descriptors = np.concatenate(desc, 0) if desc else None
    kp3d_scores = np.concatenate(scores, 0).astype(np.float64) if scores else np.empty((0, 1))
    return kp3d_position, kp3d_descriptors, kp3d_scores, np.array(idxs)


def mean_descriptors_and_scores(descriptors, scores, idxs, device=None):
    """-> (avg_descriptors [N,D] float32, avg_scores [N,1] ones ("Fake score!" upstream), idxs) like
    feature_process.py:527-541; the spans' means are computed on the device."""
    lib = _lib.load()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idxs = np.asarray(idxs)
    rows = torch.from_numpy(np.ascontiguousarray(descriptors, dtype=np.float32)).to(device)
    n_seg, D = int(len(idxs)), int(rows.shape[1])
    offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(idxs)]).astype(np.int64)).to(device)
    out = torch.empty((n_seg, D), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(lib.opp_segmented_mean(rows.data_ptr(), D, offsets.data_ptr(), n_seg, out.data_ptr(), stream), "opp_segmented_mean")
    avg = out.cpu().numpy()
    return avg, np.ones((avg.shape[0], 1)), idxs


def save_3d_anno(xyzs, descriptors, scores, out_path):
    """feature_process.py:316-319"""
    np.savez(out_path, keypoints3d=xyzs, descriptors3d=descriptors.transpose(1, 0), scores3d=scores)


def build_object_bank(kp3d_id_feature, kp3d_id_score, xyzs, points_idxs, out_path=None, device=None):
    """Steps 2-3 of get_kpt_ann (:605-649) in one call: gather -> mean -> (optionally) anno_3d_average npz.
    -> (keypoints3d [N,3], descriptors3d [N,D], scores3d [N,1])"""
    pos, desc, scores, idxs = gather_3d_ann(kp3d_id_feature, kp3d_id_score, xyzs, points_idxs, verbose=False)
    avg, avg_scores, _ = mean_descriptors_and_scores(desc, scores, idxs, device=device)
    if out_path is not None:
        save_3d_anno(pos, avg, avg_scores, out_path)
    return pos, avg, avg_scores
