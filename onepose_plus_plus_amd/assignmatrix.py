"""Ground-truth matrices of a training sample, built on the device (the input side of the training step).

Mirror of `OnePosePlusDataset.build_assignmatrix` (/root/reference/src/datasets/OnePosePlus_dataset.py:174-236).  The reference's
loader forms `conf_matrix_gt` [N, L] int16 and `fine_location_matrix_gt` [N, L, 2] fp32 on the host for every sample -- 229 MB at the
training pad size N = 7000, L = 4096, i.e. ~0.9 GB of host-to-device traffic per 4-sample step -- out of k <= a few thousand
(2D keypoint, 3D point) pairs.  Here only the pairs (and the two small keypoint arrays) travel; `libopp_hip.so` scatters them into the
two matrices in HBM (`opp_build_assignmatrix`: memset + fill + two index-sized kernels).  Same dropping rules and the same
last-write-wins behaviour for duplicate pairs as the reference's CPU `index_put`.
"""
import torch

from . import _lib


def build_assignmatrix(keypoints2D_coarse, keypoints2D_fine, assign_matrix, shape3d, n_query_coarse_grid, w_c, query_img_scale,
                       coarse_scale=1.0 / 8, device=None, strict=True):
    """-> (conf_matrix [shape3d, n_query_coarse_grid] int16, fine_location_matrix [shape3d, n_query_coarse_grid, 2] float32) on `device`.

    The reference method reads `self.shape3d`, `self.n_query_coarse_grid`, `self.w_c`, `self.query_img_scale` and `self.coarse_scale`
    (set by `read_anno`, :271-283); they are explicit arguments here.  `assign_matrix` [2, k]: row 0 = index into the 2D keypoints,
    row 1 = (padded) 3D point index.  strict: raise IndexError where the reference's indexing would (one device sync)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("build_assignmatrix runs on the device: the HIP path has no CPU fallback")
    lib = _lib.load()
    kc = torch.as_tensor(keypoints2D_coarse, dtype=torch.float32).to(device).contiguous()
    kf = torch.as_tensor(keypoints2D_fine, dtype=torch.float32).to(device).contiguous()
    am = torch.as_tensor(assign_matrix).long().to(device).contiguous()            # assign_matrix.long()  (:183)
    if am.dim() != 2 or am.shape[0] != 2 or kc.dim() != 2 or kc.shape[1] != 2 or kf.shape != kc.shape:
        raise RuntimeError("build_assignmatrix: keypoints must be [n, 2] and assign_matrix [2, k]")
    scale = torch.as_tensor(query_img_scale, dtype=torch.float32).flatten()
    N, L, k = int(shape3d), int(n_query_coarse_grid), int(am.shape[1])
    conf = torch.empty((N, L), dtype=torch.int16, device=device)
    floc = torch.empty((N, L, 2), dtype=torch.float32, device=device)
    keys = torch.empty(max(k, 1), dtype=torch.int64, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.opp_build_assignmatrix(kc.data_ptr(), kf.data_ptr(), int(kc.shape[0]), am.data_ptr(), k, N, L, int(w_c),
                                              float(scale[1]), float(scale[0]), float(coarse_scale), conf.data_ptr(), floc.data_ptr(),
                                              keys.data_ptr(), status.data_ptr(), torch.cuda.current_stream(device).cuda_stream),
                   "opp_build_assignmatrix")
    if strict and int(status.item()) & 1:
        raise IndexError("build_assignmatrix: a keypoint / point / grid index is out of range (the reference raises here as well)")
    return conf, floc
