"""Pose from 2D-3D matches on the GPU: mirror of the reference's `ransac_PnP`
(/root/reference/src/utils/metric_utils.py:121-204) on top of `opp_pnp_ransac` (csrc/pnp.hip).

Same call signature and return tuple `(pose [3,4], pose_homo [4,4], inliers, state)`; the
matches may be CUDA tensors (they then never leave the device until the 12-number pose is read)
or numpy arrays.  Deterministic for a given `seed`.  No CPU fallback.

Differences from the reference's `cv2.solvePnPRansac(..., flags=SOLVEPNP_EPNP, iterationsCount=10000)`
(parity is accuracy-level only -- cv2 is not installed here, so there are no OpenCV fixtures: "parity unpinned"):
  * minimal solver: Grunert P3P on 3 matches + a 4th for disambiguation, Gauss-Newton refinement on the inliers,
    instead of EPnP on 4+ matches; same hypothesis budget by default (10000), but every hypothesis is evaluated
    (OpenCV stops early at confidence 0.99) and the sampling sequence is a hash of (seed, hypothesis), not cv::RNG;
  * failure convention: fewer than 4 matches, or a degenerate configuration, returns the reference's own
    `cv2.error` branch -- identity pose, empty inliers, state False (metric_utils.py:197-204); when no hypothesis
    gathers 4 inliers the best hypothesis is returned with state True and whatever inliers it has, like OpenCV,
    which reports success with an empty inlier set there;
  * the pycolmap branch (`use_pycolmap_ransac=True`) is not reproduced: the argument is accepted and ignored.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _dev_f32(a, device):
    if torch.is_tensor(a):
        t = a
    else:
        t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device=device, dtype=torch.float32).contiguous()


def ransac_PnP(K, pts_2d, pts_3d, scale=1, pnp_reprojection_error=5, img_hw=None, use_pycolmap_ransac=False,
               iterations=10000, refine_iters=8, seed=0, device=None):
    """K [3,3]; pts_2d [M,2] pixels; pts_3d [M,3].  `img_hw` / `use_pycolmap_ransac` are accepted for
    signature compatibility (the pycolmap branch of the reference is not reproduced)."""
    lib = _lib.load()
    if device is None:
        device = pts_2d.device if torch.is_tensor(pts_2d) and pts_2d.is_cuda else torch.device("cuda", torch.cuda.current_device())
    Kn = K.detach().cpu().numpy() if torch.is_tensor(K) else np.asarray(K)
    Kn = Kn.astype(np.float64)
    K4 = (ctypes.c_double * 4)(Kn[0, 0], Kn[1, 1], Kn[0, 2], Kn[1, 2])
    p2 = _dev_f32(pts_2d, device).reshape(-1, 2)
    p3 = _dev_f32(pts_3d, device).reshape(-1, 3)
    n = int(p2.shape[0])
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        out = torch.empty(12, dtype=torch.float64, device=device)
        mask = torch.empty(max(n, 1), dtype=torch.int32, device=device)
        cnt = torch.zeros(2, dtype=torch.int32, device=device)          # n_inliers, ok
        nbytes = lib.opp_pnp_workspace_bytes(int(iterations))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _lib.check(lib.opp_pnp_ransac(p2.data_ptr(), p3.data_ptr(), n, K4, float(pnp_reprojection_error), float(scale),
                                      int(iterations), int(seed) & 0xFFFFFFFF, int(refine_iters), out.data_ptr(),
                                      mask.data_ptr(), cnt.data_ptr(), cnt.data_ptr() + 4, ws.data_ptr(), nbytes, stream),
                   "opp_pnp_ransac")
        host = torch.cat([out, cnt.double()]).cpu().numpy()              # one D2H copy
    pose = host[:12].reshape(3, 4).copy()
    state = bool(host[13] != 0)
    pose_homo = np.concatenate([pose, np.array([[0.0, 0.0, 0.0, 1.0]])], axis=0)
    if not state:
        return np.eye(4)[:3], np.eye(4), np.array([]).astype(bool), False
    inliers = torch.nonzero(mask[:n]).to(torch.int32).cpu().numpy().reshape(-1, 1)   # OpenCV returns [n_inl, 1] indices
    return pose, pose_homo, inliers, True
