"""Resident per-object 3D feature bank.

The reference re-reads nothing per image but re-UPLOADS both descriptor banks for every query
(`inference_OnePosePlus_worker.py:53-56`: `data_c = {k: v.cuda() ...}`), 7.7 MB at 5k points.  Here the
bank is read once from the reference's on-disk format, kept on the device in the layout the module
consumes in place, and handed to every forward; the module's per-object token cache (model.py
`_object_tokens`) then also skips the image-independent keypoint encoding.

On-disk format (written by `save_3d_anno`, src/sfm_utils/postprocess/feature_process.py:316-319, read by
`OnePosePlusInferenceDataset.read_anno3d`, src/datasets/OnePosePlus_inference_dataset.py:109-160):
    anno_3d_average.npz          keypoints3d [N,3] (float64), descriptors3d [128,N], scores3d [N,1]
    anno_3d_average_coarse.npz   descriptors3d [256,N], scores3d [N,1]          (same stem + "_coarse")
"""
import os.path as osp

import numpy as np
import torch


class ObjectBank:
    def __init__(self, keypoints3d, descriptors3d, descriptors3d_coarse=None, scores3d=None, device=None):
        """keypoints3d [N,3]; descriptors3d [128,N] (fine); descriptors3d_coarse [256,N] or None."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        f32 = lambda a: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(torch.float32)  # noqa: E731
        kp, df = f32(keypoints3d), f32(descriptors3d)
        if kp.dim() != 2 or kp.shape[1] != 3 or df.dim() != 2 or df.shape[1] != kp.shape[0]:
            raise ValueError("bank shapes: keypoints3d [N,3], descriptors3d [dim,N]")
        self.num_3d_orig = kp.shape[0]
        self.keypoints3d = kp[None].contiguous().to(dev)                    # [1,N,3]
        self.descriptors3d_db = df[None].contiguous().to(dev)               # [1,128,N] channel-major, as the dataset yields it
        self.descriptors3d_coarse_db = None
        if descriptors3d_coarse is not None:
            dc = f32(descriptors3d_coarse)
            if dc.dim() != 2 or dc.shape[1] != kp.shape[0]:
                raise ValueError("coarse bank must be [dim,N]")
            self.descriptors3d_coarse_db = dc[None].contiguous().to(dev)    # [1,256,N]
        self.scores3d = None if scores3d is None else f32(scores3d)

    @classmethod
    def from_npz(cls, avg_anno3d_file, shape3d=None, pad=True, load_3d_coarse=True, device=None, generator=None):
        """Mirrors `read_anno3d` (OnePosePlus_inference_dataset.py:109-160), including its only size rule:
        with `pad` and more than `shape3d` points, `shape3d` points are drawn WITH replacement
        (`pad_keypoints3d_random`, src/utils/data_utils.py:212-223: torch.randint); fewer points are kept as
        they are (nothing is padded despite the name)."""
        avg = np.load(avg_anno3d_file)
        kp = torch.Tensor(avg["keypoints3d"])
        desc = torch.Tensor(avg["descriptors3d"])
        scores = torch.Tensor(avg["scores3d"])
        coarse = None
        if load_3d_coarse:
            stem, ext = osp.splitext(avg_anno3d_file)
            coarse = torch.Tensor(np.load(stem + "_coarse" + ext)["descriptors3d"])
        n_orig = kp.shape[0]
        if pad and shape3d is not None and shape3d - n_orig < 0:
            idx = torch.randint(n_orig, (shape3d,), generator=generator)
            kp, desc, scores = kp[idx], desc[:, idx], scores[idx, :]
            if coarse is not None:
                coarse = coarse[:, idx]
        bank = cls(kp, desc, coarse, scores, device=device)
        bank.num_3d_orig = n_orig
        return bank

    @staticmethod
    def save_npz(avg_anno3d_file, keypoints3d, descriptors3d, scores3d, descriptors3d_coarse=None, scores3d_coarse=None):
        """Writes the reference's format (save_3d_anno, feature_process.py:316-319); used by tests and tools."""
        np.savez(avg_anno3d_file, keypoints3d=np.asarray(keypoints3d), descriptors3d=np.asarray(descriptors3d),
                 scores3d=np.asarray(scores3d))
        if descriptors3d_coarse is not None:
            stem, ext = osp.splitext(avg_anno3d_file)
            np.savez(stem + "_coarse" + ext, keypoints3d=np.asarray(keypoints3d),
                     descriptors3d=np.asarray(descriptors3d_coarse),
                     scores3d=np.asarray(scores3d if scores3d_coarse is None else scores3d_coarse))

    def data(self, query_image, query_image_scale=None, **extra):
        """The dict `OnePosePlus_model(data)` expects (dataset `__getitem__`, :178-222), around tensors that
        already live on the device: `query_image` [1,H,W] or [1,1,H,W] float32."""
        if query_image.dim() == 3:
            query_image = query_image[None]
        d = {"keypoints3d": self.keypoints3d, "descriptors3d_db": self.descriptors3d_db, "query_image": query_image}
        if self.descriptors3d_coarse_db is not None:
            d["descriptors3d_coarse_db"] = self.descriptors3d_coarse_db
        if query_image_scale is not None:
            s = torch.as_tensor(query_image_scale, dtype=torch.float32)
            d["query_image_scale"] = (s[None] if s.dim() == 1 else s).to(query_image.device)
        d.update(extra)
        return d
