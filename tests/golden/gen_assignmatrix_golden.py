"""Fixtures for the device-side ground-truth builder, produced by the REFERENCE'S OWN method: the source of
`OnePosePlusDataset.build_assignmatrix` is read from /root/reference/src/datasets/OnePosePlus_dataset.py at generation time (the
class itself cannot be imported here: pycocotools / kornia / cv2 are absent), compiled unchanged and called with a stand-in `self`
that carries the five attributes the method reads.  Runs only in the build container:

    python tests/golden/gen_assignmatrix_golden.py

Stores the inputs and the two matrices in sparse form (positions + values of the entries that differ from the fill value).
"""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden.cases import ASSIGN_CASES, make_assign_inputs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("OPP_REFERENCE_ROOT", "/root/reference")


def reference_method():
    path = os.path.join(REF, "src", "datasets", "OnePosePlus_dataset.py")
    tree = ast.parse(open(path).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "OnePosePlusDataset")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "build_assignmatrix")
    mod = ast.Module(body=[fn], type_ignores=[])
    logger = types.SimpleNamespace(warning=lambda *a, **k: None)
    ns = {"torch": torch, "np": np, "logger": logger}
    exec(compile(mod, path, "exec"), ns)
    return ns["build_assignmatrix"]


def main():
    fn = reference_method()
    for name, case in ASSIGN_CASES.items():
        kc, kf, am, meta = make_assign_inputs(case)
        self = types.SimpleNamespace(shape3d=meta["shape3d"], n_query_coarse_grid=meta["L"], w_c=meta["w_c"],
                                     query_img_scale=meta["scale"], coarse_scale=meta["coarse_scale"])
        conf, floc = fn(self, kc, kf, am)
        pos = torch.nonzero(conf)
        fpos = torch.nonzero((floc != -50).any(-1))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), conf_shape=np.array(conf.shape), conf_pos=pos.numpy(),
                            floc_pos=fpos.numpy(), floc_val=floc[fpos[:, 0], fpos[:, 1]].numpy())
        print(name, tuple(conf.shape), "ones:", len(pos), "fine entries:", len(fpos))


if __name__ == "__main__":
    main()
