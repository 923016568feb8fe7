"""Kernel-level view of the full coarse-to-fine forward on the committed high-confidence bank (~1200-1500 matches):
    rocprofv3 --kernel-trace --stats -d gpurun_out/fine_prof -- python tools/fine_profile.py [--steps 20] [--repeat-matches K]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from onepose_plus_plus_amd import OnePosePlus_model, default_config  # noqa: E402
from onepose_plus_plus_amd.synthetic import make_inputs, make_state_dict  # noqa: E402
from tests.golden.cases import HIGHCONF_CASES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--precision", default="bf16x3")
    args = ap.parse_args()
    name = "highconf_512x512_n3000"
    hw, n, n_planted, thr, wseed, iseed = HIGHCONF_CASES[name]
    cfg = default_config(thr=thr)
    dev = torch.device("cuda:0")
    model = OnePosePlus_model(cfg).eval().set_gemm_precision(args.precision).to(dev)
    model.load_state_dict(make_state_dict(cfg, wseed), strict=True)
    data = make_inputs(n, hw, iseed)
    data["descriptors3d_coarse_db"] = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))["bank_c_f16"]).float()
    data = {k: v.to(dev) for k, v in data.items()}
    for _ in range(3):
        d = dict(data)
        with torch.no_grad():
            model(d)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(args.steps):
        d = dict(data)
        with torch.no_grad():
            model(d)
    torch.cuda.synchronize()
    print("M = %d, %.3f ms per forward" % (d["mconf"].numel(), (time.perf_counter() - t) / args.steps * 1e3))


if __name__ == "__main__":
    main()
