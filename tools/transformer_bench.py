"""Times the coarse / fine transformer stage alone (opp_transformer), fused encoder layers vs launch-per-Linear.
    python tools/transformer_bench.py [--n 5000] [--reps 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onepose_plus_plus_amd import OnePosePlus_model, default_config, _lib       # noqa: E402
from onepose_plus_plus_amd.synthetic import make_state_dict                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--L", type=int, default=4096)
    ap.add_argument("--m", type=int, default=1500, help="matches of the fine-level run")
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    cfg = default_config()
    sd = make_state_dict(cfg, 0)
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    for which, n_seg, len0, len1, C in ((0, 1, args.L, args.n, 256), (1, args.m, 25, 1, 128)):
        for fusion in (2, 1, 0):
            m = OnePosePlus_model(cfg).eval().set_encoder_fusion(fusion)
            m.load_state_dict(sd, strict=True)
            m = m.to(dev)
            _, ctx = m._ensure_ready(dev)
            x0 = torch.randn(n_seg * (len0 + len1), C, device=dev)
            nb = lib.opp_transformer_workspace_bytes(ctx, which, n_seg, len0, len1)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            s = torch.cuda.current_stream(dev).cuda_stream
            xs = [x0.clone() for _ in range(args.reps + 3)]
            for i in range(3):
                _lib.check(lib.opp_transformer(ctx, which, xs[i].data_ptr(), n_seg, len0, len1, ws.data_ptr(), nb, s), "transformer")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.reps):
                _lib.check(lib.opp_transformer(ctx, which, xs[3 + i].data_ptr(), n_seg, len0, len1, ws.data_ptr(), nb, s), "transformer")
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.reps
            nl = cfg["loftr_coarse" if which == 0 else "loftr_fine"]
            layers = len(nl["layer_names"]) * nl["layer_iter_n"]
            T = n_seg * (len0 + len1)
            gflop = 2.0 * T * 10.25 * C * C * layers / 1e9
            print("%s transformer  T=%d  fusion=%s : %.1f us  (%.1f us/layer, %.1f TFLOP/s)" %
                  ("coarse" if which == 0 else "fine", T, fusion, us, us / layers, gflop / us * 1e-3 * 1e3))


if __name__ == "__main__":
    main()
