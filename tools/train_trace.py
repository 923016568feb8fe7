"""Training steps alone (bench.py train_leg, no profiler) for `rocprofv3 --kernel-trace --stats`: the per-kernel table of the whole step.
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -o tr -- python tools/train_trace.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    r = bench.train_leg(torch, torch.device("cuda:0"), "bf16x3", nsteps=n)
    print({k: v for k, v in r.items() if k in ("forward_ms", "step_ms")}, flush=True)
