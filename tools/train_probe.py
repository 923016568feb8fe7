"""Where the training step (bench.py train_leg: BASELINE configs[4] per-GPU shape) spends its device time:
torch.profiler table of one step (HIP forward, fine_supervision, Loss, PyTorch-ops backward, AdamW)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    holder = {}
    real_timed = None

    # reuse train_leg's construction: run it once for warm-up, keeping its closures through a tiny hook
    import onepose_plus_plus_amd.losses as L
    r = bench.train_leg(torch, dev, "bf16x3", nsteps=1)
    print({k: v for k, v in r.items() if k in ("forward_ms", "step_ms")}, flush=True)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        bench.train_leg(torch, dev, "bf16x3", nsteps=1)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))


if __name__ == "__main__":
    main()
