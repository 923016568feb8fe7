"""Where the training step (bench.py train_leg: BASELINE configs[4] per-GPU shape) spends its device time:
torch.profiler table of one train_leg pass (HIP forward, fine_supervision, Loss, backward, AdamW; the pass also
contains the forward-only and loss-only timings of that leg).
    python tools/train_probe.py > gpurun_out/train_probe.txt"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    r = bench.train_leg(torch, dev, "bf16x3", nsteps=1)          # warm-up (allocator, MIOpen find)
    print({k: v for k, v in r.items() if k in ("forward_ms", "step_ms")}, flush=True)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        bench.train_leg(torch, dev, "bf16x3", nsteps=1)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))


if __name__ == "__main__":
    main()
