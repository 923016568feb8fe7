"""Where the training step (bench.py train_leg: BASELINE configs[4] per-GPU shape) spends its device time:
torch.profiler table of two warmed-up steps (HIP forward, fine_supervision, Loss, backward, AdamW; set-up and the
forward-only / loss-only timings of that leg stay outside the profiler).
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
    nsteps = 2
    prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
    r = bench.train_leg(torch, dev, "bf16x3", nsteps=nsteps, step_profiler=prof)     # the profiler sees warmed-up steps only
    print({k: v for k, v in r.items() if k in ("forward_ms", "step_ms")}, "-- table below: %d steps" % nsteps, flush=True)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=32, max_name_column_width=70))


if __name__ == "__main__":
    main()
