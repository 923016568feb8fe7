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
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=70))
    phases(dev)


def phases(dev):
    """wall time of the step's phases with a device synchronisation after each (host-bound vs device-bound)"""
    import time
    from onepose_plus_plus_amd import OnePosePlus_model, default_config
    from onepose_plus_plus_amd.losses import Loss, fine_supervision
    from onepose_plus_plus_amd.synthetic import make_inputs, make_state_dict
    B, N, hw = 4, 7000, (512, 512)
    cfg = default_config(thr=0.2)
    model = OnePosePlus_model(cfg).set_gemm_precision("bf16x3").to(dev)
    model.load_state_dict(make_state_dict(cfg, 0), strict=True)
    model.train()
    parts = [make_inputs(N, hw, 30 + b) for b in range(B)]
    base = {k: torch.cat([p[k] for p in parts], 0).to(dev) for k in parts[0]}
    g = torch.Generator().manual_seed(9)
    gt = torch.zeros(B, N, 4096, dtype=torch.int16)
    for b in range(B):
        gt[b, torch.randperm(N, generator=g)[:1500], torch.randperm(4096, generator=g)[:1500]] = 1
    base["conf_matrix_gt"] = gt.to(dev)
    base["fine_location_matrix_gt"] = torch.full((B, N, 4096, 2), -50.0, device=dev)
    hparams = {"OnePosePlus": cfg, "loss": {"coarse_type": "focal", "coarse_weight": 1.0, "fine_type": "l2_with_std", "fine_weight": 0.81,
                                            "focal_alpha": 0.5, "focal_gamma": 2.0, "pos_weight": 1.0, "neg_weight": 1.0, "fine_correct_thr": 1.0}}
    loss_mod = Loss(hparams["loss"]).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-6)
    acc = {}
    for it in range(4):
        marks = []

        def mark(name):
            t_host = time.perf_counter()
            torch.cuda.synchronize(dev)
            marks.append((name, t_host, time.perf_counter()))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        d = dict(base)
        model(d)
        mark("forward")
        fine_supervision(d, hparams)
        loss_mod(d)
        mark("supervision+loss")
        opt.zero_grad(set_to_none=True)
        d["loss"].backward()
        mark("backward")
        opt.step()
        mark("adamw")
        prev = t0
        if it:
            for name, th, ts in marks:
                a = acc.setdefault(name, [0.0, 0.0])
                a[0] += (th - prev) * 1e3           # host time to ENQUEUE the phase
                a[1] += (ts - prev) * 1e3           # until the device finished it
                prev = ts
    print("phase: host-enqueue ms / until-device-done ms (mean of 3 steps, a sync after every phase)")
    for name, (h, t) in acc.items():
        print("  %-18s %7.2f / %7.2f" % (name, h / 3, t / 3))


if __name__ == "__main__":
    main()
