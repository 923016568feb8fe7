"""Data-parallel training throughput at the BASELINE configs[4] shape (train_onepose_plus.py with DDP over 8 GPUs, "15k-point clouds"):
every rank runs the training step of bench.py's train leg on its own B = 4 samples of 512 x 512 x N points (train()-mode forward as a graph
of HIP nodes, fine_supervision, Loss, backward, AdamW), gradients averaged by ONE flat all-reduce over RCCL / xGMI
(sharding.GradientAverager) -- the one real exchange step of the whole system.  Launch:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/train_scale.py [--n-points 15000]

Rank 0 prints one JSON line: samples/s over all ranks (weak scaling: per-GPU batch fixed), step time = max over the ranks, and the share of
the step spent in the all-reduce.  tools/scale_check.sh runs it at N = 1, 2, 4, 8 on the first multi-GPU lease (never executed with N > 1:
the build box and the gpurun boxes expose one device)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-points", type=int, default=15000)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    from onepose_plus_plus_amd import OnePosePlus_model, default_config
    from onepose_plus_plus_amd.losses import Loss, fine_supervision
    from onepose_plus_plus_amd.sharding import GradientAverager
    from onepose_plus_plus_amd.synthetic import make_inputs, make_state_dict
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    B, N, hw = args.batch, args.n_points, (512, 512)
    cfg = default_config(thr=0.2)
    model = OnePosePlus_model(cfg).to(dev)
    model.load_state_dict(make_state_dict(cfg, rank), strict=True)          # different on purpose: the averager broadcasts rank 0's state
    model.train()
    parts = [make_inputs(N, hw, 1000 * rank + b) for b in range(B)]          # every rank its own samples
    base = {k: torch.cat([p[k] for p in parts], 0).to(dev) for k in parts[0]}
    g = torch.Generator().manual_seed(9 + rank)
    gt = torch.zeros(B, N, 4096, dtype=torch.int16)
    for b in range(B):
        gt[b, torch.randperm(N, generator=g)[:1500], torch.randperm(4096, generator=g)[:1500]] = 1
    base["conf_matrix_gt"] = gt.to(dev)
    loc = torch.full((B, N, 4096, 2), -50.0, device=dev)
    pos = torch.nonzero(base["conf_matrix_gt"] == 1)
    cell = torch.stack([pos[:, 2] % 64, pos[:, 2] // 64], 1).float() * 8.0
    loc[pos[:, 0], pos[:, 1], pos[:, 2]] = cell + torch.rand(len(pos), 2, device=dev) * 4.0 - 2.0
    base["fine_location_matrix_gt"] = loc
    hparams = {"OnePosePlus": cfg, "loss": {"coarse_type": "focal", "coarse_weight": 1.0, "fine_type": "l2_with_std", "fine_weight": 0.81,
                                            "focal_alpha": 0.5, "focal_gamma": 2.0, "pos_weight": 1.0, "neg_weight": 1.0, "fine_correct_thr": 1.0}}
    loss_mod = Loss(hparams["loss"]).train()
    avg = GradientAverager(model)                                            # broadcast of rank 0's parameters / buffers + the flat gradient buffer
    opt = torch.optim.AdamW(model.parameters(), lr=1e-6)
    t_reduce = [0.0]

    def step():
        d = dict(base)
        model(d)
        fine_supervision(d, hparams)
        loss_mod(d)
        d["loss"].backward()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        avg.average()                                                        # ONE all-reduce of the 40.9 MB flat buffer
        torch.cuda.synchronize(dev)
        t_reduce[0] += time.perf_counter() - t
        opt.step()
        avg.zero()
        return float(d["loss"].detach())

    for _ in range(args.warmup):
        step()
    t_reduce[0] = 0.0
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize(dev)
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    # after the averaged step every rank holds the same parameters
    chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        step_ms = float(el.item()) / args.steps * 1e3
        print(json.dumps({"metric": "training samples/s, data-parallel (BASELINE configs[4] shape)", "value": round(world * B / step_ms * 1e3, 2),
                          "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_ms, 2),
                          "scaling": "weak", "per_gpu_batch": B, "n_points": N, "allreduce_ms_per_step": round(t_reduce[0] / args.steps * 1e3, 3),
                          "parameters_identical_across_ranks": bool(float(lo.item()) == float(hi.item())), "loss": round(loss, 5)}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
