#!/usr/bin/env python
"""Throughput bench of the 2D-3D matching forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): one object, 512x512 query image x 5000-point cloud,
coarse-match only (`fine_matching.enable=False`), B=1 per forward, synthetic image/bank and
seeded random weights (no datasets or checkpoints are available offline).  A step is one
`OnePosePlus_model(data)` forward through the HIP path with image and descriptor banks already
resident in HBM.  With N GPUs every rank owns a different synthetic object (per-object sharding,
SURVEY.md §8e): weights are broadcast once from rank 0 over RCCL, then there is no collective in
the data path ("scaling": "weak").  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N=1 only):
  roofline     - HIP-event timing (events recorded on the launch stream by libopp_hip.so) of
                 every launch of the dominant kernel symbol, the 128x128-tile implicit-GEMM
                 conv, during the timed steps; algorithmic FLOPs use unpadded channel counts.
  cpu_baseline - the CPU oracle (oracle/onepose_oracle.py, a port of the reference PyTorch
                 path) timed on the host cores over a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense (32x32x16)
# GEMM kernel symbols profiled per precision: (tile cfg, conv, name, peak in ALGORITHMIC TFLOP/s).  The one with
# the largest total time per forward is reported as `roofline`, the rest under `roofline.other_kernels`.
_P32 = PEAK_F32_MFMA_TFLOPS
_P16 = PEAK_F16_MFMA_TFLOPS / 3.0   # three fp16 MFMA products per algorithmic multiply-add
SYMBOLS = {
    "fp32": [(25, 1, "opp_gemm_kernel<128,128,4,2,conv> (fp32 MFMA implicit-GEMM 3x3 conv, 8 waves)", _P32, "128, 128, 4, 2, true, 0, 2, false"),
             (11, 1, "opp_gemm_kernel<128,128,2,2,conv,depth4> (fp32 MFMA implicit-GEMM 3x3 conv)", _P32, "128, 128, 2, 2, true, 0, 4, false"),
             (1, 1, "opp_gemm_kernel<64,128,2,2,conv> (fp32 MFMA implicit-GEMM conv)", _P32, "64, 128, 2, 2, true, 0, 2, false")],
    "fp16x2": [(25, 1, "opp_gemm_kernel<128,128,4,2,conv,fp16x2> (3x fp16 MFMA implicit-GEMM conv, 8 waves)", _P16, "128, 128, 4, 2, true, 0, 2, true"),
               (2, 1, "opp_gemm_kernel<64,64,2,2,conv,fp16x2> (3x fp16 MFMA implicit-GEMM conv)", _P16, "64, 64, 2, 2, true, 0, 2, true"),
               (25, 0, "opp_gemm_kernel<128,128,4,2,dense,fp16x2> (3x fp16 MFMA GEMM, 8 waves)", _P16, "128, 128, 4, 2, false, 0, 2, true")],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--n-points", type=int, default=5000)
    ap.add_argument("--hw", type=int, default=512)
    ap.add_argument("--fine", action="store_true", help="full coarse-to-fine forward instead of configs[1]")
    ap.add_argument("--thr", type=float, default=0.1)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the short exact-fp32 GEMM comparison run")
    ap.add_argument("--precision", default=None, choices=["fp32", "fp16x2", "fp16x2_all"],
                    help="GEMM arithmetic (default: the module default / OPP_GEMM_PRECISION)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("OPP_BENCH_STREAMS", "3")),
                    help="B=1 forwards kept in flight per GPU on separate HIP streams (the reference runs "
                         "2 Ray workers per GPU, inference_OnePosePlus.py:18-26); steps are split evenly")
    args = ap.parse_args()

    from onepose_plus_plus_amd import OnePosePlus_model, default_config, _lib
    from onepose_plus_plus_amd.synthetic import make_state_dict, make_inputs

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = default_config(thr=args.thr, fine=args.fine)
    model = OnePosePlus_model(cfg).eval()
    if args.precision:
        model.set_gemm_precision(args.precision)
    precision = model.gemm_precision
    symbols = SYMBOLS["fp32" if precision == "fp32" else "fp16x2"]
    sd = make_state_dict(cfg, 0) if rank == 0 else None
    model = model.to(dev)
    if dist is not None:
        from onepose_plus_plus_amd.sharding import broadcast_weights
        broadcast_weights(model, sd, src=0)       # ONE broadcast of the flat weight buffer (RCCL over xGMI)
    else:
        model.load_state_dict(sd, strict=True)
    n_streams = max(1, args.streams)
    models = [model]
    for _ in range(1, n_streams):        # one module (own workspace / outputs) per in-flight forward
        m = OnePosePlus_model(cfg).eval().set_gemm_precision(precision).to(dev)
        m.load_state_dict(model.state_dict(), strict=True)
        models.append(m)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [None]

    # one synthetic object per rank (seed = rank), a small pool of query images resident in HBM
    n_img = 4
    datas = []
    for i in range(n_img):
        d = make_inputs(args.n_points, (args.hw, args.hw), seed=1 + 1000 * rank)
        if i:
            g = torch.Generator().manual_seed(77 + i + 1000 * rank)
            d["query_image"] = torch.rand(1, 1, args.hw, args.hw, generator=g)
        datas.append({k: v.to(dev) for k, v in d.items()})
    bank = {k: datas[0][k] for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    for d in datas:
        d.update(bank)          # the bank is per-object constant: one resident copy

    def step(i, slot=0):
        d = dict(datas[i % n_img])
        with torch.no_grad():
            if streams[slot] is None:
                models[slot](d)
            else:
                with torch.cuda.stream(streams[slot]):
                    models[slot](d)
        return d

    def run_steps(n):
        """n forwards; with several streams one host thread per stream keeps its forward in flight
        (the forward's single D2H sync of the match count releases the GIL)."""
        if n_streams == 1:
            last = None
            for i in range(n):
                last = step(i)
            return last
        import threading
        out = [None] * n_streams

        def worker(slot):
            torch.cuda.set_device(dev)
            for i in range(n // n_streams + (1 if slot < n % n_streams else 0)):     # exactly n forwards in total
                out[slot] = step(i * n_streams + slot, slot)
            streams[slot].synchronize()
        th = [threading.Thread(target=worker, args=(k,)) for k in range(n_streams)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return out[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        for k in range(n_streams):
            step(i, k)
    lib = _lib.load()
    prof = (not args.no_roofline) and rank == 0 and world == 1
    prof_in_timed = prof and n_streams == 1

    def prof_stop():
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        _lib.check(lib.opp_profile_stop(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)), "profile_stop")
        return ms.value, fl.value, n.value

    barrier()
    if prof_in_timed:
        _lib.check(lib.opp_profile_start(symbols[0][0], symbols[0][1], args.steps * 32), "profile_start")
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    roof = None
    if prof:
        # HIP events recorded by the library on the launch stream around every launch of one kernel symbol.
        # With several forwards in flight kernels of different streams share the CUs, so a kernel's own launch
        # duration is only meaningful on its own: the symbols are then timed in single-stream passes of the
        # same steps right after the timed region (`measured` says which).
        meas = []
        for si, (cfg_id, conv, name, peak, tmpl) in enumerate(symbols):
            if si == 0 and prof_in_timed:
                ms, fl, n = prof_stop()
                how = "timed region"
                nsteps = args.steps
            else:
                nsteps = min(args.steps, 40)
                _lib.check(lib.opp_profile_start(cfg_id, conv, nsteps * 32), "profile_start")
                for i in range(nsteps):
                    step(i, 0)
                torch.cuda.synchronize(dev)
                ms, fl, n = prof_stop()
                how = "single-stream pass of %d steps after the timed region" % nsteps
            if n > 0 and ms > 0:
                ach = fl / (ms * 1e-3) / 1e12
                meas.append({"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                             "frac": round(ach / peak, 4), "traffic": None, "kernel": name, "symbol": tmpl,
                             "measured": "HIP events on the launch stream, " + how,
                             "launches": n, "launches_per_step": round(n / nsteps, 2), "avg_launch_us": round(ms * 1e3 / n, 2),
                             "us_per_step": round(ms * 1e3 / nsteps, 1), "alg_gflop_per_launch": round(fl / n / 1e9, 3)})
        if meas:
            meas.sort(key=lambda m: -m["us_per_step"])
            roof = meas[0]
            tpath = os.path.join(ROOT, "profiles", "traffic_gemm_symbols_%s.json" % ("fp32" if precision == "fp32" else "fp16x2"))
            if os.path.exists(tpath):     # HBM bytes per launch from the committed rocprofv3 --pmc passes
                with open(tpath) as f:
                    tr = json.load(f)
                for m in meas:
                    m["traffic"] = tr.get(m["symbol"])
            roof["other_kernels"] = meas[1:]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    fp32_leg = None
    if rank == 0 and world == 1 and precision != "fp32" and not args.no_fp32_leg:
        # the same forwards with the exact-fp32 MFMA GEMMs, for reference beside the headline value
        for m in models:
            m.set_gemm_precision("fp32").to(dev)
        for k in range(n_streams):
            step(0, k)
        torch.cuda.synchronize(dev)
        n32 = min(args.steps, 30)
        t1 = time.perf_counter()
        run_steps(n32)
        torch.cuda.synchronize(dev)
        fp32_leg = {"value": round(n32 / (time.perf_counter() - t1), 3), "unit": "images/s", "steps": n32,
                    "gemm_precision": "fp32"}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cpu = cpu_baseline(cfg, make_state_dict(cfg, 0), args, make_inputs)

    if rank == 0:
        total = args.steps * world
        flops_img = 2 * (126.726e9 + 6 * (4096 + args.n_points) * 671744 + args.n_points * 4096 * 256)
        out = {
            "metric": "query images/sec (2D-3D match fwd) at 512x512 img x 5k pts",
            "value": round(total / elapsed, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if precision == "fp32" else "f32 (GEMM operands as hi+lo fp16 pairs, fp32 accumulate)",
            "data": "synthetic (seeded random image, descriptor bank and weights)",
            "config": {"workload": "configs[1]: single object, %dx%d image x %d points, coarse-match only%s, B=1 per forward, "
                                   "%d forward(s) in flight per GPU, one object per GPU"
                                   % (args.hw, args.hw, args.n_points, " + fine refine" if args.fine else "", n_streams),
                       "streams_per_gpu": n_streams, "gemm_precision": precision, "fp32_gemm_leg": fp32_leg,
                       "matches_last_step": int(last["mconf"].numel()),
                       "model_gflop_per_image": round(flops_img / 1e9, 1),
                       "model_tflops": round(flops_img * total / elapsed / 1e12, 2),
                       "model_frac_of_mfma_peak": round(flops_img * total / elapsed / 1e12 / world /
                                                        (PEAK_F32_MFMA_TFLOPS if precision == "fp32" else PEAK_F16_MFMA_TFLOPS / 3.0), 4),
                       "mfma_peak_tflops_for_this_arithmetic": round(PEAK_F32_MFMA_TFLOPS if precision == "fp32" else PEAK_F16_MFMA_TFLOPS / 3.0, 1)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(cfg, sd, args, make_inputs):
    """Oracle (port of the reference PyTorch CPU path) timed on the host cores over a bounded
    sample: forwards of the SAME workload for ~cpu_seconds.  The thread count is picked by a
    short calibration (a 256-thread pool on this class of box is slower than a small one), and
    the count actually used is reported as `cores`."""
    from oracle import onepose_oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    data = make_inputs(args.n_points, (args.hw, args.hw), seed=1)
    small = make_inputs(500, (128, 128), seed=1)
    best = None
    for th in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        O.forward(sd, dict(small), cfg)
        t = time.perf_counter()
        O.forward(sd, dict(small), cfg)
        dt = time.perf_counter() - t
        if best is None or dt < best[1]:
            best = (th, dt)
    torch.set_num_threads(best[0])
    O.forward(sd, dict(data), cfg)            # warm-up at full size
    times = []
    t_end = time.perf_counter() + args.cpu_seconds
    while len(times) < 2 or (time.perf_counter() < t_end and len(times) < 50):
        d = dict(data)
        t = time.perf_counter()
        O.forward(sd, d, cfg)
        times.append(time.perf_counter() - t)
    med = sorted(times)[len(times) // 2]
    return {"value": round(1.0 / med, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d forwards of the same %dx%d x %d-pt workload through oracle/onepose_oracle.py "
                      "(fp32 PyTorch CPU, %d of %d available threads), median; min %.3f s"
                      % (len(times), args.hw, args.hw, args.n_points, best[0], avail, min(times))}


if __name__ == "__main__":
    main()
