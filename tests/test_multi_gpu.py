"""GPU: the multi-process paths of bench.py / per-object sharding on real devices.

  * world_size 1 over nccl (= RCCL) through the SAME self-launch code path `python bench.py --gpus N` takes (runs on
    the 1-GPU box): the launcher starts torch.distributed.run, the rank initialises the process group, broadcasts
    the weights and reports `n_ranks_seen` / per-rank devices;
  * world_size 2 over nccl when at least two devices are visible (skipped otherwise): weights broadcast from rank 0,
    every rank matches its own object, results gathered -- must equal the single-process results bit for bit."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line_and_detail(stdout, detail_path):
    """bench.py's ONE line (<= 6 KB: numbers only) + the sidecar it wrote (per-rank device records, full legs)"""
    line = [ln for ln in stdout.splitlines() if ln.startswith("{")][-1]
    assert len(line) < 6000
    out = json.loads(line)
    with open(detail_path) as f:
        out["detail_record"] = json.load(f)
    return out


def _run_bench(extra, env_extra=None):
    import tempfile
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.update(env_extra or {})
    env["OPP_BENCH_DETAIL"] = os.path.join(tempfile.mkdtemp(), "bench_detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--cpu-seconds", "0",
                        "--no-roofline", "--no-legs", "--hw", "128", "--n-points", "300"] + extra,
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return _line_and_detail(r.stdout, env["OPP_BENCH_DETAIL"])


def test_bench_plain_and_torchrun_paths_agree():
    plain = _run_bench(["--gpus", "1"])
    assert plain["n_gpus"] == 1 and plain["config"]["n_ranks_seen"] == 1 and plain["value"] > 0
    # the N > 1 launcher path, exercised with N = 1: RANK / WORLD_SIZE set by torch.distributed.run, RCCL process group
    port = str(29000 + os.getpid() % 2000)
    import tempfile
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OPP_BENCH_PIN="1",     # also exercise the NUMA pinning of the N > 1 path
               OPP_BENCH_DETAIL=os.path.join(tempfile.mkdtemp(), "bench_detail.json"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--cpu-seconds", "0", "--no-roofline", "--no-legs", "--hw", "128", "--n-points", "300"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    tr = _line_and_detail(r.stdout, env["OPP_BENCH_DETAIL"])
    devs = tr["detail_record"]["config"]["rank_devices"]
    assert tr["config"]["n_ranks_seen"] == 1 and devs[0]["rank"] == 0
    assert tr["config"]["ranks"] == [[0, 0, devs[0]["device"], devs[0]["images_per_s"]]]      # [rank, local_rank, device, images/s] on the line
    assert tr["value"] > 0 and tr["n_gpus"] == 1 and tr["scaling"] == "weak" and tr["value"] == tr["detail_record"]["value"]
    assert "host_affinity" in devs[0] and tr["config"]["per_rank_images_per_s"]["sum"] > 0
    assert tr["config"]["images_per_step"] == 16 and abs(tr["ms_per_step"] - 16 * tr["ms_per_image"]) < 1e-2 * tr["ms_per_step"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible GPUs")
def test_bench_self_launch_two_ranks():
    out = _run_bench(["--gpus", "2"])
    assert out["n_gpus"] == 2 and out["config"]["n_ranks_seen"] == 2
    recs = out["detail_record"]["config"]["rank_devices"]
    devs = sorted(d["device"] for d in recs)
    assert devs == [0, 1] and len({d["pid"] for d in recs}) == 2
    assert sorted(r[2] for r in out["config"]["ranks"]) == [0, 1]


def _nccl_worker(rank, world, port, q):
    import torch.distributed as dist
    from onepose_plus_plus_amd import OnePosePlus_model, default_config
    from onepose_plus_plus_amd.sharding import broadcast_weights
    from onepose_plus_plus_amd.synthetic import make_state_dict, make_inputs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        cfg = default_config(thr=0.0)
        model = OnePosePlus_model(cfg).eval().to(dev)
        broadcast_weights(model, make_state_dict(cfg, 0) if rank == 0 else None, src=0)
        d = {k: v.to(dev) for k, v in make_inputs(300, (128, 128), 10 + rank).items()}
        with torch.no_grad():
            model(d)
        torch.cuda.synchronize(dev)
        q.put((rank, d["i_ids"].cpu(), d["j_ids"].cpu(), d["mconf"].cpu()))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible GPUs")
def test_sharded_objects_over_rccl_match_single_process():
    import torch.multiprocessing as mp
    from tests import hip_ops as ops
    from onepose_plus_plus_amd import default_config
    from onepose_plus_plus_amd.synthetic import make_state_dict, make_inputs
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() + 7) % 2000
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, i, j, c = q.get(timeout=600)
        res[r] = (i, j, c)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    cfg = default_config(thr=0.0)
    single = ops.make_model(cfg, make_state_dict(cfg, 0))
    for r in range(2):
        out = ops.run_model(single, make_inputs(300, (128, 128), 10 + r))
        assert torch.equal(out["i_ids"].cpu(), res[r][0]) and torch.equal(out["j_ids"].cpu(), res[r][1])
        assert torch.equal(out["mconf"].cpu(), res[r][2])


def _nccl_train_worker(rank, world, port, q, mode):
    """One data-parallel training step (train()-mode HIP forward, fine_supervision, Loss, backward) per rank on a
    rank-specific batch; `mode` = 'averager' (sharding.GradientAverager: one flat all-reduce) or 'ddp'."""
    import torch.distributed as dist
    from tests import helpers as H
    from tests import hip_ops as ops
    from tests.golden.cases import LOSS_CONFIG
    from onepose_plus_plus_amd.losses import Loss, fine_supervision
    from onepose_plus_plus_amd.sharding import GradientAverager
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        cfg, sd, data = H.train_setup("train_b4_64x96_n150")
        model = ops.make_model(cfg, sd).to(dev).train()
        g = torch.Generator().manual_seed(500 + rank)
        data["query_image"] = (data["query_image"] + 0.05 * torch.randn(data["query_image"].shape, generator=g)).clamp(0, 1)
        B, N = data["keypoints3d"].shape[:2]
        data["fine_location_matrix_gt"] = torch.full((B, N, data["conf_matrix_gt"].shape[2], 2), -50.0)
        d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
        hparams = {"OnePosePlus": cfg, "loss": dict(LOSS_CONFIG)}
        loss_mod = Loss(hparams["loss"]).train()
        if mode == "ddp":
            wrapped = torch.nn.parallel.DistributedDataParallel(model, device_ids=[rank], find_unused_parameters=False)
            avg = None
        else:
            wrapped = model
            avg = GradientAverager(model)
        d = wrapped(d)          # DDP copies dict inputs: the results come back in the RETURNED dict (forward returns `data`)
        fine_supervision(d, hparams)
        loss_mod(d)
        d["loss"].backward()
        params = [p for p in model.parameters() if p.requires_grad]
        if avg is not None:
            own = torch.cat([p.grad.flatten() for p in params]).clone()
            avg.average()
        flat = torch.cat([p.grad.flatten() for p in params])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same = all(torch.equal(gathered[0], t) for t in gathered[1:])
        mean_ok = True
        if avg is not None:
            owns = [torch.empty_like(own) for _ in range(world)]
            dist.all_gather(owns, own)
            want = sum(owns) / world
            mean_ok = bool(torch.allclose(flat, want, rtol=1e-6, atol=1e-9)) and not torch.equal(owns[0], owns[1])
        q.put((rank, same, mean_ok, bool(torch.isfinite(flat).all()), float(flat.abs().sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible GPUs")
@pytest.mark.parametrize("mode", ["averager", "ddp"])
def test_data_parallel_training_step_over_rccl(mode):
    """BASELINE configs[4] (Lightning DDP upstream): after the gradient exchange both ranks hold the same gradients (and,
    for the flat averager, exactly the mean of the two ranks' own gradients)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() + (11 if mode == "ddp" else 13)) % 2000
    procs = [ctx.Process(target=_nccl_train_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=900)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in (0, 1):
        same, mean_ok, finite, mass = res[r]
        assert same and mean_ok and finite and mass > 0
