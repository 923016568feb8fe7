"""CPU, world_size 2 over gloo: the N>1 path of bench.py / per-object sharding -- weight
broadcast from rank 0, object partition, result gather.  (On the GPU box the same code runs
with backend "nccl" = RCCL.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from onepose_plus_plus_amd import OnePosePlus_model, default_config
from onepose_plus_plus_amd.sharding import (broadcast_weights, chunk_index, run_sharded, shard_objects)
from onepose_plus_plus_amd.synthetic import make_state_dict


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                 # different random init on every rank
        cfg = default_config()
        model = OnePosePlus_model(cfg).eval()
        sd = make_state_dict(cfg, 0) if rank == 0 else None
        broadcast_weights(model, sd, src=0)
        gold = make_state_dict(cfg, 0)
        same = all(torch.equal(v, gold[k]) for k, v in model.state_dict().items())
        objects = {"obj%02d" % i: i for i in range(7)}

        def fwd(name, payload):                       # stand-in for the per-object forward
            return {"rank": rank, "value": payload * payload}

        merged = run_sharded(objects, fwd)
        q.put((rank, same, merged))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, same, merged = q.get(timeout=300)
        res[rank] = (same, merged)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][0] and res[1][0]                    # both ranks hold rank 0's weights
    merged = res[0][1]
    assert res[1][1] is None
    assert sorted(merged) == ["obj%02d" % i for i in range(7)]
    for i in range(7):
        assert merged["obj%02d" % i]["value"] == i * i
        assert merged["obj%02d" % i]["rank"] == i % 2   # objects -> ranks round-robin


def test_partition_helpers():
    objs = list(range(10))
    parts = [shard_objects(objs, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == objs and max(map(len, parts)) - min(map(len, parts)) <= 1
    ch = chunk_index(10, 3)
    assert sum(ch, []) == list(range(10)) and [len(c) for c in ch] == [4, 3, 3]
    assert chunk_index(2, 4) == [[0], [1], [], []]


def _train_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from onepose_plus_plus_amd.sharding import GradientAverager
        cfg = default_config()
        model = OnePosePlus_model(cfg).train()
        model.load_state_dict(make_state_dict(cfg, rank), strict=True)     # ranks start from DIFFERENT states ...
        avg = GradientAverager(model)                                      # ... construction broadcasts rank 0's
        gold = make_state_dict(cfg, 0)
        init_ok = all(torch.equal(v, gold[k]) for k, v in model.state_dict().items())
        with torch.no_grad():                                              # per-rank BatchNorm running statistics drift
            model.get_buffer("backbone.bn1.running_mean").add_(float(rank + 1))
        avg.sync_buffers()                                                 # DDP broadcast_buffers=True equivalent
        init_ok = init_ok and torch.equal(model.get_buffer("backbone.bn1.running_mean"), gold["backbone.bn1.running_mean"] + 1.0)
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        opt = torch.optim.SGD(model.parameters(), lr=0.5)

        def weights(r):                               # rank-dependent pseudo-loss: grad of p = w_r(p)
            g = torch.Generator().manual_seed(1000 + r)
            return [torch.randn(p.shape, generator=g) for p in avg.params]

        loss = sum((p * w).sum() for p, w in zip(avg.params, weights(rank)))
        loss.backward()
        attached_after_backward = avg.attached()
        own = [p.grad.clone() for p in avg.params]
        avg.average()
        want = [sum(weights(r)[i] for r in range(world)) / world for i in range(len(avg.params))]
        mean_ok = all(torch.allclose(p.grad, w, rtol=0, atol=1e-6) for p, w in zip(avg.params, want))
        own_ok = all(torch.equal(o, w) for o, w in zip(own, weights(rank)))
        opt.step()
        digest = float(sum(p.detach().double().sum() for p in avg.params))
        avg.zero()
        zero_ok = all(float(p.grad.abs().max()) == 0.0 for p in avg.params) and avg.attached()
        opt.zero_grad(set_to_none=True)               # detaches the views -> average() must refuse
        try:
            avg.average()
            refused = False
        except RuntimeError:
            refused = True
        q.put((rank, len(names), int(avg.flat.numel()), attached_after_backward, own_ok, mean_ok, zero_ok, refused and init_ok, digest))
    finally:
        dist.destroy_process_group()


def test_gradient_averager_world2():
    """Data-parallel training step: ONE all-reduce of the flat 40.9 MB gradient buffer, identical parameters afterwards."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=300)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
    assert set(res) == {0, 1}
    for r in (0, 1):
        n, numel, attached, own_ok, mean_ok, zero_ok, refused, _ = res[r]
        assert n == 144 and numel * 4 > 40e6                       # the whole model in one buffer
        assert attached and own_ok and mean_ok and zero_ok and refused
    assert res[0][-1] == res[1][-1]                                 # same parameters on both ranks after the step
