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


# ---- DistributedDataParallel around a module shaped like OnePosePlus_model in train() mode -----------------------------------
class _ScaleFn(torch.autograd.Function):
    """stand-in for the HIP nodes of train_autograd.py: a custom autograd.Function that takes PARAMETERS as inputs and returns
    their gradients from its own backward (the reducer hooks of DDP must fire on them like on any leaf)"""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return g @ w, g.t() @ x, g.sum(0)


class _DictModule(torch.nn.Module):
    """same calling convention as OnePosePlus_model.forward: results are written into the `data` dict, which is also returned"""

    def __init__(self):
        super().__init__()
        self.w1 = torch.nn.Parameter(torch.randn(8, 6))
        self.b1 = torch.nn.Parameter(torch.zeros(8))
        self.w2 = torch.nn.Parameter(torch.randn(3, 8))
        self.b2 = torch.nn.Parameter(torch.zeros(3))

    def forward(self, data):
        h = torch.relu(_ScaleFn.apply(data["x"], self.w1, self.b1))
        data["out"] = _ScaleFn.apply(h, self.w2, self.b2)
        return data


def _ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(7 + rank)                       # ranks start from DIFFERENT parameters: DDP must broadcast rank 0's
        model = _DictModule()
        ddp = torch.nn.parallel.DistributedDataParallel(model)
        g = torch.Generator().manual_seed(40 + rank)
        d = {"x": torch.randn(5, 6, generator=g), "meta": "kept"}
        out = ddp(d)                                      # DDP rebuilds dict inputs: the RETURNED dict carries the results
        returned_has_out = "out" in out
        caller_dict_has_out = "out" in d                  # documents why callers must use the return value
        out["out"].square().sum().backward()
        flat = torch.cat([p.grad.flatten() for p in model.parameters()])
        # the mean of the two ranks' own gradients, from an un-wrapped copy of the synchronised parameters
        ref = _DictModule()
        ref.load_state_dict(model.state_dict())
        o = ref({"x": d["x"]})
        o["out"].square().sum().backward()
        own = torch.cat([p.grad.flatten() for p in ref.parameters()])
        owns = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(owns, own)
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        psame = [torch.empty(sum(p.numel() for p in model.parameters())) for _ in range(world)]
        dist.all_gather(psame, torch.cat([p.detach().flatten() for p in model.parameters()]))
        q.put((rank, returned_has_out, caller_dict_has_out, bool(torch.equal(gathered[0], gathered[1])),
               bool(torch.allclose(flat, sum(owns) / world, rtol=1e-6, atol=1e-8)), bool(torch.equal(psame[0], psame[1])),
               not torch.equal(owns[0], owns[1])))
    finally:
        dist.destroy_process_group()


def test_ddp_around_a_dict_mutating_module_with_custom_autograd_functions():
    """The two DDP behaviours the training path depends on (tests/test_multi_gpu.py runs the real module over RCCL when two GPUs
    are visible): (1) DistributedDataParallel copies dict inputs, so results written into `data` reach the caller only through the
    RETURN value -- OnePosePlus_model.forward returns `data` for that reason; (2) gradients produced by custom autograd.Functions
    that take parameters as inputs are averaged by DDP's reducer like any others."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=300)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        returned_has_out, caller_has_out, same, is_mean, params_same, ranks_differ = res[r]
        assert returned_has_out and same and is_mean and params_same and ranks_differ, (r, res[r])
