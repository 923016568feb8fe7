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
