"""Per-object sharding of the matching forward across the GPUs of one node.

Replaces the reference's Ray fan-out (src/utils/ray_utils.py:10-110,
src/inference/inference_OnePosePlus.py:62-99, inference.py:83-106): one process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests),
objects -> ranks, ONE broadcast of the weights at start-up, no collective in the data path,
results gathered on the host at the end.
"""
import torch
import torch.distributed as dist


def shard_objects(objects, rank, world_size):
    """Objects are independent (inference.py:137-191 loops over them); rank r takes every
    world_size-th object starting at r, which balances clouds of different size on average."""
    return list(objects)[rank::world_size]


def chunk_index(n_items, n_chunks):
    """Contiguous index chunks, sizes differing by at most 1 (the split the reference uses for
    image subsets, src/utils/ray_utils.py:95-103 semantics: every item exactly once)."""
    base, rem = divmod(n_items, n_chunks)
    out, start = [], 0
    for i in range(n_chunks):
        size = base + (1 if i < rem else 0)
        out.append(list(range(start, start + size)))
        start += size
    return out


def broadcast_weights(model, state_dict=None, src=0, group=None):
    """Rank `src` holds `state_dict`; every rank ends with it loaded (strict).  The float
    tensors travel as ONE flat fp32 buffer (40.9 MB for the shipped config: a single RCCL
    broadcast over xGMI); integer counters (num_batches_tracked) as one int64 buffer."""
    rank = dist.get_rank(group)
    ref = model.state_dict()
    device = next(model.parameters()).device
    fkeys = [k for k, v in ref.items() if v.is_floating_point()]
    ikeys = [k for k, v in ref.items() if not v.is_floating_point()]
    fbuf = torch.empty(sum(ref[k].numel() for k in fkeys), dtype=torch.float32, device=device)
    ibuf = torch.empty(len(ikeys), dtype=torch.int64, device=device)
    if rank == src:
        if state_dict is None:
            raise ValueError("source rank needs the state dict")
        missing = [k for k in ref if k not in state_dict]
        if missing:
            raise KeyError("state dict is missing %s" % missing[:3])
        fbuf.copy_(torch.cat([state_dict[k].reshape(-1).float() for k in fkeys]).to(device))
        if ikeys:
            ibuf.copy_(torch.stack([state_dict[k].reshape(()).long() for k in ikeys]).to(device))
    dist.broadcast(fbuf, src=src, group=group)
    if ikeys:
        dist.broadcast(ibuf, src=src, group=group)
    out, off = {}, 0
    for k in fkeys:
        n = ref[k].numel()
        out[k] = fbuf[off:off + n].view(ref[k].shape)
        off += n
    for i, k in enumerate(ikeys):
        out[k] = ibuf[i].clone()
    model.load_state_dict(out, strict=True)
    return model


def gather_results(local_results, dst=0, group=None):
    """Host-side gather of per-object results (small python objects: ids, confidences, poses)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bucket = [None] * world if rank == dst else None
    dist.gather_object(local_results, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    return merged


def run_sharded(objects, forward_fn, group=None, dst=0):
    """objects: {name: payload}.  Each rank runs forward_fn(name, payload) on its shard; rank
    `dst` receives {name: result} for every object."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    names = sorted(objects)
    mine = shard_objects(names, rank, world)
    local = {n: forward_fn(n, objects[n]) for n in mine}
    return gather_results(local, dst=dst, group=group)


class GradientAverager:
    """Data-parallel training (BASELINE configs[4]; the reference wraps `PL_OnePosePlus` in Lightning DDP,
    train_onepose_plus.py): every rank runs the training step on its own B = 4 samples, then the gradients are averaged.

    The 144 parameter gradients (40.9 MB fp32) live in ONE flat buffer - each `p.grad` is a view into it - so a step needs
    ONE all-reduce over xGMI (per-link bound ring: one large message instead of DDP's 25 MB buckets and their per-bucket
    latency) and no copy in or out:

        avg = GradientAverager(model)          # after model.cuda(), before the first backward
        loss.backward(); avg.average(); optimiser.step(); avg.zero()

    No SyncBN, like the reference (plain nn.BatchNorm2d, backbone/resnet.py:25-26): BatchNorm batch statistics stay
    per rank.  What Lightning's DistributedDataParallel does besides averaging is done here as well:
      * construction broadcasts rank `src`'s parameters AND buffers to every rank (`sync_initial=True`), so ranks that
        were initialised or loaded differently start from one state;
      * `sync_buffers()` re-broadcasts the buffers (the BatchNorm running statistics) from rank `src` - DDP's default
        `broadcast_buffers=True` does that before every forward; call it per step for the same behaviour, or before a
        checkpoint is written by a rank other than `src`.
    """

    def __init__(self, module, group=None, src=0, sync_initial=True):
        self.group = group
        self.src = src
        self.module = module
        if sync_initial and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=src, group=group)
            if hasattr(module, "repack"):
                module.repack()                              # packed weights / cached tokens follow the new values
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("parameters must share one device and dtype")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)      # autograd accumulates into the view in place
            off += n

    def sync_buffers(self):
        """broadcast the module's buffers (BatchNorm running statistics, num_batches_tracked) from rank `src`
        (DistributedDataParallel(broadcast_buffers=True) does this before every forward)"""
        if dist.get_world_size(self.group) > 1:
            with torch.no_grad():
                for b in self.module.buffers():
                    dist.broadcast(b.data, src=self.src, group=self.group)
            if hasattr(self.module, "repack"):
                self.module.repack()

    def zero(self):
        """use instead of optimiser.zero_grad(set_to_none=True), which would detach the views"""
        self.flat.zero_()

    def attached(self):
        """True while every p.grad still is its view of the flat buffer"""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                return False
            off += p.numel()
        return True

    def average(self):
        """one all-reduce of the flat buffer; afterwards every rank holds the mean gradient"""
        if not self.attached():
            raise RuntimeError("a parameter's .grad was replaced (zero_grad(set_to_none=True)?): call zero() instead")
        world = dist.get_world_size(self.group)
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(world)
        return self.flat
