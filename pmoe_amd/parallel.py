"""Data parallelism for the gradient arena: bucketed all-reduce overlapped with backward.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, xGMI between the 8
GPUs of a node).  The reference has no distributed code at all; data parallelism shards the
minibatch (samples are independent except for BatchNorm statistics, which stay per replica exactly
as the reference's plain nn.BatchNorm2d does) and the only exchange is the mean of the parameter
gradients.

The engine writes gradients into ONE flat f32 arena laid out in backward order (heads first, stem
last).  ``BucketedAllReduce`` cuts it into a few large contiguous buckets and launches each
bucket's all-reduce as soon as the backward pass has moved past it, on RCCL's own stream, while
the remaining (largest: stem, layer1) backward kernels still run.  Few, large collectives are what
the point-to-point xGMI mesh wants (7 links x ~153 GB/s per GPU): a 221 MB f32 arena in 6 buckets
is ~37 MB per collective.
"""
import torch
import torch.distributed as dist


class BucketedAllReduce:
    last_issued = 0          # collectives launched by the most recent backward of this process (bench.py / tests report it)

    def __init__(self, group=None, n_buckets=6, always=False):
        self.group = group
        self.n_buckets = max(1, int(n_buckets))
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.single = self.world == 1 and not (always and dist.is_initialized())    # nothing to exchange
        self._avg_native = dist.is_initialized() and dist.get_backend(group) == "nccl"

    def begin(self, arena):
        self.arena = arena
        n = arena.numel()
        self.bucket = (n + self.n_buckets - 1) // self.n_buckets
        self.sent = 0
        self.works = []

    def ready(self, upto, final=False):
        """Elements [0, upto) of the arena are final: every kernel that writes them is enqueued on the CURRENT stream or on
        a stream the current one already waits for.  (The collective orders itself after the current stream at the moment
        of the call; with weight gradients on a side stream the engine calls this from that side stream after making it
        wait for the main one -- engine.backward -- so both producers are covered.)"""
        if self.single:
            return
        n = self.arena.numel()
        limit = n if final else (min(upto, n) // self.bucket) * self.bucket
        while self.sent < limit:
            hi = min(n, self.sent + self.bucket)
            chunk = self.arena[self.sent:hi]
            if self._avg_native:
                w = dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            else:
                w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.works.append((w, chunk))
            self.sent = hi

    def finish(self):
        """Flush the tail and make the current stream wait for every bucket."""
        if self.single:
            return
        self.ready(self.arena.numel(), final=True)
        for w, chunk in self.works:
            w.wait()
            if not self._avg_native:
                chunk.mul_(1.0 / self.world)
        BucketedAllReduce.last_issued = len(self.works)
        self.works = []


def shard_batch(global_batch, rank, world):
    """Rows [lo, hi) of the global minibatch owned by ``rank`` (equal shards; weak scaling keeps the
    per-GPU batch fixed, so the bench passes global_batch = per_gpu * world)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


@torch.no_grad()
def sync_bn_buffers(model, group=None, mode="mean"):
    """BatchNorm running statistics diverge per replica (local batch stats, as in the reference).
    Before checkpointing/evaluation average them (mode="mean") or take rank 0's (mode="rank0")."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    bufs = [b for n, b in model.named_buffers() if n.endswith("running_mean") or n.endswith("running_var")]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    if mode == "mean":
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= dist.get_world_size(group)
    else:
        dist.broadcast(flat, src=0, group=group)
    o = 0
    for b in bufs:
        b.copy_(flat[o:o + b.numel()].view_as(b))
        o += b.numel()
