"""Data-parallel plumbing on CPU (gloo, world_size 2): the bucketed gradient all-reduce that the engine
drives from backward, batch sharding, and BatchNorm-buffer sync.  No GPU kernels are involved."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pmoe_amd.parallel import BucketedAllReduce, shard_batch, sync_bn_buffers


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, n_buckets, q, mode="rs_ag", cuts=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        arena = torch.randn(n, generator=g)
        mine = arena.clone()
        red = BucketedAllReduce(None, n_buckets, mode=mode)
        red.begin(arena, cuts)
        # backward fills the arena front to back; buckets fly as soon as the prefix passes them
        sent_before_finish = 0
        for upto in range(0, n + 1, max(1, n // 7)):
            red.ready(upto)
            sent_before_finish = red.sent
        red.finish()
        others = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        expect = sum(others) / world
        ok = torch.allclose(arena, expect, atol=1e-6)
        # shard_batch partitions the global batch without overlap
        lo, hi = shard_batch(8 * world, rank, world)
        # BN buffer sync
        bn = torch.nn.BatchNorm2d(4)
        bn.running_mean.fill_(float(rank))
        sync_bn_buffers(bn, mode="mean")
        ok_bn = torch.allclose(bn.running_mean, torch.full((4,), (world - 1) / 2.0))
        q.put((rank, bool(ok), sent_before_finish, (lo, hi), bool(ok_bn), float((mine - arena).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,n_buckets,mode,cuts", [
    (1000, 6, "rs_ag", None),                  # reduce-scatter + all-gather per bucket, equal buckets (the last one odd-sized: ring fallback)
    (1024, 4, "rs_ag", [512, 768, 1000]),      # engine-chosen cuts: small last bucket
    (17, 3, "rs_ag", None),                    # ragged buckets: all-reduce fallback where length % world != 0
    (1000, 6, "ring", None), (64, 1, "ring", None)])
def test_bucketed_allreduce_world2(n, n_buckets, mode, cuts):
    """the mean of the arena over 2 gloo ranks through both collective shapes (RCCL's in-place reduce-scatter is emulated
    by per-slice reduces on gloo: same slice arithmetic, same all-gather)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, n_buckets, q, mode, cuts)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    for rank, ok, sent, (lo, hi), ok_bn, delta in res:
        assert ok and ok_bn
        assert (lo, hi) == (8 * rank, 8 * rank + 8)
        if n_buckets > 1 and n >= 100:
            assert sent > 0          # some buckets were launched before finish(): overlap with "backward"
        assert delta > 0


def test_shard_batch_rejects_ragged():
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def test_world1_is_a_noop():
    red = BucketedAllReduce(None, 4)
    a = torch.arange(10.0)
    red.begin(a)
    red.ready(5)
    red.finish()
    assert torch.equal(a, torch.arange(10.0))
