"""Data-parallel plumbing on CPU (gloo, world_size 2): the bucketed gradient all-reduce that the engine
drives from backward, batch sharding, and BatchNorm-buffer sync.  No GPU kernels are involved."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pmoe_amd.parallel import BucketedAllReduce, checkpoint_state_dict, shard_batch, sync_bn_buffers


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


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the bucket layout the ENGINE really produces for the E=4 and E=8 gradient arenas, driven through
# world-size 4 and 8 groups.  Payload is scaled down (one float stands for STRIDE arena elements: every real cut is a multiple
# of 256, so cut // STRIDE keeps every bucket length a multiple of 8 = the largest world size here).
STRIDE = 32


def _engine_cuts(n_experts, n_buckets=6):
    from pmoe_amd.model.moe import get_model
    from pmoe_amd.utils import stage2_model_cfg
    eng = get_model(stage2_model_cfg("moe", n_experts, dropout=0.0))._engine()
    eng._layout_arena()
    return eng, eng._bucket_cuts(n_buckets)


@pytest.mark.parametrize("n_experts", [4, 8])
def test_engine_bucket_cuts_divide_for_every_world_size(n_experts):
    eng, cuts = _engine_cuts(n_experts)
    n = eng._arena_numel
    assert n >= eng._arena_used and n % 256 == 0
    assert cuts == sorted(set(cuts)) and cuts[-1] == n and len(cuts) == 6          # strictly increasing, ends at the arena's end
    lo = 0
    for hi in cuts:
        for world in (2, 4, 8):
            assert (hi - lo) % world == 0 and (hi - lo) > 0, (lo, hi, world)        # reduce-scatter needs length % world == 0
        assert hi % 256 == 0
        lo = hi
    # the last bucket is the backward TAIL only (layer1 + stem + the two measurement encoders, whose backward closures run
    # last: ~5 % of the parameters for ~45 % of the backward time), cut at a slot boundary
    tail = n - cuts[-2]
    assert 0 < tail < 0.06 * n, (tail, n)
    # backward order: the heads' slots sit at the front of the arena, the stem's at its end
    first, last = eng._order[0], eng._order[-1]
    assert eng._slots[first][0] == 0 and eng._slots[last][0] + eng._slots[last][1] == eng._arena_used


def _worker_engine_cuts(rank, world, port, cuts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = cuts[-1]
        g = torch.Generator().manual_seed(500 + rank)
        arena = torch.randn(n, generator=g)
        red = BucketedAllReduce(None, len(cuts))
        red.begin(arena, cuts)
        mode = red.mode
        launched = []
        for upto in cuts:                        # backward passes one bucket boundary after the other
            red.ready(upto - 1)                  # one element short: the bucket must NOT fly yet
            before = red.sent
            red.ready(upto)
            launched.append((before, red.sent))
        red.finish()
        expect = sum(torch.randn(n, generator=torch.Generator().manual_seed(500 + r)) for r in range(world)) / world
        # checkpoint hook: BatchNorm buffers averaged, state_dict identical on every rank afterwards
        bn = torch.nn.BatchNorm2d(3)
        bn.running_mean.fill_(float(rank))
        bn.running_var.fill_(1.0 + rank)
        sd = checkpoint_state_dict(bn)
        q.put((rank, bool(torch.allclose(arena, expect, atol=1e-5)), mode, launched, BucketedAllReduce.last_issued,
               sd["running_mean"].tolist(), sd["running_var"].tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_experts", [(4, 4), (8, 8)])
def test_engine_buckets_through_world_4_and_8(world, n_experts):
    eng, cuts = _engine_cuts(n_experts)
    scaled = [c // STRIDE for c in cuts]
    assert all((b - a) % world == 0 for a, b in zip([0] + scaled, scaled))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_engine_cuts, args=(r, world, port, scaled, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, mode, launched, issued, rm, rv in res:
        assert ok, rank
        assert mode == "rs_ag"                                   # agreed by all ranks at start-up (_decide_mode)
        assert issued == len(scaled)
        lo = 0
        for (before, after), hi in zip(launched, scaled):        # every bucket flew exactly when its end was reached, in order
            assert before == lo and after == hi
            lo = hi
        assert rm == [(world - 1) / 2.0] * 3 and rv == [1.0 + (world - 1) / 2.0] * 3


def _worker_mode_vote(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if rank == 1:                            # on one rank the reduce-scatter shape runs but yields a WRONG result
            real = BucketedAllReduce._launch

            class _Corrupt:
                def __init__(self, t):
                    self.t = t

                def wait(self):
                    self.t[3] += 1.0

            def wrong(self, lo, hi):
                real(self, lo, hi)
                if self.n_buckets == 1 and self.arena.numel() == 256 * world:       # the start-up probe's buffer
                    self.works.append((_Corrupt(self.arena), None))
            BucketedAllReduce._launch = wrong
        arena = torch.full((64,), float(rank))
        red = BucketedAllReduce(None, 2)
        red.begin(arena)
        red.finish()
        q.put((rank, red.mode, arena.tolist() == [0.5] * 64))
    finally:
        dist.destroy_process_group()


def test_collective_shape_is_agreed_by_all_ranks():
    """the collective shape is decided ONCE, by all ranks together, before the first real bucket: a start-up probe
    whose RESULT is wrong on one rank moves every rank to all_reduce (a per-rank, per-bucket fallback would issue mismatched
    collective sequences: a hang, not a recovery)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mode_vote, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [m for _, m, _ in res] == ["ring", "ring"] and all(ok for *_, ok in res)
