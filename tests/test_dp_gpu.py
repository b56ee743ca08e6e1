"""Data-parallel gradient averaging driven from the engines' backward (pmoe_amd.parallel.BucketedAllReduce), rehearsed with
two ranks that SHARE cuda:0 over gloo (RCCL refuses two ranks on one device; the real run is one rank per GPU): the
all-reduced gradients of every rank must equal the mean of the ranks' local gradients, for the mixture (buckets fly during
backward) and for stage-1 PU-Net training (shared weights accumulate over the roll-out, so buckets fly at the end)."""
import os
import socket
import tempfile
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, tmp, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = "cuda"
        torch.manual_seed(0)                       # identical replicas
        g = torch.Generator().manual_seed(10 + rank)            # each rank its own shard
        if kind == "moe":
            from pmoe_amd.loss import moe_loss
            from pmoe_amd.model.moe import get_model
            from pmoe_amd.utils import stage2_model_cfg
            model = get_model(stage2_model_cfg("moe", 2, dropout=0.0)).to(dev)
            images = torch.rand(2, 4, 3, 64, 64, generator=g).to(dev)
            speed, tgt = torch.rand(2, 1, generator=g).to(dev), torch.rand(2, 1, generator=g).to(dev)
            cmd = torch.nn.functional.one_hot(torch.randint(0, 6, (2,), generator=g), 6).float().to(dev)
            act = (torch.rand(2, 2, generator=g) * 2 - 1).to(dev)

            def loss_fn():
                d, sp = model(images, speed, cmd)
                return moe_loss(d, sp, act, tgt, [0.7, 0.3])
        else:
            from pmoe_amd.loss import AutoregressiveCriterion
            from pmoe_amd.model import blocks as B
            from pmoe_amd.model.punet import PredictiveUnet
            path = Path(tmp) / f"unet{rank}.pth"
            torch.manual_seed(1)
            torch.save({"unet": B.UNet().state_dict()}, path)
            torch.manual_seed(0)
            model = PredictiveUnet(4, 2, model_name="unet", model_path=str(path)).to(dev)
            images = torch.rand(2, 4, 3, 32, 32, generator=g).to(dev)
            target = torch.randint(0, 23, (2, 2, 32, 32), generator=g).to(dev)
            crit = AutoregressiveCriterion(2, "tversky")

            def loss_fn():
                return crit(model(images), target)
        model.compute_dtype = torch.float32
        model.train()
        params = [p for p in model.parameters() if p.requires_grad]
        loss_fn().backward()                       # local gradients (no exchange yet)
        local = torch.cat([p.grad.flatten() for p in params]).clone()
        for p in params:
            p.grad = None
        model.enable_data_parallel(n_buckets=3)
        loss_fn().backward()
        torch.cuda.synchronize()
        got = torch.cat([p.grad.flatten() for p in params])
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world
        err = ((got - want).abs().max() / want.abs().max()).item()
        differs = ((gathered[0] - gathered[1]).abs().max() / want.abs().max()).item()
        q.put((rank, err, differs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["moe", "stage1"])
def test_engine_gradient_allreduce_two_ranks_one_gpu(kind):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    tmp = tempfile.mkdtemp()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, tmp, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, differs in res:
        assert differs > 1e-3, "the two shards must produce different local gradients"
        # deterministic backward (no float atomics): the all-reduced gradient IS the mean of the local gradients, up to
        # the f32 rounding of gloo's own sum
        assert err <= 1e-6, (kind, rank, err)


def test_rccl_branch_runs_on_one_gpu():
    """the `nccl` (= RCCL) branch of bench.py / pmoe_amd.parallel had never executed anywhere.  bench.py
    under torch.distributed.run with ONE rank: RCCL communicator init (before the first GPU call of the rank), the weight
    broadcast, ReduceOp.AVG on every arena bucket launched from the engine's backward, the bucket waits, barrier and
    max-over-ranks timing -- everything the N-GPU job does except moving bytes between devices."""
    import json
    import subprocess
    import sys
    repo = Path(__file__).resolve().parents[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(repo / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch", "8", "--size", "128", "--no-cpu-baseline", "--no-stage1", "--no-sub-configs"]
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["parallelism"] == "dp1"
    assert out["allreduce_model"]["gradient_bytes"] == 4 * 55373360          # E=4 MixtureOfExperts
    # the collectives really ran: bench logs the backend and the bucket count it issued
    assert out.get("dp", {}).get("backend") == "nccl" and out["dp"]["buckets_issued_per_backward"] == 6, out.get("dp")


def test_bench_launches_itself(tmp_path):
    """`python bench.py --gpus N` (no launcher, the way the driver invokes `--gpus 1`) starts its own
    one-rank-per-GPU job as a child BEFORE touching the GPU and relays rank 0's line.  Rehearsed with two ranks sharing this
    box's GPU over gloo (PMOE_BENCH_SHARE_GPU=1); without that switch the same command must refuse with the device count."""
    import json
    import subprocess
    import sys
    repo = Path(__file__).resolve().parents[1]
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    args = [sys.executable, str(repo / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--size", "64",
            "--no-cpu-baseline", "--no-stage1", "--no-sub-configs", "--no-kernel-profile"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(args, cwd=repo, env=dict(base, PMOE_BENCH_SHARE_GPU="0"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "needs 2 devices" in r.stderr, r.stderr[-2000:]
    r = subprocess.run(args, cwd=repo, env=dict(base, PMOE_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert out["dp"]["rccl_ranks"] == 2 and out["dp"]["rank_devices"] == [0, 0] and out["dp"]["backend"] == "gloo"
    assert out["dp"]["buckets_issued_per_backward"] == 6
