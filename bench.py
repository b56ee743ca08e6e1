#!/usr/bin/env python3
"""Headline benchmark: samples/sec, forward + moe_loss + backward (+ gradient all-reduce for N > 1),
4-expert MixtureOfExperts ("PMoE" experts), 256x256 RGB x 4 frames, batch 64 per GPU, bf16.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One process per GPU; weak scaling (64 samples per GPU); inputs are synthetic, resident in HBM before
the timed region; rank 0 prints ONE JSON line.  A "step" = zero_grad + forward + moe_loss + backward
(incl. the bucketed RCCL all-reduce).  The full reference step recipe (clip_grad_norm_ 1.0 + Adam
amsgrad, train_2.py:157-165) is timed separately and reported as `h1_step`.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work per sample (SURVEY.md section 8d / BASELINE.md section 3), E=4, 256x256
MAC_FWD_PER_EXPERT_SAMPLE_256 = 11.7299e9
HBM_BYTES_PER_SAMPLE_E4_256_BF16 = 787e6
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--experts", type=int, default=4)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8 = BASELINE config 5: e4m3 weights + activations on the block-scaled fp8 matrix instruction for the "
                         "dense 3x3 stride-1 forward convolutions of ResNet layer2-4, bf16 elsewhere (quote it with --batch 128)")
    ap.add_argument("--dropout", type=float, default=0.3, help="stage_2*.yaml value")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage1", action="store_true", help="skip the stage-1 PU-Net training step (section 8f N4)")
    ap.add_argument("--no-sub-configs", action="store_true",
                    help="skip the short sub-records of BASELINE configs C3 (per-GPU shard, E=8), C4 (punet) and C5 (fp8, B=128)")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--only-steps", action="store_true",
                    help="run warm-up + timed steps and print the basic line only (PMC traffic collection: tools/collect_traffic.py)")
    ap.add_argument("--measure-overlap", action="store_true",
                    help="also time the step with weight gradients on a side stream (co-running kernels: keep it out of "
                         "runs that are profiled per kernel)")
    ap.add_argument("--detail", action="store_true", help="per-layer launch table on stderr")
    return ap.parse_args()


def make_batch(B, size, seed, device):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(B, 4, 3, size, size, generator=g)
    speed = torch.rand(B, 1, generator=g)
    target = torch.rand(B, 1, generator=g)
    command = torch.nn.functional.one_hot(torch.randint(0, 6, (B,), generator=g), 6).float()
    control = torch.rand(B, 2, generator=g) * 2 - 1
    return [t.to(device) for t in (images, speed, command, control, target)]


def _cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """The CPU oracle (oracle/pmoe_oracle.py, a port of the reference path checked against goldens of the
    imported reference) timed on this box's host cores on bounded samples of the workloads SURVEY.md section 8d names:
    the headline shape at batch 4 (the `value`), BASELINE config 1 exactly (E=4, B=2, 128x128, the full H1 step:
    fwd + moe_loss + zero_grad + bwd + clip_grad_norm_(1.0) + Adam(amsgrad), train_2.py:149-165) and B=8 at 256x256."""
    from oracle import pmoe_oracle as O
    from oracle import weights as W
    # the box grants one GPU a 16-CPU share; os.cpu_count() reports the whole host and oversubscribes
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    n = max(1, min(n, 16))
    torch.set_num_threads(n)
    model = O.get_model(O.stage2_cfg("moe", args.experts, dropout=0.0))
    model.train()

    def timed(Bc, size, budget_s, max_iters, h1=False):
        inp = W.make_inputs(Bc, size, size, seed=7)
        opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, amsgrad=True) if h1 else None

        def step():
            d, s = model(inp["images"], inp["speed"], inp["command"])
            loss = O.moe_loss(d, s, inp["control"], inp["target_speed"].clone(), [0.7, 0.3])
            model.zero_grad(set_to_none=True)
            loss.backward()
            if h1:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
                opt.step()
        step()                                                 # thread-pool / allocator warm-up at this shape
        t0 = time.perf_counter()
        iters = 0
        while iters < 2 or (time.perf_counter() - t0 < budget_s and iters < max_iters):
            step()
            iters += 1
        dt = time.perf_counter() - t0
        return round(Bc * iters / dt, 4), iters

    small = W.make_inputs(1, 64, 64, seed=7)
    d, s = model(small["images"], small["speed"], small["command"])
    O.moe_loss(d, s, small["control"], small["target_speed"], [0.7, 0.3]).backward()
    v4, it4 = timed(4, args.size, 8, 5)
    v8, it8 = timed(8, args.size, 8, 3)
    vc1, itc1 = timed(2, 128, 4, 12, h1=True)
    return {"value": v4, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": _cpu_model_string(), "host_cpus": os.cpu_count(),
            "sample": f"{it4} x (fwd+moe_loss+bwd) at batch 4, {args.size}x{args.size}, E={args.experts}, fp32, "
                      f"torch {torch.__version__} CPU",
            "batch8": {"value": v8, "unit": "samples/s",
                       "sample": f"{it8} x (fwd+moe_loss+bwd) at batch 8, {args.size}x{args.size}, E={args.experts}, fp32"},
            "config1": {"value": vc1, "unit": "samples/s",
                        "sample": f"{itc1} x full H1 step (fwd+moe_loss+zero_grad+bwd+clip_grad_norm_(1.0)+Adam amsgrad) at "
                                  f"batch 2, 128x128, E={args.experts}, fp32 -- BASELINE config 1"}}


def sub_configs(dev, args):
    """Short driver-timed records of the other single-GPU BASELINE configs (they used to live in DESIGN.md prose only):
    C3's per-GPU shard (E=8, B=64), C4 (PUNetExpert T=4 F=6, B=64) and C5 (fp8 policy, B=128), all 256x256, fwd+loss+bwd."""
    import gc
    import tempfile
    from pmoe_amd.loss import moe_loss, punet_loss
    from pmoe_amd.model.moe import get_model
    from pmoe_amd.utils import stage2_model_cfg
    out = {}

    def timed(step, n=10):
        """-> (mean, median) ms over ``n`` timed iterations after 2 warm-up ones (HIP events on the launch stream)"""
        step()
        step()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
        return ev[0].elapsed_time(ev[n]) / n, per[n // 2]

    def moe_case(experts, batch, fp8, dtype=torch.bfloat16):
        model = get_model(stage2_model_cfg("moe", experts, dropout=args.dropout)).to(dev)
        model.compute_dtype = dtype
        model.fp8_weights = fp8
        model.train()
        images, speed, command, control, target = make_batch(batch, args.size, 99, dev)

        def step():
            model.zero_grad(set_to_none=True)
            d, s = model(images, speed, command)
            moe_loss(d, s, control, target, [0.7, 0.3]).backward()
        ms, med = timed(step)
        flop = 6 * MAC_FWD_PER_EXPERT_SAMPLE_256 * experts * (args.size / 256.0) ** 2 * batch
        return {"ms_per_step": round(med, 2), "ms_per_step_mean": round(ms, 2), "timed_iterations": 10,
                "samples_per_s": round(batch / med * 1e3, 1), "tflops_algorithmic": round(flop / med / 1e9, 1)}

    out["C3_shard_e8_b64"] = dict(moe_case(8, 64, False), what="8-expert MoE, batch 64 (one GPU's shard of the B=512 DP config), bf16")
    gc.collect(); torch.cuda.empty_cache()
    out["C5_fp8_b128"] = dict(moe_case(args.experts, 128, True), what="4-expert MoE, batch 128, e4m3 weights + e4m3 activations "
                              "on the block-scaled fp8 matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4) for the dense 3x3 "
                              "stride-1 forward convolutions of layer2-4 (46 % of the forward MACs); bf16 elsewhere and backward")
    gc.collect(); torch.cuda.empty_cache()
    out["C2_bf16_b128"] = dict(moe_case(args.experts, 128, False), what="same shape as C5 in plain bf16 (the A/B partner)")
    gc.collect(); torch.cuda.empty_cache()
    out["C2_f32_b16"] = dict(moe_case(args.experts, 16, False, torch.float32), what="4-expert MoE, batch 16, exact-f32 path "
                             "(v_mfma_f32_32x32x2_f32, f32 activations): the path the 1e-4 parity claims rest on; its roof is "
                             "the 157.3 TFLOP/s f32 matrix rate")
    out["C2_f32_b16"]["frac_of_f32_mfma_peak"] = round(out["C2_f32_b16"]["tflops_algorithmic"] / 157.3, 4)
    gc.collect(); torch.cuda.empty_cache()
    # C4: PUNetExpert (the constructor reads checkpoint files in the reference's layouts: write random-init ones)
    from pmoe_amd.utils import build_product
    tmp = Path(tempfile.mkdtemp())
    model = build_product(tmp, dict(type="punet", n_experts=2, future_frames=6), dropout=args.dropout).to(dev)
    model.compute_dtype = torch.bfloat16
    model.train()
    images, speed, command, control, target = make_batch(64, args.size, 98, dev)

    def pstep():
        model.zero_grad(set_to_none=True)
        act, sp = model(images, speed, command)
        punet_loss(act, sp, control, target, [0.7, 0.3]).backward()
    ms, med = timed(pstep)
    out["C4_punet_b64"] = {"ms_per_step": round(med, 2), "ms_per_step_mean": round(ms, 2), "timed_iterations": 10,
                           "samples_per_s": round(64 / med * 1e3, 1),
                           "tflops_algorithmic": round((2 * 477.27e9 + 6 * 16.4836e9) * 64 / med / 1e9, 1),
                           "what": "PUNetExpert T=4 F=6, batch 64, 256x256, bf16, fwd+punet_loss+bwd (PU-Net frozen, forward only)"}
    del model
    gc.collect(); torch.cuda.empty_cache()
    return out


def stage1_step(dev, batch=10, size=224, frames=6, steps=10):
    import tempfile
    from pmoe_amd.loss import AutoregressiveCriterion
    from pmoe_amd.model import blocks as B
    from pmoe_amd.model.punet import PredictiveUnet
    from pmoe_amd.optim import FusedAdam
    path = os.path.join(tempfile.mkdtemp(), "unet.pth")
    torch.save({"unet": B.UNet().state_dict()}, path)          # the constructor reads a stage-0 checkpoint (punet.py:40)
    pu = PredictiveUnet(4, frames, model_name="unet", model_path=path).to(dev)
    pu.train()
    opt = FusedAdam([p for p in pu.parameters() if p.requires_grad], lr=1e-4)
    crit = AutoregressiveCriterion(frames, "tversky")
    img = torch.rand(batch, 4, 3, size, size, device=dev)
    tgt = torch.randint(0, 23, (batch, frames, size, size), device=dev)

    def one():
        loss = crit(pu(img), tgt)
        opt.zero_grad()
        loss.backward()
        opt.step()
    one()
    one()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        one()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))[steps // 2]
    return {"ms_per_step": round(ms, 2), "timed_iterations": steps, "samples_per_s": round(batch / ms * 1e3, 1),
            "what": f"PredictiveUnet fwd + AutoregressiveCriterion('tversky') + bwd through the roll-out + Adam, B={batch} "
                    f"{size}x{size} T=4 F={frames} bf16 (conf/stage_1.yaml)"}


def self_launch(args):
    """--gpus N > 1 without a launcher: run this script under torch.distributed.run (one rank per GPU, RCCL) as a child
    process and relay it.  PMOE_BENCH_SHARE_GPU=1 (rehearsal: every rank on device 0 over gloo) lifts the device check."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("PMOE_BENCH_SHARE_GPU") != "1":
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices, this node shows {have}", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    t_start = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` launches itself: one rank per GPU as a FRESH child job, started before this process
        # has made any GPU call (torch is imported, nothing else: device_count() does not initialise the runtime), the
        # child's output relayed, its exit code returned.  Never an exec of a process that holds the GPU.
        raise SystemExit(self_launch(args))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus {args.gpus} does it)")
    # rehearsal of the N-rank control flow on a ONE-GPU box: PMOE_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # uses gloo (RCCL refuses two ranks on one device); the real multi-GPU run is one rank per GPU over RCCL
    share = os.environ.get("PMOE_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # under torch.distributed.run the process group is created even for ONE rank: the RCCL communicator, ReduceOp.AVG and
    # the bucket waits of pmoe_amd.parallel then run exactly as in the N-GPU job (tests/test_dp_gpu.py runs this on one GPU)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from pmoe_amd import hip, ops
    from pmoe_amd.loss import moe_loss
    from pmoe_amd.model.moe import get_model
    from pmoe_amd.utils import stage2_model_cfg
    hip.load()

    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    torch.manual_seed(0)
    model = get_model(stage2_model_cfg("moe", args.experts, dropout=args.dropout)).to(dev)
    model.compute_dtype = dtype
    model.fp8_weights = args.dtype == "fp8"
    model.train()
    if use_dist:
        # identical replicas: broadcast rank 0's random init
        for p in model.parameters():
            dist.broadcast(p.data, 0)
        model.enable_data_parallel(always=True)
    images, speed, command, control, target = make_batch(args.batch, args.size, 1234 + rank, dev)
    coefs = [0.7, 0.3]

    def step():
        model.zero_grad(set_to_none=True)
        d, s = model(images, speed, command)
        loss = moe_loss(d, s, control, target, coefs)
        loss.backward()
        return loss

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    log("model + inputs resident; warm-up")
    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    fence()
    log("timing")
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step durations for the median
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    rank_ms = None
    if use_dist:
        own = marks[0].elapsed_time(marks[args.steps]) / args.steps       # this rank's device time per step (no barrier wait)
        rank_ms = [None] * world
        dist.all_gather_object(rank_ms, round(own, 3))
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms = elapsed / args.steps * 1e3
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2]
    total_samples = args.batch * world * args.steps
    value = total_samples / elapsed

    out = {
        "metric": "samples/sec fwd+bwd, 256x256 RGB 4-expert PMoE", "value": round(value, 2), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
        "ms_per_step_median": round(ms_median, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.experts}-expert MoE (PMoE experts), {args.size}x{args.size}x3 x4 frames, "
                               f"batch {args.batch}/GPU, fwd+moe_loss+bwd, dropout {args.dropout}"
                               + (", fp8 policy (e4m3 weights + activations, layer2-4 3x3 stride-1 forward convolutions)" if args.dtype == "fp8" else ""),
                   "global_batch": args.batch * world, "parallelism": f"dp{world}"},
        "loss": round(float(loss.item()), 5),
    }

    if use_dist:
        devs = [None] * world
        dist.all_gather_object(devs, int(local))
    if use_dist and rank == 0:
        from pmoe_amd import parallel as _par
        out["dp"] = {"backend": dist.get_backend(), "world": world, "rccl_ranks": dist.get_world_size(),
                     "rank_devices": devs, "rank_ms_per_step": rank_ms,
                     "collective": {"rs_ag": "reduce_scatter_tensor + all_gather_into_tensor per bucket, in place (agreed by all "
                                             "ranks at start-up: parallel._decide_mode)",
                                    "ring": "all_reduce per bucket"}.get(_par.BucketedAllReduce.last_mode, _par.BucketedAllReduce.last_mode),
                     "collective_mode": _par.BucketedAllReduce.last_mode,
                     "buckets_issued_per_backward": _par.BucketedAllReduce.last_issued,
                     "bucket_ends": model._engine()._bucket_cuts(model._engine().dp_buckets)}
    if rank == 0:
        # ---- whole-step rooflines from the algorithmic work model
        scale = (args.size / 256.0) ** 2 * args.experts / 4.0
        flop_per_sample = 6 * MAC_FWD_PER_EXPERT_SAMPLE_256 * 4 * scale
        out["step_mfma_frac"] = round(flop_per_sample * value / world / 1e12 / PEAK_BF16_TFLOPS, 4)
        out["step_hbm_frac_model"] = round(HBM_BYTES_PER_SAMPLE_E4_256_BF16 * scale * value / world / 1e9 / PEAK_HBM_GBS, 4)

    log(f"timed: {ms:.2f} ms/step")
    if args.only_steps:
        if rank == 0:
            print(json.dumps(out), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    # ---- per-kernel timing with HIP events on the launch stream (one instrumented step, untimed region)
    if not args.no_kernel_profile:
        # every rank runs this step (its backward holds collectives); only rank 0 records the per-launch events
        if rank == 0:
            ops.profile_begin()
        step()
        torch.cuda.synchronize()
    if rank == 0 and not args.no_kernel_profile:
        recs = ops.profile_end()
        if args.detail:
            agg = {}
            for name, meta, ms_k in recs:
                key = (name, meta.get("name", ""))
                a = agg.setdefault(key, [0.0, 0, 0.0, 0.0])
                a[0] += ms_k
                a[1] += 1
                a[2] += meta.get("flop", 0.0)
                a[3] += meta.get("bytes", 0.0)
            for (name, lname), (ms_k, n, fl, nb) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:120]:
                tf = fl / (ms_k * 1e-3) / 1e12 if ms_k > 0 and fl else 0.0
                tb = f"{nb / (ms_k * 1e-3) / 1e12:6.2f} TB/s" if ms_k > 0 and nb else ""
                print(f"  {name:18s} {lname:24s} n={n:3d} {ms_k:8.3f} ms  {tf:8.1f} TF/s {tb}", file=sys.stderr)
        by = {}
        for name, meta, ms_k in recs:
            k = by.setdefault(name, {"ms": 0.0, "n": 0, "flop": 0.0, "bytes": 0.0})
            k["ms"] += ms_k
            k["n"] += 1
            k["flop"] += meta.get("flop", 0.0)
            k["bytes"] += meta.get("bytes", 0.0)
        tot = sum(v["ms"] for v in by.values())
        out["kernel_ms"] = {k: round(v["ms"], 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"])}
        out["kernel_ms_total"] = round(tot, 3)
        # ---- roofline of the dominant kernel: conv launches attributed to the kernel instantiation that served them
        # (pmoe_conv2d_plan), so that the numbers line up with the rows of a rocprofv3 kernel trace
        peak = 157.3 if args.dtype == "f32" else PEAK_BF16_TFLOPS    # (the non-scaled fp8 MFMA runs at the bf16 rate)
        tname = "f" if args.dtype == "f32" else "DF16b"
        kdt = "f32" if args.dtype == "f32" else "bf16"

        def symbol(code):
            if code == 7309:                      # ... with separated roles (round 4): 4 accumulating waves + 1 request-only wave
                return "conv_wgrad_dma2_kernel<*>", "void conv_wgrad_dma2_kernel<", 1
            if code == 7009:                      # LDS-DMA staged weight gradient (conv_wgrad.hip)
                return "conv_wgrad_dma_kernel<1, 2, false, *>", "void conv_wgrad_dma_kernel<1, 2, false", 1   # (last parameter: request code, PMOE_WGRAD_REQ)
            if code == 7109:                      # ... its 2 x 4 wave layout for <= 32 input channels (the stem's first convolution)
                return "conv_wgrad_dma_kernel<1, 1, *> (<= 32 input channels)", "void conv_wgrad_dma_kernel<1, 1", 1
            if code == 7209:                      # ... with the stem's first BatchNorm backward applied on load (round 4)
                return "conv_wgrad_bnbwd_kernel", "conv_wgrad_bnbwd_kernel", 1
            if 6000 <= code < 7000:               # conv_wgrad_kernel<T, taps, MAXV>
                taps, maxv = (code - 6000) // 100, (code - 6000) % 100
                return (f"conv_wgrad_kernel<{kdt},{taps},{maxv}>", f"_Z17conv_wgrad_kernelI{tname}Li{taps}ELi{maxv}EEv9WgradArgs", 1)
            if code == 9207:                      # the four parity classes of a stride-2 3x3 data gradient on the LDS-DMA kernel
                return ("conv3x3s2_dma_kernel<true> (4 parity-class launches per stride-2 dgrad)",
                        "void (anonymous namespace)::conv3x3s2_dma_kernel<true>", 4)
            if code == 8507:                      # block-scaled fp8 MFMA kernel (conv_dma.hip)
                return "conv3x3_dma_f8_kernel", "void (anonymous namespace)::conv3x3_dma_f8_kernel", 1
            if code >= 8000:                      # e4m3 operands (BASELINE config 5): same tiles, TL = fp8
                r, m, n = symbol(code - 8000)
                return r.replace("<bf16,", "<bf16+e4m3,"), m.replace(f"E{tname}Ev8ConvArgs", "E3fp8Ev8ConvArgs"), n
            if code in (5047, 5057):              # ... persistent, the producer wave's request stream running across tiles (round 4)
                tf = "true" if code == 5057 else "false"
                return f"conv3x3_dma_stream_kernel<{tf}>", f"void (anonymous namespace)::conv3x3_dma_stream_kernel<{tf}>", 1
            if code in (5007, 5017, 5027, 5037):  # LDS-DMA staged 3x3 kernel (conv_dma.hip): 32x32x16 | 16x16x32 MFMA instantiation;
                tf = "true" if code in (5017, 5037) else "false"          # + 20: the producer-wave instantiation (forward launches)
                pr = "true" if code >= 5027 else "false"
                return f"conv3x3_dma_kernel<{tf}, {pr}>", f"void (anonymous namespace)::conv3x3_dma_kernel<{tf}, {pr}>", 1
            if code == 5207:                      # ... its stride-2 forward sibling
                return "conv3x3s2_dma_kernel<false>", "void (anonymous namespace)::conv3x3s2_dma_kernel<false>", 1
            four = code >= 4000
            code %= 4000
            if code == 3000:
                return "gemm_skinny_kernel", "gemm_skinny_kernel", 1
            if code >= 2000:
                return (f"conv_igemm_lite_kernel<{kdt},{code - 2000}>" + (" (4 parity-class launches per stride-2 dgrad)" if four else ""),
                        f"_Z22conv_igemm_lite_kernelI{tname}Li{code - 2000}E{tname}Ev8ConvArgs", 4 if four else 1)
            if code in (1402, 1404):              # 1x1 convolutions over many pixels, direct form (conv_c1x1.hip)
                return f"conv1x1_direct_kernel<{code - 1400}>", f"void (anonymous namespace)::conv1x1_direct_kernel<{code - 1400}>", 1
            if code == 1316:                      # 16-channel stem convolution, direct form (conv_c16.hip)
                return "conv3x3_c16_kernel", "void (anonymous namespace)::conv3x3_c16_kernel", 1
            if 1200 <= code < 1300:               # ... with the read-out of tile t under the MFMAs of tile t+1: <bias, mode>
                tf, mode = ("true" if (code - 1207) // 10 % 2 else "false"), (code - 1207) // 20
                return f"conv3x3_respipe_kernel<{tf}, {mode}>", f"void conv3x3_respipe_kernel<{tf}, {mode}>", 1
            if code >= 1100:                      # filter bank resident in LDS, halo patches by LDS-DMA (conv_res.hip); <true>: bias row
                tf = "true" if code == 1117 else "false"
                return f"conv3x3_resdma_kernel<{tf}>", f"void conv3x3_resdma_kernel<{tf}>", 1
            if code >= 1000:
                return f"conv3x3_res_kernel<{code - 1000}>", f"void conv3x3_res_kernel<{code - 1000}", 1
            rb, wm, wn = code // 100, code // 10 % 10, code % 10
            return (f"conv_igemm_kernel<{kdt},{rb},{wm},{wn}>" + (" (4 parity-class launches per stride-2 dgrad)" if four else ""),
                    f"_Z17conv_igemm_kernelI{tname}Li{rb}ELi{wm}ELi{wn}E{tname}Ev8ConvArgs", 4 if four else 1)
        groups = {}
        for name, meta, ms_k in recs:
            if name not in ("conv2d", "conv2d_wgrad") or "kernel" not in meta:
                continue
            readable, mangled, per_call = symbol(int(meta["kernel"]))
            g = groups.setdefault(mangled, {"kernel": readable, "ms": 0.0, "launches": 0, "flop": 0.0})
            g["ms"] += ms_k
            g["launches"] += per_call
            g["flop"] += meta.get("flop", 0.0)
        if groups:
            out["conv_kernels"] = {g["kernel"]: {"ms_per_step": round(g["ms"], 3), "launches": g["launches"],
                                                 "tflops": round(g["flop"] / (g["ms"] * 1e-3) / 1e12, 1)}
                                   for g in sorted(groups.values(), key=lambda g: -g["ms"])}
            mangled, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
            achieved = dom["flop"] / (dom["ms"] * 1e-3) / 1e12
            traffic = None
            tf = REPO / "profiles" / "traffic.json"
            if tf.exists():
                pk = json.loads(tf.read_text()).get("per_kernel", {})
                norm = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "")
                hit = [v for k, v in pk.items() if norm(k).startswith(norm(mangled))]
                if hit:          # (several instantiations of one template: launch-weighted mean)
                    traffic = (sum((v["fetch_bytes_per_launch_x2"] + v["write_bytes_per_launch"]) * v["launches"] for v in hit)
                               / sum(v["launches"] for v in hit))
            # the dominant kernel = the MFMA kernel instantiation with the most time per step (forward / data-gradient and
            # weight-gradient kernels alike); every instantiation's own figure is in `conv_kernels`
            out["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "rocprof_name": mangled,
                               "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(achieved / peak, 4), "traffic": traffic,
                               "traffic_source": ("profiles/traffic.json: FETCH_SIZE x 2 + WRITE_SIZE from separate rocprofv3 --pmc "
                                                  "passes over this command (tools/collect_traffic.py), NOT counted in this run")
                               if traffic is not None else None,
                               "launches_per_step": dom["launches"],
                               "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                               "flop_per_launch": dom["flop"] / dom["launches"],
                               "all_conv_fwd_dgrad": {"ms_per_step": round(by["conv2d"]["ms"], 3),
                                                      "tflops": round(by["conv2d"]["flop"] / (by["conv2d"]["ms"] * 1e-3) / 1e12, 1)}}
        wg = by.get("conv2d_wgrad")
        if wg and wg["ms"] > 0:
            out["wgrad_tflops"] = round(wg["flop"] / (wg["ms"] * 1e-3) / 1e12, 2)

    # ---- reference step recipe H1 (train_2.py:157-165): + clip_grad_norm_(1.0) + grad-norm + Adam(amsgrad), with the
    # fused optimizer tail (pmoe_amd.optim) and, for comparison, torch's own per-tensor kernels
    if rank == 0 or world > 1:
        from pmoe_amd import optim as fused_optim

        def time_h1(step_fn, label):
            step_fn()
            step_fn()
            fence()
            n_h1 = max(10, args.steps)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_h1 + 1)]
            t0 = time.perf_counter()
            ev[0].record()
            for i in range(n_h1):
                step_fn()
                ev[i + 1].record()
            fence()
            ms_mean = (time.perf_counter() - t0) / n_h1 * 1e3
            ms_h1 = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n_h1))[n_h1 // 2]
            return {"ms_per_step": round(ms_h1, 3), "ms_per_step_mean": round(ms_mean, 3), "timed_iterations": n_h1,
                    "samples_per_s": round(args.batch * world / ms_h1 * 1e3, 1), "what": label}

        fopt = fused_optim.FusedAdam(model.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, amsgrad=True)

        def h1_fused():
            d, s = model(images, speed, command)
            loss = moe_loss(d, s, control, target, coefs)
            fopt.zero_grad()
            loss.backward()
            gn = fused_optim.clip_grad_norm_(model.parameters(), 1.0, scale=False)
            fopt.step(clip=gn)
        log("kernel profile done; H1 step (fused optimizer tail)")
        out["h1_step"] = time_h1(h1_fused, "fwd+moe_loss+bwd+clip_grad_norm_(1.0)+Adam(amsgrad), fused multi-tensor HIP "
                                           "optimizer tail (pmoe_amd.optim); weights repacked every step")
        # what a trainer sees: `value` is SURVEY section 8d's metric -- fwd+loss+bwd, optimizer step
        # reported separately -- and its timed loop never changes the weights, so the per-layer weight repack is skipped
        # there; this is the same step with the optimizer tail and the repack it forces, every step
        out["h1_samples_per_s"] = out["h1_step"]["samples_per_s"]
        out["h1_ms_per_step"] = out["h1_step"]["ms_per_step"]
        out["h1_note"] = "full reference training step (train_2.py:149-165), weights repacked every step"
        del fopt
        opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, amsgrad=True)

        def h1_torch():
            d, s = model(images, speed, command)
            loss = moe_loss(d, s, control, target, coefs)
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
        log("H1 step (torch optimizer kernels)")
        out["h1_step_torch_optim"] = time_h1(h1_torch, "same step with torch.nn.utils.clip_grad_norm_ + torch.optim.Adam")
        del opt

    # ---- the gradient all-reduce against the backward pass that has to hide it: measured backward
    # time of this rank vs a model of the bucketed ring all-reduce over xGMI (7 links x ~153 GB/s per GPU, point to point)
    if rank == 0:
        d, sp = model(images, speed, command)
        lossb = moe_loss(d, sp, control, target, coefs)
        model.zero_grad(set_to_none=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lossb.backward()
        e1.record()
        torch.cuda.synchronize()
        nbytes = 4 * sum(p.numel() for p in model.parameters() if p.requires_grad)
        link = 153e9
        ring = lambda n: 2 * (n - 1) / n * nbytes / link * 1e3           # one ring: per-link bound
        mesh = lambda n: 2 * nbytes / n / link * 1e3                      # reduce-scatter + all-gather, one slice per peer link
        out["allreduce_model"] = {
            "gradient_bytes": nbytes, "buckets": model._engine().dp_buckets, "backward_ms_measured": round(e0.elapsed_time(e1), 2),
            "xgmi_link_GBs": 153, "collective": "reduce_scatter_tensor + all_gather_into_tensor per bucket (pmoe_amd/parallel.py)",
            "mesh_ms": {str(n): round(mesh(n), 2) for n in (2, 4, 8)}, "ring_ms": {str(n): round(ring(n), 2) for n in (2, 4, 8)},
            "note": "ARITHMETIC, not a measurement (no multi-GPU node was available to the builder): f32 gradient arena over "
                    "153 GB/s xGMI links; mesh = every rank exchanges its 1/N slice with each peer over that peer's own link "
                    "(2 x bytes / N / link), ring = one link per hop (2(N-1)/N x bytes / link).  Buckets are cut by backward "
                    "TIME (engine._bucket_cuts): each starts as soon as the tape has moved past it (heads -> layer4 -> ... -> "
                    "stem); the last one (the stem) is small and is the only one not covered by remaining backward work"}
    elif use_dist:
        d, sp = model(images, speed, command)                             # keep the ranks' collectives aligned
        model.zero_grad(set_to_none=True)
        moe_loss(d, sp, control, target, coefs).backward()

    # ---- optional mode: weight gradients on a side stream (engine.overlap_wgrad), reported beside the standard path
    if world == 1 and args.measure_overlap:
        eng = model._engine()
        eng.overlap_wgrad = True
        step()
        fence()
        t0 = time.perf_counter()
        n_ov = max(2, min(10, args.steps))
        for _ in range(n_ov):
            step()
        fence()
        eng.overlap_wgrad = False
        out["overlap_wgrad_ms_per_step"] = round((time.perf_counter() - t0) / n_ov * 1e3, 3)

    # ---- next-row N4: one stage-1 PU-Net training step at the reference's own configuration
    # (conf/stage_1.yaml: batch 10, 224x224, 4 past / 6 predicted frames, 'tversky' criterion; train_1.py:129-141)
    if rank == 0 and world == 1 and not args.no_stage1:
        log("stage-1 PU-Net training step")
        out["stage1_step"] = stage1_step(dev)
    if rank == 0 and world == 1 and not use_dist and not args.no_sub_configs and args.dtype == "bf16" and args.batch == 64:
        log("sub-records of BASELINE configs C3 / C4 / C5")
        del model
        out["configs"] = sub_configs(dev, args)

    log("H1 done; CPU baseline")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
