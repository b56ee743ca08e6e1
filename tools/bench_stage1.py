#!/usr/bin/env python3
"""Stage-1 PU-Net training step (trainer/train_1.py:129-141, conf/stage_1.yaml): PredictiveUnet
(T=4 past frames, F=6 predicted), AutoregressiveCriterion('tversky'), backward through the roll-out, Adam step.
  python tools/bench_stage1.py [--batch 10] [--size 224] [--frames 6] [--steps 5] [--dtype bf16] [--profile]"""
import argparse
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pmoe_amd import ops                                   # noqa: E402
from pmoe_amd.loss import AutoregressiveCriterion          # noqa: E402
from pmoe_amd.model import blocks as B                     # noqa: E402
from pmoe_amd.model.punet import PredictiveUnet            # noqa: E402
from pmoe_amd.optim import FusedAdam                       # noqa: E402


def unet_gmac(h, w, cin=3, classes=23):
    """forward GMAC of one U-Net application (blocks/unet.py) at h x w."""
    chans = [(cin, 64), (64, 128), (128, 256), (256, 512), (512, 512)]
    mac, hh, ww = 0, h, w
    for i, (a, b) in enumerate(chans):
        mac += hh * ww * 9 * (a * b + b * b)
        if i < 4:
            hh, ww = hh // 2, ww // 2
    for (cu_in, cu_out, cf_in) in [(512, 512, 1024), (512, 256, 512), (256, 128, 256), (128, 64, 128)]:
        mac += hh * ww * cu_in * cu_out * 4
        hh, ww = hh * 2, ww * 2
        mac += hh * ww * 9 * (cf_in * cu_out + cu_out * cu_out)
    mac += hh * ww * 64 * classes
    return mac / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    dev = "cuda"
    tmp = Path(tempfile.mkdtemp())
    torch.save({"unet": B.UNet().state_dict()}, tmp / "unet.pth")
    torch.manual_seed(0)
    model = PredictiveUnet(4, a.frames, model_name="unet", model_path=str(tmp / "unet.pth")).to(dev)
    model.compute_dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    model.train()
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    crit = AutoregressiveCriterion(a.frames, "tversky")
    images = torch.rand(a.batch, 4, 3, a.size, a.size, device=dev)
    target = torch.randint(0, 23, (a.batch, a.frames, a.size, a.size), device=dev)

    def step():
        out = model(images)
        loss = crit(out, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    print(f"peak memory GiB after warm-up: {torch.cuda.max_memory_allocated() / 2**30:.2f}")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    u = unet_gmac(a.size, a.size)
    entry = a.size * a.size * 9 * (92 * 64 + 64 * 3) / 1e9
    gflop = 2 * (4 * u + 3 * a.frames * (u + entry))          # frozen: forward only; trained: forward + dgrad + wgrad
    print(f"PredictiveUnet stage-1 step B={a.batch} {a.size}x{a.size} T=4 F={a.frames} {a.dtype}: {ms:.1f} ms/step = "
          f"{a.batch / ms * 1e3:.1f} samples/s, {a.batch * gflop / ms:.0f} TFLOP/s "
          f"(algorithmic {gflop:.1f} GFLOP/sample), loss {loss.item():.4f}")
    if a.profile:
        ops.profile_begin()
        step()
        rows = ops.profile_end()
        agg = {}
        for name, meta, ms_ in rows:
            d = agg.setdefault(name, [0.0, 0])
            d[0] += ms_
            d[1] += 1
        for k, (ms_, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:24]:
            print(f"  {k:24s} {ms_:8.2f} ms {n:5d} launches")
        layers = {}
        for name, meta, ms_ in rows:
            if name in ("conv2d", "conv2d_wgrad") and "name" in meta:
                key = (name, meta["name"].replace("punet.", ""))
                d = layers.setdefault(key, [0.0, 0, 0.0])
                d[0] += ms_
                d[1] += 1
                d[2] += meta.get("flop", 0.0)
        for (op, nm), (ms_, n, fl) in sorted(layers.items(), key=lambda kv: -kv[1][0])[:40]:
            print(f"  {op:13s} {nm:28s} {ms_:7.2f} ms {n:4d} launches {fl / ms_ / 1e9:7.0f} TFLOP/s")

if __name__ == "__main__":
    main()
