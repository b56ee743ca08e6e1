#!/usr/bin/env python3
"""Config C4 of BASELINE.json (PUNetExpert, T=4 past / F=6 future frames, 256x256, bf16): ms per fwd+punet_loss+bwd step.
  python tools/bench_punet.py [--batch 64] [--size 256] [--steps 5] [--detail]"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pmoe_amd import ops  # noqa: E402
from pmoe_amd.loss import punet_loss  # noqa: E402
from pmoe_amd.utils import build_product  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--future", type=int, default=6)
    ap.add_argument("--detail", action="store_true")
    a = ap.parse_args()
    tmp = Path(__file__).resolve().parents[1] / "build" / "probe_punet"
    model = build_product(tmp, dict(type="punet", n_experts=2, future_frames=a.future), dropout=0.3).cuda()
    model.compute_dtype = torch.bfloat16
    model.train()
    g = torch.Generator().manual_seed(0)
    B, S = a.batch, a.size
    images = torch.rand(B, 4, 3, S, S, generator=g).cuda()
    speed, target = torch.rand(B, 1, generator=g).cuda(), torch.rand(B, 1, generator=g).cuda()
    command = torch.nn.functional.one_hot(torch.randint(0, 6, (B,), generator=g), 6).float().cuda()
    control = (torch.rand(B, 2, generator=g) * 2 - 1).cuda()

    def step():
        model.zero_grad(set_to_none=True)
        act, sp = model(images, speed, command)
        loss = punet_loss(act, sp, control, target, [0.7, 0.3])
        loss.backward()
        return loss
    step()
    torch.cuda.synchronize()
    print("peak memory GiB after warm-up:", round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), flush=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    flop = (2 * 477.27e9 + 6 * 16.4836e9) * B if (S == 256 and a.future == 6) else float("nan")
    print(f"PUNetExpert B={B} {S}x{S} F={a.future} bf16: {ms:.1f} ms/step = {B / ms * 1e3:.1f} samples/s, "
          f"{flop / ms / 1e9:.0f} TFLOP/s (algorithmic 1053.4 GFLOP/sample), loss {loss.item():.4f}", flush=True)
    if a.detail:
        ops.profile_begin()
        step()
        rows = ops.profile_end()
        by = {}
        for name, meta, ms_ in rows:
            d = by.setdefault(name, [0.0, 0])
            d[0] += ms_
            d[1] += 1
        for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:32]:
            print(f"  {k:22s} {t:8.2f} ms  {n:4d} launches")
        # conv launches grouped by (layer name, kernel plan code): where the 300 conv launches spend their time
        conv = {}
        for name, meta, ms_ in rows:
            if name in ("conv2d", "conv2d_wgrad"):
                d = conv.setdefault((name, meta.get("name", "?"), meta.get("kernel", "?")), [0.0, 0, 0.0])
                d[0] += ms_; d[1] += 1; d[2] += meta.get("flop", 0.0)
        for (op, lname, code), (t, n, fl) in sorted(conv.items(), key=lambda kv: -kv[1][0])[:28]:
            print(f"  {op:13s} {lname:34s} plan {code!s:>6}  {t:7.2f} ms  {n:3d} launches  {fl / max(t, 1e-9) / 1e9:7.0f} TFLOP/s")


if __name__ == "__main__":
    main()
