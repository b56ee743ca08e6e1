#!/usr/bin/env python3
"""Interleaved A/B of kernel variants in ONE process on ONE device (cdna_hip_programming.md rule 24: devices differ by
10 %+, so builds must never be ranked across runs).  Variants are environment switches read per launch:
  python tools/ab_conv.py PMOE_DMA_VARIANT 0 1 2 [-- l2 l3 l4]
Prints median ms and TFLOP/s per variant per layer (forward conv with fused BatchNorm statistics + data gradient)."""
import os
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pmoe_amd import hip, ops  # noqa: E402

E, B = 4, 64
SHAPES = {"l2": (128, 128, 64), "l3": (256, 256, 32), "l4": (512, 512, 16), "l1": (64, 64, 128), "conv2": (64, 64, 256)}


def main():
    args = sys.argv[1:]
    names = ["l2", "l3", "l4"]
    if "--" in args:
        names = args[args.index("--") + 1:]
        args = args[:args.index("--")]
    var, values = args[0], args[1:]
    hip.load()
    dt = torch.bfloat16
    N = E * B
    for name in names:
        cin, cout, H = SHAPES[name]
        x = torch.randn(N, H, H, cin, device="cuda").to(dt)
        ws = [torch.randn(cout, cin, 3, 3, device="cuda") * 0.05 for _ in range(E)]
        wf = torch.empty(E, cout, 9, cin, dtype=dt, device="cuda")
        wd = torch.empty(E, cin, 9, cout, dtype=dt, device="cuda")
        ops.pack_conv_weights(hip.ptr_table(ws, "cuda"), wf, wd, E, cout, cin, 3, cout, cin, cin, cout, dt)
        y = torch.empty(N, H, H, cout, dtype=dt, device="cuda")
        dy = torch.randn(N, H, H, cout, device="cuda").to(dt)
        dx = torch.empty_like(x)
        rows = ops.conv2d_stat_rows(N, H, H, H, H, cin, cout, cout, B, 3, 1, 1, dt)
        stats = torch.empty(rows, 2, cout, device="cuda")
        flop = 2.0 * N * H * H * cin * cout * 9

        def fwd():
            ops.conv2d(x, wf, y, cin=cin, cout=cout, coutp=cout, ipe=B, ks=3, stride=1, pad=1, stats=stats)

        def dgrad():
            ops.conv2d(dy, wd, dx, cin=cout, cout=cin, coutp=cin, ipe=B, ks=3, stride=1, pad=1)
        wsb = torch.empty(E, 9, cout, cin, device="cuda")

        def wgrad():
            ops.conv2d_wgrad(x, dy, wsb, cin=cin, cout=cout, cinp=cin, coutp=cout, ipe=B, ks=3, stride=1, pad=1)
        kinds = (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad))
        times = {(v, k): [] for v in values for k, _ in kinds}
        for rnd in range(12):
            for v in values:
                os.environ[var] = v
                for k, fn in kinds:
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd >= 2:
                        times[(v, k)].append(e0.elapsed_time(e1) / 5)
        for v in values:
            f, d, w = (statistics.median(times[(v, k)]) for k in ("fwd", "dgrad", "wgrad"))
            print(f"{name} {var}={v}: fwd {f:.3f} ms {flop / f / 1e9:7.1f} TF/s (min {min(times[(v, 'fwd')]):.3f}) | "
                  f"dgrad {d:.3f} ms {flop / d / 1e9:7.1f} TF/s (min {min(times[(v, 'dgrad')]):.3f}) | "
                  f"wgrad {w:.3f} ms {flop / w / 1e9:7.1f} TF/s (min {min(times[(v, 'wgrad')]):.3f})", flush=True)


if __name__ == "__main__":
    main()
