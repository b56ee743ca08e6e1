#!/usr/bin/env python3
"""Micro-benchmark of single conv launches at the headline shapes (B=64, E=4, 256x256 network).
  python tools/bench_conv.py [name ...]      names: conv2 conv1 l1 l2 l3 l4 l2s2 (default: all)
Prints ms and TFLOP/s for forward, data-gradient and weight-gradient of each layer."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pmoe_amd import hip, ops  # noqa: E402

E, B = 4, 64
SHAPES = {  # name: (cin, cout, H, ks, stride)
    "conv1": (12, 64, 256, 3, 1), "conv2": (64, 64, 256, 3, 1), "l1": (64, 64, 128, 3, 1),
    "l2s2": (64, 128, 128, 3, 2), "l2": (128, 128, 64, 3, 1), "l3": (256, 256, 32, 3, 1), "l4": (512, 512, 16, 3, 1),
    "l2d": (64, 128, 128, 1, 2), "l3s2": (128, 256, 64, 3, 2), "l4s2": (256, 512, 32, 3, 2),
}


def r16(c): return (c + 15) // 16 * 16
def r64(c): return (c + 63) // 64 * 64


def bench(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    names = sys.argv[1:] or list(SHAPES)
    hip.load()
    dt = torch.bfloat16
    N = E * B
    for name in names:
        cin, cout, H, ks, stride = SHAPES[name]
        pad = ks // 2
        Ho = (H + 2 * pad - ks) // stride + 1
        cinp, coutp, cst = r16(cin), r64(cout), r16(cout)
        x = torch.randn(N, H, H, cinp, device="cuda").to(dt)
        ws = [torch.randn(cout, cin, ks, ks, device="cuda") * 0.05 for _ in range(E)]
        tab = hip.ptr_table(ws, "cuda")
        wf = torch.empty(E, coutp, ks * ks, cinp, dtype=dt, device="cuda")
        wd = torch.empty(E, r64(cin), ks * ks, cst, dtype=dt, device="cuda")
        ops.pack_conv_weights(tab, wf, wd, E, cout, cin, ks, coutp, cinp, r64(cin), cst, dt)
        y = torch.empty(N, Ho, Ho, cst, dtype=dt, device="cuda")
        rows = ops.conv2d_stat_rows(N, H, H, Ho, Ho, cinp, cst, coutp, B, ks, stride, pad, dt)
        stats = torch.empty(rows, 2, coutp, device="cuda")
        dy = torch.randn(N, Ho, Ho, cst, device="cuda").to(dt)
        dx = torch.empty_like(x)
        ckw = 64
        cpw, cow = (cinp + ckw - 1) // ckw * ckw, (cst + ckw - 1) // ckw * ckw
        wsb = torch.zeros(E, ks * ks, cow, cpw, device="cuda")
        flop = 2.0 * N * Ho * Ho * cout * cin * ks * ks
        t_f = bench(lambda: ops.conv2d(x, wf, y, cin=cinp, cout=cst, coutp=coutp, ipe=B, ks=ks, stride=stride, pad=pad, stats=stats))
        t_d = bench(lambda: ops.conv2d(dy, wd, dx, cin=cst, cout=cinp, coutp=r64(cin), ipe=B, ks=ks, stride=1,
                                       pad=ks - 1 - pad, dilate=(stride == 2)))
        t_w = bench(lambda: ops.conv2d_wgrad(x, dy, wsb, cin=cinp, cout=cst, cinp=cpw, coutp=cow, ipe=B, ks=ks, stride=stride, pad=pad))
        print(f"{name:6s} {cin:3d}->{cout:3d} {H:3d}^2 k{ks}s{stride}  fwd {t_f:7.3f} ms {flop / t_f / 1e9:7.1f} TF/s | "
              f"dgrad {t_d:7.3f} ms {flop / t_d / 1e9:7.1f} TF/s | wgrad {t_w:7.3f} ms {flop / t_w / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
