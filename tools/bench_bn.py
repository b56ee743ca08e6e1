#!/usr/bin/env python3
"""TB/s of the BatchNorm streaming kernels at the headline layer shapes (E=4, B=64)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pmoe_amd import ops  # noqa: E402

E, B = 4, 64


def bench(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    for (H, C) in ((128, 64), (64, 128), (32, 256), (16, 512), (8, 512), (4, 512)):
        N = E * B
        rpe = B * H * H
        x = torch.randn(N, H, H, C, device="cuda").to(torch.bfloat16)
        res = torch.randn_like(x)
        dy = torch.randn_like(x)
        y = torch.empty_like(x)
        dz = torch.empty_like(x)
        gm = torch.empty_like(x)
        sc, sh, mu, inv, c1, c2 = (torch.rand(E, C, device="cuda") + 0.5 for _ in range(6))
        by = x.numel() * 2
        t = bench(lambda: ops.bn_apply(x, None, y, sc, sh, mu, rpe, E, C, True))
        t2 = bench(lambda: ops.bn_apply(x, res, y, sc, sh, mu, rpe, E, C, True))
        nparts = min(1024, rpe // 256)
        part = torch.empty(E, nparts, 2, C, device="cuda")
        t3 = bench(lambda: ops.bn_bwd_reduce(dy, None, x, mu, inv, sc, sh, rpe, E, C, True, part, nparts))
        t4 = bench(lambda: ops.bn_bwd_apply(dy, None, x, mu, inv, sc, sh, c1, c2, dz, None, rpe, E, C, True))
        t5 = bench(lambda: ops.bn_bwd_apply(dy, y, x, mu, inv, sc, sh, c1, c2, dz, gm, rpe, E, C, True))
        print(f"   times us: apply {t*1e6:.0f} apply+res {t2*1e6:.0f} bwd_reduce {t3*1e6:.0f} bwd_apply {t4*1e6:.0f} bwd_apply+res {t5*1e6:.0f}")
        print(f"{H:4d}^2 C={C:3d} ({by / 2**20:5.0f} MB): apply {2 * by / t / 1e12:5.2f}  apply+res {3 * by / t2 / 1e12:5.2f}  "
              f"bwd_reduce {2 * by / t3 / 1e12:5.2f}  bwd_apply {3 * by / t4 / 1e12:5.2f}  bwd_apply+res {5 * by / t5 / 1e12:5.2f} TB/s",
              flush=True)


if __name__ == "__main__":
    main()
