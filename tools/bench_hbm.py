#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this GPU (calibrates the HBM-bound BatchNorm / pooling passes):
torch copy / add, our pmoe_copy_window, a read-only reduction.  Prints TB/s of algorithmic bytes."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pmoe_amd import ops  # noqa: E402


def bench(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    for mb in (256, 1024, 2048):
        n = mb * 2 ** 20 // 2
        x = torch.randn(n // 64, 64, device="cuda").to(torch.bfloat16)
        y = torch.empty_like(x)
        z = torch.empty_like(x)
        by = x.numel() * 2
        t = bench(lambda: y.copy_(x))
        print(f"{mb:5d} MB  torch copy (1R+1W)   {2 * by / t / 1e12:5.2f} TB/s")
        t = bench(lambda: ops.copy_window(x, 0, y, 0, 64))
        print(f"{mb:5d} MB  pmoe_copy_window     {2 * by / t / 1e12:5.2f} TB/s")
        t = bench(lambda: torch.add(x, y, out=z))
        print(f"{mb:5d} MB  torch add (2R+1W)    {3 * by / t / 1e12:5.2f} TB/s")
        t = bench(lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32))
        print(f"{mb:5d} MB  torch sum (1R)       {by / t / 1e12:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
