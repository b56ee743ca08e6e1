#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline step: the reference trainer hands CPU batches to the model
(train_2.py:138-145 `.to(device)`); here each step first copies its 201 MB of f32 frames (+ measurements) from pinned
host memory, on the compute stream (no overlap) and on a side stream one step ahead (overlap)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from bench import make_batch  # noqa: E402
from pmoe_amd.loss import moe_loss  # noqa: E402
from pmoe_amd.model.moe import get_model  # noqa: E402
from pmoe_amd.utils import stage2_model_cfg  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = get_model(stage2_model_cfg("moe", 4, dropout=0.3)).to(dev).train()
    gpu = make_batch(64, 256, 1234, dev)
    host = [t.cpu().pin_memory() for t in gpu]

    def step(batch):
        images, speed, command, control, target = batch
        model.zero_grad(set_to_none=True)
        d, s = model(images, speed, command)
        moe_loss(d, s, control, target, [0.7, 0.3]).backward()

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_res = timeit(lambda: step(gpu))
    t_sync = timeit(lambda: step([t.to(dev, non_blocking=True) for t in host]))
    side = torch.cuda.Stream()
    nxt = [None]

    def prefetch():
        with torch.cuda.stream(side):
            nxt[0] = [t.to(dev, non_blocking=True) for t in host]

    prefetch()

    def overlapped():
        torch.cuda.current_stream().wait_stream(side)
        batch = nxt[0]
        for t in batch:
            t.record_stream(torch.cuda.current_stream())
        prefetch()
        step(batch)
    t_ovl = timeit(overlapped)
    mb = sum(t.numel() * t.element_size() for t in host) / 1e6
    print(f"resident inputs      : {t_res:6.2f} ms/step  {64 / t_res * 1e3:7.1f} samples/s")
    print(f"H2D on compute stream: {t_sync:6.2f} ms/step  {64 / t_sync * 1e3:7.1f} samples/s   ({mb:.0f} MB per step from pinned host memory)")
    print(f"H2D one step ahead   : {t_ovl:6.2f} ms/step  {64 / t_ovl * 1e3:7.1f} samples/s")


if __name__ == "__main__":
    main()
