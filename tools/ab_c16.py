#!/usr/bin/env python3
"""First stem convolution (16 -> 64 channels, 256x256, input shared by the experts): conv3x3_c16_kernel against
conv3x3_res_kernel<5>, interleaved in one process (PMOE_CONV_C16 is read per launch), outputs and BatchNorm sums compared.
  python tools/ab_c16.py        (GPU box)"""
import os
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pmoe_amd import hip, ops  # noqa: E402


def main():
    E, B, H, cin, cout, dt = 4, 64, 256, 16, 64, torch.bfloat16
    N = E * B
    x = torch.randn(B, H, H, cin, device="cuda").to(dt)
    x[..., 12:] = 0
    ws = [torch.randn(cout, cin, 3, 3, device="cuda") * 0.1 for _ in range(E)]
    wf = torch.empty(E, cout, 9, cin, dtype=dt, device="cuda")
    wd = torch.empty(E, cin, 9, cout, dtype=dt, device="cuda")
    ops.pack_conv_weights(hip.ptr_table(ws, "cuda"), wf, wd, E, cout, cin, 3, cout, cin, cin, cout, dt)
    res, times = {}, {"0": [], "1": [], "2": []}
    for rnd in range(9):
        for v in ("0", "1"):
            os.environ["PMOE_CONV_C16"] = "0" if v == "0" else "1"
            os.environ["PMOE_C16_LDS_STORE"] = "0" if v == "2" else "1"
            rows = ops.conv2d_stat_rows(N, H, H, H, H, cin, cout, cout, B, 3, 1, 1, dt, in_ld=cin, out_ld=cout, in_shared=True)
            stats = torch.zeros(rows, 2, cout, device="cuda")
            y = torch.empty(N, H, H, cout, dtype=dt, device="cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(x, wf, y, cin=cin, cout=cout, coutp=cout, ipe=B, ks=3, stride=1, pad=1, stats=stats, in_shared=True)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[v].append(e0.elapsed_time(e1))
            else:
                res[v] = (y, stats.view(E, rows // E, 2, cout).sum(1))
            del y
    gb = (N * H * H * cout * 2 + B * H * H * cin * 2) / 1e9
    for v, nm in (("0", "conv3x3_res_kernel<5>       "), ("1", "conv3x3_c16_kernel          ")):
        t = statistics.median(times[v])
        print(f"{nm}: {t:.3f} ms   {gb / t:.2f} TB/s   {2.0 * N * H * H * 12 * cout * 9 / t / 1e9:.0f} TFLOP/s")
    (ya, sa), (yb, sb) = res["0"], res["1"]
    print("outputs bit-identical:", torch.equal(ya, yb), "  max |diff|", (ya.float() - yb.float()).abs().max().item(),
          "  stats rel diff", ((sa - sb).abs().max() / sa.abs().max()).item())


if __name__ == "__main__":
    main()
