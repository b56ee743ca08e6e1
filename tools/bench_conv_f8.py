#!/usr/bin/env python3
"""conv3x3_dma_f8_kernel (e4m3 weights + e4m3 activations, block-scaled MFMA) against conv3x3_dma_kernel (bf16) on the layer2-4
forward shapes of the headline step, interleaved in one process.   python tools/bench_conv_f8.py [--batch 64]"""
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pmoe_amd import hip, ops  # noqa: E402

E = 4
SHAPES = {"l2": (128, 64), "l3": (256, 32), "l4": (512, 16)}


def main():
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 64
    hip.load()
    dt = torch.bfloat16
    N = E * B
    for name, (c, H) in SHAPES.items():
        x = torch.randn(N, H, H, c, device="cuda").abs().to(dt)
        x8 = (x.float() * 16).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        ws = [torch.randn(c, c, 3, 3, device="cuda") * 0.05 for _ in range(E)]
        tab = hip.ptr_table(ws, "cuda")
        wf = torch.empty(E, c, 9, c, dtype=dt, device="cuda")
        wd = torch.empty(E, c, 9, c, dtype=dt, device="cuda")
        ops.pack_conv_weights(tab, wf, wd, E, c, c, 3, c, c, c, c, dt)
        w8 = torch.empty(E, c, 9, c, dtype=torch.uint8, device="cuda")
        wsc, osc = torch.empty(E, c, device="cuda"), torch.empty(E, c, device="cuda")
        ops.pack_conv_weights_fp8(tab, w8, wd, wsc, osc, 16.0, E, c, c, 3, c, c, c, c)
        y = torch.empty(N, H, H, c, dtype=dt, device="cuda")
        rows = ops.conv2d_stat_rows(N, H, H, H, H, c, c, c, B, 3, 1, 1, dt)
        st16 = torch.empty(rows, 2, c, device="cuda")
        rows8 = ops.conv2d_stat_rows(N, H, H, H, H, c, c, c, B, 3, 1, 1, dt, w_fp8=True, in_fp8=True, in_ld=c)
        st8 = torch.empty(rows8, 2, c, device="cuda")
        kw = dict(cin=c, cout=c, coutp=c, ipe=B, ks=3, stride=1, pad=1)
        fns = {"bf16": lambda: ops.conv2d(x, wf, y, stats=st16, **kw),
               "fp8 ": lambda: ops.conv2d(x8, w8, y, stats=st8, out_scale=osc, in_scale=16.0, **kw)}
        flop = 2.0 * N * H * H * c * c * 9
        times = {k: [] for k in fns}
        for rnd in range(12):
            for k, fn in fns.items():
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rnd >= 2:
                    times[k].append(e0.elapsed_time(e1) / 5)
        print(name, f"B={B}", " | ".join(f"{k} {statistics.median(v):.3f} ms {flop / statistics.median(v) / 1e9:7.1f} TF/s (min {min(v):.3f})"
                                        for k, v in times.items()), flush=True)


if __name__ == "__main__":
    main()
