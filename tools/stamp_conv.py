#!/usr/bin/env python3
"""Where do the waves of the dominant conv kernel spend their cycles?  Builds a CYCLE-STAMPED variant of conv_igemm.hip
(-DPMOE_STAMP: s_memtime laps in scalar registers around the five phases of the main loop; never the product build), loads it
through PMOE_HIP_LIB and runs single launches at the headline shapes.

  python tools/stamp_conv.py --build          # here (hipcc cross-compiles): pmoe_amd/libpmoe_hip_stamp.so
  python tools/stamp_conv.py [l3 l4 l2]       # on the GPU box: per-phase share of wave cycles, forward and data gradient
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "pmoe_amd" / "libpmoe_hip_stamp.so"
PHASES = ["halo patch staging (+ its barrier)", "fragment reads + MFMA issue", "weight-tile wait + LDS write", "per-tap barrier",
          "epilogue"]


def build():
    csrc, bld = ROOT / "pmoe_amd" / "csrc", ROOT / "build"
    subprocess.check_call([str(ROOT / "build.sh")])
    objs = []
    for src in ("conv_igemm", "conv_dma", "conv_wgrad", "conv_res"):
        obj = bld / f"{src}_stamp.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-Wno-unused-result", "-DPMOE_STAMP", "-c", str(csrc / f"{src}.hip"), "-o", str(obj)])
        objs.append(str(obj))
    skip = ("conv_igemm.o", "conv_igemm_stamp.o", "conv_dma.o", "conv_dma_stamp.o", "conv_wgrad.o", "conv_wgrad_stamp.o",
            "conv_res.o", "conv_res_stamp.o")
    others = [str(o) for o in sorted(bld.glob("*.o")) if o.name not in skip]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + objs + others)
    print("built", LIB)


def main():
    if "--build" in sys.argv:
        return build()
    os.environ["PMOE_HIP_LIB"] = str(LIB)
    sys.path.insert(0, str(ROOT))
    import torch
    from pmoe_amd import hip, ops
    hip.load()
    E, B = 4, 64
    shapes = {"l2": (128, 128, 64), "l3": (256, 256, 32), "l4": (512, 512, 16), "conv2": (64, 64, 256), "l1": (64, 64, 128)}
    dt = torch.bfloat16
    for name in [a for a in sys.argv[1:] if a in shapes] or ["l3", "l4"]:
        cin, cout, H = shapes[name]
        N = E * B
        x = torch.randn(N, H, H, cin, device="cuda").to(dt)
        ws = [torch.randn(cout, cin, 3, 3, device="cuda") * 0.05 for _ in range(E)]
        wf = torch.empty(E, cout, 9, cin, dtype=dt, device="cuda")
        wd = torch.empty(E, cin, 9, cout, dtype=dt, device="cuda")
        ops.pack_conv_weights(hip.ptr_table(ws, "cuda"), wf, wd, E, cout, cin, 3, cout, cin, cin, cout, dt)
        y = torch.empty(N, H, H, cout, dtype=dt, device="cuda")
        rows = ops.conv2d_stat_rows(N, H, H, H, H, cin, cout, cout, B, 3, 1, 1, dt)
        plan = ops.conv2d_plan(N, H, H, H, H, cin, cout, cout, B, 3, 1, 1, dt) if hasattr(ops, "conv2d_plan") else None
        runs = {"forward": (x, wf), "data gradient": (y, wd)}
        if "--dbn" in sys.argv and cin == cout:      # + the data gradient with the BatchNorm-backward epilogue (PMOE_RES_DBN)
            runs["data gradient + BatchNorm reductions"] = (y, wd)
        zz = torch.randn(N, H, H, cin, device="cuda").to(dt)
        coef = torch.rand(4, E, cin, device="cuda") + 0.5
        for what, (src, w) in runs.items():
            stats = torch.full((rows, 2, cout), -1.0, device="cuda")
            extra = dict(res=zz, res_mode=hip.RES_DBN, bn_coef=coef) if what.endswith("reductions") else {}
            for _ in range(3):          # warm caches / clocks, keep the last
                stats.fill_(-1.0)
                ops.conv2d(src, w, y if what == "forward" else x, cin=cin if what == "forward" else cout,
                           cout=cout if what == "forward" else cin, coutp=cout if what == "forward" else cin, ipe=B, ks=3,
                           stride=1, pad=1, stats=stats, **extra)
            torch.cuda.synchronize()
            v = stats.flatten().cpu()
            if cout == 64:                                        # conv3x3_resdma_kernel: 5 laps per tile, summed over the workgroup's tiles
                rec = v[:rows * 64].view(rows * 8, 8)
                assert (rec[:, :7] >= 0).all(), "stamped library not loaded or a different kernel ran"
                cyc, wall, nt = rec[:, :5], rec[:, 5] / 100.0, rec[:, 6]
                per = (cyc / nt[:, None]).mean(0).tolist()
                pipe = os.environ.get("PMOE_RES_PIPE", "1") != "0"
                names = (["requests+init", "36 MFMA steps + the previous tile's read-out", "round/swap + wait + barrier", "-", "-"] if pipe else
                         ["requests+init", "36 MFMA steps", "barrier (patch read)", "staging+wait+barrier", "read-out+stores+barrier"])
                print(f"{name} {what} [{'conv3x3_respipe_kernel' if pipe else 'conv3x3_resdma_kernel'}]: {rows} workgroups, {nt.mean():.1f} tiles each; cycles per tile per wave: "
                      + ", ".join(f"{n} {c:.0f}" for n, c in zip(names, per)) + f" = {sum(per):.0f}; workgroup wall {wall.mean():.1f} us "
                      f"(=> {cyc.sum(1).mean() / wall.mean() / 1e3:.2f} GHz); slowest wave's main loop {(cyc[:, 1] / nt).max():.0f}")
                continue
            nwg = rows * (cout // 128)
            if os.environ.get("PMOE_CONV_DMA", "1") != "0":       # conv3x3_dma_kernel: 3 laps + wall time per wave
                rec = v[:nwg * 8 * 8].view(nwg * 8, 8)[:, :4]
                assert (rec >= 0).all(), "stamped library not loaded or a different kernel ran"
                cyc, wall = rec[:, :3], rec[:, 3] / 100.0
                print(f"{name} {what} [conv3x3_dma_kernel]: {nwg} workgroups; per wave: prologue {cyc[:, 0].mean():.0f}, main loop "
                      f"{cyc[:, 1].mean():.0f}, epilogue {cyc[:, 2].mean():.0f} cycles; workgroup wall time {wall.mean():.2f} us "
                      f"(=> {cyc.sum(1).mean() / wall.mean() / 1e3:.2f} GHz)")
                continue
            rec = v[:nwg * 8 * 8].view(nwg * 8, 8)[:, :5]
            assert (rec >= 0).all(), "stamped library not loaded or a different kernel ran"
            tot = rec.sum(1)
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                ops.conv2d(src, w, y if what == "forward" else x, cin=cin if what == "forward" else cout,
                           cout=cout if what == "forward" else cin, coutp=cout if what == "forward" else cin, ipe=B, ks=3,
                           stride=1, pad=1, stats=stats)
            t1.record(); torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / 10
            print(f"{name} {what}: {ms:.3f} ms per launch = {2.0 * N * H * H * cin * cout * 9 / ms / 1e9:.0f} TFLOP/s (stamped build)")
            share = (rec / tot[:, None]).mean(0)
            print(f"{name} {what}: kernel plan {plan}, {nwg} workgroups x 8 waves, {tot.mean().item():.0f} stamped cycles per wave "
                  f"(min {tot.min().item():.0f}, max {tot.max().item():.0f})")
            for ph, s in zip(PHASES, share.tolist()):
                print(f"    {100 * s:5.1f} %  {ph}")


if __name__ == "__main__":
    main()
