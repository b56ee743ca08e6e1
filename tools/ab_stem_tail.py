#!/usr/bin/env python3
"""Stem tail (BN+ReLU, bn1+ReLU, max-pool 3/2/1 of the 256x256x64 stem output): the row-walking pool / dz2 kernels against
the gather kernels they replace, interleaved in one process (PMOE_STEM_WALK / PMOE_STEM_WALK_KO are read per launch), and
bit-compared.   python tools/ab_stem_tail.py [E B H W]          (GPU box)"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from pmoe_amd import ops


def main():
    E, B, H, W = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (4, 64, 256, 256)
    C, dev, dt = 64, "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    N = E * B
    z2 = torch.randn(N, H, W, C, device=dev, generator=g).to(dt)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dp = torch.randn(N, Ho, Wo, C, device=dev, generator=g).to(dt)
    f = lambda lo, hi: torch.rand(E, C, device=dev, generator=g) * (hi - lo) + lo
    sc2, sh2, sc1, sh1, mu1, is1, mu2, is2 = f(.5, 1.5), f(-.3, .3), f(.5, 1.5), f(-.3, .3), f(.2, .5), f(.8, 1.2), f(-.1, .1), f(.8, 1.2)
    c11, c21, c12, c22 = f(-.01, .01), f(-.01, .01), f(-.01, .01), f(-.01, .01)
    consts = [sc2, sh2, sc1, sh1, mu1, is1, mu2, is2, c11, c21, c12, c22]
    part = torch.empty(E, 1024, 2, C, device=dev)
    variants = {"gather": {"PMOE_STEM_WALK": "0"}, "walk KO=2": {"PMOE_STEM_WALK": "1", "PMOE_STEM_WALK_KO": "2"},
                "walk KO=4": {"PMOE_STEM_WALK": "1", "PMOE_STEM_WALK_KO": "4"}}
    out = {}
    gb_pool = (z2.numel() * 2 + dp.numel() * 3) / 1e9
    gb_dz = (z2.numel() * 4 + dp.numel() * 3) / 1e9
    times = {k: [[], []] for k in variants}
    for rep in range(6):
        for name, env in variants.items():
            os.environ.update(env)
            y = torch.empty(N, Ho, Wo, C, dtype=dt, device=dev)
            am = torch.empty(N, Ho, Wo, C, dtype=torch.uint8, device=dev)
            dz = torch.empty_like(z2)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            ops.stem_tail_pool(z2, y, am, sc2, sh2, sc1, sh1, mu2, mu1, B)
            ev[1].record()
            ops.stem_tail_bwd(3, z2, dp, am, dz, consts, part, 1024, E, B)
            ev[2].record()
            torch.cuda.synchronize()
            if rep:
                times[name][0].append(ev[0].elapsed_time(ev[1]))
                times[name][1].append(ev[1].elapsed_time(ev[2]))
            if rep == 0:
                out[name] = (y, am, dz)
            del y, am, dz
    ref = out["gather"]
    for name in variants:
        tp, tb = (sorted(t)[len(t) // 2] for t in times[name])
        same = all(torch.equal(a, b) for a, b in zip(out[name], ref))
        print(f"{name:10s}: pool {tp:.3f} ms ({gb_pool / tp:.2f} TB/s)   dz2 {tb:.3f} ms ({gb_dz / tb:.2f} TB/s)   "
              f"bit-identical to gather: {same}")
        if not same:
            for nm, a, b in zip(("y", "argmax", "dz2"), out[name], ref):
                bad = a != b
                print("    ", nm, "mismatches:", bad.sum().item(), "of", a.numel(), " max |diff|",
                      (a.float() - b.float()).abs().max().item(), " max |ref| there", b[bad].float().abs().max().item() if bad.any() else 0,
                      " first at", bad.nonzero()[0].tolist() if bad.any() else None,
                      a[bad][:4].tolist() if bad.any() else None, b[bad][:4].tolist() if bad.any() else None)


if __name__ == "__main__":
    main()
