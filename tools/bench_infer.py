#!/usr/bin/env python3
"""Closed-loop inference shape of autoagents/image_agent.py:127-177: B=1, eval mode, 224x224,
`model.sample(...)` per tick.  Eager launch chain vs the same chain captured once into a HIP graph."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pmoe_amd.model.moe import get_model  # noqa: E402
from pmoe_amd.utils import stage2_model_cfg  # noqa: E402


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
    model = get_model(stage2_model_cfg("moe", E, dropout=0.3)).cuda().eval()
    img = torch.rand(1, 4, 3, 224, 224, device="cuda")
    spd = torch.rand(1, 1, device="cuda")
    cmd = torch.nn.functional.one_hot(torch.tensor([2]), 6).float().cuda()

    def tick():
        with torch.no_grad():
            return model.mixture_params(img, spd, cmd)

    for _ in range(3):
        tick()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        out = tick()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n * 1e3
    print(f"E={E} B=1 224x224 eval  eager: {eager:.3f} ms/tick", flush=True)

    from pmoe_amd.infer import GraphedMixture
    gm = GraphedMixture(model, img, spd, cmd)
    ref = [t.clone() for t in out]
    got = gm(img, spd, cmd)
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b), "graph replay differs from the eager chain"
    t0 = time.perf_counter()
    for _ in range(n):
        got = gm(img, spd, cmd)
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / n * 1e3
    print(f"E={E} B=1 224x224 eval  HIP graph: {graphed:.3f} ms/tick ({eager / graphed:.1f}x)", flush=True)
    a = gm.sample(img, spd, cmd)
    assert a.shape == (1, 2)

    from pmoe_amd.infer import PlannedMixture
    pm = PlannedMixture(model, img, spd, cmd)
    got = pm(img, spd, cmd)
    torch.cuda.synchronize()
    for x, y in zip(ref, got):
        assert torch.equal(x, y), "launch-plan replay differs from the eager chain"
    t0 = time.perf_counter()
    for _ in range(n):
        got = pm(img, spd, cmd)
    torch.cuda.synchronize()
    planned = (time.perf_counter() - t0) / n * 1e3
    print(f"E={E} B=1 224x224 eval  recorded launch plan ({len(pm.plan.calls)} C-ABI calls, no capture): {planned:.3f} ms/tick "
          f"({eager / planned:.1f}x)", flush=True)
    if "--profile" in sys.argv:
        from pmoe_amd import ops
        ops.profile_begin()
        tick()
        rows = ops.profile_end()
        print(f"  {len(rows)} launches, {sum(r[2] for r in rows):.3f} ms of kernel time")
        for name, meta, ms in sorted(rows, key=lambda r: -r[2])[:16]:
            print(f"  {name:22s} {meta.get('name', ''):28s} {ms * 1e3:7.1f} us  kernel={meta.get('kernel', '')}")


if __name__ == "__main__":
    main()
