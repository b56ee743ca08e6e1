#!/usr/bin/env python3
"""Pick WELL-CONDITIONED golden cases by measurement.  TEST INFRASTRUCTURE.

For each candidate (type, E, B, size) the CPU oracle is evaluated in float32 and in float64 on the same seeded weights and
inputs; printed: the f32-vs-f64 drift of every forward output (max |d| / max |ref|) and of every parameter gradient
(rel-L2 per tensor: median, worst, name of the worst).  A case qualifies for the flat north_star tolerances when the
forward drift is <= 1e-5 and EVERY gradient tensor drifts <= 1e-3: then an exact-f32 implementation has no excuse.

    python oracle/probe_conditioning.py moe:4:8:128 moe:4:8:192 moe:4:8:256 punet:2:8:96:2
"""
import copy
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from oracle import pmoe_oracle as O      # noqa: E402
from oracle import weights as W          # noqa: E402


def moe_case(typ, E, B, size):
    cfg = O.stage2_cfg(typ, E, dropout=0.0)
    m = O.get_model(cfg)
    W.fill_state_dict(m, seed=0)
    m.train()
    inp = W.make_inputs(B, size, size, seed=1234)

    def run(model, cast):
        model.zero_grad()
        d, s = model(cast(inp["images"]), cast(inp["speed"]), cast(inp["command"]))
        loss = O.moe_loss(d, s, cast(inp["control"]), cast(inp["target_speed"]).clone(), cfg.loss_coefs)
        loss.backward()
        outs = dict(probs=d.mixture_distribution.probs, mean=d.component_distribution.base_dist.loc,
                    std=d.component_distribution.base_dist.scale, speeds=s, loss=loss.reshape(1))
        return {k: v.detach().double() for k, v in outs.items()}, {k: p.grad.double() for k, p in model.named_parameters()}
    return m, run


def punet_case(typ, B, size, F):
    cfg = O.stage2_cfg(typ, 2, dropout=0.0, future_frames=F)
    m = O.get_model(cfg)
    W.fill_state_dict(m, seed=0)
    m.train()
    inp = W.make_inputs(B, size, size, seed=1234)

    def run(model, cast):
        model.zero_grad()
        a, s = model(cast(inp["images"]), cast(inp["speed"]), cast(inp["command"]))
        loss = O.punet_loss(a, s, cast(inp["control"]), cast(inp["target_speed"]), cfg.loss_coefs)
        loss.backward()
        return ({"actions": a.detach().double(), "speeds": s.detach().double(), "loss": loss.detach().double().reshape(1)},
                {k: p.grad.double() for k, p in model.named_parameters() if p.grad is not None})
    return m, run


def main():
    torch.set_num_threads(8)
    for spec in sys.argv[1:]:
        f = spec.split(":")
        t0 = time.time()
        if f[0] in ("punet", "punet_inter"):
            m, run = punet_case(f[0], int(f[2]), int(f[3]), int(f[4]))
        else:
            m, run = moe_case(f[0], int(f[1]), int(f[2]), int(f[3]))
        o32, g32 = run(m, lambda t: t)
        m64 = copy.deepcopy(m).double()
        o64, g64 = run(m64, lambda t: t.double())
        fwd = {k: ((o32[k] - o64[k]).abs().max() / (o64[k].abs().max() + 1e-30)).item() for k in o64}
        tot = sum(g.norm().item() ** 2 for g in g64.values()) ** 0.5
        rows = sorted(((g32[k] - g64[k]).norm().item() / (g64[k].norm().item() + 1e-30), k) for k in g64
                      if g64[k].norm().item() > 1e-6 * tot)
        print(f"{spec}: {time.time() - t0:.0f}s  forward drift " + " ".join(f"{k}={v:.1e}" for k, v in fwd.items()))
        print(f"    gradient drift: median {rows[len(rows) // 2][0]:.1e}  worst {rows[-1][0]:.1e} ({rows[-1][1]})  "
              f"tensors > 1e-3: {sum(r > 1e-3 for r, _ in rows)} / {len(rows)}", flush=True)


if __name__ == "__main__":
    main()
