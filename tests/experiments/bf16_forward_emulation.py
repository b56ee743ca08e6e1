#!/usr/bin/env python3
"""Is the bf16 path's end-to-end forward error the storage format or the kernels?  CPU experiment: the oracle with every
parameter matrix and every layer output rounded to bf16 (f32 arithmetic in between) against the float64 oracle, metric
max |got - ref| / (1 + |ref|) on probs / mean / std / speeds -- the same metric tests/parity_util.py applies to the HIP
path.  Measured (build container): B=32 64x64 (golden g10): 6.1e-3 / 3.6e-2 / 2.2e-2 / 2.4e-2 (HIP bf16 path on the same
case: worst 3.5e-2); B=2 128x128 (g1): 1.7e-3 / 1.1e-2 / 7.0e-3 / 1.1e-2; B=16 128x128: 4.1e-3 / 1.6e-2 / 9.0e-3 / 1.1e-2.
  python tests/experiments/bf16_forward_emulation.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch, torch.nn as nn
from oracle import pmoe_oracle as O, weights as W
class RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x): return x.to(torch.bfloat16).to(x.dtype)
    @staticmethod
    def backward(ctx, g): return g.to(torch.bfloat16).to(g.dtype)
def run(mode, B_, S, seed_in=1234):
    cfg = O.stage2_cfg("moe", 4, dropout=0.0)
    model = O.get_model(cfg); W.fill_state_dict(model, seed=0); model.train()
    inp = W.make_inputs(B_, S, S, seed=seed_in)
    if mode == "f64":
        model = model.double(); inp = {k: v.double() for k, v in inp.items()}
    if mode == "bf16":
        for mod in model.modules():
            if isinstance(mod, (nn.Conv2d, nn.BatchNorm2d, nn.Linear, nn.MaxPool2d, O.EfficientBlock)):
                mod.register_forward_hook(lambda m_, i, o: RoundBF16.apply(o))
        with torch.no_grad():
            for p in model.parameters():
                if p.dim() > 1: p.copy_(p.to(torch.bfloat16).float())
        inp["images"] = inp["images"].to(torch.bfloat16).float()
    with torch.no_grad():
        d, s = model(inp["images"], inp["speed"], inp["command"])
    return [t.double() for t in (d.mixture_distribution.probs, d.component_distribution.base_dist.loc, d.component_distribution.base_dist.scale, s)]
for (B_, S) in [(32, 64), (2, 128), (16, 128)]:
    ref = run("f64", B_, S)
    for mode in ("f32", "bf16"):
        got = run(mode, B_, S)
        print(B_, S, mode, ["%.2e" % ((a - b).abs() / (1 + b.abs())).max().item() for a, b in zip(got, ref)], flush=True)
