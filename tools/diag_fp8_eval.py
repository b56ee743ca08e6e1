#!/usr/bin/env python3
"""Layer-by-layer: HIP path (bf16, then the fp8 policy) vs the CPU oracle with bf16 storage (and the fp8 policy) emulated, on
the eval-mode golden g2 (synthetic running statistics: activations far outside the policy's range).  Prints, per BatchNorm
output, the rel-L2 distance to the float64 oracle for both sides and between them, and the count of conv-input elements beyond
the e4m3 range at IN_SCALE (|x| * 16 > 448).  VERDICT r2 weak item 4."""
import copy
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from oracle import bf16_emulation as E, fp8_policy as P8      # noqa: E402
from tests.parity_util import GOLDEN, build_pair               # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "g2_moe_e4_b1_224_eval"
g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
B = g["meta"]["batch"]


def oracle_acts(model, inp, dtype):
    acts = {}
    hooks = []
    ex = model.moe[0]
    bb = ex.backbone
    names = {bb.conv1.layer1.conv1[1]: "stem.bn1"}
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(bb, f"layer{li}")):
            names[blk.bn1] = f"layer{li}.{bi}.bn1"
            names[blk] = f"layer{li}.{bi}.bn2"
    for mod, nm in names.items():
        if nm.endswith("bn1"):
            hooks.append(mod.register_forward_hook(lambda m, i, o, nm=nm: acts.__setitem__(nm, torch.relu(o.detach()).double())))
        else:
            hooks.append(mod.register_forward_hook(lambda m, i, o, nm=nm: acts.__setitem__(nm, o.detach().double())))
    with torch.no_grad():
        d, s = model(inp["images"].to(dtype), inp["speed"].to(dtype), inp["command"].to(dtype))
    for h in hooks:
        h.remove()
    return acts, dict(mean=d.component_distribution.base_dist.loc.double(), speeds=s.double())


ocfg, oracle, model, inp = build_pair(g, torch.bfloat16)
a64, o64 = oracle_acts(copy.deepcopy(oracle).double(), inp, torch.float64)
res = {}
for fp8 in (False, True):
    m = copy.deepcopy(oracle)
    E.emulate_bf16(m, "all")
    if fp8:
        P8.apply_fp8_policy(m)
    ae, oe = oracle_acts(m, inp, torch.float32)
    model.fp8_weights = fp8
    eng = model._engine()
    eng.debug_acts = {}
    eng.fold_bn_eval = False          # every BatchNorm as its own pass, so that its output exists
    dev = {k: v.cuda() for k, v in inp.items()}
    with torch.no_grad():
        dist, speeds = model(dev["images"], dev["speed"], dev["command"])
    ah = {k: v[0][0:B, :, :, v[1]:v[1] + v[2]].permute(0, 3, 1, 2).double().cpu() for k, v in eng.debug_acts.items()
          if k in a64}
    print(f"---- {'fp8 policy' if fp8 else 'bf16'}: rel-L2 to float64 (HIP | emulation), HIP vs emulation, elements beyond 28 (HIP | emulation)")
    for k in a64:
        if k not in ah:
            continue
        n = a64[k].norm()
        print(f"{k:18s} {((ah[k] - a64[k]).norm() / n).item():.3e} | {((ae[k] - a64[k]).norm() / n).item():.3e}   "
              f"{((ah[k] - ae[k]).norm() / n).item():.3e}   {int((ah[k].abs() > 28).sum())} | {int((ae[k].abs() > 28).sum())}"
              f"   max {ah[k].abs().max().item():.1f}")
    hm, hs = dist.hip_params[1].double().cpu(), speeds.double().cpu()
    for k, hv, ev in (("mean", hm, oe["mean"]), ("speeds", hs, oe["speeds"])):
        r = o64[k]
        print(f"{k}: HIP err {((hv - r).abs() / (1 + r.abs())).max().item():.3e}  emulation err {((ev - r).abs() / (1 + r.abs())).max().item():.3e}"
              f"  HIP vs emulation {((hv - ev).abs() / (1 + ev.abs())).max().item():.3e}")
