"""Layer-by-layer comparison of the HIP path with the CPU oracle under emulated storage / quantisation policies (bf16 storage,
bf16 + the fp8 policy of oracle/fp8_policy.py): per BatchNorm(+ReLU) output of expert 0, the rel-L2 distance of each side to the
float64 oracle and to each other, and the number of convolution-input elements beyond the e4m3 range at the policy's input
scale.  Used by tests/test_fp8_gpu.py and tests/experiments/diag_fp8_eval.py."""
import copy

import torch

from oracle import bf16_emulation as E, fp8_policy as P8
from tests.parity_util import GOLDEN, build_pair

SAT = P8.FP8_MAX / P8.IN_SCALE          # 28: |x| * 16 > 448 saturates


def oracle_acts(model, inp, dtype):
    acts, hooks = {}, []
    bb = model.moe[0].backbone
    names = {bb.conv1.layer1.conv1[1]: "stem.bn1"}
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(bb, f"layer{li}")):
            names[blk.bn1] = f"layer{li}.{bi}.bn1"
            names[blk] = f"layer{li}.{bi}.bn2"
    for mod, nm in names.items():
        post = torch.relu if nm.endswith("bn1") else (lambda t: t)          # the block's own output is already post-ReLU
        hooks.append(mod.register_forward_hook(lambda m, i, o, nm=nm, post=post: acts.__setitem__(nm, post(o.detach()).double())))
    with torch.no_grad():
        d, s = model(inp["images"].to(dtype), inp["speed"].to(dtype), inp["command"].to(dtype))
    for h in hooks:
        h.remove()
    return acts, dict(mean=d.component_distribution.base_dist.loc.double(), speeds=s.double())


def layerwise(name, fp8):
    """-> rows [(layer, d_hip, d_emul, d_between, sat_hip, sat_emul, max_hip)], outputs {k: (err_hip, err_emul, hip_vs_emul)}"""
    g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
    B = g["meta"]["batch"]
    ocfg, oracle, model, inp = build_pair(g, torch.bfloat16)
    a64, o64 = oracle_acts(copy.deepcopy(oracle).double(), inp, torch.float64)
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
    ah = {k: v[0][0:B, :, :, v[1]:v[1] + v[2]].permute(0, 3, 1, 2).double().cpu() for k, v in eng.debug_acts.items() if k in a64}
    eng.debug_acts = None
    rows = []
    for k in a64:
        if k not in ah:
            continue
        n = a64[k].norm()
        rows.append((k, ((ah[k] - a64[k]).norm() / n).item(), ((ae[k] - a64[k]).norm() / n).item(),
                     ((ah[k] - ae[k]).norm() / n).item(), int((ah[k].abs() > SAT).sum()), int((ae[k].abs() > SAT).sum()),
                     ah[k].abs().max().item()))
    outs = {}
    for k, hv in (("mean", dist.hip_params[1].double().cpu()), ("speeds", speeds.double().cpu())):
        r, ev = o64[k], oe[k]
        met = lambda a, b: ((a - b).abs() / (1 + b.abs())).max().item()
        outs[k] = (met(hv, r), met(ev, r), met(hv, ev))
    return rows, outs
