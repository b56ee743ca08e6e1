#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates tests/golden/bf16_bounds.pt from the CPU oracle (no reference import needed: the
oracle itself is pinned to the imported reference by tests/test_oracle_golden.py):

  * per golden case: the float64 oracle's forward outputs (the truth the bf16 checks measure against) and the error of
    the bf16-STORAGE-emulating oracle (oracle/bf16_emulation.py) against it, per output -- the bound of the -m gpu bf16
    parity tests is 1.25 x that measured figure;
  * for the two largest f32 gradient cases (g3, g10): per parameter tensor the norm of the float64 gradient and the f32
    oracle's own relative drift from it, so that the GPU box does not pay for the float64 backward.

  python oracle/make_bounds.py          (about 10 minutes on 8 cores)
"""
import copy
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import bf16_emulation as E  # noqa: E402
from oracle import pmoe_oracle as O  # noqa: E402
from oracle import weights as W  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"
MOE_CASES = ["g1_moe_e4_b2_128", "g4_moealt_e4_b2_64", "g5_moe_e3_b3_96", "g6_moeshared_k4_b6_96", "g10_moe_e4_b32_64",
             "g2_moe_e4_b1_224_eval", "g7_moeshared_k6_b1_224_eval"]
PUNET_CASES = ["p1_punet_b2_64_f2", "p2_punet_b1_64_f6_eval", "p3_punetinter_b2_64_f2"]
GRAD_CASES = ["g3_moe_e8_b2_128", "g10_moe_e4_b32_64"]
VARIANTS = ("all", "fused")


def build(meta):
    kw = dict(dropout=0.0)
    if "future_frames" in meta:
        kw["future_frames"] = meta["future_frames"]
    cfg = O.stage2_cfg(meta["type"], meta["n_experts"], **kw)
    model = O.get_model(cfg)
    W.fill_state_dict(model, seed=meta["weight_seed"])
    model.train(meta["train"])
    inp = W.make_inputs(meta["batch"], meta["size"], meta["size"], seed=meta["input_seed"])
    return cfg, model, inp


def outputs(model, inp, dtype):
    m = copy.deepcopy(model).to(dtype)
    with torch.no_grad():
        r = m(inp["images"].to(dtype), inp["speed"].to(dtype), inp["command"].to(dtype))
    if isinstance(r[0], torch.Tensor):                          # PUNetExpert: (actions, speed)
        return {"actions": r[0].double(), "speed": r[1].double()}
    d, s = r
    return {"probs": d.mixture_distribution.probs.double(), "mean": d.component_distribution.base_dist.loc.double(),
            "std": d.component_distribution.base_dist.scale.double(), "speeds": s.double()}


def forward_bounds(name):
    meta = torch.load(GOLDEN / f"{name}.pt", weights_only=False)["meta"]
    cfg, model, inp = build(meta)
    ref = outputs(model, inp, torch.float64)
    f32 = outputs(model, inp, torch.float32)
    rec = {"f64": ref, "f32_oracle": {k: (E.metric(f32[k], ref[k]), E.rms_metric(f32[k], ref[k])) for k in ref}, "emul": {}}
    for var in VARIANTS:
        m = copy.deepcopy(model)
        E.emulate_bf16(m, var)
        inp_b = dict(inp)
        inp_b["images"] = inp["images"].to(torch.bfloat16).float()
        got = outputs(m, inp_b, torch.float32)
        rec["emul"][var] = {k: (E.metric(got[k], ref[k]), E.rms_metric(got[k], ref[k])) for k in ref}
    print(name, {v: {k: "%.2e" % e[0] for k, e in rec["emul"][v].items()} for v in VARIANTS}, flush=True)
    return rec


def grad_bounds(name):
    meta = torch.load(GOLDEN / f"{name}.pt", weights_only=False)["meta"]
    cfg, model, inp = build(meta)
    m32 = copy.deepcopy(model)
    d, s = m32(inp["images"], inp["speed"], inp["command"])
    O.moe_loss(d, s, inp["control"], inp["target_speed"].clone(), cfg.loss_coefs).backward()
    m64 = copy.deepcopy(model).double()
    d, s = m64(inp["images"].double(), inp["speed"].double(), inp["command"].double())
    O.moe_loss(d, s, inp["control"].double(), inp["target_speed"].double(), cfg.loss_coefs).backward()
    g32 = {k: p.grad for k, p in m32.named_parameters()}
    rec = {}
    for k, p in m64.named_parameters():
        n64 = p.grad.norm().item()
        rec[k] = (n64, ((g32[k].double() - p.grad).norm() / (n64 + 1e-300)).item())
    print(name, "f64 gradients:", len(rec), "tensors, worst f32-oracle drift %.2e" % max(v[1] for v in rec.values()), flush=True)
    return rec


def main():
    torch.manual_seed(0)
    out = {"forward": {}, "grad": {}}
    for name in MOE_CASES + PUNET_CASES:
        out["forward"][name] = forward_bounds(name)
    for name in GRAD_CASES:
        out["grad"][name] = grad_bounds(name)
    torch.save(out, GOLDEN / "bf16_bounds.pt")
    print("wrote", GOLDEN / "bf16_bounds.pt")


if __name__ == "__main__":
    main()
