#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates tests/golden/bf16_bounds.pt from the CPU oracle (no reference import needed: the
oracle itself is pinned to the imported reference by tests/test_oracle_golden.py):

  * per golden case: the float64 oracle's forward outputs (the truth the bf16 checks measure against) and the error of
    the bf16-STORAGE-emulating oracle (oracle/bf16_emulation.py) against it, per output -- the bound of the -m gpu bf16
    parity tests is 1.25 x that measured figure;
  * for the two largest f32 gradient cases (g3, g10): per parameter tensor the norm of the float64 gradient and the f32
    oracle's own relative drift from it, so that the GPU box does not pay for the float64 backward.

  python oracle/make_bounds.py [--keep-grads]         (about 25 minutes on 8 cores)
"""
import copy
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import bf16_emulation as E  # noqa: E402
from oracle import fp8_policy as P8  # noqa: E402
from oracle import pmoe_oracle as O  # noqa: E402
from oracle import weights as W  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"
MOE_CASES = ["g1_moe_e4_b2_128", "g4_moealt_e4_b2_64", "g5_moe_e3_b3_96", "g6_moeshared_k4_b6_96", "g10_moe_e4_b32_64",
             "g11_moe_e4_b8_128", "g2_moe_e4_b1_224_eval", "g7_moeshared_k6_b1_224_eval"]
PUNET_CASES = ["p1_punet_b2_64_f2", "p2_punet_b1_64_f6_eval", "p3_punetinter_b2_64_f2", "p5_pmoe_e2_b2_64_f2",
               "p6_punet_b8_96_f2"]
GRAD_CASES = ["g3_moe_e8_b2_128", "g10_moe_e4_b32_64"]
VARIANTS = ("all", "fused")
JITTERS = 6        # extra draws of the emulation error: the same case with its images jittered by 1 % (seeds 1..JITTERS)
FP8_CASES = ["g1_moe_e4_b2_128", "g5_moe_e3_b3_96", "g10_moe_e4_b32_64", "g2_moe_e4_b1_224_eval"]   # BASELINE config 5 policy


def build(meta):
    kw = dict(dropout=0.0)
    if "future_frames" in meta:
        kw["future_frames"] = meta["future_frames"]
    if meta["type"].startswith("pmoe"):
        kw["exclude_freeze"] = ["lat_weights", "long_weights"]
    cfg = O.stage2_cfg(meta["type"], meta["n_experts"], **kw)
    model = O.get_model(cfg)
    W.fill_state_dict(model, seed=meta["weight_seed"])
    model.train(meta["train"])
    inp = W.make_inputs(meta["batch"], meta["size"], meta["size"], seed=meta["input_seed"])
    return cfg, model, inp


def outputs(model, inp, dtype, clone=True):
    m = copy.deepcopy(model).to(dtype) if clone else model     # (instance-patched forwards do not survive a deepcopy)
    with torch.no_grad():
        if hasattr(m, "blend"):                                 # PMoE: its deterministic parts (tests/punet_parity.run_pmoe_case)
            args = (inp["images"].to(dtype), inp["speed"].to(dtype), inp["command"].to(dtype))
            pa, _ = m.punet(*args)
            d, _ = m.moe(*args)
            return {"punet_actions": pa.double(), "probs": d.mixture_distribution.probs.double(),
                    "mean": d.component_distribution.base_dist.loc.double(),
                    "std": d.component_distribution.base_dist.scale.double()}
        r = m(inp["images"].to(dtype), inp["speed"].to(dtype), inp["command"].to(dtype))
    if isinstance(r[0], torch.Tensor):                        # PUNetExpert: (actions, speed)
        return {"actions": r[0].double(), "speed": r[1].double()}
    d, s = r
    return {"probs": d.mixture_distribution.probs.double(), "mean": d.component_distribution.base_dist.loc.double(),
            "std": d.component_distribution.base_dist.scale.double(), "speeds": s.double()}


def jittered(inp, seed):
    """draw `seed` of the case: the golden inputs themselves (0) or their images jittered by 1 % (the emulation error is a
    random quantity -- a maximum over a few dozen output values -- so its upper range is estimated from several draws)."""
    if seed == 0:
        return inp
    g = torch.Generator().manual_seed(7000 + seed)
    out = dict(inp)
    out["images"] = (inp["images"] + 0.02 * (torch.rand(inp["images"].shape, generator=g) - 0.5)).clamp(0.0, 1.0)
    return out


def emulation_draws(name, fp8):
    """{variant: [per draw {output: (max metric, rms metric)}]}, the float64 outputs of draw 0 and the f32 oracle's drift."""
    meta = torch.load(GOLDEN / f"{name}.pt", weights_only=False)["meta"]
    cfg, model, inp0 = build(meta)
    draws = {v: [] for v in VARIANTS}
    ref0 = f32_0 = None
    for seed in range(JITTERS + 1):
        inp = jittered(inp0, seed)
        ref = outputs(model, inp, torch.float64)
        if seed == 0:
            ref0, f32_0 = ref, outputs(model, inp, torch.float32)
        for var in VARIANTS:
            m = copy.deepcopy(model)
            E.emulate_bf16(m, var)
            if fp8:
                P8.apply_fp8_policy(m)
            inp_b = dict(inp)
            inp_b["images"] = inp["images"].to(torch.bfloat16).float()
            got = outputs(m, inp_b, torch.float32, clone=False)
            draws[var].append({k: (E.metric(got[k], ref[k]), E.rms_metric(got[k], ref[k])) for k in ref})
    worst = {k: max(d[k][0] for v in VARIANTS for d in draws[v]) for k in ref0}
    print(name, "fp8" if fp8 else "bf16", "draw 0:", {k: "%.2e" % draws["all"][0][k][0] for k in ref0},
          "worst of %d draws:" % (2 * (JITTERS + 1)), {k: "%.2e" % v for k, v in worst.items()}, flush=True)
    return draws, ref0, f32_0


def forward_bounds(name):
    draws, ref, f32 = emulation_draws(name, fp8=False)
    return {"f64": ref, "f32_oracle": {k: (E.metric(f32[k], ref[k]), E.rms_metric(f32[k], ref[k])) for k in ref},
            "emul": draws}


def fp8_bounds(name, rec):
    """error of the oracle with bf16 storage AND the fp8 policy of oracle/fp8_policy.py, against the float64 oracle; for the
    train-mode cases also how well that emulation's parameter gradients align with the f32 oracle's (median cosine)."""
    rec["emul_fp8"], _, _ = emulation_draws(name, fp8=True)
    meta = torch.load(GOLDEN / f"{name}.pt", weights_only=False)["meta"]
    if not meta["train"]:
        return
    cfg, model, inp = build(meta)
    grads = []
    for fp8 in (False, True):
        m = copy.deepcopy(model)
        if fp8:
            E.emulate_bf16(m, "all")
            P8.apply_fp8_policy(m)
        d, s = m(inp["images"], inp["speed"], inp["command"])
        O.moe_loss(d, s, inp["control"], inp["target_speed"].clone(), cfg.loss_coefs).backward()
        grads.append({k: p.grad for k, p in m.named_parameters()})
    cos = sorted(torch.nn.functional.cosine_similarity(grads[1][k].flatten(), grads[0][k].flatten(), dim=0).item()
                 for k in grads[0] if grads[0][k].numel() >= 1024)
    rec["emul_fp8_grad_median_cos"] = cos[len(cos) // 2]
    print(name, "fp8 emulation: median gradient cosine vs the f32 oracle %.3f" % cos[len(cos) // 2], flush=True)


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
    path = GOLDEN / "bf16_bounds.pt"
    keep_grads = "--keep-grads" in sys.argv and path.exists()   # reuse the (slow) float64 gradient records
    if "--fp8-only" in sys.argv:                                # refresh the fp8-policy records after a policy change (round 3)
        out = torch.load(path, weights_only=False)
        for name in FP8_CASES:
            fp8_bounds(name, out["forward"][name])
        torch.save(out, path)
        return
    if "--only" in sys.argv:                                    # add / refresh the forward record of one case
        name = sys.argv[sys.argv.index("--only") + 1]
        out = torch.load(path, weights_only=False)
        out["forward"][name] = forward_bounds(name)
        torch.save(out, path)
        return
    out = {"forward": {}, "grad": torch.load(path, weights_only=False)["grad"] if keep_grads else {}}
    for name in MOE_CASES + PUNET_CASES:
        out["forward"][name] = forward_bounds(name)
    if not keep_grads:
        for name in GRAD_CASES:
            out["grad"][name] = grad_bounds(name)
    for name in FP8_CASES:
        fp8_bounds(name, out["forward"][name])
    torch.save(out, GOLDEN / "bf16_bounds.pt")
    print("wrote", GOLDEN / "bf16_bounds.pt")


if __name__ == "__main__":
    main()
