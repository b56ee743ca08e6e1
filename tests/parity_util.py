"""Shared by tests/test_model_gpu.py and __graft_entry__.smoke(): run one golden case through the HIP
model on cuda:0 and compare with (a) the committed golden vectors produced by the imported reference
and (b) the CPU oracle run live on the same seeded weights/inputs (full gradient tensors)."""
from pathlib import Path

import torch

from oracle import pmoe_oracle as O
from oracle import weights as W
from pmoe_amd.loss import moe_loss
from pmoe_amd.model.moe import get_model
from pmoe_amd.utils import stage2_model_cfg

GOLDEN = Path(__file__).resolve().parent / "golden"

# Forward outputs -- north_star: within 1e-4 fp32 / 1e-2 bf16.
#   f32 : max|got-ref| <= 1e-4 * max|ref|                      (measured ~2e-6)
#   bf16: |got-ref| <= 3e-2*(1+|ref|) elementwise.  The 1e-2 of north_star is NOT met end to end on these
#         tiny-batch goldens (measured up to 2.8e-2); the same CPU oracle with bf16-rounded layer outputs is
#         off by the same amount (tests/experiments/bf16_conditioning.py), i.e. it is the storage format, not the kernels:
#         every bf16 kernel meets 1e-2 on its own (tests/test_ops_gpu.py).
# Gradients:
#   f32 : run_forced_case (below): the float64 oracle differentiates the same piecewise-linear function as the HIP path
#         (ReLU / max-pool decisions forced, tests/forced_masks.py); EVERY tensor of EVERY expert within FORCED_GRAD_TOL,
#         every decision disagreement a near-tie.  This is what proves the backward algorithm.
#   bf16: statistical agreement only (median cosine >= 0.85, total norm within 20%: the bf16-storage-emulating oracle has a
#         median rel-L2 of 0.4 against f64, DESIGN.md "numerics"); the bf16 KERNELS are held to 1e-2 per op in
#         tests/test_ops_gpu.py.
TOL = {torch.float32: 1e-4, torch.bfloat16: 3e-2}
# bf16 forward bound, measured instead of flat: tests/golden/bf16_bounds.pt (oracle/make_bounds.py)
# holds, per golden case and output, the float64 oracle's value and the error of the bf16-STORAGE-emulating oracle
# (oracle/bf16_emulation.py: weights and every layer output rounded to bf16, f32 arithmetic) against it.  The HIP bf16
# path must be within north_star's 1e-2 of the float64 truth, or -- where the storage format itself makes that impossible
# -- within BF16_SLACK x the emulation's own error, output by output (metric max |d| / (1 + |ref|)).  That error is a
# maximum over a few dozen values, i.e. a random quantity: the yardstick is the largest of 14 draws of it (two emulation
# variants, "every layer stored" / "BatchNorm passes fused", on the golden inputs and on 6 copies with images jittered by 1 %).
BF16_SLACK = 1.25
BF16_FLOOR = 1e-2
_BOUNDS = None


def bf16_bounds(name):
    global _BOUNDS
    if _BOUNDS is None:
        _BOUNDS = torch.load(GOLDEN / "bf16_bounds.pt", weights_only=False)
    return _BOUNDS["forward"].get(name)


def bf16_fwd_report(name, outs):
    """{output: (HIP error vs float64, emulating-oracle error vs float64)} for the outputs of golden case ``name``."""
    b = bf16_bounds(name)
    rep = {}
    for k, got in outs.items():
        ref = b["f64"][k]
        err = ((got.detach().double().cpu() - ref).abs() / (1 + ref.abs())).max().item()
        rep[k] = (err, max(BF16_FLOOR / BF16_SLACK, emul_worst(b["emul"], k)))
    return rep


def emul_worst(draws, k):
    """largest max-metric error of output ``k`` over all variants and draws of an emulation record."""
    return max(d[k][0] for v in draws for d in draws[v])


def fwd_err(got, ref, dtype):
    """error in units of the tolerance (<= 1 passes)."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    if dtype == torch.float32:
        return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item() / TOL[dtype]
    return ((got - ref).abs() / (TOL[dtype] + TOL[dtype] * ref.abs())).max().item()


def rel_err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def rel_l2(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).norm() / (ref.norm() + 1e-20)).item()


_ORACLES = {}      # (type, experts, weight seed, dropout) -> the filled CPU oracle: building + filling one costs ~2 s, a deepcopy 0.1 s


def cached_oracle(key, make):
    """a private deepcopy of the oracle `make()` builds for `key` (built once per test process: the suite constructs the same
    handful of configurations about a hundred times)"""
    import copy
    if key not in _ORACLES:
        _ORACLES[key] = make()
    return copy.deepcopy(_ORACLES[key])


def build_pair(g, dtype, dropout=0.0):
    m = g["meta"]
    ocfg = O.stage2_cfg(m["type"], m["n_experts"], dropout=dropout)

    def make():
        o = O.get_model(ocfg)
        W.fill_state_dict(o, seed=m["weight_seed"])
        return o
    oracle = cached_oracle((m["type"], m["n_experts"], m["weight_seed"], dropout), make)
    oracle.train(m["train"])
    # (the product module before its first forward is parameter containers only: a pristine instance per configuration, deep-copied)
    model = cached_oracle(("product", m["type"], m["n_experts"], dropout),
                          lambda: get_model(stage2_model_cfg(m["type"], m["n_experts"], dropout=dropout)))
    model.load_state_dict(oracle.state_dict(), strict=True)
    model = model.to("cuda")
    model.compute_dtype = dtype
    model.train(m["train"])
    inp = W.make_inputs(m["batch"], m["size"], m["size"], seed=m["input_seed"])
    return ocfg, oracle, model, inp


def run_parity_case(name, dtype=torch.float32, check_grads=True, verbose=True, fwd_tol_mult=1.0):
    g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
    if dtype == torch.float32 and g["meta"]["train"] and check_grads:
        return run_forced_case(name, verbose)            # flat bounds, every expert, every tensor (tests/forced_masks.py)
    ocfg, oracle, model, inp = build_pair(g, dtype)
    dev = {k: v.to("cuda") for k, v in inp.items()}
    tol = TOL[dtype]
    report = {}
    if g["meta"]["train"]:
        dist, speeds = model(dev["images"], dev["speed"], dev["command"])
        loss = moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs)
        loss.backward()
    else:
        with torch.no_grad():
            dist, speeds = model(dev["images"], dev["speed"], dev["command"])
        loss = None
    probs, mean, std = dist.hip_params
    if dtype == torch.bfloat16 and bf16_bounds(name) is not None:
        # error vs the float64 oracle in units of BF16_SLACK x the bf16-emulating oracle's own error (<= 1 passes)
        rep = bf16_fwd_report(name, dict(probs=probs, mean=mean, std=std, speeds=speeds))
        for k, (err, emul) in rep.items():
            report[k] = err / (BF16_SLACK * emul)
            report[k + "_abs"] = (err, emul)
    else:
        report["probs"] = fwd_err(probs, g["probs"], dtype)
        report["mean"] = fwd_err(mean, g["mean"], dtype)
        report["std"] = fwd_err(std, g["std"], dtype)
        report["speeds"] = fwd_err(speeds, g["speeds"], dtype)
    if loss is not None:
        report["loss"] = abs(loss.item() - g["loss"].item()) / max(1.0, abs(g["loss"].item())) / tol
    for k, v in report.items():
        if not k.endswith("_abs"):
            assert v <= fwd_tol_mult, f"{name} [{dtype}] {k}: {v:.3f} x tolerance ({report.get(k + '_abs', tol)})"
    if loss is not None and check_grads:
        named = dict(model.named_parameters())
        od, os_ = oracle(inp["images"], inp["speed"], inp["command"])
        ol = O.moe_loss(od, os_, inp["control"], inp["target_speed"], ocfg.loss_coefs)
        ol.backward()
        onamed = dict(oracle.named_parameters())
        total_ref = sum(p.grad.norm().item() ** 2 for p in onamed.values()) ** 0.5
        total = sum(p.grad.float().norm().item() ** 2 for p in named.values()) ** 0.5
        errs, cosines = [], []
        for k, p in named.items():
            assert p.grad is not None, f"no gradient for {k}"
            assert torch.isfinite(p.grad).all(), k
            ref = onamed[k].grad
            if ref.norm().item() < 1e-6 * total_ref:      # numerically zero in the reference: absolute check
                assert p.grad.norm().item() < 1e-3 * total_ref, k
                continue
            errs.append((rel_l2(p.grad, ref), k))
            if p.numel() >= 1024:
                cosines.append(torch.nn.functional.cosine_similarity(p.grad.flatten().cpu().float(), ref.flatten(), dim=0).item())
        errs.sort()
        report["grad_median_rel_l2"], report["grad_worst"] = errs[len(errs) // 2][0], errs[-1]
        if dtype == torch.float32:
            raise AssertionError("f32 gradient parity runs through run_forced_case")
        else:
            cosines.sort()
            report["grad_median_cos"] = cosines[len(cosines) // 2]
            assert cosines[len(cosines) // 2] >= 0.85, cosines[len(cosines) // 2]
            assert abs(total - total_ref) <= 0.2 * total_ref, (total, total_ref)
        # BN running statistics after one training step (checkpoint parity)
        sd = model.state_dict()
        for k, v in g["bn_after_1"].items():
            if v.dtype == torch.long:
                assert int(sd[k].item()) == int(v.item()), k
            else:
                # bf16: same 3e-2 as the forward tolerance (the deepest running means average 32 bf16 activations here)
                assert rel_err(sd[k], v) <= (1e-4 if dtype == torch.float32 else 3e-2), k
    if verbose:
        print(name, dtype, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in report.items()})
    return report


# ---- gradient parity with the discrete decisions forced (tests/forced_masks.py): flat bounds, every expert, every tensor
# rel-L2 of a gradient tensor against the float64 oracle run on the HIP path's ReLU / max-pool decisions.  Measured on the nine
# golden cases (profiles/r03_parity_report.log): worst tensor of >= 64 elements 5.4e-5 (a 3-tap ECA filter: heavy
# cancellation), median 4e-6 .. 2e-5; tensors of a handful of elements (a [1] bias) up to 9e-4 -> 10 x the bound for those.
FORCED_GRAD_TOL = 2e-4
FLIP_ZONE = 1e-4              # a disagreement between the float64 oracle's own decision and the HIP one must sit this close to the tie


def run_forced_case(name, verbose=True, tol=FORCED_GRAD_TOL):
    """f32 HIP path vs (a) the reference's golden forward vectors at north_star's 1e-4 and (b) the float64 oracle
    differentiating the same piecewise-linear function (forced_masks.py): EVERY parameter gradient of EVERY expert within
    ``tol``, every decision disagreement a near-tie, and -- for the experts without any disagreement, where the forced
    oracle IS the reference network -- the reference's own golden gradient slices and norms."""
    import re
    from tests import forced_masks as FM
    g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
    m = g["meta"]
    ocfg, oracle, model, inp = build_pair(g, torch.float32)
    eng = model._engine()
    eng.debug_acts = {}
    dev = {k: v.to("cuda") for k, v in inp.items()}
    dist, speeds = model(dev["images"], dev["speed"], dev["command"])
    loss = moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs)
    loss.backward()
    probs, mean, std = dist.hip_params
    report = {k: fwd_err(v, g[k], torch.float32) for k, v in (("probs", probs), ("mean", mean), ("std", std), ("speeds", speeds))}
    report["loss"] = abs(loss.item() - g["loss"].item()) / max(1.0, abs(g["loss"].item())) / TOL[torch.float32]
    for k, v in report.items():
        assert v <= 1.0, f"{name} forward {k}: {v:.3f} x 1e-4"

    def run(m64, cast):
        d, s = m64(cast(inp["images"]), cast(inp["speed"]), cast(inp["command"]))
        O.moe_loss(d, s, cast(inp["control"]), cast(inp["target_speed"]).clone(), ocfg.loss_coefs).backward()
        return dict(mean=d.component_distribution.base_dist.loc.detach(), speeds=s.detach())
    out64, g64, log = FM.forced_float64(oracle, eng, inp, m["batch"], run, alt=m["type"] == "moe_alt",
                                        shared=m["type"] == "moe_shared")
    eng.debug_acts = None
    # the forced function coincides with the network at the HIP path's operating point: its float64 outputs are the HIP outputs
    assert rel_err(mean, out64["mean"]) <= 1e-4 and rel_err(speeds, out64["speeds"]) <= 1e-4
    report["flips"] = [(e, nm, n, f"{z:.1e}") for e, nm, n, z, _ in log]
    for e, nm, n, z, numel in log:
        assert z <= FLIP_ZONE, f"{name}: expert {e} {nm}: {n} decisions differ from the float64 oracle's, |pre-activation| up to {z:.2e}"
        assert n <= max(4, numel // 100000), f"{name}: expert {e} {nm}: {n} of {numel} decisions differ"
    named = dict(model.named_parameters())
    total = sum(v.norm().item() ** 2 for v in g64.values()) ** 0.5
    errs = []
    for k, p in named.items():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        ref = g64[k].float()
        if ref.norm().item() < 1e-6 * total:
            assert p.grad.norm().item() < 1e-4 * total, k
            continue
        errs.append((rel_l2(p.grad, ref), k))
    errs.sort()
    report["grad_median_rel_l2"], report["grad_worst"] = errs[len(errs) // 2][0], errs[-1]
    bad = [(f"{e:.2e}", k) for e, k in errs if e > (tol if named[k].numel() >= 64 else 10 * tol)]
    assert not bad, f"{name}: {len(bad)} gradient tensors beyond {tol:g} of the float64 oracle on the same decisions: {bad[-6:]}"
    # golden slices / norms of the REFERENCE: valid yardstick for the experts whose decisions all agree
    flipped = {e for e, *_ in log}

    def expert_of(k):
        mt = re.match(r"moe\.(\d+)\.", k)
        return int(mt.group(1)) if mt else 0
    checked = 0
    for k, sl in g["grad_slices"].items():
        if expert_of(k) in flipped:
            continue
        scale = max(sl.abs().max().item(), g["grad_norms"][k] / max(1, named[k].numel()) ** 0.5)
        e = (named[k].grad.flatten()[:64].cpu() - sl).abs().max().item() / (scale + 1e-20)
        assert e <= 2e-2, f"{name} reference gradient slice {k}: {e:.3e}"      # the f32 reference's own noise (flips on ITS side)
        checked += 1
    report["golden_slices_checked"] = checked
    sd = model.state_dict()
    for k, v in g["bn_after_1"].items():
        if v.dtype == torch.long:
            assert int(sd[k].item()) == int(v.item()), k
        else:
            assert rel_err(sd[k], v) <= 1e-4, k
    if verbose:
        print(name, "forced", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in report.items()})
    return report
