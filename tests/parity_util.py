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

# north_star tolerances: 1e-4 fp32, 1e-2 bf16 (relative to the tensor's scale)
TOL = {torch.float32: 1e-4, torch.bfloat16: 1e-2}
# gradients accumulate rounding through 20 conv+BN layers in both directions; bf16 storage of every
# activation and activation-gradient gives a few percent on individual weight-gradient tensors
GRAD_TOL = {torch.float32: 1e-3, torch.bfloat16: 6e-2}


def rel_err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def rel_l2(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).norm() / (ref.norm() + 1e-20)).item()


def build_pair(g, dtype, dropout=0.0):
    m = g["meta"]
    ocfg = O.stage2_cfg(m["type"], m["n_experts"], dropout=dropout)
    oracle = O.get_model(ocfg)
    W.fill_state_dict(oracle, seed=m["weight_seed"])
    oracle.train(m["train"])
    model = get_model(stage2_model_cfg(m["type"], m["n_experts"], dropout=dropout))
    model.load_state_dict(oracle.state_dict(), strict=True)
    model = model.to("cuda")
    model.compute_dtype = dtype
    model.train(m["train"])
    inp = W.make_inputs(m["batch"], m["size"], m["size"], seed=m["input_seed"])
    return ocfg, oracle, model, inp


def run_parity_case(name, dtype=torch.float32, check_grads=True, verbose=True):
    g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
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
    report["probs"] = rel_err(probs, g["probs"])
    report["mean"] = rel_err(mean, g["mean"])
    report["std"] = rel_err(std, g["std"])
    report["speeds"] = rel_err(speeds, g["speeds"])
    report["log_prob"] = rel_err(dist.log_prob(dev["control"]), g["log_prob"])
    if loss is not None:
        report["loss"] = abs(loss.item() - g["loss"].item()) / max(1.0, abs(g["loss"].item()))
    for k, v in report.items():
        assert v <= tol, f"{name} [{dtype}] {k}: {v:.3e} > {tol}"
    if loss is not None and check_grads:
        # (a) reference goldens: per-parameter gradient norms and 64-element slices
        named = dict(model.named_parameters())
        gt = GRAD_TOL[dtype]
        worst = ("", 0.0)
        for k, sl in g["grad_slices"].items():
            e = rel_err(named[k].grad.flatten()[:64], sl) if sl.abs().max() > 0 else named[k].grad.flatten()[:64].abs().max().item()
            # slices are compared relative to the whole tensor's scale
            scale = max(sl.abs().max().item(), g["grad_norms"][k] / max(1, named[k].numel()) ** 0.5)
            e = (named[k].grad.flatten()[:64].cpu() - sl).abs().max().item() / (scale + 1e-20)
            if e > worst[1]:
                worst = (k, e)
            assert e <= gt * 4, f"{name} [{dtype}] grad slice {k}: {e:.3e}"
        # (b) live oracle: full tensors, relative L2 per parameter
        od, os_ = oracle(inp["images"], inp["speed"], inp["command"])
        ol = O.moe_loss(od, os_, inp["control"], inp["target_speed"], ocfg.loss_coefs)
        ol.backward()
        onamed = dict(oracle.named_parameters())
        total_ref = sum(p.grad.norm().item() ** 2 for p in onamed.values()) ** 0.5
        worst_l2 = ("", 0.0)
        for k, p in named.items():
            assert p.grad is not None, f"no gradient for {k}"
            ref = onamed[k].grad
            # tensors whose gradient is numerically ~0 relative to the whole model are judged on absolute size
            if ref.norm().item() < 1e-6 * total_ref:
                assert p.grad.norm().item() < 1e-4 * total_ref, k
                continue
            e = rel_l2(p.grad, ref)
            if e > worst_l2[1]:
                worst_l2 = (k, e)
            assert e <= gt, f"{name} [{dtype}] grad {k}: rel L2 {e:.3e} > {gt}"
        total = sum(p.grad.float().norm().item() ** 2 for p in named.values()) ** 0.5
        assert abs(total - total_ref) <= gt * total_ref
        report["worst_grad_slice"], report["worst_grad_l2"] = worst, worst_l2
        # BN running statistics after one training step (checkpoint parity)
        sd = model.state_dict()
        for k, v in g["bn_after_1"].items():
            if v.dtype == torch.long:
                assert int(sd[k].item()) == int(v.item()), k
            else:
                assert rel_err(sd[k], v) <= max(tol, 1e-3 if dtype == torch.bfloat16 else tol), k
    if verbose:
        print(name, dtype, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in report.items()})
    return report
