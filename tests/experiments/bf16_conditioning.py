"""How much of the bf16 gradient mismatch is intrinsic?  CPU experiment: the oracle with every layer
output (and its incoming gradient) rounded to bf16 versus the plain f32 oracle, and f32 vs f64."""
import sys, statistics
sys.path.insert(0, '.')
import torch, torch.nn as nn
from oracle import pmoe_oracle as O, weights as W

class RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x): return x.to(torch.bfloat16).to(x.dtype)
    @staticmethod
    def backward(ctx, g): return g.to(torch.bfloat16).to(g.dtype)

def run(name, mode, batch=None, size=None):
    g = torch.load(f"tests/golden/{name}.pt", weights_only=False)
    m = g["meta"]
    cfg = O.stage2_cfg(m["type"], m["n_experts"], dropout=0.0)
    model = O.get_model(cfg); W.fill_state_dict(model, seed=0); model.train()
    B_ = batch or m["batch"]; S = size or m["size"]
    inp = W.make_inputs(B_, S, S, seed=1234)
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
    d, s = model(inp["images"], inp["speed"], inp["command"])
    loss = O.moe_loss(d, s, inp["control"], inp["target_speed"], cfg.loss_coefs); loss.backward()
    return {k: p.grad.double() for k, p in model.named_parameters()}, d.mixture_distribution.probs.double(), s.double()

for name, b, sz in [("g4_moealt_e4_b2_64", None, None), ("g1_moe_e4_b2_128", None, None), ("g1_moe_e4_b2_128", 16, 128)]:
    ref, p64, s64 = run(name, "f64", b, sz)
    for mode in ("f32", "bf16"):
        got, p, s = run(name, mode, b, sz)
        errs = sorted(((got[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)).item() for k in ref)
        print(name, b, sz, mode, "probs err %.2e speeds err %.2e" % ((p - p64).abs().max().item(), ((s - s64).abs().max() / s64.abs().max()).item()),
              "grad rel-L2: median %.2e p90 %.2e max %.2e" % (statistics.median(errs), errs[int(0.9 * len(errs))], errs[-1]))
