"""CPU experiment: does var = E[x^2] - mean^2 in f32 (one-pass statistics) explain the f32 gradient drift?"""
import sys, copy, torch, torch.nn as nn
sys.path.insert(0, '.')
from oracle import pmoe_oracle as O, weights as W
name = sys.argv[1] if len(sys.argv) > 1 else "g4_moealt_e4_b2_64"
g = torch.load(f"tests/golden/{name}.pt", weights_only=False); m = g["meta"]
cfg = O.stage2_cfg(m["type"], m["n_experts"], dropout=0.0)
def make():
    model = O.get_model(cfg); W.fill_state_dict(model, seed=0); model.train(); return model
inp = W.make_inputs(m["batch"], m["size"], m["size"], seed=1234)
def grads(model, dt):
    i = {k: v.to(dt) for k, v in inp.items()}
    d, s = model(i["images"], i["speed"], i["command"])
    O.moe_loss(d, s, i["control"], i["target_speed"], cfg.loss_coefs).backward()
    return {k: p.grad.double() for k, p in model.named_parameters()}
ref = grads(make().double(), torch.float64)
class OnePassBN(nn.Module):
    def __init__(s, bn): super().__init__(); s.bn = bn
    def forward(s, x):
        mean = x.mean(dim=(0, 2, 3)); ex2 = (x * x).mean(dim=(0, 2, 3)); var = (ex2 - mean * mean).clamp_min(0)
        xh = (x - mean[None, :, None, None]) * torch.rsqrt(var + s.bn.eps)[None, :, None, None]
        return xh * s.bn.weight[None, :, None, None] + s.bn.bias[None, :, None, None]
def patch(mod):
    for n, c in list(mod.named_children()):
        if isinstance(c, nn.BatchNorm2d): setattr(mod, n, OnePassBN(c))
        else: patch(c)
m32 = make(); g32 = grads(m32, torch.float32)
m1p = make(); patch(m1p); g1p = {k.replace(".bn.", "."): v for k, v in grads(m1p, torch.float32).items()}
def worst(gx):
    rows = sorted((((gx[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)).item(), k) for k in ref if ref[k].norm() > 1e-9)
    return rows[-5:], rows[len(rows)//2]
print("two-pass f32 :", worst(g32)); print("one-pass f32 :", worst(g1p))
