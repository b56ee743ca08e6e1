import sys, copy, torch
sys.path.insert(0, '.')
from tests.parity_util import *
from oracle import pmoe_oracle as O
import torch.distributions as D
name = sys.argv[1] if len(sys.argv) > 1 else "g4_moealt_e4_b2_64"
g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
ocfg, oracle, model, inp = build_pair(g, torch.float32)
dev = {k: v.cuda() for k, v in inp.items()}
probs, mean, std, speeds = model.mixture_params(dev["images"], dev["speed"], dev["command"])
for t in (probs, mean, std, speeds): t.retain_grad()
from pmoe_amd.model.moe import MixtureDistribution
dist = MixtureDistribution(probs, mean, std)
loss = moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs); loss.backward()
o64 = copy.deepcopy(oracle).double()
i64 = {k: v.double() for k, v in inp.items()}
p64, m64, s64, sp64 = o64.mixture_params(i64["images"], i64["speed"], i64["command"])
d64 = D.MixtureSameFamily(D.Categorical(p64), D.Independent(D.Normal(m64, s64), 1))
l64 = O.moe_loss(d64, sp64, i64["control"], i64["target_speed"], ocfg.loss_coefs)
gp, gm, gs, gsp = torch.autograd.grad(l64, [p64, m64, s64, sp64], retain_graph=True)
torch.set_printoptions(precision=4, sci_mode=True, linewidth=200)
print("probs64", p64); print("probs hip err", (probs.detach().cpu().double() - p64).abs())
for nm, a, b in (("dprobs", probs.grad, gp), ("dmean", mean.grad, gm), ("dstd", std.grad, gs), ("dspeeds", speeds.grad, gsp)):
    a = a.cpu().double()
    print(nm, "ref", b.flatten()[:16]); print(nm, "relerr", ((a - b).abs() / (b.abs() + 1e-30)).flatten()[:16])
