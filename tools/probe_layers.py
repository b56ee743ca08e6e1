"""Layer-by-layer backward check: gradient entering every BasicBlock output (HIP f32) vs the float64 oracle."""
import sys, copy, torch
sys.path.insert(0, '.')
from tests.parity_util import *
from oracle import pmoe_oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "g4_moealt_e4_b2_64"
g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
ocfg, oracle, model, inp = build_pair(g, torch.float32)
eng = model._engine(); eng.debug_grads = {}
for kv in sys.argv[2:]:
    k, v = kv.split("="); setattr(eng, k, v == "1")
dev = {k: v.cuda() for k, v in inp.items()}
dist, speeds = model(dev["images"], dev["speed"], dev["command"])
moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs).backward()
o64 = copy.deepcopy(oracle).double()
store = {}
for mod in o64.modules():
    if isinstance(mod, torch.nn.ReLU): mod.inplace = False
E = len(o64.moe)
for e, ex in enumerate(o64.moe):
    bb = ex.backbone
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(bb, f"layer{li}")):
            blk.register_full_backward_hook(lambda m, gi, go, key=(e, f"layer{li}.{bi}.bn2"): store.__setitem__(key, go[0]))
i64 = {k: v.double() for k, v in inp.items()}
d64, s64 = o64(i64["images"], i64["speed"], i64["command"])
O.moe_loss(d64, s64, i64["control"], i64["target_speed"], ocfg.loss_coefs).backward()
B = inp["images"].shape[0]
print("note: block-level hook = grad wrt block OUTPUT (after final relu); bn1 hook = grad wrt bn1 output (before relu)")
for key in sorted(eng.debug_grads, reverse=True):
    if not key.endswith("bn2"): continue
    t = eng.debug_grads[key].float().cpu()            # [E*B,H,W,C]
    row = []
    for e in range(E):
        ref = store[(e, key)].permute(0, 2, 3, 1).float()
        got = t[e * B:(e + 1) * B]
        row.append(((got - ref).norm() / (ref.norm() + 1e-30)).item())
    print("%-18s " % key + " ".join("%.2e" % r for r in row))
