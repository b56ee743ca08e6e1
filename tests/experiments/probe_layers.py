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
bnin = {}
for e, ex in enumerate(o64.moe):
    bb = ex.backbone
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(bb, f"layer{li}")):
            for nm, mod in (("bn1", blk.bn1), ("bn2", blk.bn2)) + ((("downbn", blk.downsample[1]),) if blk.downsample is not None else ()):
                def fh(m, args, key=(e, f"layer{li}.{bi}.{nm}")):
                    x = args[0]; x.retain_grad(); bnin[key] = x
                mod.register_forward_pre_hook(fh)
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

# ---- detail for the first (deepest) block whose gradient is off
bad = None
for key in sorted(eng.debug_grads, reverse=True):
    if not key.endswith("bn2"): continue
    t = eng.debug_grads[key].float().cpu()
    for e in range(E):
        ref = store[(e, key)].permute(0, 2, 3, 1).float(); got = t[e * B:(e + 1) * B]
        if ((got - ref).norm() / ref.norm()).item() > 1e-4 and bad is None: bad = (key, e, got, ref)
if bad:
    key, e, got, ref = bad
    print("first bad:", key, "expert", e, "shape", tuple(got.shape))
    d = (got - ref)
    perc = d.pow(2).sum(dim=(0, 1, 2)).sqrt() / (ref.pow(2).sum(dim=(0, 1, 2)).sqrt() + 1e-30)
    top = perc.topk(6)
    print("per-channel rel err top:", [(int(i), float(v)) for v, i in zip(top.values, top.indices)])
    print("channels with err>1e-4:", int((perc > 1e-4).sum()), "of", perc.numel())
    pp = d.pow(2).sum(dim=3).sqrt() / (ref.pow(2).sum(dim=3).sqrt() + 1e-30)
    print("per-pixel rel err (img0):"); print(pp[0])
    ratio = (got / (ref + 1e-30))
    c = int(top.indices[0]); print("ratio got/ref channel", c, ratio[0, :, :, c])

if bad:
    key, e, _, _ = bad
    li, bi = int(key[5]), int(key[7])
    # the block AFTER `key` (deeper) is the one whose backward produced this gradient
    nxt = f"layer{li}.{bi+1}" if (f"layer{li}.{bi+1}.bn2" in eng.debug_grads) else f"layer{li+1}.0"
    for nm in ("bn2", "bn1", "downbn"):
        k2 = f"{nxt}.{nm}"
        if (e, k2) not in bnin: continue
        x = bnin[(e, k2)]; gx = x.grad.permute(0, 2, 3, 1).float()
        dz = eng.debug_grads[k2 + ":dz"].float().cpu()[e * B:(e + 1) * B]
        mean, invstd, c1, c2 = (t[e].cpu() for t in eng.debug_grads[k2 + ":stats"])
        xe = x.detach().permute(1, 0, 2, 3).reshape(x.shape[1], -1)
        rmean, ris = xe.mean(1), 1 / torch.sqrt(xe.var(1, unbiased=False) + 1e-5)
        perc = (dz - gx).pow(2).sum(dim=(0, 1, 2)).sqrt() / (gx.pow(2).sum(dim=(0, 1, 2)).sqrt() + 1e-30)
        top = perc.topk(4)
        print(k2, "dz rel err total %.2e" % ((dz - gx).norm() / gx.norm()).item(), "worst channels", [(int(i), "%.1e" % float(v)) for v, i in zip(top.values, top.indices)])
        for i in top.indices[:3]:
            i = int(i)
            print("    ch %d: mean hip %.6e ref %.6e | invstd hip %.6e ref %.6e | |dz| ch norm %.3e of total %.3e" % (i, mean[i], rmean[i], invstd[i], ris[i], gx[..., i].norm(), gx.norm()))
