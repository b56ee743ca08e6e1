import sys, torch
sys.path.insert(0, '.')
from tests.parity_util import *
name = sys.argv[1] if len(sys.argv) > 1 else "g4_moealt_e4_b2_64"
g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
runs = []
for r in range(4):
    ocfg, oracle, model, inp = build_pair(g, torch.float32)
    dev = {k: v.cuda() for k, v in inp.items()}
    dist, speeds = model(dev["images"], dev["speed"], dev["command"])
    loss = moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs)
    loss.backward()
    runs.append((loss.item(), [t.clone() for t in dist.hip_params], {k: p.grad.clone() for k, p in model.named_parameters()}))
for r in range(1, 4):
    d = sorted(((rel_l2(runs[r][2][k], runs[0][2][k]), k) for k in runs[0][2]), reverse=True)
    print("run", r, "loss", runs[r][0], runs[0][0], "fwd diff", [rel_err(a, b) for a, b in zip(runs[r][1], runs[0][1])])
    for e, k in d[:5]: print("   %.3e %s" % (e, k))
