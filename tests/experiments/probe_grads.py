import sys, torch
sys.path.insert(0, '.')
from tests.parity_util import *
from oracle import pmoe_oracle as O
for name, dtype in [("g4_moealt_e4_b2_64", torch.float32), ("g1_moe_e4_b2_128", torch.float32), ("g1_moe_e4_b2_128", torch.bfloat16), ("g4_moealt_e4_b2_64", torch.bfloat16)]:
    g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
    ocfg, oracle, model, inp = build_pair(g, dtype)
    dev = {k: v.cuda() for k, v in inp.items()}
    dist, speeds = model(dev["images"], dev["speed"], dev["command"])
    loss = moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs)
    loss.backward()
    od, os_ = oracle(inp["images"], inp["speed"], inp["command"])
    ol = O.moe_loss(od, os_, inp["control"], inp["target_speed"], ocfg.loss_coefs)
    ol.backward()
    on = dict(oracle.named_parameters())
    rows = []
    for k, p in model.named_parameters():
        ref = on[k].grad
        rows.append((rel_l2(p.grad, ref), k, ref.norm().item(), p.numel()))
    rows.sort(reverse=True)
    tot = sum(r[2]**2 for r in rows) ** 0.5
    print("==", name, dtype, "loss", loss.item(), ol.item(), "total grad norm", tot)
    print("outputs:", rel_err(dist.hip_params[0], od.mixture_distribution.probs), rel_err(dist.hip_params[1], od.component_distribution.base_dist.loc), rel_err(speeds, os_))
    for r in rows[:14]:
        print("  %.3e  %-55s norm=%.3e n=%d" % r)
    import statistics
    print("  median rel l2: %.3e" % statistics.median(r[0] for r in rows))
