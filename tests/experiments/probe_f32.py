import sys, copy, torch
sys.path.insert(0, '.')
from tests.parity_util import *
from oracle import pmoe_oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "g4_moealt_e4_b2_64"
g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
for cfg in [dict(), dict(fold_stem_input=False), dict(fuse_stem_tail=False), dict(fuse_conv_stats=False), dict(fold_stem_input=False, fuse_stem_tail=False, fuse_conv_stats=False)]:
    ocfg, oracle, model, inp = build_pair(g, torch.float32)
    for k, v in cfg.items(): setattr(model._engine(), k, v)
    dev = {k: v.cuda() for k, v in inp.items()}
    dist, speeds = model(dev["images"], dev["speed"], dev["command"])
    moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs).backward()
    o64 = copy.deepcopy(oracle).double()
    d64, s64 = o64(inp["images"].double(), inp["speed"].double(), inp["command"].double())
    O.moe_loss(d64, s64, inp["control"].double(), inp["target_speed"].double(), ocfg.loss_coefs).backward()
    od, os_ = oracle(inp["images"], inp["speed"], inp["command"])
    O.moe_loss(od, os_, inp["control"], inp["target_speed"], ocfg.loss_coefs).backward()
    g64 = {k: p.grad.float() for k, p in o64.named_parameters()}
    on = dict(oracle.named_parameters())
    rows = []
    for k, p in model.named_parameters():
        if g64[k].norm() < 1e-9: continue
        rows.append((rel_l2(p.grad, g64[k]), rel_l2(on[k].grad, g64[k]), k))
    rows.sort(reverse=True)
    print("== cfg", cfg)
    for r in rows[:8]: print("   hip %.2e  oracle32 %.2e  %s" % r)
