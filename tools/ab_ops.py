#!/usr/bin/env python3
"""Per-launch times of the headline step's ops under two values of an environment switch, in ONE process (switches are read per
launch): which launches a kernel change moves, including its neighbours'.  python tools/ab_ops.py VAR a b [op substring ...]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import bench  # noqa: E402
from pmoe_amd import hip, ops  # noqa: E402
from pmoe_amd.loss import moe_loss  # noqa: E402
from pmoe_amd.model.moe import get_model  # noqa: E402
from pmoe_amd.utils import stage2_model_cfg  # noqa: E402

var, va, vb = sys.argv[1:4]
pick = sys.argv[4:]
hip.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = get_model(stage2_model_cfg("moe", 4, dropout=0.3)).to(dev)
model.compute_dtype = torch.bfloat16
model.train()
images, speed, command, control, target = bench.make_batch(64, 256, 1234, dev)


def step():
    model.zero_grad(set_to_none=True)
    d, s = model(images, speed, command)
    moe_loss(d, s, control, target, [0.7, 0.3]).backward()


def profile(val, reps=4):
    os.environ[var] = val
    for _ in range(2):
        step()
    acc = None
    for _ in range(reps):
        ops.profile_begin()
        step()
        rows = ops.profile_end()
        if acc is None:
            acc = [[n, m, t] for n, m, t in rows]
        else:
            for r, (_, _, t) in zip(acc, rows):
                r[2] += t
    return [(n, m, t / reps) for n, m, t in acc]


ra, rb = profile(va), profile(vb)
ra2, rb2 = profile(va), profile(vb)


def table(rows):
    d = {}
    for n, m, t in rows:
        key = (n, str(m.get("name", "")), str(m.get("kernel", "")))
        e = d.setdefault(key, [0.0, 0])
        e[0] += t
        e[1] += 1
    return d


ta, tb = table(ra + ra2), table(rb + rb2)
print(f"launches per step: {len(ra)} vs {len(rb)}")
groups = {}
for key in sorted(set(ta) | set(tb)):
    a_, b_ = ta.get(key, [0.0, 0]), tb.get(key, [0.0, 0])
    x, y = a_[0] / 2, b_[0] / 2
    g = groups.setdefault(key[0], [0.0, 0.0])
    g[0] += x
    g[1] += y
    if (not pick and abs(x - y) > 0.012) or any(p in key[0] for p in pick):
        print(f"{key[0]:18s} {key[1]:30s} plan {key[2]:>6s}  x{a_[1] // 2}/{b_[1] // 2}  {var}={va}: {x:.3f} ms   {var}={vb}: {y:.3f} ms   {y - x:+.3f}")
print("-- by op")
for n, (x, y) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:22s} {x:7.3f} {y:7.3f}  {y - x:+.3f}")
print(f"sum of all launches: {sum(g[0] for g in groups.values()):.2f} vs {sum(g[1] for g in groups.values()):.2f} ms")
