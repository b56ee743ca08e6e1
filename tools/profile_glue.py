#!/usr/bin/env python3
"""Which torch-native kernels (fills, copies, ...) a headline step launches beside the library's own, with the Python lines that
cause them.  python tools/profile_glue.py [--batch 64]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from pmoe_amd import hip  # noqa: E402
from pmoe_amd.loss import moe_loss  # noqa: E402
from pmoe_amd.model.moe import get_model  # noqa: E402
from pmoe_amd.utils import stage2_model_cfg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
hip.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = get_model(stage2_model_cfg("moe", 4, dropout=0.3)).to(dev)
model.compute_dtype = torch.bfloat16
model.train()
images, speed, command, control, target = bench.make_batch(a.batch, 256, 1234, dev)


def step():
    model.zero_grad(set_to_none=True)
    d, s = model(images, speed, command)
    loss = moe_loss(d, s, control, target, [0.7, 0.3])
    loss.backward()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cpu_parent is None or (
            ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and not ev.cpu_parent.name.startswith("aten::")):
        st = [f for f in (ev.stack or []) if "/root/repo" in f or "pmoe_amd" in f or "bench" in f]
        key = (ev.name, st[0] if st else "?")
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += ev.device_time_total if hasattr(ev, "device_time_total") else 0.0
for (name, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{n:4d} x {name:28s} {us:8.1f} us  {where}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
