"""Scratch diagnosis of PMOE_RES_DBN mismatches on conv3x3_respipe_kernel (not collected by pytest)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from tests import test_ops_gpu as T
from pmoe_amd import hip, ops

case = (2, 2, 64, 128, 128, False, False)
Ebn, ipb, C, H, W, per_image, with_bias = case
g = torch.Generator().manual_seed(sum(case[:5]))
N = Ebn * ipb
E, ipe = Ebn, ipb
BF = torch.bfloat16
dy = T.rnd((N, C, H, W), g, BF)
z = T.rnd((N, C, H, W), g, BF, 2.0) + 0.3
ws = [T.rnd((C, C, 3, 3), g, torch.float32, (2.0 / (C * 9)) ** 0.5) for _ in range(E)]
_, wd, _keep = T.pack(ws, 3, BF, want_dgrad=True)
coef = torch.empty(4, Ebn, C)
coef[0] = torch.randn(Ebn, C, generator=g) * 0.5 + 0.3
coef[1] = torch.rand(Ebn, C, generator=g) + 0.5
coef[2] = coef[1] * (torch.randn(Ebn, C, generator=g) * 0.5 + 1.0)
coef[3] = torch.randn(Ebn, C, generator=g) * 0.3
coefd = coef.to("cuda")
dyd, zd = T.nhwc(dy, C, BF), T.nhwc(z, C, BF)
kw = dict(cin=C, cout=C, coutp=64, ipe=ipe, ks=3, stride=1, pad=1, bias=None)
plain = torch.empty(N, H, W, C, dtype=BF, device="cuda")
ops.conv2d(dyd, wd, plain, **kw)
got = torch.full((N, H, W, C), 7.0, dtype=BF, device="cuda")
rows = ops.conv2d_stat_rows(N, H, W, H, W, C, C, 64, ipe, 3, 1, 1, BF)
stats = torch.full((rows, 2, 64), 5.0, device="cuda")
print("plan", ops.conv2d(dyd, wd, got, plan_only=True, res=zd, res_mode=hip.RES_DBN, bn_coef=coefd, bn_ipe=ipb, **kw))
ops.conv2d(dyd, wd, got, stats=stats, res=zd, res_mode=hip.RES_DBN, bn_coef=coefd, bn_ipe=ipb, **kw)
first = got.clone()
nbad = 0
for it in range(200):
    got.fill_(7.0)
    ops.conv2d(dyd, wd, got, stats=stats, res=zd, res_mode=hip.RES_DBN, bn_coef=coefd, bn_ipe=ipb, **kw)
    nbad += int((got != first).sum())
print("elements differing from the first launch over 200 repeats:", nbad)
pl2 = torch.empty_like(plain)
nb0 = 0
for it in range(200):
    ops.conv2d(dyd, wd, pl2, **kw)
    nb0 += int((pl2 != plain).sum())
print("plain launch: elements differing over 200 repeats:", nb0)
cb = coef.repeat_interleave(ipb, dim=1).view(4, N, 1, 1, C)
zf = zd.float().cpu()
d = zf - cb[0]
y = d * cb[2] + cb[3]
mask = y > 0
want = torch.where(mask, plain.float().cpu(), torch.zeros(()))
gotc = got.float().cpu()
bad = gotc != want
print("mismatches", int(bad.sum()), "channels", bad.sum((0, 1, 2)).nonzero().flatten().tolist())
idx = bad.nonzero()[:12]
for n, yy, xx, c in idx.tolist():
    print((n, yy, xx, c), "got", gotc[n, yy, xx, c].item(), "want", want[n, yy, xx, c].item(), "plain", plain[n, yy, xx, c].item(),
          "z", zf[n, yy, xx, c].item(), "y", y[n, yy, xx, c].item())
# which alternative z explains the kernel's decision?  (neighbouring pixel / channel)
kmask = gotc != 0
for name, alt in (("z shifted +1 px", torch.roll(zf, -1, 2)), ("z shifted -1 px", torch.roll(zf, 1, 2)), ("z ch+2", torch.roll(zf, -2, 3)),
                  ("z ch+4", torch.roll(zf, -4, 3)), ("z ch+8", torch.roll(zf, -8, 3)), ("z ch+16", torch.roll(zf, -16, 3))):
    ya = (alt - cb[0]) * cb[2] + cb[3]
    sel = bad
    agree = ((ya > 0) == kmask)[sel].float().mean().item()
    print(name, "explains", agree)
print("|y| at mismatches: median", y[bad].abs().median().item(), "max", y[bad].abs().max().item())
b = bad.view(N, H // 16, 16, W // 16, 16, C)
print("by row in tile:", b.sum((0, 1, 3, 4, 5)).tolist())
print("by column in tile:", b.sum((0, 1, 2, 3, 5)).tolist())
print("by image:", b.sum((1, 2, 3, 4, 5)).tolist())
print("by tile row:", b.sum((0, 2, 3, 4, 5)).tolist())
print("by tile col:", b.sum((0, 1, 2, 4, 5)).tolist())
# is the wrong value some other element of the plain result?
pf = plain.float().cpu()
n, yy, xx, c = bad.nonzero()[0].tolist()
val = gotc[n, yy, xx, c].item()
cand = (pf[n] == val).nonzero()[:10].tolist()
print("first bad", (n, yy, xx, c), "value", val, "appears in plain at", cand)
