"""repeat the fused (BatchNorm backward on load) and the unfused per-image filter gradient on identical inputs: which of the two is
not bit-reproducible?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from pmoe_amd import ops

DEV = "cuda"
BF = torch.bfloat16
for (E, ipb, H, W) in [(2, 2, 64, 64), (1, 3, 40, 72), (2, 1, 256, 256), (4, 8, 128, 128)]:
    g = torch.Generator().manual_seed(1)
    N, C = E * ipb, 64
    xd = torch.rand((ipb, H, W, 16), generator=g).to(BF).to(DEV)
    xd[..., 12:] = 0
    gd = torch.randn((N, H, W, C), generator=g).to(BF).to(DEV)
    zd = (torch.randn((N, H, W, C), generator=g) * 2 + 0.3).to(BF).to(DEV)
    coef = torch.rand(4, E, C, generator=g).to(DEV) + 0.5
    c1, c2 = (torch.randn(E, C, generator=g) * 0.1).to(DEV), (torch.randn(E, C, generator=g) * 0.1).to(DEV)
    kw = dict(cin=16, cout=C, cinp=64, coutp=64, ipe=ipb, ks=3, stride=1, pad=1, x_shared=True, per_image=True)
    outs_f, outs_u = [], []
    for it in range(12):
        G = torch.full((N, 9, 64, 64), float(it), device=DEV)
        ops.conv2d_wgrad(xd, gd, G, bn_fuse=(zd, coef, c1, c2), **kw)
        outs_f.append(G)
        dz = torch.empty_like(gd)
        ops.bn_bwd_apply(gd, None, zd, coef[0], coef[1], coef[2], coef[3], c1, c2, dz, None, ipb * H * W, E, C, False)
        G2 = torch.full((N, 9, 64, 64), float(-it), device=DEV)
        ops.conv2d_wgrad(xd, dz, G2, **kw)
        outs_u.append(G2)
    torch.cuda.synchronize()
    sc = outs_u[0].abs().max().item()
    print((E, ipb, H, W), "fused repeat max diff", max((o - outs_f[0]).abs().max().item() for o in outs_f) / sc,
          "unfused repeat max diff", max((o - outs_u[0]).abs().max().item() for o in outs_u) / sc,
          "fused vs unfused", [round((a - b).abs().max().item() / sc, 6) for a, b in zip(outs_f, outs_u)][:6], flush=True)

# which one is right on the ragged shape?  CPU f32 statement of dz and the per-image filter gradient
import torch.nn.functional as F
E, ipb, H, W = 1, 3, 40, 72
g = torch.Generator().manual_seed(1)
N, C = E * ipb, 64
x = torch.rand((ipb, H, W, 16), generator=g).to(BF)
x[..., 12:] = 0
gy = torch.randn((N, H, W, C), generator=g).to(BF)
z = (torch.randn((N, H, W, C), generator=g) * 2 + 0.3).to(BF)
coef = torch.rand(4, E, C, generator=g) + 0.5
c1, c2 = torch.randn(E, C, generator=g) * 0.1, torch.randn(E, C, generator=g) * 0.1
kw = dict(cin=16, cout=C, cinp=64, coutp=64, ipe=ipb, ks=3, stride=1, pad=1, x_shared=True, per_image=True)
G = torch.zeros((N, 9, 64, 64), device=DEV)
ops.conv2d_wgrad(x.to(DEV), gy.to(DEV), G, bn_fuse=(z.to(DEV), coef.to(DEV), c1.to(DEV), c2.to(DEV)), **kw)
dz = torch.empty_like(gy.to(DEV))
cd = coef.to(DEV)
ops.bn_bwd_apply(gy.to(DEV), None, z.to(DEV), cd[0], cd[1], cd[2], cd[3], c1.to(DEV), c2.to(DEV), dz, None, ipb * H * W, E, C, False)
G2 = torch.zeros((N, 9, 64, 64), device=DEV)
ops.conv2d_wgrad(x.to(DEV), dz, G2, **kw)
A = coef[2].view(1, 1, 1, C)
dzr = gy.float() * A + ((z.float() - coef[0].view(1, 1, 1, C)) * (-A * coef[1].view(1, 1, 1, C) * c2.view(1, 1, 1, C)) + (-A * c1.view(1, 1, 1, C)))
print("dz kernel vs CPU", (dz.float().cpu() - dzr.to(BF).float()).abs().max().item())
dzr = dz.float().cpu()
for n in range(N):
    w = torch.zeros(C, 16, 3, 3, requires_grad=True)
    F.conv2d(x[n % ipb:n % ipb + 1].float().permute(0, 3, 1, 2), w, padding=1).backward(dzr[n:n + 1].permute(0, 3, 1, 2))
    ref = w.grad.permute(2, 3, 0, 1).reshape(9, C, 16)
    sc = ref.abs().max().item()
    ef = (G[n].cpu()[:, :, :16] - ref).abs()
    eu = (G2[n].cpu()[:, :, :16] - ref).abs()
    print("image", n, "fused err", ef.max().item() / sc, "per tap", [round(ef[t].max().item() / sc, 4) for t in range(9)],
          "unfused err", eu.max().item() / sc)
print("plan unfused:", ops.conv2d_wgrad.__name__)
