"""A/B of PMOE_RES_INBN (BatchNorm + ReLU of the input applied on load: conv3x3_respipe_kernel<false, 3>, 64 channels) against the
plain launch and the pmoe_bn_apply pass it replaces, at the U-Net level-1 shape of BASELINE config 4 (64 images).  (The 128- and
256-channel rows are what a caller pays there for the pair; the persistent kernel's variant was measured and removed, conv_dma.hip.)
python tools/ab_inbn.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from pmoe_amd import hip, ops  # noqa: E402

DEV, BF, N = "cuda", torch.bfloat16, 64


def t(f, n=20):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for C, H in ((64, 256), (128, 128), (256, 64)):
    z = torch.randn(N, H, H, C, device=DEV).to(BF)
    w = (torch.randn(1, C, 9, C, device=DEV) * 0.05).to(BF)
    out, a_ = torch.empty_like(z), torch.empty_like(z)
    coef = torch.rand(4, 1, C, device=DEV) + 0.5
    kw = dict(cin=C, cout=C, coutp=C, ipe=N, ks=3, stride=1, pad=1)
    rows = ops.conv2d_stat_rows(N, H, H, H, H, C, C, C, N, 3, 1, 1, BF)
    st = torch.zeros(rows, 2, C, device=DEV)
    for rep in range(2):
        p = t(lambda: ops.conv2d(z, w, out, stats=st, **kw))
        code = ops.conv2d(z, w, out, res_mode=hip.RES_INBN, bn_coef=coef, plan_only=True, **kw)
        i = t(lambda: ops.conv2d(z, w, out, stats=st, res_mode=hip.RES_INBN, bn_coef=coef, **kw)) if code > 0 else float("nan")
        b = t(lambda: ops.bn_apply(z, None, a_, coef[2], coef[3], coef[0], N * H * H, 1, C, True))
        print(f"C={C:3d} {H}x{H}: plain {p:.3f} ms (plan {ops.conv2d(z, w, out, plan_only=True, **kw)})  on load {i:.3f} ms "
              f"(plan {code})  bn_apply {b:.3f} ms  "
              f"pair {p + b:.3f} -> {i:.3f}", flush=True)
