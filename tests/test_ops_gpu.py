"""Per-kernel parity (-m gpu): each C-ABI entry point against a plain PyTorch fp32 CPU reference of
the same op, called through the same ctypes binding the model uses.

Tolerances: f32 kernels 1e-4 (exact-f32 MFMA chains vs CPU summation order); bf16 kernels 1e-2
relative to the tensor's scale with the inputs pre-rounded to bf16 (north_star: 1e-4 fp32 / 1e-2 bf16).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from pmoe_amd import hip, ops  # noqa: E402

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def rnd(shape, g, dtype, scale=1.0):
    x = torch.randn(shape, generator=g) * scale
    return x.to(dtype).float() if dtype == torch.bfloat16 else x


def close(got, ref, dtype, what="", floor=2.0 ** -4):
    """``floor``: share of the tensor's scale added to |ref| in the elementwise bound.  1/16 where the kernel sees exactly the
    operands of the reference (inputs representable in the storage type: only summation order and the output rounding differ);
    1/4 where the kernel re-rounds a DERIVED operand to bf16 that the reference keeps in f32 (BatchNorm-scaled / gated weight
    packs): that noise is 2^-9 sqrt(K) rms(w x) -- measured 0.2 % of the tensor's scale -- whatever the element's own size."""
    got = got.float().cpu()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item() / scale
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    assert err <= tol, f"{what}: rel-to-max err {err:.3e} > {tol} (scale {scale:.3e})"
    # ... and elementwise (rel-to-max alone is blind to errors on small elements): every element within
    # tol of ITS OWN magnitude plus a floor of 1/16 of the tensor's scale
    bad = (got - ref).abs() > tol * (ref.abs() + scale * floor)
    if bad.any():
        i = ((got - ref).abs() / (ref.abs() + scale * floor)).argmax()
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} elements beyond {tol} x (|ref| + scale x {floor}); worst "
                             f"got {got.flatten()[i].item():.6e} ref {ref.flatten()[i].item():.6e} (scale {scale:.3e})")


def r16(c):
    return (c + 15) // 16 * 16


def r64(c):
    return (c + 63) // 64 * 64


def nhwc(x, cpad, dtype):
    """CPU NCHW f32 -> device NHWC padded."""
    n, c, h, w = x.shape
    out = torch.zeros(n, h, w, cpad)
    out[..., :c] = x.permute(0, 2, 3, 1)
    return out.to(dtype).to(DEV).contiguous()


def from_nhwc(t, c):
    return t.float().cpu()[..., :c].permute(0, 3, 1, 2).contiguous()


def pack(ws, ks, dtype, want_dgrad=False):
    """ws: list (per expert) of CPU f32 [cout,cin,ks,ks] -> packed device tensors via the pack kernel."""
    E = len(ws)
    cout, cin = ws[0].shape[:2]
    dev_ws = [w.to(DEV).contiguous() for w in ws]
    tab = hip.ptr_table(dev_ws, DEV)
    coutp, cinp = r64(cout), r16(cin)
    fwd = torch.empty(E, coutp, ks * ks, cinp, dtype=dtype, device=DEV)
    dg = None
    cinp2, coutp2 = r64(cin), r16(cout)
    if want_dgrad:
        dg = torch.empty(E, cinp2, ks * ks, coutp2, dtype=dtype, device=DEV)
    ops.pack_conv_weights(tab, fwd, dg, E, cout, cin, ks, coutp, cinp, cinp2, coutp2, dtype)
    torch.cuda.synchronize()
    return fwd, dg, dev_ws


CONV_CASES = [
    # E, ipe, cin, cout, H, W, ks, stride
    (2, 3, 64, 64, 32, 32, 3, 1),
    (2, 2, 12, 64, 40, 24, 3, 1),
    (1, 2, 64, 128, 32, 32, 3, 2),
    (2, 2, 64, 128, 32, 32, 1, 2),
    (1, 3, 128, 128, 16, 16, 3, 1),
    (2, 2, 128, 256, 14, 14, 3, 2),
    (1, 2, 64, 64, 13, 9, 3, 2),          # odd sizes: the parity classes of the stride-2 data gradient differ in extent
    (2, 1, 64, 128, 7, 10, 3, 2),
    (1, 5, 256, 256, 7, 7, 3, 1),
    (2, 3, 512, 512, 4, 4, 3, 1),
    (2, 3, 64, 64, 40, 24, 3, 1),         # resident-filter kernel (register read-out): ragged tiles on both axes
    (2, 3, 64, 64, 8, 16, 3, 1),          # ... two images per tile, the expert's last tile half empty
]


def _conv_case(case, dtype, plan=None):
    """forward (+ fused BatchNorm partial sums), data gradient, weight gradient of one grouped conv against CPU fp32 autograd.
    ``plan`` = (forward code, data-gradient code, weight-gradient K-split x channel-tile pairs): asserted BEFORE running, so
    a routing change cannot silently move a shape off the kernel this case exists to cover."""
    E, ipe, cin, cout, H, W, ks, stride = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    pad = ks // 2
    N = E * ipe
    x = rnd((N, cin, H, W), g, dtype)
    ws = [rnd((cout, cin, ks, ks), g, dtype, (2.0 / (cin * ks * ks)) ** 0.5) for _ in range(E)]
    Ho, Wo = ops.conv_out_size(H, ks, stride, pad), ops.conv_out_size(W, ks, stride, pad)
    dy = rnd((N, cout, Ho, Wo), g, dtype)

    # reference (CPU fp32 autograd)
    xr = x.clone().requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in ws]
    yr = torch.cat([F.conv2d(xr[e * ipe:(e + 1) * ipe], wr[e], stride=stride, padding=pad) for e in range(E)])
    yr.backward(dy)

    cinp, coutp = r16(cin), r64(cout)
    wf, wd, _ = pack(ws, ks, dtype, want_dgrad=True)
    xd = nhwc(x, cinp, dtype)
    out = torch.full((N, Ho, Wo, r16(cout)), 7.0, dtype=dtype, device=DEV)
    rows = ops.conv2d_stat_rows(N, H, W, Ho, Wo, cinp, cout, coutp, ipe, ks, stride, pad, dtype)
    stats = torch.zeros(rows, 2, coutp, device=DEV)
    if plan is not None:
        got = ops.conv2d(xd, wf, out, cin=cinp, cout=cout, coutp=coutp, ipe=ipe, ks=ks, stride=stride, pad=pad, plan_only=True)
        assert got == plan[0], f"forward routed to kernel code {got}, this case is meant for {plan[0]}"
    ops.conv2d(xd, wf, out, cin=cinp, cout=cout, coutp=coutp, ipe=ipe, ks=ks, stride=stride, pad=pad, stats=stats)
    y = from_nhwc(out, cout)
    close(y, yr.detach(), dtype, "conv fwd")
    # fused BN partial sums == column sums of the stored output
    st = stats.view(E, rows // E, 2, coutp).sum(1).cpu()
    yo = out.float().cpu()[..., :cout].reshape(E, -1, cout)
    close(st[:, 0, :cout], yo.sum(1), torch.float32 if dtype == torch.float32 else dtype, "fused stats sum")
    close(st[:, 1, :cout], (yo * yo).sum(1), torch.float32 if dtype == torch.float32 else dtype, "fused stats sumsq")

    # data gradient: stride 1 = conv with flipped weights; stride 2 = dilated source
    dyd = nhwc(dy, r16(cout), dtype)
    dx = torch.empty(N, H, W, cinp, dtype=dtype, device=DEV)
    if plan is not None:
        got = ops.conv2d(dyd, wd, dx, cin=r16(cout), cout=cinp, coutp=r64(cin), ipe=ipe, ks=ks, stride=1,
                         pad=ks - 1 - pad, dilate=(stride == 2), plan_only=True)
        assert plan[1] is None or got == plan[1], f"data gradient routed to kernel code {got}, this case is meant for {plan[1]}"
    ops.conv2d(dyd, wd, dx, cin=r16(cout), cout=cinp, coutp=r64(cin), ipe=ipe, ks=ks, stride=1,
               pad=ks - 1 - pad, dilate=(stride == 2))
    close(from_nhwc(dx, cin), xr.grad, dtype, "conv dgrad")
    if stride == 1 and ks == 3 and plan is not None and plan[1] in (5007, 5017, 5027, 5037, 5047, 5057, 1207):
        # a data gradient accumulating into the gradient another consumer left (BasicBlock `.1.conv1`: the identity branch's):
        # PMOE_RES_ADD with the residual prefetched under the MFMAs on both LDS-DMA kernels
        prev = rnd((N, cin, H, W), torch.Generator().manual_seed(9), dtype)
        acc = nhwc(prev, cinp, dtype)
        ops.conv2d(dyd, wd, acc, cin=r16(cout), cout=cinp, coutp=r64(cin), ipe=ipe, ks=ks, stride=1, pad=ks - 1 - pad,
                   res=acc, res_mode=hip.RES_ADD)
        # (the sum is formed on the bf16-ROUNDED accumulator, like the stored gradient it is added to: an absolute error of half
        #  an ulp of the larger operand whatever the size of the sum -- hence the floor of a quarter of the scale)
        close(from_nhwc(acc, cin), xr.grad + prev, dtype, "conv dgrad accumulated onto an existing gradient", floor=0.25)
    if stride == 2:
        # second consumer: accumulated IN PLACE into an existing gradient (for the 1x1 case only the even pixels are touched)
        prev = rnd((N, cin, H, W), torch.Generator().manual_seed(9), dtype)
        acc = nhwc(prev, cinp, dtype)
        ops.conv2d(dyd, wd, acc, cin=r16(cout), cout=cinp, coutp=r64(cin), ipe=ipe, ks=ks, stride=1, pad=ks - 1 - pad,
                   dilate=True, res=acc, res_mode=hip.RES_ADD)
        close(from_nhwc(acc, cin), xr.grad + prev, dtype, "conv dgrad accumulated in place")

    # weight gradient
    ckw = 64 if dtype == torch.bfloat16 else 32
    cpw, cow = (cinp + ckw - 1) // ckw * ckw, (r16(cout) + ckw - 1) // ckw * ckw
    ws_buf = torch.full((E, ks * ks, cow, cpw), 3.0, device=DEV)      # overwritten, not accumulated onto
    if plan is not None:
        nsplit = ops.conv2d_wgrad(xd, dyd, ws_buf, cin=cinp, cout=r16(cout), cinp=cpw, coutp=cow, ipe=ipe, ks=ks, stride=stride,
                                  pad=pad, plan_only=True)
        assert plan[2] is None or nsplit * E * (cow // ckw) * (cpw // ckw) == plan[2], (nsplit, plan[2])
    ops.conv2d_wgrad(xd, dyd, ws_buf, cin=cinp, cout=r16(cout), cinp=cpw, coutp=cow, ipe=ipe, ks=ks, stride=stride, pad=pad)
    grads = torch.empty(E, cout, cin, ks, ks, device=DEV)
    ops.unpack_conv_wgrad(ws_buf, grads, E, cout, cin, ks, cow, cpw)
    for e in range(E):
        close(grads[e], wr[e].grad, dtype, f"conv wgrad e{e}")
    # round 3: the launch's own K-split fold writes the parameter-layout gradient (WgradDesc.grads): bit-identical to the
    # workspace + pmoe_unpack_conv_wgrad pair (same fixed-order fold)
    direct = torch.full((E, cout, cin, ks, ks), 9.0, device=DEV)
    ops.conv2d_wgrad(xd, dyd, ws_buf, cin=cinp, cout=r16(cout), cinp=cpw, coutp=cow, ipe=ipe, ks=ks, stride=stride, pad=pad,
                     grads=direct.view(-1), grads_cout=cout, grads_cin=cin)
    # (the fold sums the K-split slabs in four interleaved chains -- a fixed order, but not the sequential one of
    #  wgrad_reduce_kernel: equal to f32 rounding, bit-identical between its own two call forms below)
    assert (direct - grads).abs().max().item() <= 1e-5 * grads.abs().max().item() + 1e-7, "fold + unpack differs from the two-launch path"
    first = direct.clone()
    # ... and as two calls (the engine's form: the MFMA kernel and the fold are then timed apart)
    direct.fill_(9.0)
    d = ops.conv2d_wgrad(xd, dyd, ws_buf, cin=cinp, cout=r16(cout), cinp=cpw, coutp=cow, ipe=ipe, ks=ks, stride=stride, pad=pad,
                         grads=direct.view(-1), grads_cout=cout, grads_cin=cin, defer_fold=True)
    ops.conv2d_wgrad_fold(d)
    assert torch.equal(direct, first)




@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dtype):
    _conv_case(case, dtype)


# The layer shapes of BASELINE config 2 (E=4, B=64, 256x256) at reduced image counts, each pinned to the kernel
# instantiation that serves it in the benchmark (VERDICT r1 "weak" 3 / ADVICE r1: the dominant kernels had no isolated
# parity test).  Codes: include/pmoe_hip.h pmoe_conv2d_plan; the third number = workgroups of the weight-gradient launch.
BASELINE_CONV_CASES = [
    # (E, ipe, cin, cout, H, W, ks, stride), (fwd, dgrad, wgrad workgroups)
    ((1, 4, 128, 128, 64, 64, 3, 1), (5047, 5047, 256)),       # layer2: conv3x3_dma_stream_kernel<false> (LDS-DMA, persistent, producer wave), 2 channel chunks
    ((1, 8, 256, 256, 32, 32, 3, 1), (5057, 5057, 256)),       # layer3: 4 chunks, 32 x 8 pixel tiles; the 16x16x32 MFMA instantiation of the persistent kernel
    ((2, 32, 512, 512, 16, 16, 3, 1), (5037, 5037, 256)),      # layer4 (eight chunks: the one-tile producer-wave kernel): 8 chunks, one 16 x 16 image per tile (two patch rows per
                                                               # 32-lane fragment: the column-keyed swizzle), 2 experts
    ((1, 5, 128, 256, 40, 24, 3, 1), (5047, 5057, 200)),       # ragged: 24-wide rows in 32-wide tiles, 40 rows in strips of 8, 5 images
    ((1, 1, 64, 64, 256, 256, 3, 1), (1207, 1207, 256)),       # stem conv2: conv3x3_respipe_kernel, 256 tiles of one image
    ((2, 2, 64, 64, 128, 128, 3, 1), (1207, 1207, 256)),       # layer1: resident kernel, persistent workgroups per expert
    ((1, 4, 64, 128, 128, 128, 3, 2), (5207, 4741, 256)),      # layer2.0.conv1: conv3x3s2_dma_kernel (parity planes by LDS-DMA); its data gradient (64 gradient rows) stays on the generic kernel's 4 parity-class launches
    ((1, 8, 128, 256, 64, 64, 3, 2), (5207, 9207, None)),      # layer3.0.conv1: 2 channel chunks per plane; 4 chunks of dy per class
    ((2, 16, 256, 512, 32, 32, 3, 2), (5207, 9207, None)),     # layer4.0.conv1: 4 chunks (single-tap steps back to back), 4 cout blocks, 2 experts
    ((1, 6, 64, 128, 71, 55, 3, 2), (5207, None, None)),       # odd image sides: the last block row / column has only its plane-0 pixels; ragged tiles
    ((1, 4, 64, 128, 128, 128, 1, 2), (1404, 741, 128)),       # layer2.0.downsample (1x1 stride 2: conv1x1_direct_kernel<4>)
    ((1, 1, 12, 64, 256, 256, 3, 1), (1316, 1207, 256)),       # stem conv1 (12 -> 16 input channels): direct-form conv3x3_c16_kernel
    ((2, 3, 12, 64, 40, 40, 3, 1), (1316, None, None)),        # ... ragged: 40-wide rows = one full + one 8-pixel tile, 2 experts x 3 images
    ((1, 2, 9, 48, 33, 96, 3, 1), (1316, None, None)),         # ... 9 input / 48 output channels (zero-padded filter rows, masked stores)
]


@pytest.mark.parametrize("case,plan", BASELINE_CONV_CASES)
def test_conv_baseline_layer_shapes_bf16(case, plan):
    _conv_case(case, torch.bfloat16, plan)


# BASELINE config 4: the 64-output-channel convolutions over >= 128 input channels (the U-Nets' `up_forw_4.0`, 128 -> 64 at full
# resolution; model/blocks/unet.py:57-63) on conv3x3_dma_stream_kernel<false, true>: 256-pixel x 64-cout tiles
NARROW_CONV_CASES = [
    ((1, 2, 128, 64, 128, 128, 3, 1), (5067, None, None)),     # two chunks, 32 x 8 pixel tiles, 128 tiles for the persistent grid
    ((2, 3, 192, 64, 40, 56, 3, 1), (5067, None, None)),       # three chunks, ragged tiles, 2 experts x 3 images
    ((1, 17, 128, 56, 16, 16, 3, 1), (5067, None, None)),      # 56 of 64 output channels, one 16 x 16 image per tile, 17 images
]


@pytest.mark.parametrize("case,plan", NARROW_CONV_CASES)
def test_conv_64_output_channel_tiles(case, plan, monkeypatch):
    _conv_case(case, torch.bfloat16, plan)
    monkeypatch.setenv("PMOE_DMA_NARROW", "0")              # the A/B partner: the generic register-staged kernel
    _conv_case(case, torch.bfloat16, (741, None, None))


# (E_bn, images per BN set, channels, H, W, conv runs per image?, bias?) -> kernel code
DBN_CASES = [
    ((2, 2, 64, 128, 128, False, False), 1247),        # layer1 conv2 data gradient: conv3x3_respipe_kernel<false, 2>, persistent workgroups
    ((2, 3, 64, 64, 96, True, True), 1257),            # stem conv2 (ECA gate folded: one "expert" per image, per-image bias)
    ((1, 4, 128, 64, 64, False, False), 5007),         # layer2: conv3x3_dma_kernel
    ((2, 16, 512, 16, 16, False, False), 5017),        # layer4: one image per tile row group, 4 output-channel blocks
    ((1, 5, 256, 40, 24, False, False), 5017),         # ragged tiles
]


@pytest.mark.parametrize("case", [(2, 3, 64, 96), (1, 5, 40, 56), (3, 14, 16, 16)])
def test_conv_applies_input_batchnorm_relu_on_load(case):
    """PMOE_RES_INBN (round 4, the frozen U-Nets' conv -> BatchNorm -> ReLU -> conv pairs): the 64 -> 64 resident-filter kernel
    evaluates relu(BatchNorm(z)) on its halo patch in LDS.  Against pmoe_bn_apply + the plain launch of the same kernel: output
    and fused statistics BIT-identical (same arithmetic and rounding, same accumulation order); pixels outside the image must
    stay zero (relu(bn(0)) is not), which the ragged cases and the shifted coefficients exercise."""
    E, ipe, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    N, C_ = E * ipe, 64
    BF = torch.bfloat16
    z = nhwc(rnd((N, C_, H, W), g, BF, 2.0) + 0.3, C_, BF)
    ws = [rnd((C_, C_, 3, 3), g, BF, (2.0 / (C_ * 9)) ** 0.5) for _ in range(E)]
    wf, _, _ = pack(ws, 3, BF)
    coef = torch.empty(4, E, C_, device=DEV)
    coef[0] = rnd((E, C_), g, torch.float32, 0.5).to(DEV) + 0.3            # mean
    coef[1] = 1.0                                                            # invstd (unused by the forward)
    coef[2] = (rnd((E, C_), g, torch.float32, 0.3).abs() + 0.5).to(DEV)      # gamma * invstd
    coef[3] = rnd((E, C_), g, torch.float32, 0.4).to(DEV) + 0.2              # beta: relu(bn(0)) > 0 for most channels
    kw = dict(cin=C_, cout=C_, coutp=C_, ipe=ipe, ks=3, stride=1, pad=1)
    rows = ops.conv2d_stat_rows(N, H, W, H, W, C_, C_, C_, ipe, 3, 1, 1, BF)
    # the pair
    a_ = torch.empty_like(z)
    ops.bn_apply(z, None, a_, coef[2], coef[3], coef[0], ipe * H * W, E, C_, True)
    ref, st_ref = torch.full_like(z, 7.0), torch.zeros(rows, 2, C_, device=DEV)
    assert ops.conv2d(a_, wf, ref, plan_only=True, **kw) == 1207
    ops.conv2d(a_, wf, ref, stats=st_ref, **kw)
    # on load
    out, st = torch.full_like(z, 5.0), torch.zeros(rows, 2, C_, device=DEV)
    assert ops.conv2d(z, wf, out, res_mode=hip.RES_INBN, bn_coef=coef, plan_only=True, **kw) == 1267
    ops.conv2d(z, wf, out, res_mode=hip.RES_INBN, bn_coef=coef, stats=st, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()
    assert torch.equal(st, st_ref)
    # and the pair itself against CPU fp32 (so that "identical" is not "identically wrong")
    zc = z.float().cpu().permute(0, 3, 1, 2)
    cc = coef.cpu()
    yr = torch.cat([F.conv2d(torch.relu((zc[e * ipe:(e + 1) * ipe] - cc[0, e].view(1, -1, 1, 1)) * cc[2, e].view(1, -1, 1, 1)
                                        + cc[3, e].view(1, -1, 1, 1)).to(BF).float(), ws[e].to(BF).float(), padding=1) for e in range(E)])
    close(from_nhwc(out, C_), yr, BF, "conv over relu(bn(z)) applied on load")
    # what the kernel does not serve says so (the caller then runs the pair)
    w128, _, _ = pack([rnd((128, C_, 3, 3), g, BF, 0.1) for _ in range(E)], 3, BF)
    o128 = torch.empty(N, H, W, 128, dtype=BF, device=DEV)
    assert ops.conv2d(z, w128, o128, res_mode=hip.RES_INBN, bn_coef=coef, plan_only=True, cin=C_, cout=128, coutp=128, ipe=ipe,
                      ks=3, stride=1, pad=1) == hip.ERR_UNSUPPORTED


@pytest.mark.parametrize("case,plan", DBN_CASES)
def test_conv_dgrad_with_batchnorm_reductions(case, plan):
    _dbn_case(case, plan)


def test_conv_resident_lds_staged_fallback(monkeypatch):
    """PMOE_RES_PIPE=0 (read per launch): the 64-channel layers back on conv3x3_resdma_kernel (read-out staged through LDS
    between the tiles), the A/B partner of conv3x3_respipe_kernel: same parity bar."""
    monkeypatch.setenv("PMOE_RES_PIPE", "0")
    _conv_case((2, 2, 64, 64, 128, 128, 3, 1), torch.bfloat16, (1107, 1107, 256))
    _dbn_case(DBN_CASES[0][0], 1107)
    _dbn_case(DBN_CASES[1][0], 1117)


def _dbn_case(case, plan):
    """PMOE_RES_DBN (round 3): the data gradient into a = relu(BatchNorm(z)) masks itself with the recomputed ReLU decision
    and leaves the BatchNorm backward's two channel reductions in `stats`.  Against the plain data gradient of the same
    launch masked on the host (BIT-identical: same accumulators, the mask only selects) and host sums of that."""
    Ebn, ipb, C, H, W, per_image, with_bias = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    N = Ebn * ipb
    E = N if per_image else Ebn
    ipe = 1 if per_image else ipb
    BF = torch.bfloat16
    dy = rnd((N, C, H, W), g, BF)
    z = rnd((N, C, H, W), g, BF, 2.0) + 0.3
    ws = [rnd((C, C, 3, 3), g, torch.float32, (2.0 / (C * 9)) ** 0.5) for _ in range(E)]
    _, wd, _keep = pack(ws, 3, BF, want_dgrad=True)
    coef = torch.empty(4, Ebn, C)
    coef[0] = torch.randn(Ebn, C, generator=g) * 0.5 + 0.3             # mean
    coef[1] = torch.rand(Ebn, C, generator=g) + 0.5                     # invstd
    coef[2] = coef[1] * (torch.randn(Ebn, C, generator=g) * 0.5 + 1.0)  # gamma * invstd (some negative gammas too)
    coef[3] = torch.randn(Ebn, C, generator=g) * 0.3                    # beta
    coefd = coef.to(DEV)
    bias = (torch.randn(E, r64(C), generator=g) * 0.05).to(DEV) if with_bias else None
    dyd, zd = nhwc(dy, C, BF), nhwc(z, C, BF)
    kw = dict(cin=C, cout=C, coutp=r64(C), ipe=ipe, ks=3, stride=1, pad=1, bias=bias)
    plain = torch.empty(N, H, W, C, dtype=BF, device=DEV)
    ops.conv2d(dyd, wd, plain, **kw)
    got = torch.full((N, H, W, C), 7.0, dtype=BF, device=DEV)
    common = dict(res=zd, res_mode=hip.RES_DBN, bn_coef=coefd, bn_ipe=ipb, **kw)
    assert ops.conv2d(dyd, wd, got, plan_only=True, **common) == plan
    rows = ops.conv2d_stat_rows(N, H, W, H, W, C, C, r64(C), ipe, 3, 1, 1, BF)
    stats = torch.full((rows, 2, r64(C)), 5.0, device=DEV)
    ops.conv2d(dyd, wd, got, stats=stats, **common)
    cb = coef.repeat_interleave(ipb, dim=1).view(4, N, 1, 1, C)          # per image
    zf = zd.float().cpu()
    d = zf - cb[0]
    mask = (d * cb[2] + cb[3]) > 0
    assert 0.2 < mask.float().mean() < 0.8
    want = torch.where(mask, plain.float().cpu(), torch.zeros(()))
    bad = (got.float().cpu() != want)
    assert not bad.any(), ("masked data gradient differs from the plain one", int(bad.sum()), bad.nonzero()[:8].tolist(),
                           bad.sum((0, 1, 2)).nonzero().flatten().tolist()[:64])
    st = stats.view(Ebn, rows // Ebn, 2, r64(C)).sum(1).cpu()
    s1 = want.view(Ebn, -1, C).sum(1)
    s2 = (want * d * cb[1]).view(Ebn, -1, C).sum(1)
    for nm, a, b in (("sum g", st[:, 0, :C], s1), ("sum g*xhat", st[:, 1, :C], s2)):
        scale = b.abs().max().item()
        assert ((a - b).abs().max().item() <= 2e-4 * scale + 1e-3), (nm, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("case", [(1, 4, 128, 128, 64, 64, 3, 1), (2, 32, 512, 512, 16, 16, 3, 1), (1, 5, 128, 256, 40, 24, 3, 1)])
def test_conv_dma_mfma16_variant(monkeypatch, case):
    """PMOE_DMA_MF16=1 (read per launch): conv3x3_dma_kernel on v_mfma_f32_16x16x32_bf16 -- other fragment / accumulator layouts,
    same tile, same parity bar (forward with fused statistics, data gradient)."""
    monkeypatch.setenv("PMOE_DMA_MF16", "1")
    monkeypatch.setenv("PMOE_DMA_STREAM", "0")
    monkeypatch.setenv("PMOE_DMA_PRODUCER", "2")            # (2: the producer-wave instantiation for every channel count)
    _conv_case(case, torch.bfloat16, (5037, 5037, None))


@pytest.mark.parametrize("case,codes", [((1, 4, 128, 128, 64, 64, 3, 1), (5007, 5007, None)), ((2, 32, 512, 512, 16, 16, 3, 1), (5017, 5017, None)),
                                        ((1, 5, 128, 256, 40, 24, 3, 1), (5007, 5017, None))])
def test_conv_dma_without_producer_wave(monkeypatch, case, codes):
    """PMOE_DMA_PRODUCER=0 (read per launch): plain forward / data-gradient launches back on the 8-wave conv3x3_dma_kernel<MF16> in
    which every wave issues its own LDS-DMA requests -- the A/B partner of the round-4 producer-wave instantiation, and the kernel
    the side-input epilogue modes (PMOE_RES_DBN / PMOE_RES_ADD) always run on."""
    monkeypatch.setenv("PMOE_DMA_PRODUCER", "0")
    monkeypatch.setenv("PMOE_DMA_STREAM", "0")
    _conv_case(case, torch.bfloat16, codes)


@pytest.mark.parametrize("case", [(1, 4, 128, 128, 64, 64, 3, 1), (2, 32, 512, 512, 16, 16, 3, 1), (1, 5, 128, 256, 40, 24, 3, 1),
                                  (3, 70, 128, 128, 64, 64, 3, 1)])
def test_conv_dma_persistent_stream(monkeypatch, case):
    """conv3x3_dma_stream_kernel (round 4): persistent workgroups, the producer wave's request stream running across tile boundaries,
    the epilogue staged in the patch buffer / ring slots the stream is not filling.  Same parity bar as the one-tile kernels, and --
    same accumulation order, same rounding -- BIT-IDENTICAL forward output and BatchNorm partial sums; the last case gives every
    workgroup several tiles incl. a ragged tail (3 experts x 70 images of 64 x 64: 3360 tiles on 256 workgroups)."""
    E, ipe, cin, cout, H, W, ks, stride = case
    monkeypatch.setenv("PMOE_DMA_STREAM", "2")              # (2: also for more than four channel chunks)
    if E * ipe <= 64:
        _conv_case(case, torch.bfloat16, None)
    g = torch.Generator().manual_seed(5)
    BF = torch.bfloat16
    N = E * ipe
    x = nhwc(rnd((N, cin, H, W), g, BF), cin, BF)
    ws = [rnd((cout, cin, 3, 3), g, BF, (2.0 / (cin * 9)) ** 0.5) for _ in range(E)]
    wf, _, _keep = pack(ws, 3, BF)
    outs = []
    for stream in ("2", "0"):
        monkeypatch.setenv("PMOE_DMA_STREAM", stream)
        kw = dict(cin=cin, cout=cout, coutp=r64(cout), ipe=ipe, ks=3, stride=1, pad=1)
        y = torch.full((N, H, W, cout), 7.0, dtype=BF, device=DEV)
        code = ops.conv2d(x, wf, y, plan_only=True, **kw)
        assert code in ((5047, 5057) if stream == "2" else (5007, 5017, 5027, 5037)), code
        rows = ops.conv2d_stat_rows(N, H, W, H, W, cin, cout, r64(cout), ipe, 3, 1, 1, BF)
        st = torch.full((rows, 2, r64(cout)), 3.0, device=DEV)
        ops.conv2d(x, wf, y, stats=st, **kw)
        outs.append((y, st))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])



@pytest.mark.parametrize("req", ["0", "2"])
def test_conv_wgrad_request_modes(monkeypatch, req):
    """PMOE_WGRAD_REQ (read per launch): the weight-gradient kernel's round-2 request code (0) and the variant that spreads the
    requests between the k-blocks (2) -- the A/B partners of the shipped mode 1 (tile-invariant piece geometry, guarded range
    checks): same parity bar on a layer1 shape and on ragged tiles with an odd image count per expert."""
    monkeypatch.setenv("PMOE_WGRAD_REQ", req)
    _conv_case((2, 2, 64, 64, 128, 128, 3, 1), torch.bfloat16)
    _conv_case((1, 5, 128, 256, 40, 24, 3, 1), torch.bfloat16)
    _conv_case((2, 3, 256, 128, 16, 16, 3, 1), torch.bfloat16)


def test_conv_resident_pingpong_fallback(monkeypatch):
    """PMOE_RES_DMA=0 (read per launch) routes the 64-channel layers back to conv3x3_res_kernel<7>, the kernel that also
    serves shapes the LDS-DMA variant declines: same parity bar, plan code 1007."""
    monkeypatch.setenv("PMOE_RES_DMA", "0")
    _conv_case((2, 2, 64, 64, 128, 128, 3, 1), torch.bfloat16, (1007, 1007, 256))
    monkeypatch.setenv("PMOE_CONV_C16", "0")            # ... and the 16-channel stem convolution back on conv3x3_res_kernel<5>
    _conv_case((1, 2, 12, 64, 64, 64, 3, 1), torch.bfloat16, (1005, None, None))


@pytest.mark.parametrize("case", [c for c, _ in BASELINE_CONV_CASES[:4]])
def test_conv_baseline_layer_shapes_f32(case):
    _conv_case(case, torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_shared_input_and_epilogues(dtype):
    """in_shared (ECA1-style input read by all experts), bias+ReLU/ELU, residual add, column offsets."""
    g = torch.Generator().manual_seed(3)
    E, ipe, K, Nn = 3, 5, 6, 512
    x = rnd((ipe, K), g, dtype)
    ws = [rnd((Nn, K, 1, 1), g, dtype, 0.5) for _ in range(E)]
    bs = [rnd((Nn,), g, torch.float32, 0.3) for _ in range(E)]
    wf, _, _ = pack(ws, 1, dtype)
    dev_b = [b.to(DEV) for b in bs]
    bias = torch.empty(E, r64(Nn), device=DEV)
    ops.pack_bias(hip.ptr_table(dev_b, DEV), bias, E, Nn, r64(Nn))
    xd = torch.zeros(ipe, 1, 1, 16, dtype=dtype, device=DEV)
    xd[:, 0, 0, :K] = x.to(dtype).to(DEV)
    for act, fn in ((hip.ACT_RELU, torch.relu), (hip.ACT_ELU, F.elu), (hip.ACT_TANH, torch.tanh), (hip.ACT_SIGMOID, torch.sigmoid),
                    (hip.ACT_NONE, lambda t: t)):
        out = torch.zeros(E * ipe, 1, 1, 1536, dtype=dtype, device=DEV)
        ops.conv2d(xd, wf, out, cin=16, cout=Nn, coutp=r64(Nn), ipe=ipe, ks=1, stride=1, pad=0, in_shared=True,
                   out_coff=512, bias=bias, act=act)
        ref = torch.cat([fn(x @ ws[e][:, :, 0, 0].t() + bs[e]) for e in range(E)])
        close(out.view(E * ipe, 1536)[:, 512:1024], ref, dtype, f"linear act{act}")
        assert out.view(E * ipe, 1536)[:, :512].abs().max().item() == 0
        assert out.view(E * ipe, 1536)[:, 1024:].abs().max().item() == 0
    # residual add + reading a column slice
    res = rnd((E * ipe, Nn), g, dtype)
    resd = res.to(dtype).to(DEV).view(E * ipe, 1, 1, Nn).contiguous()
    out2 = torch.empty(E * ipe, 1, 1, Nn, dtype=dtype, device=DEV)
    ops.conv2d(xd, wf, out2, cin=16, cout=Nn, coutp=r64(Nn), ipe=ipe, ks=1, stride=1, pad=0, in_shared=True,
               res=resd, res_mode=hip.RES_ADD)
    ref = torch.cat([x @ ws[e][:, :, 0, 0].t() for e in range(E)]) + res
    close(out2.view(E * ipe, Nn), ref, dtype, "res add")
    # activation-derivative epilogues (MLP backward)
    y_saved = rnd((E * ipe, Nn), g, dtype)
    ysd = y_saved.to(dtype).to(DEV).view(E * ipe, 1, 1, Nn).contiguous()
    lin = torch.cat([x @ ws[e][:, :, 0, 0].t() for e in range(E)])
    out3 = torch.empty_like(out2)
    ops.conv2d(xd, wf, out3, cin=16, cout=Nn, coutp=r64(Nn), ipe=ipe, ks=1, stride=1, pad=0, in_shared=True,
               res=ysd, res_mode=hip.RES_DRELU)
    close(out3.view(E * ipe, Nn), lin * (y_saved > 0), dtype, "drelu")
    ops.conv2d(xd, wf, out3, cin=16, cout=Nn, coutp=r64(Nn), ipe=ipe, ks=1, stride=1, pad=0, in_shared=True,
               res=ysd, res_mode=hip.RES_DELU)
    close(out3.view(E * ipe, Nn), lin * torch.where(y_saved > 0, torch.ones_like(y_saved), y_saved + 1), dtype, "delu")
    # tanh' / sigmoid' from the saved output (make_mlp act choices of basics.py:23-28)
    yt = torch.tanh(y_saved).to(dtype).float()
    ops.conv2d(xd, wf, out3, cin=16, cout=Nn, coutp=r64(Nn), ipe=ipe, ks=1, stride=1, pad=0, in_shared=True,
               res=yt.to(dtype).to(DEV).view(E * ipe, 1, 1, Nn).contiguous(), res_mode=hip.RES_DTANH)
    close(out3.view(E * ipe, Nn), lin * (1 - yt * yt), dtype, "dtanh")
    ysg = torch.sigmoid(y_saved).to(dtype).float()
    ops.conv2d(xd, wf, out3, cin=16, cout=Nn, coutp=r64(Nn), ipe=ipe, ks=1, stride=1, pad=0, in_shared=True,
               res=ysg.to(dtype).to(DEV).view(E * ipe, 1, 1, Nn).contiguous(), res_mode=hip.RES_DSIGMOID)
    close(out3.view(E * ipe, Nn), lin * ysg * (1 - ysg), dtype, "dsigmoid")


@pytest.mark.parametrize("case", [(4, 64, 512, 512), (4, 64, 1536, 512), (2, 37, 512, 5), (3, 70, 512, 1536), (1, 130, 48, 64)])
def test_expert_mlp_gemm_shapes_bf16(case):
    """The grouped skinny GEMM (csrc/gemm_skinny.hip) that serves the expert MLP layers in bf16: batch rows per expert that
    are not a multiple of its 64-row tile, K split over the 4 waves (48 .. 1536), 5-row head, bias + ELU + dropout, and the
    activation-derivative epilogue of the data gradient reading the saved (dropped) output."""
    E, ipe, K, Nn = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(K + Nn)
    x = rnd((E * ipe, K), g, dtype)
    ws = [rnd((Nn, K, 1, 1), g, dtype, K ** -0.5) for _ in range(E)]
    bs = [rnd((Nn,), g, torch.float32, 0.3) for _ in range(E)]
    wf, wd, _ = pack(ws, 1, dtype, want_dgrad=True)
    bias = torch.empty(E, r64(Nn), device=DEV)
    ops.pack_bias(hip.ptr_table([b.to(DEV) for b in bs], DEV), bias, E, Nn, r64(Nn))
    xd = x.to(dtype).to(DEV).view(E * ipe, 1, 1, K).contiguous()
    cst = r16(Nn)
    lin = torch.cat([x[e * ipe:(e + 1) * ipe] @ ws[e][:, :, 0, 0].t() + bs[e] for e in range(E)])
    out = torch.full((E * ipe, 1, 1, cst), 9.0, dtype=dtype, device=DEV)
    ops.conv2d(xd, wf, out, cin=K, cout=cst, coutp=r64(Nn), ipe=ipe, ks=1, stride=1, pad=0, bias=bias, act=hip.ACT_ELU)
    close(out.view(E * ipe, cst)[:, :Nn], F.elu(lin), dtype, "mlp elu")
    if cst > Nn:
        assert out.view(E * ipe, cst)[:, Nn:].abs().max().item() == 0           # padded columns of the head stay zero
    # dropout: same seed -> same mask, survivors scaled by 1/(1-p)
    a = torch.empty_like(out)
    b = torch.empty_like(out)
    ops.conv2d(xd, wf, a, cin=K, cout=cst, coutp=r64(Nn), ipe=ipe, ks=1, stride=1, pad=0, bias=bias, act=hip.ACT_ELU,
               drop_p=0.3, seed=77)
    ops.conv2d(xd, wf, b, cin=K, cout=cst, coutp=r64(Nn), ipe=ipe, ks=1, stride=1, pad=0, bias=bias, act=hip.ACT_ELU,
               drop_p=0.3, seed=77)
    assert torch.equal(a, b)
    av, ov = a.view(E * ipe, cst)[:, :Nn].float(), out.view(E * ipe, cst)[:, :Nn].float()
    kept = av != 0
    if Nn >= 64:
        assert abs(kept.float().mean().item() - 0.7) < 0.03
    torch.testing.assert_close(av[kept], (ov / 0.7)[kept], rtol=1e-2, atol=1e-2)
    # data gradient through the dropped ELU layer: dx = (dy * elu'(z) * mask / (1-p)) @ W
    dy = rnd((E * ipe, Nn), g, dtype)
    dyd = torch.zeros(E * ipe, 1, 1, cst, dtype=dtype, device=DEV)
    dyd.view(E * ipe, cst)[:, :Nn] = dy.to(dtype).to(DEV)
    dx = torch.empty(E * ipe, 1, 1, r16(K), dtype=dtype, device=DEV)
    ops.conv2d(dyd, wd, dx, cin=cst, cout=r16(K), coutp=r64(K), ipe=ipe, ks=1, stride=1, pad=0)
    ref_dx = torch.cat([dy[e * ipe:(e + 1) * ipe] @ ws[e][:, :, 0, 0] for e in range(E)])
    close(dx.view(E * ipe, -1)[:, :K], ref_dx, dtype, "mlp dgrad")
    # weight gradient: dW[e] = dy[e]^T x[e]  (a dedicated column-gather kernel for these was measured: -0.1 ms per step, dropped)
    cpw, cow = (r16(K) + 63) // 64 * 64, (cst + 63) // 64 * 64
    wsb = torch.zeros(E, 1, cow, cpw, device=DEV)
    ops.conv2d_wgrad(xd, dyd, wsb, cin=r16(K), cout=cst, cinp=cpw, coutp=cow, ipe=ipe, ks=1, stride=1, pad=0)
    grads = torch.empty(E, Nn, K, 1, 1, device=DEV)
    ops.unpack_conv_wgrad(wsb, grads, E, Nn, K, 1, cow, cpw)
    for e in range(E):
        close(grads[e, :, :, 0, 0], dy[e * ipe:(e + 1) * ipe].t() @ x[e * ipe:(e + 1) * ipe], dtype, f"mlp wgrad e{e}")
    if K % 16 == 0 and Nn >= 16:
        # ... with the activation derivative of the PREVIOUS layer's saved output in the epilogue (RES_DELU, dropout 0.3)
        ysaved = torch.empty(E * ipe, 1, 1, r16(K), dtype=dtype, device=DEV)
        yprev = rnd((E * ipe, K), g, dtype)
        mask = torch.rand(E * ipe, K, generator=g) >= 0.3
        ys = (torch.where(yprev > 0, yprev, torch.expm1(yprev)) * mask / 0.7).to(dtype).float()
        ysaved.view(E * ipe, -1)[:, :K] = ys.to(dtype).to(DEV)
        ops.conv2d(dyd, wd, dx, cin=cst, cout=r16(K), coutp=r64(K), ipe=ipe, ks=1, stride=1, pad=0, res=ysaved,
                   res_mode=hip.RES_DELU, drop_p=0.3)
        yy = ys * 0.7
        want = torch.where(ys == 0, torch.zeros_like(ref_dx), ref_dx * torch.where(yy > 0, torch.ones_like(yy), yy + 1) / 0.7)
        close(dx.view(E * ipe, -1)[:, :K], want, dtype, "mlp dgrad + delu'")


@pytest.mark.parametrize("case", [(3, 1, 512, 512, 7, 7, 3, 1), (3, 1, 256, 256, 14, 14, 3, 1), (2, 1, 128, 256, 14, 14, 3, 2),
                                  (2, 2, 128, 256, 10, 6, 1, 2), (1, 3, 64, 128, 9, 5, 3, 1), (4, 1, 256, 512, 13, 13, 3, 2)])
def test_small_feature_map_convs_bf16(case):
    """Convolutions with at most 256 output pixels per expert (closed-loop inference, B = 1: layer3 / layer4) run on the
    tap-looping skinny kernel: 3x3 / 1x1, stride 1 / 2, zero padding at the borders, folded-BatchNorm epilogue (bias +
    residual add + ReLU), and the stride-1 data gradient."""
    E, ipe, cin, cout, H, W, ks, stride = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(sum(case))
    pad = ks // 2
    N = E * ipe
    x = rnd((N, cin, H, W), g, dtype)
    ws = [rnd((cout, cin, ks, ks), g, dtype, (2.0 / (cin * ks * ks)) ** 0.5) for _ in range(E)]
    bs = [rnd((cout,), g, torch.float32, 0.2) for _ in range(E)]
    Ho, Wo = ops.conv_out_size(H, ks, stride, pad), ops.conv_out_size(W, ks, stride, pad)
    assert ipe * Ho * Wo <= 256
    res = rnd((N, cout, Ho, Wo), g, dtype)
    ref = torch.cat([F.conv2d(x[e * ipe:(e + 1) * ipe], ws[e], bs[e], stride=stride, padding=pad) for e in range(E)])
    wf, wd, _ = pack(ws, ks, dtype, want_dgrad=True)
    bias = torch.empty(E, r64(cout), device=DEV)
    ops.pack_bias(hip.ptr_table([b.to(DEV) for b in bs], DEV), bias, E, cout, r64(cout))
    xd = nhwc(x, r16(cin), dtype)
    out = torch.full((N, Ho, Wo, r16(cout)), 5.0, dtype=dtype, device=DEV)
    ops.conv2d(xd, wf, out, cin=r16(cin), cout=r16(cout), coutp=r64(cout), ipe=ipe, ks=ks, stride=stride, pad=pad, bias=bias,
               act=hip.ACT_RELU, res=nhwc(res, r16(cout), dtype), res_mode=hip.RES_ADD)
    close(from_nhwc(out, cout), torch.relu(ref + res), dtype, "small-map conv + bias + res + relu")
    if stride == 1:
        dy = rnd((N, cout, Ho, Wo), g, dtype)
        xr = x.clone().requires_grad_(True)
        torch.cat([F.conv2d(xr[e * ipe:(e + 1) * ipe], ws[e], stride=1, padding=pad) for e in range(E)]).backward(dy)
        dx = torch.empty(N, H, W, r16(cin), dtype=dtype, device=DEV)
        ops.conv2d(nhwc(dy, r16(cout), dtype), wd, dx, cin=r16(cout), cout=r16(cin), coutp=r64(cin), ipe=ipe, ks=ks, stride=1,
                   pad=ks - 1 - pad)
        close(from_nhwc(dx, cin), xr.grad, dtype, "small-map dgrad")


def test_dropout_epilogue_statistics():
    g = torch.Generator().manual_seed(5)
    E, ipe, K, Nn = 1, 64, 16, 512
    dtype = torch.float32
    x = rnd((ipe, K), g, dtype)
    ws = [rnd((Nn, K, 1, 1), g, dtype, 0.5)]
    wf, _, _ = pack(ws, 1, dtype)
    xd = x.to(DEV).view(ipe, 1, 1, K).contiguous()
    a = torch.empty(ipe, 1, 1, Nn, device=DEV)
    b = torch.empty_like(a)
    ops.conv2d(xd, wf, a, cin=16, cout=Nn, coutp=512, ipe=ipe, ks=1, stride=1, pad=0)
    ops.conv2d(xd, wf, b, cin=16, cout=Nn, coutp=512, ipe=ipe, ks=1, stride=1, pad=0, drop_p=0.3, seed=1234)
    kept = (b != 0)
    frac = kept.float().mean().item()
    assert abs(frac - 0.7) < 0.02, frac
    torch.testing.assert_close(b[kept], a[kept] / 0.7, rtol=1e-5, atol=1e-6)
    c = torch.empty_like(a)
    ops.conv2d(xd, wf, c, cin=16, cout=Nn, coutp=512, ipe=ipe, ks=1, stride=1, pad=0, drop_p=0.3, seed=1234)
    assert torch.equal(b, c)          # same seed -> same mask
    ops.conv2d(xd, wf, c, cin=16, cout=Nn, coutp=512, ipe=ipe, ks=1, stride=1, pad=0, drop_p=0.3, seed=99)
    assert not torch.equal(b, c)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 3, 64, 20, 12), (1, 2, 128, 9, 9), (3, 2, 512, 4, 4)])
def test_batchnorm_fwd_bwd(shape, dtype):
    E, ipe, C_, H, W = shape
    g = torch.Generator().manual_seed(11)
    N = E * ipe
    x = (rnd((N, C_, H, W), g, dtype) * 1.5 + 0.3).to(dtype).float()
    res = rnd((N, C_, H, W), g, dtype)
    dy = rnd((N, C_, H, W), g, dtype)
    gam = [1 + 0.1 * torch.randn(C_, generator=g) for _ in range(E)]
    bet = [0.1 * torch.randn(C_, generator=g) for _ in range(E)]
    rm = [0.05 * torch.randn(C_, generator=g) for _ in range(E)]
    rv = [1 + 0.1 * torch.rand(C_, generator=g) for _ in range(E)]
    # reference
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True)
    gr = [t.clone().requires_grad_(True) for t in gam]
    br = [t.clone().requires_grad_(True) for t in bet]
    rmr, rvr = [t.clone() for t in rm], [t.clone() for t in rv]
    yr = torch.cat([torch.relu(F.batch_norm(xr[e * ipe:(e + 1) * ipe], rmr[e], rvr[e], gr[e], br[e], True, 0.1, 1e-5)
                               + rr[e * ipe:(e + 1) * ipe]) for e in range(E)])
    yr.backward(dy)
    # device
    xd, resd, dyd = nhwc(x, C_, dtype), nhwc(res, C_, dtype), nhwc(dy, C_, dtype)
    dg, db, drm, drv = ([t.to(DEV) for t in l] for l in (gam, bet, rm, rv))
    tabs = [hip.ptr_table(l, DEV) for l in (dg, db, drm, drv)]
    rpe = ipe * H * W
    nparts = 4
    part = torch.empty(E, nparts, 2, C_, device=DEV)
    shiftc = torch.empty(E, C_, device=DEV)
    ops.colstats(rpe, xd, E, C_, part, nparts, shiftc=shiftc)
    part2 = torch.empty(E, 2, 2, C_, device=DEV)
    ops.reduce_partials(part, part2, E, nparts, 2, 2 * C_)
    scale, shift, mean, invstd = (torch.empty(E, C_, device=DEV) for _ in range(4))
    ops.bn_finalize(part2, 2, rpe, tabs[0], tabs[1], tabs[2], tabs[3], 0.1, 1e-5, True, scale, shift, mean, invstd, E, C_, shiftc)
    y = torch.empty_like(xd)
    ops.bn_apply(xd, resd, y, scale, shift, mean, rpe, E, C_, True)
    close(from_nhwc(y, C_), yr.detach(), dtype, "bn fwd")
    for e in range(E):
        close(drm[e], rmr[e], torch.float32, "running_mean")
        close(drv[e], rvr[e], torch.float32, "running_var")
    # backward
    bpart = torch.empty(E, nparts, 2, C_, device=DEV)
    ops.bn_bwd_reduce(dyd, y, xd, mean, invstd, scale, shift, rpe, E, C_, True, bpart, nparts)
    dgam, dbet, c1, c2 = (torch.empty(E, C_, device=DEV) for _ in range(4))
    ops.bn_bwd_finalize(bpart, nparts, rpe, dgam, dbet, c1, c2, E, C_)
    dx, gm = torch.empty_like(xd), torch.empty_like(xd)
    ops.bn_bwd_apply(dyd, y, xd, mean, invstd, scale, shift, c1, c2, dx, gm, rpe, E, C_, True)
    close(from_nhwc(dx, C_), xr.grad, dtype, "bn dx")
    close(from_nhwc(gm, C_), rr.grad, dtype, "bn residual grad")
    for e in range(E):
        close(dgam[e], gr[e].grad, dtype, "dgamma")
        close(dbet[e], br[e].grad, dtype, "dbeta")
    # the reduce pass can store the masked gradient itself; the apply pass then takes it as dy (relu off, saved output unread):
    # bit-identical partial sums, residual gradient and dx
    bpart2, gm2, dx2 = torch.empty_like(bpart), torch.empty_like(xd), torch.empty_like(xd)
    ops.bn_bwd_reduce(dyd, y, xd, mean, invstd, scale, shift, rpe, E, C_, True, bpart2, nparts, gmask=gm2)
    assert torch.equal(bpart2, bpart) and torch.equal(gm2, gm)
    ops.bn_bwd_apply(gm2, None, xd, mean, invstd, scale, shift, c1, c2, dx2, None, rpe, E, C_, False)
    assert torch.equal(dx2, dx)
    # no-residual ReLU: the mask is recomputed from x (y = None) and must equal the y-based mask
    y2 = torch.empty_like(xd)
    ops.bn_apply(xd, None, y2, scale, shift, mean, rpe, E, C_, True)
    pa, pb = torch.empty_like(bpart), torch.empty_like(bpart)
    ops.bn_bwd_reduce(dyd, y2, xd, mean, invstd, scale, shift, rpe, E, C_, True, pa, nparts)
    ops.bn_bwd_reduce(dyd, None, xd, mean, invstd, scale, shift, rpe, E, C_, True, pb, nparts)
    close(pb, pa.cpu(), torch.float32, "mask-from-x reduce")
    dxa, dxb = torch.empty_like(xd), torch.empty_like(xd)
    ops.bn_bwd_finalize(pa, nparts, rpe, dgam, dbet, c1, c2, E, C_)
    ops.bn_bwd_apply(dyd, y2, xd, mean, invstd, scale, shift, c1, c2, dxa, None, rpe, E, C_, True)
    ops.bn_bwd_apply(dyd, None, xd, mean, invstd, scale, shift, c1, c2, dxb, None, rpe, E, C_, True)
    close(dxb, dxa.float().cpu(), dtype, "mask-from-x apply")
    # eval mode: scale/shift from running buffers
    ops.bn_finalize(part2, 2, rpe, tabs[0], tabs[1], tabs[2], tabs[3], 0.1, 1e-5, False, scale, shift, mean, invstd, E, C_)
    ops.bn_apply(xd, None, y, scale, shift, mean, rpe, E, C_, False)
    yev = torch.cat([F.batch_norm(x[e * ipe:(e + 1) * ipe], rmr[e], rvr[e], gam[e], bet[e], False, 0.1, 1e-5)
                     for e in range(E)])
    close(from_nhwc(y, C_), yev, dtype, "bn eval")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw", [(16, 16), (13, 9)])
def test_maxpool(hw, dtype):
    g = torch.Generator().manual_seed(2)
    N, C_ = 3, 64
    H, W = hw
    x = torch.relu(rnd((N, C_, H, W), g, dtype))       # post-ReLU input as in the model: many exact ties at 0
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = rnd(tuple(yr.shape), g, dtype)
    yr.backward(dy)
    xd = nhwc(x, C_, dtype)
    Ho, Wo = yr.shape[-2:]
    y = torch.empty(N, Ho, Wo, C_, dtype=dtype, device=DEV)
    am = torch.empty(N, Ho, Wo, C_, dtype=torch.uint8, device=DEV)
    ops.maxpool_fwd(xd, y, am)
    close(from_nhwc(y, C_), yr.detach(), dtype, "maxpool fwd")
    dx = torch.empty_like(xd)
    ops.maxpool_bwd(nhwc(dy, C_, dtype), am, dx)
    # ties at zero may route differently; compare where the input is positive (the ReLU mask upstream kills the rest)
    got, ref = from_nhwc(dx, C_), xr.grad
    m = x > 0
    close(got * m, ref * m, dtype, "maxpool bwd")
    assert abs(got.sum().item() - ref.sum().item()) <= 1e-2 * ref.abs().sum().item()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(12, True), (64, False)])
def test_eca(cfg, dtype):
    creal, shared = cfg
    g = torch.Generator().manual_seed(4)
    E, ipe, H, W = 2, 3, 10, 6
    C_ = r16(creal)
    N = E * ipe
    k = 3
    nx = ipe if shared else N
    x = rnd((nx, creal, H, W), g, dtype)
    wk = [0.5 * torch.randn(1, 1, k, generator=g) for _ in range(E)]
    dy = rnd((N, creal, H, W), g, dtype)

    xr = x.clone().requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in wk]
    outs = []
    for e in range(E):
        xe = xr if shared else xr[e * ipe:(e + 1) * ipe]
        gmean = xe.mean(dim=(2, 3))
        s = torch.sigmoid(F.conv1d(gmean.unsqueeze(1), wr[e], padding=k // 2).squeeze(1))
        outs.append(xe * s[:, :, None, None])
    yr = torch.cat(outs)
    yr.backward(dy)

    xd = nhwc(x, C_, dtype)
    dwk = [w.to(DEV).contiguous() for w in wk]
    wtab = hip.ptr_table(dwk, DEV)
    nparts = 3
    part = torch.empty(nx, nparts, C_, device=DEV)
    ops.gap_partial(xd, None, part, nparts)
    gate = torch.empty(N, C_, device=DEV)
    gapmean = torch.empty(N, C_, device=DEV)
    ops.eca_gate(part, nparts, H * W, wtab, k, gate, gapmean, N, ipe, ipe if shared else 0, C_, creal)
    y = torch.empty(N, H, W, C_, dtype=dtype, device=DEV)
    ops.eca_scale(xd, gate, y, ipe if shared else 0)
    close(from_nhwc(y, creal), yr.detach(), dtype, "eca fwd")
    # backward
    dyd = nhwc(dy, C_, dtype)
    dot = torch.empty(N, nparts, C_, device=DEV)
    ops.gap_partial(dyd, xd, dot, nparts, ipe if shared else 0)
    dgap = torch.empty(N, C_, device=DEV)
    dw = torch.empty(E, k, device=DEV)
    ops.eca_bwd_small(dot, nparts, gate, gapmean, wtab, k, dgap, dw, N, ipe, C_, creal)
    for e in range(E):
        close(dw[e], wr[e].grad.flatten(), dtype, "eca dw")
    if not shared:
        dx = torch.empty_like(xd)
        ops.eca_bwd_apply(dyd, gate, dgap, dx)
        close(from_nhwc(dx, creal), xr.grad, dtype, "eca dx")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gap_and_layout(dtype):
    g = torch.Generator().manual_seed(6)
    N, C_, H, W = 4, 512, 4, 4
    x = rnd((N, C_, H, W), g, dtype)
    xd = nhwc(x, C_, dtype)
    part = torch.empty(N, 2, C_, device=DEV)
    ops.gap_partial(xd, None, part, 2)
    feat = torch.zeros(N, 1536, dtype=dtype, device=DEV)
    ops.gap_finish(part, feat, N, C_, 2, H * W, 1536, 0)
    close(feat[:, :512], x.mean(dim=(2, 3)), dtype, "gap fwd")
    dfe = rnd((N, 1536), g, dtype)
    dfd = dfe.to(dtype).to(DEV)
    dx = torch.empty_like(xd)
    ops.gap_bwd(dfd, dx, 1536, 0)
    close(from_nhwc(dx, C_), (dfe[:, :512] / (H * W))[:, :, None, None].expand(N, C_, H, W), dtype, "gap bwd")
    img = torch.rand(3, 12, 9, 7, generator=g)
    dst = torch.empty(3, 9, 7, 16, dtype=dtype, device=DEV)
    ops.nchw_to_nhwc(img.to(DEV), dst)
    close(from_nhwc(dst, 12), img.to(dtype).float(), dtype, "nchw->nhwc")
    assert dst[..., 12:].abs().max().item() == 0
    sp = torch.rand(5, 6, generator=g)
    pd = torch.empty(5, 16, dtype=dtype, device=DEV)
    ops.pad_rows(sp.to(DEV), pd)
    close(pd[:, :6], sp.to(dtype).float(), dtype, "pad_rows")
    assert pd[:, 6:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("E", [3, 4, 8])
@pytest.mark.parametrize("alpha_relu", [True, False])
def test_gate_mixture_and_loss(E, alpha_relu, dtype):
    import torch.distributions as D
    g = torch.Generator().manual_seed(E)
    B = 37
    head = rnd((E * B, 16), g, dtype)
    spd = rnd((E * B, 16), g, dtype)
    act = torch.rand(B, 2, generator=g) * 2 - 1
    tgt = torch.rand(B, 1, generator=g)
    hr = head.clone().requires_grad_(True)
    sr = spd.clone().requires_grad_(True)
    h3 = hr.view(E, B, 16).permute(1, 0, 2)
    alpha = torch.relu(h3[..., 4]) if alpha_relu else h3[..., 4]
    probs_r = torch.softmax(alpha, dim=1)
    mean_r = h3[..., 0:2]
    std_r = F.elu(h3[..., 2:4]) + 1
    speeds_r = sr.view(E, B, 16).permute(1, 0, 2)[..., 0:1]
    dist = D.MixtureSameFamily(D.Categorical(probs_r), D.Independent(D.Normal(mean_r, std_r), 1))
    nll = -dist.log_prob(act).mean()
    loss_r = 0.7 * nll + 0.3 * F.mse_loss(speeds_r, tgt.unsqueeze(1).expand_as(speeds_r)) / E
    loss_r.backward()

    hd, sd_ = head.to(dtype).to(DEV), spd.to(dtype).to(DEV)
    probs = torch.empty(B, E, device=DEV)
    mean = torch.empty(B, E, 2, device=DEV)
    std = torch.empty(B, E, 2, device=DEV)
    speeds = torch.empty(B, E, 1, device=DEV)
    ops.gate_mixture_fwd(hd, sd_, probs, mean, std, speeds, B, E, alpha_relu)
    close(probs, probs_r.detach(), torch.float32, "probs")
    close(std, std_r.detach(), torch.float32, "std")
    loss = torch.empty(1, device=DEV)
    ll = torch.empty(B, device=DEV)
    dp, dm, ds, dsp = torch.empty_like(probs), torch.empty_like(mean), torch.empty_like(std), torch.empty_like(speeds)
    ops.moe_loss(probs, mean, std, speeds, act.to(DEV), tgt.to(DEV), 0.7, 0.3, loss, ll, dp, dm, ds, dsp, B, E)
    assert abs(loss.item() - loss_r.item()) <= 2e-5 * max(1, abs(loss_r.item()))
    close(ll, dist.log_prob(act).detach(), torch.float32, "loglik")
    dhead = torch.empty_like(hd)
    dspd = torch.empty_like(sd_)
    ops.gate_mixture_bwd(hd, probs, dp, dm, ds, dsp, dhead, dspd, B, E, alpha_relu)
    tol_dt = torch.float32 if dtype == torch.float32 else dtype
    close(dhead, hr.grad, tol_dt, "dhead")
    close(dspd, sr.grad, tol_dt, "dspd")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K", [3, 4, 6, 12])
def test_gate_mixture_and_loss_shared_trunk(K, dtype):
    """MixtureOfExpertsShared head layout (moe.py:217-226) and the [B,1] speed branch of moe_loss (loss.py:129-130)."""
    import torch.distributions as D
    g = torch.Generator().manual_seed(K)
    B = 29
    ld = (5 * K + 15) // 16 * 16
    head = rnd((B, ld), g, dtype)
    spd = rnd((B, 16), g, dtype)
    act = torch.rand(B, 2, generator=g) * 2 - 1
    tgt = torch.rand(B, 1, generator=g)
    hr = head.clone().requires_grad_(True)
    sr = spd.clone().requires_grad_(True)
    mean_r, raw = hr[:, :4 * K].view(B, K, 4).split(2, dim=-1)
    std_r = F.elu(raw) + 1
    probs_r = torch.softmax(hr[:, 4 * K:5 * K], dim=1)
    speeds_r = sr[:, 0:1]
    dist = D.MixtureSameFamily(D.Categorical(probs_r), D.Independent(D.Normal(mean_r, std_r), 1))
    loss_r = 0.7 * -dist.log_prob(act).mean() + 0.3 * F.mse_loss(speeds_r, tgt)
    loss_r.backward()

    hd, sd_ = head.to(dtype).to(DEV), spd.to(dtype).to(DEV)
    probs = torch.empty(B, K, device=DEV)
    mean = torch.empty(B, K, 2, device=DEV)
    std = torch.empty(B, K, 2, device=DEV)
    speeds = torch.empty(B, 1, device=DEV)
    ops.gate_mixture_fwd(hd, sd_, probs, mean, std, speeds, B, K, False, True)
    close(probs, probs_r.detach(), torch.float32, "probs")
    close(mean, mean_r.detach(), torch.float32, "mean")
    close(std, std_r.detach(), torch.float32, "std")
    close(speeds, speeds_r.detach(), torch.float32, "speeds")
    loss = torch.empty(1, device=DEV)
    ll = torch.empty(B, device=DEV)
    dp, dm, ds, dsp = torch.empty_like(probs), torch.empty_like(mean), torch.empty_like(std), torch.empty_like(speeds)
    ops.moe_loss(probs, mean, std, speeds, act.to(DEV), tgt.to(DEV), 0.7, 0.3, loss, ll, dp, dm, ds, dsp, B, K, True)
    assert abs(loss.item() - loss_r.item()) <= 2e-5 * max(1, abs(loss_r.item()))
    dhead = torch.full_like(hd, 7.0)
    dspd = torch.full_like(sd_, 7.0)
    ops.gate_mixture_bwd(hd, probs, dp, dm, ds, dsp, dhead, dspd, B, K, False, True)
    tol_dt = torch.float32 if dtype == torch.float32 else dtype
    close(dhead, hr.grad, tol_dt, "dhead")      # padding columns 5K.. must come back as zeros
    close(dspd, sr.grad, tol_dt, "dspd")


def test_loss_golden_vectors():
    """moe_loss kernel against the values the imported reference produced (tests/golden/micro.pt)."""
    from pathlib import Path
    from pmoe_amd.loss import moe_loss
    import torch.distributions as D
    g = torch.load(Path(__file__).resolve().parent / "golden" / "micro.pt", weights_only=False)
    for key in ("loss_case", "loss_case_shared"):
        lc = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in g[key].items()}
        dist = D.MixtureSameFamily(D.Categorical(lc["probs"]), D.Independent(D.Normal(lc["mean"], lc["std"]), 1))
        got = moe_loss(dist, lc["speeds"], lc["act"], lc["tgt"], [0.7, 0.3])
        assert abs(got.item() - lc["loss"].item()) <= 1e-5 * max(1.0, abs(lc["loss"].item())), key


@pytest.mark.parametrize("dtype", DTYPES)
def test_unet_pool_upconv_concat_kernels(dtype):
    """MaxPool2d(2), ConvTranspose2d(k2,s2) = 1x1 GEMM + pixel shuffle into a concat window, channel-window copies."""
    g = torch.Generator().manual_seed(11)
    n, h, w_, cin, cout = 2, 6, 10, 32, 16
    x = rnd((n, cin, h, w_), g, dtype)
    # maxpool
    xh = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)
    y = torch.empty(n, h // 2, w_ // 2, cin, dtype=dtype, device=DEV)
    ops.maxpool2_fwd(xh, y)
    close(y.permute(0, 3, 1, 2), F.max_pool2d(x, 2, 2), torch.float32, "maxpool2")
    # transposed conv: weight [cin, cout, 2, 2]
    wt = rnd((cin, cout, 2, 2), g, dtype) * 0.25          # (a power of two: the weights stay representable in bf16)
    bias = rnd((cout,), g, torch.float32)
    ref = F.conv_transpose2d(x, wt, bias, stride=2)
    w1 = wt.permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1).contiguous().to(DEV)
    coutp = 64
    wf = torch.empty(1, coutp, 1, cin, dtype=dtype, device=DEV)
    ops.pack_conv_weights(hip.ptr_table([w1], DEV), wf, None, 1, 4 * cout, cin, 1, coutp, cin, 64, 64, dtype)
    bp = torch.empty(1, coutp, device=DEV)
    ops.pack_bias(hip.ptr_table([bias.repeat(4).contiguous().to(DEV)], DEV), bp, 1, 4 * cout, coutp)
    t = torch.empty(n, h, w_, 4 * cout, dtype=dtype, device=DEV)
    ops.conv2d(xh, wf, t, cin=cin, cout=4 * cout, coutp=coutp, ipe=n, ks=1, stride=1, pad=0, bias=bp)
    skip = rnd((n, 2 * h, 2 * w_, cout), g, dtype).to(dtype).to(DEV)
    cat = torch.zeros(n, 2 * h, 2 * w_, 2 * cout, dtype=dtype, device=DEV)
    ops.copy_window(skip, 0, cat, 0, cout)
    ops.pixel_shuffle2(t, cat, cout, dst_coff=cout)
    assert torch.equal(cat[..., :cout], skip)
    close(cat[..., cout:].permute(0, 3, 1, 2), ref, dtype, "conv_transpose2d")
    # unaligned channel windows (23-class masks packed 4 x 23 -> 96)
    m = rnd((n, h, w_, 32), g, dtype).to(dtype).to(DEV)
    packed = torch.zeros(n, h, w_, 96, dtype=dtype, device=DEV)
    for k in range(4):
        ops.copy_window(m, 0, packed, 23 * k, 23)
    for k in range(4):
        assert torch.equal(packed[..., 23 * k:23 * k + 23], m[..., :23])
    assert packed[..., 92:].abs().max() == 0


def test_action_head_loss_and_blend_kernels():
    """tanh action head, punet_loss / pmoe_loss (golden values of the imported reference) and the PMoE blend."""
    from pathlib import Path
    from pmoe_amd.loss import pmoe_loss, punet_loss
    g = torch.Generator().manual_seed(5)
    B = 19
    head = torch.randn(B, 16, generator=g)
    spd = torch.randn(B, 16, generator=g)
    act = torch.rand(B, 2, generator=g) * 2 - 1
    tgt = torch.rand(B, 1, generator=g)
    hr, sr = head.clone().requires_grad_(True), spd.clone().requires_grad_(True)
    a_ref, s_ref = torch.tanh(hr[:, :2]), sr[:, :1]
    loss_ref = 0.7 * F.l1_loss(a_ref, act) + 0.3 * F.mse_loss(s_ref, tgt)
    loss_ref.backward()
    a = torch.empty(B, 2, device=DEV)
    s = torch.empty(B, 1, device=DEV)
    ops.action_head_fwd(head.to(DEV), spd.to(DEV), a, s, B)
    close(a, a_ref.detach(), torch.float32, "actions")
    a.requires_grad_(True); s.requires_grad_(True)
    loss = punet_loss(a, s, act.to(DEV), tgt.to(DEV), [0.7, 0.3])
    assert abs(loss.item() - loss_ref.item()) < 1e-6
    loss.backward()
    dhead, dspd = torch.full((B, 16), 3.0, device=DEV), torch.full((B, 16), 3.0, device=DEV)
    ops.action_head_bwd(a.detach(), a.grad, s.grad, dhead, dspd, B)
    close(dhead, hr.grad, torch.float32, "dhead")
    close(dspd, sr.grad, torch.float32, "dspd")
    mic = torch.load(Path(__file__).resolve().parent / "golden" / "micro.pt", weights_only=False)["action_loss_case"]
    c = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in mic.items()}
    assert abs(punet_loss(c["actions"], c["speeds"], c["act"], c["tgt"], [0.7, 0.3]).item() - mic["punet_loss"].item()) < 1e-6
    assert abs(pmoe_loss(c["actions"], -1, c["act"], c["tgt"], [0.7, 0.3]).item() - mic["pmoe_loss"].item()) < 1e-6
    # blend
    from pmoe_amd.model.moe import _BlendFn
    moe_a, pu_a = torch.randn(B, 2, generator=g), torch.tanh(torch.randn(B, 2, generator=g))
    lw, lb = torch.randn(1, 2, generator=g), torch.randn(1, generator=g)
    gw, gb = torch.randn(1, 2, generator=g), torch.randn(1, generator=g)
    ps = [t.clone().requires_grad_(True) for t in (pu_a, lw, lb, gw, gb)]
    lat = F.linear(torch.cat([moe_a[:, 0:1], ps[0][:, 0:1]], -1), ps[1], ps[2])
    lon = F.linear(torch.cat([moe_a[:, 1:], ps[0][:, 1:]], -1), ps[3], ps[4])
    out_ref = torch.tanh(torch.cat([lat, lon], -1))
    F.l1_loss(out_ref, act).backward()
    pd = [t.detach().to(DEV).requires_grad_(True) for t in (pu_a, lw, lb, gw, gb)]
    out = _BlendFn.apply(moe_a.to(DEV), *pd)
    close(out, out_ref.detach(), torch.float32, "blend")
    F.l1_loss(out, act.to(DEV)).backward()
    for got, ref, nm in zip(pd, ps, ("dpunet", "dlat_w", "dlat_b", "dlong_w", "dlong_b")):
        close(got.grad, ref.grad, torch.float32, nm)


@pytest.mark.parametrize("dtype", DTYPES)
def test_folded_weight_packs(dtype):
    """pmoe_pack_conv_weights_scaled (eval-mode BatchNorm folded into the conv, with residual add + ReLU in the epilogue)
    and pmoe_pack_conv_weights_gated (ECA gate folded into per-image weights) against the unfused torch chains."""
    g = torch.Generator().manual_seed(21)
    E, ipe, cin, cout, H, W = 2, 3, 64, 64, 10, 12
    N = E * ipe
    x = rnd((N, cin, H, W), g, dtype)
    res = rnd((N, cout, H, W), g, dtype)
    ws = [rnd((cout, cin, 3, 3), g, dtype, (2.0 / (cin * 9)) ** 0.5) for _ in range(E)]
    gamma, beta = torch.rand(E, cout, generator=g) + 0.5, torch.randn(E, cout, generator=g) * 0.3
    rm, rv = torch.randn(E, cout, generator=g) * 0.2, torch.rand(E, cout, generator=g) + 0.5
    eps = 1e-5
    ref = torch.cat([torch.relu(F.batch_norm(F.conv2d(x[e * ipe:(e + 1) * ipe], ws[e], padding=1), rm[e].clone(), rv[e].clone(),
                                             gamma[e], beta[e], False, 0.1, eps) + res[e * ipe:(e + 1) * ipe])
                     for e in range(E)])
    dev_ws = [w.to(DEV).contiguous() for w in ws]
    tab = hip.ptr_table(dev_ws, DEV)
    scale = (gamma / torch.sqrt(rv + eps)).to(DEV).contiguous()
    wf = torch.empty(E, 64, 9, cin, dtype=dtype, device=DEV)
    bf = torch.empty(E, 64, device=DEV)
    ops.pack_conv_weights_scaled(tab, scale, beta.to(DEV).contiguous(), rm.to(DEV).contiguous(), wf, bf, E, cout, cin, 3, 64, cin, dtype)
    xd, rd = nhwc(x, cin, dtype), nhwc(res, cout, dtype)
    y = torch.empty(N, H, W, cout, dtype=dtype, device=DEV)
    ops.conv2d(xd, wf, y, cin=cin, cout=cout, coutp=64, ipe=ipe, ks=3, stride=1, pad=1, bias=bf, act=hip.ACT_RELU,
               res=rd, res_mode=hip.RES_ADD)
    close(from_nhwc(y, cout), ref, dtype, "conv with folded BatchNorm + residual + ReLU", floor=0.25)
    # gate fold: conv(x * g[n, c], W[e]) == conv(x, W[e] * g[n, c]) with one weight pack per image
    gate = torch.rand(N, cin, generator=g)
    ref2 = torch.cat([F.conv2d(x[n:n + 1] * gate[n].view(1, -1, 1, 1), ws[n // ipe], padding=1) for n in range(N)])
    wg = torch.empty(N, 64, 9, cin, dtype=dtype, device=DEV)
    wd = torch.empty(N, 64, 9, cout, dtype=dtype, device=DEV)
    ops.pack_conv_weights_gated(tab, gate.to(DEV).contiguous(), wg, wd, N, ipe, cout, cin, 3, 64, cin, 64, cout, dtype)
    y2 = torch.empty(N, H, W, cout, dtype=dtype, device=DEV)
    ops.conv2d(xd, wg, y2, cin=cin, cout=cout, coutp=64, ipe=1, ks=3, stride=1, pad=1)
    close(from_nhwc(y2, cout), ref2, dtype, "conv with per-image gate-folded weights", floor=0.25)
    # and its data gradient (flipped pack) against autograd through the gated input
    xr = x.clone().requires_grad_(True)
    dy = rnd((N, cout, H, W), g, dtype)
    torch.cat([F.conv2d(xr[n:n + 1] * gate[n].view(1, -1, 1, 1), ws[n // ipe], padding=1) for n in range(N)]).backward(dy)
    dx = torch.empty(N, H, W, cin, dtype=dtype, device=DEV)
    ops.conv2d(nhwc(dy, cout, dtype), wd, dx, cin=cout, cout=cin, coutp=64, ipe=1, ks=3, stride=1, pad=1)
    close(from_nhwc(dx, cin), xr.grad, dtype, "gate-folded data gradient", floor=0.25)


def test_batchnorm_statistics_are_centred():
    """|mean| >> std (a near-constant channel): one-pass E[x^2]-mean^2 in f32 loses the variance; the centred sums
    (deviations from a sample of the channel) must reproduce the float64 statistics."""
    g = torch.Generator().manual_seed(0)
    E, ipe, C_, H, W = 2, 4, 64, 16, 16
    x = torch.randn(E * ipe, C_, H, W, generator=g)
    x[:, 0] = 100.0 + 1e-3 * x[:, 0]                 # mean^2 / var = 1e10
    x[:, 1] = -7.0 + 1e-2 * x[:, 1]
    xd = nhwc(x, C_, torch.float32)
    rpe = ipe * H * W
    part = torch.empty(E, 8, 2, C_, device=DEV)
    shiftc = torch.empty(E, C_, device=DEV)
    ops.colstats(rpe, xd, E, C_, part, 8, shiftc=shiftc)
    scale, shift, mean, invstd = (torch.empty(E, C_, device=DEV) for _ in range(4))
    ops.bn_finalize(part, 8, rpe, None, None, None, None, 0.1, 1e-5, True, scale, shift, mean, invstd, E, C_, shiftc)
    xe = x.double().view(E, ipe, C_, H * W).permute(0, 2, 1, 3).reshape(E, C_, -1)
    ref_is = 1.0 / torch.sqrt(xe.var(dim=2, unbiased=False) + 1e-5)
    assert ((invstd.cpu().double() - ref_is).abs() / ref_is).max().item() < 1e-5
    assert ((mean.cpu().double() - xe.mean(dim=2)).abs().max().item()) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 3, 75, 53), (1, 2, 64, 96), (3, 1, 17, 8)])
def test_stem_tail_row_walking_kernels_match_gather_kernels(dtype, shape, monkeypatch):
    """stem_tail_pool_walk_kernel / stem_tail_dz_walk_kernel (each z2 element loaded and evaluated once, taps as max-keys / SWAR
    bytes) against the per-output gather kernels they replace (PMOE_STEM_WALK=0, themselves checked against the unfused chain in
    test_model_gpu): pooled output and winning taps bit-identical, dz2 within one rounding of the output type; odd sizes, image
    and expert boundaries."""
    E, B, H, W = shape
    C_, N = 64, E * B
    g = torch.Generator().manual_seed(11)
    z2 = rnd((N, H, W, C_), g, dtype).to(dtype).to(DEV)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dp = rnd((N, Ho, Wo, C_), g, dtype).to(dtype).to(DEV)
    u = lambda lo, hi: (torch.rand(E, C_, generator=g) * (hi - lo) + lo).to(DEV)
    sc2, sh2, sc1, sh1, mu1, is1, mu2, is2 = u(.5, 1.5), u(-.3, .3), u(.5, 1.5), u(-.3, .3), u(.2, .5), u(.8, 1.2), u(-.1, .1), u(.8, 1.2)
    consts = [sc2, sh2, sc1, sh1, mu1, is1, mu2, is2, u(-.01, .01), u(-.01, .01), u(-.01, .01), u(-.01, .01)]
    part = torch.empty(E, 4, 2, C_, device=DEV)
    out = {}
    for walk in ("0", "1"):
        monkeypatch.setenv("PMOE_STEM_WALK", walk)
        y = torch.full((N, Ho, Wo, C_), 7.0, dtype=dtype, device=DEV)
        am = torch.full((N, Ho, Wo, C_), 99, dtype=torch.uint8, device=DEV)
        dz = torch.full_like(z2, 7.0)
        ops.stem_tail_pool(z2, y, am, sc2, sh2, sc1, sh1, mu2, mu1, B)
        ops.stem_tail_bwd(3, z2, dp, am, dz, consts, part, 4, E, B)
        out[walk] = (y, am, dz)
    assert torch.equal(out["1"][0], out["0"][0]) and torch.equal(out["1"][1], out["0"][1])
    assert int(out["0"][1].max()) <= 0x88
    a, b = out["1"][2].float(), out["0"][2].float()
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -21
    assert ((a - b).abs() <= ulp * b.abs().clamp_min(1e-3)).all()


def test_conv_c16_shared_input_and_stats_shape_guard():
    """conv3x3_c16_kernel with the input shared by the experts (the stem's first convolution reads the ONE batch of frames for
    every expert, model/moe.py:90-92) against F.conv2d per expert, fused BatchNorm sums included; a statistics buffer sized for
    another descriptor is refused instead of being folded with unwritten rows."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    E, ipe, cin, cout, H, W = 3, 2, 12, 64, 40, 72
    x = rnd((ipe, cin, H, W), g, dtype)
    ws = [rnd((cout, cin, 3, 3), g, dtype, 0.2) for _ in range(E)]
    wf, _, _ = pack(ws, 3, dtype)
    xd = nhwc(x, 16, dtype)
    out = torch.full((E * ipe, H, W, cout), 7.0, dtype=dtype, device=DEV)
    kw = dict(cin=16, cout=cout, coutp=64, ipe=ipe, ks=3, stride=1, pad=1, in_shared=True)
    assert ops.conv2d(xd, wf, out, plan_only=True, **kw) == 1316
    rows = ops.conv2d_stat_rows(E * ipe, H, W, H, W, 16, cout, 64, ipe, 3, 1, 1, dtype, in_ld=16, out_ld=cout, in_shared=True)
    stats = torch.zeros(rows, 2, 64, device=DEV)
    ops.conv2d(xd, wf, out, stats=stats, **kw)
    ref = torch.cat([F.conv2d(x, w, padding=1) for w in ws])
    close(from_nhwc(out, cout), ref, dtype, "c16 shared input")
    st = stats.view(E, rows // E, 2, 64).sum(1).cpu()
    yo = out.float().cpu().reshape(E, -1, cout)
    close(st[:, 0], yo.sum(1), dtype, "c16 stats sum")
    close(st[:, 1], (yo * yo).sum(1), dtype, "c16 stats sumsq")
    with pytest.raises(ValueError, match="stats must be"):
        ops.conv2d(xd, wf, out, stats=torch.zeros(rows + 3, 2, 64, device=DEV), **kw)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_apply_with_fused_global_average_pool(dtype):
    """pmoe_bn_apply_gap = pmoe_bn_apply followed by pmoe_gap_partial over the stored activation, bit for bit (activation and
    partial sums), for a partition of the image that does not divide evenly and a channel count that leaves idle lanes."""
    g = torch.Generator().manual_seed(9)
    E, ipe, H, W = 2, 3, 23, 31
    for C_ in (64, 48):
        x = rnd((E * ipe, H, W, C_), g, dtype).to(dtype).to(DEV)
        scale, shift, mean = (torch.rand(E, C_, generator=g).to(DEV) + 0.5 for _ in range(3))
        for nparts in (1, 5):
            y1, y2 = torch.empty_like(x), torch.full_like(x, 3.0)
            p1 = torch.zeros(E * ipe, nparts, C_, device=DEV)
            p2 = torch.full_like(p1, -1.0)
            ops.bn_apply(x, None, y1, scale, shift, mean, ipe * H * W, E, C_, True) if C_ == 64 else None
            if C_ != 64:      # bn_apply keeps one channel vector per thread (power-of-two vector counts): reference in torch
                xf = x.float().view(E, ipe, H, W, C_)
                y1 = torch.relu((xf - mean.view(E, 1, 1, 1, C_)) * scale.view(E, 1, 1, 1, C_) + shift.view(E, 1, 1, 1, C_)).to(dtype).view_as(x)
            ops.gap_partial(y1, None, p1, nparts)
            ops.bn_apply_gap(x, y2, scale, shift, mean, p2, nparts, ipe, True)
            if C_ == 64:
                assert torch.equal(y1, y2) and torch.equal(p1, p2)
            else:
                close(y2, y1.float().cpu(), dtype, "bn_apply_gap y")
                # (sums of bf16 activations, a few of which round the other way between the torch expression and the kernel's FMA)
                close(p2, p1.cpu(), dtype, "bn_apply_gap sums")


def test_conv_c16_channel_windows():
    """conv3x3_c16_kernel reading a 16-channel window of wider rows and writing a 64-channel window of wider rows (buffer
    descriptors start at the window, row strides are the full row lengths): bit-identical to the dense launch, nothing written
    outside the window."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(6)
    E, ipe, cout, H, W = 2, 2, 64, 24, 72
    x = rnd((E * ipe, 16, H, W), g, dtype)
    ws = [rnd((cout, 16, 3, 3), g, dtype, 0.2) for _ in range(E)]
    wf, _, _ = pack(ws, 3, dtype)
    xd = nhwc(x, 16, dtype)
    dense = torch.empty(E * ipe, H, W, cout, dtype=dtype, device=DEV)
    kw = dict(cin=16, cout=cout, coutp=64, ipe=ipe, ks=3, stride=1, pad=1)
    assert ops.conv2d(xd, wf, dense, plan_only=True, **kw) == 1316
    ops.conv2d(xd, wf, dense, **kw)
    xw = torch.full((E * ipe, H, W, 48), 5.0, dtype=dtype, device=DEV)
    xw[..., 16:32] = xd
    ow = torch.full((E * ipe, H, W, 160), 7.0, dtype=dtype, device=DEV)
    assert ops.conv2d(xw, wf, ow, in_coff=16, out_coff=64, plan_only=True, **kw) == 1316
    ops.conv2d(xw, wf, ow, in_coff=16, out_coff=64, **kw)
    assert torch.equal(ow[..., 64:128], dense)
    assert (ow[..., :64] == 7.0).all() and (ow[..., 128:] == 7.0).all()


@pytest.mark.parametrize("case", [
    # E, ipe, cin, cout, H, W, stride, bias, expected plan
    (2, 3, 64, 120, 130, 130, 2, True, 1404),      # ragged: 65x65 outputs (last tile partial), 120 of 128 channels, bias
    (1, 2, 128, 256, 96, 96, 1, False, 1404),      # two 128-channel slabs
    (2, 2, 256, 512, 72, 72, 1, True, 1402),       # 64-channel slabs (8 of them), two k chunks
    (1, 2, 192, 64, 160, 160, 2, False, 1402),     # K not a multiple of the 128-channel chunk
])
def test_conv1x1_direct_kernel(case):
    """conv1x1_direct_kernel (downsample projections, transposed-convolution GEMMs) against F.conv2d per expert: outputs,
    bias added before the rounding, fused BatchNorm partial sums, untouched channels beyond cout."""
    E, ipe, cin, cout, H, W, stride, with_bias, code = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(17)
    N = E * ipe
    x = rnd((N, cin, H, W), g, dtype)
    ws = [rnd((cout, cin, 1, 1), g, dtype, 0.1) for _ in range(E)]
    bs = [rnd((cout,), g, torch.float32, 0.5) for _ in range(E)]
    coutp = (cout + 127) // 128 * 128 if code == 1404 else r64(cout)
    wf = torch.zeros(E, coutp, 1, cin, dtype=dtype, device=DEV)
    for e in range(E):
        wf[e, :cout, 0] = ws[e][:, :, 0, 0].to(dtype).to(DEV)
    bias = None
    if with_bias:
        bias = torch.zeros(E, coutp, device=DEV)
        for e in range(E):
            bias[e, :cout] = bs[e].to(DEV)
    xd = nhwc(x, cin, dtype)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.full((N, Ho, Wo, coutp), 7.0, dtype=dtype, device=DEV)
    kw = dict(cin=cin, cout=cout, coutp=coutp, ipe=ipe, ks=1, stride=stride, pad=0, bias=bias)
    assert ops.conv2d(xd, wf, out, plan_only=True, **kw) == code
    rows = ops.conv2d_stat_rows(N, H, W, Ho, Wo, cin, cout, coutp, ipe, 1, stride, 0, dtype, in_ld=cin, out_ld=coutp)
    stats = torch.zeros(rows, 2, coutp, device=DEV)
    ops.conv2d(xd, wf, out, stats=stats, **kw)
    ref = torch.cat([F.conv2d(x[e * ipe:(e + 1) * ipe], ws[e], bs[e] if with_bias else None, stride=stride) for e in range(E)])
    close(from_nhwc(out, cout), ref, dtype, "1x1 direct")
    if coutp > cout:
        assert (out[..., cout:] == 7.0).all()
    st = stats.view(E, rows // E, 2, coutp).sum(1).cpu()
    yo = out.float().cpu()[..., :cout].reshape(E, -1, cout)
    close(st[:, 0, :cout], yo.sum(1), dtype, "1x1 direct stats sum")
    close(st[:, 1, :cout], (yo * yo).sum(1), dtype, "1x1 direct stats sumsq")


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the stem's first BatchNorm backward applied ON LOAD by conv1's per-image filter gradient (conv_wgrad_bnbwd_kernel,
# include/pmoe_hip.h pmoe_wgrad_desc.bn_fused): dz = g*A + ((z - mean)*Bx + K) is evaluated between the global loads and the LDS
# tile, the gradient tensor dz is never written.
@pytest.mark.parametrize("E,ipb,H,W", [(2, 2, 64, 64), (1, 3, 40, 72), (2, 2, 52, 20), (2, 1, 256, 256)])
def test_wgrad_with_batchnorm_backward_on_load(E, ipb, H, W):
    """against (a) pmoe_bn_bwd_apply followed by the plain per-image filter gradient -- the operand the MFMAs see is the same
    bf16 tensor, so only the f32 summation order differs -- and (b) the CPU f32 statement of both steps."""
    g = torch.Generator().manual_seed(E * 100 + H + W)
    BF = torch.bfloat16
    N, C, CI = E * ipb, 64, 16
    x = torch.rand((ipb, 12, H, W), generator=g).to(BF).float()                    # the frames, shared by the experts
    gy = rnd((N, C, H, W), g, BF)
    gy = torch.where(torch.rand(gy.shape, generator=g) < 0.5, gy, torch.zeros(()))     # ReLU-masked, like PMOE_RES_DBN leaves it
    z = rnd((N, C, H, W), g, BF, 2.0) + 0.3
    coef = torch.empty(4, E, C)
    coef[0] = torch.randn(E, C, generator=g) * 0.5 + 0.3
    coef[1] = torch.rand(E, C, generator=g) + 0.5
    coef[2] = coef[1] * (torch.randn(E, C, generator=g) * 0.5 + 1.0)
    coef[3] = torch.randn(E, C, generator=g) * 0.3
    c1, c2 = torch.randn(E, C, generator=g) * 0.1, torch.randn(E, C, generator=g) * 0.1
    xd, gd, zd = nhwc(x, CI, BF), nhwc(gy, C, BF), nhwc(z, C, BF)
    coefd, c1d, c2d = coef.to(DEV), c1.to(DEV), c2.to(DEV)
    kw = dict(cin=CI, cout=C, cinp=64, coutp=64, ipe=ipb, ks=3, stride=1, pad=1, x_shared=True, per_image=True)
    assert ops.conv2d_wgrad(xd, gd, None, bn_fuse=(zd, None, None, None), plan_only=True, **kw) == 7209
    G = torch.full((N, 9, 64, 64), 7.0, device=DEV)
    ops.conv2d_wgrad(xd, gd, G, bn_fuse=(zd, coefd, c1d, c2d), **kw)
    # (a) the unfused pair
    dz = torch.empty_like(gd)
    ops.bn_bwd_apply(gd, None, zd, coefd[0], coefd[1], coefd[2], coefd[3], c1d, c2d, dz, None, ipb * H * W, E, C, False)
    G2 = torch.full((N, 9, 64, 64), 3.0, device=DEV)
    ops.conv2d_wgrad(xd, dz, G2, **kw)
    scale = G2.abs().max().item()
    per_img = [((G[n] - G2[n]).abs().max().item() / scale) for n in range(N)]
    per_tap = [((G[:, t] - G2[:, t]).abs().max().item() / scale) for t in range(9)]
    assert max(per_img) <= 2e-5, ("fused vs unfused, per image / per tap", per_img, per_tap)
    assert G[:, :, :, 16:].abs().max().item() == 0.0                               # columns past the 16 channels: written as zeros
    # (b) CPU f32: dz from the formula, then the per-image filter gradient by autograd
    cb = coef.repeat_interleave(ipb, dim=1).view(4, N, C, 1, 1)
    A = cb[2]
    dzr = gy * A + ((z - cb[0]) * (-A * cb[1] * c2.repeat_interleave(ipb, 0).view(N, C, 1, 1))
                    + (-A * c1.repeat_interleave(ipb, 0).view(N, C, 1, 1)))
    dzr = dzr.to(BF).float()
    for n in range(N):
        w = torch.zeros(C, 12, 3, 3, requires_grad=True)
        F.conv2d(x[n % ipb:n % ipb + 1], w, padding=1).backward(dzr[n:n + 1])
        got = G[n].cpu()[:, :C, :12].permute(1, 2, 0).reshape(C, 12, 3, 3)          # [tap][cout][cin] -> [cout][cin][kh][kw]
        close(got, w.grad, BF, f"fused per-image filter gradient, image {n}", floor=0.25)     # (dz is a re-rounded derived operand)


@pytest.mark.parametrize("E,ipe,K,Kr,Nn,Nr,shared,xcoff,ycoff", [
    (3, 64, 1536, 1536, 512, 512, False, 0, 0),          # speed_prediction.0 / action_head.0 (moe.py:62-66)
    (2, 100, 16, 1, 512, 512, True, 0, 0),               # speed_encoder.0: one real input column, input rows shared by the experts
    (2, 7, 512, 512, 16, 5, False, 0, 0),                # the fused 5-row head (action_pred + alpha), ragged batch
    (4, 64, 512, 512, 512, 512, False, 512, 1024),       # channel windows: input slot of the 1536-d feature, output slot
    (1, 130, 96, 90, 80, 72, False, 0, 0),               # ragged tiles, three row chunks
])
def test_mlp_wgrad_one_launch(E, ipe, K, Kr, Nn, Nr, shared, xcoff, ycoff):
    """pmoe_mlp_wgrad (round 4): weight + bias gradient of a Linear layer of all experts in one launch, parameter layout,
    against dY^T X / column sums in f32 on the CPU (inputs pre-rounded to bf16: only the summation order differs)."""
    g = torch.Generator().manual_seed(E + ipe + K + Nn)
    BF = torch.bfloat16
    xld, yld = max(K + xcoff, 1536 if xcoff else K), max(Nn + ycoff, 1536 if ycoff else Nn)
    nx = ipe if shared else E * ipe
    x = torch.zeros(nx, xld)
    x[:, xcoff:xcoff + Kr] = rnd((nx, Kr), g, BF)
    dy = torch.zeros(E * ipe, yld)
    dy[:, ycoff:ycoff + Nr] = rnd((E * ipe, Nr), g, BF)
    if xcoff:
        x[:, :xcoff] = 9.0                              # neighbouring slots must not leak in
    if ycoff:
        dy[:, :ycoff] = 9.0
    xd = x.to(BF).to(DEV).view(nx, 1, 1, xld)
    dyd = dy.to(BF).to(DEV).view(E * ipe, 1, 1, yld)
    grads = torch.full((E, Nr, Kr), 7.0, device=DEV)
    bg = torch.full((E, Nr), 7.0, device=DEV)
    ops.mlp_wgrad(xd, dyd, grads, bg, cin=K, cout=Nn, cin_real=Kr, cout_real=Nr, ipe=ipe, x_shared=shared, x_coff=xcoff, dy_coff=ycoff)
    g2 = torch.full((E, Nr, Kr), 7.0, device=DEV)
    ops.mlp_wgrad(xd, dyd, g2, None, cin=K, cout=Nn, cin_real=Kr, cout_real=Nr, ipe=ipe, x_shared=shared, x_coff=xcoff, dy_coff=ycoff)
    assert torch.equal(grads, g2)                       # deterministic, and the bias output is optional
    for e in range(E):
        xe = x[(0 if shared else e * ipe):(0 if shared else e * ipe) + ipe, xcoff:xcoff + Kr]
        de = dy[e * ipe:(e + 1) * ipe, ycoff:ycoff + Nr]
        close(grads[e], de.t() @ xe, BF, f"mlp wgrad e{e}")
        close(bg[e], de.sum(0), BF, f"mlp bias grad e{e}")


@pytest.mark.parametrize("E,ipe,cin,c_up,H,W", [(1, 4, 128, 64, 64, 64), (2, 8, 512, 256, 32, 32), (1, 2, 256, 128, 64, 128)])
def test_upconv_with_fused_pixel_shuffle(E, ipe, cin, c_up, H, W):
    """pmoe_conv_desc.shuffle_c (round 4): ConvTranspose2d(k=2, s=2) (blocks/unet.py:28-45) as ONE launch -- the 1x1 direct kernel
    scatters its 4*c_up output channels to the 2x2 block itself.  Bit-identical to the 1x1 launch + pmoe_pixel_shuffle2 pair it
    replaces, the other half of the concatenation buffer untouched, and equal to F.conv_transpose2d on the CPU."""
    g = torch.Generator().manual_seed(cin + c_up + H)
    BF = torch.bfloat16
    N = E * ipe
    x = rnd((N, cin, H, W), g, BF)
    wt = [rnd((cin, c_up, 2, 2), g, BF, (1.0 / cin) ** 0.5) for _ in range(E)]            # ConvTranspose2d weight layout
    bias = [torch.randn(c_up, generator=g) * 0.1 for _ in range(E)]
    ws = [w.permute(2, 3, 1, 0).reshape(4 * c_up, cin, 1, 1).contiguous() for w in wt]       # rows (dy, dx, c)
    wf, _, _keep = pack(ws, 1, BF)
    bp = torch.stack([b.repeat(4) for b in bias]).to(DEV).contiguous()
    xd = nhwc(x, cin, BF)
    kw = dict(cin=cin, cout=4 * c_up, coutp=4 * c_up, ipe=ipe, ks=1, stride=1, pad=0, bias=bp)
    cat_a = torch.full((N, 2 * H, 2 * W, 2 * c_up), 3.0, dtype=BF, device=DEV)
    assert ops.conv2d(xd, wf, cat_a, plan_only=True, out_coff=c_up, shuffle2_c=c_up, **kw) in (1452, 1454)
    ops.conv2d(xd, wf, cat_a, out_coff=c_up, shuffle2_c=c_up, **kw)
    t = torch.empty(N, H, W, 4 * c_up, dtype=BF, device=DEV)
    ops.conv2d(xd, wf, t, **kw)
    cat_b = torch.full((N, 2 * H, 2 * W, 2 * c_up), 3.0, dtype=BF, device=DEV)
    ops.pixel_shuffle2(t, cat_b, c_up, dst_coff=c_up)
    assert torch.equal(cat_a, cat_b)
    assert (cat_a[..., :c_up] == 3.0).all()
    for e in range(E):
        ref = F.conv_transpose2d(x[e * ipe:(e + 1) * ipe], wt[e], bias[e], stride=2)
        close(from_nhwc(cat_a[e * ipe:(e + 1) * ipe, :, :, c_up:], c_up), ref, BF, f"fused transposed conv e{e}")


@pytest.mark.parametrize("E,ipe,cin,cout,H,W,shuf", [(1, 4, 64, 24, 64, 64, 0), (2, 3, 128, 64, 64, 64, 64), (1, 2, 256, 128, 64, 128, 128),
                                                     (1, 5, 64, 23 + 1, 40, 56, 0)])
def test_conv1x1_applies_input_batchnorm_relu_on_load(E, ipe, cin, cout, H, W, shuf):
    """PMOE_RES_INBN on conv1x1_direct_kernel (the U-Nets' final classifier and ConvTranspose2d layers behind a frozen train-mode
    block): relu(BatchNorm(z)) evaluated in registers on the B operand.  Bit-identical to pmoe_bn_apply + the plain launch, with a
    bias, with and without the fused 2x2 scatter, several BatchNorm parameter sets per launch (E > 1), ragged pixel counts."""
    g = torch.Generator().manual_seed(cin + cout + H + shuf)
    BF = torch.bfloat16
    N = E * ipe
    z = nhwc(rnd((N, cin, H, W), g, BF, 2.0) + 0.3, cin, BF)
    co = 4 * shuf if shuf else cout
    ws = [rnd((co, cin, 1, 1), g, BF, (1.0 / cin) ** 0.5) for _ in range(E)]
    wf, _, _keep = pack(ws, 1, BF)
    bp = (torch.randn(E, r64(co), generator=g) * 0.1).to(DEV).contiguous()
    coef = torch.empty(4, E, cin, device=DEV)
    coef[0] = rnd((E, cin), g, torch.float32, 0.5).to(DEV) + 0.3
    coef[1] = 1.0
    coef[2] = (rnd((E, cin), g, torch.float32, 0.3).abs() + 0.5).to(DEV)
    coef[3] = rnd((E, cin), g, torch.float32, 0.4).to(DEV) + 0.2
    kw = dict(cin=cin, cout=co, coutp=r64(co), ipe=ipe, ks=1, stride=1, pad=0, bias=bp)
    a_ = torch.empty_like(z)
    ops.bn_apply(z, None, a_, coef[2], coef[3], coef[0], ipe * H * W, E, cin, True)
    if shuf:
        ref = torch.full((N, 2 * H, 2 * W, 2 * shuf), 3.0, dtype=BF, device=DEV)
        out = torch.full_like(ref, 3.0)
        extra = dict(out_coff=shuf, shuffle2_c=shuf)
        codes = (1452, 1454), (1462, 1464)
    else:
        ref = torch.full((N, H, W, r16(co)), 7.0, dtype=BF, device=DEV)
        out = torch.full_like(ref, 5.0)
        extra = {}
        codes = (1402, 1404), (1412, 1414)
    assert ops.conv2d(a_, wf, ref, plan_only=True, **extra, **kw) in codes[0]
    ops.conv2d(a_, wf, ref, **extra, **kw)
    assert ops.conv2d(z, wf, out, res_mode=hip.RES_INBN, bn_coef=coef, plan_only=True, **extra, **kw) in codes[1]
    ops.conv2d(z, wf, out, res_mode=hip.RES_INBN, bn_coef=coef, **extra, **kw)
    torch.cuda.synchronize()
    if shuf:
        assert torch.equal(out, ref)
    else:
        assert torch.equal(out[..., :co], ref[..., :co])
    # and against the CPU (so that "identical" is not "identically wrong")
    zc = z.float().cpu().permute(0, 3, 1, 2)
    cc = coef.cpu()
    for e in range(E):
        act = torch.relu((zc[e * ipe:(e + 1) * ipe] - cc[0, e].view(1, -1, 1, 1)) * cc[2, e].view(1, -1, 1, 1)
                         + cc[3, e].view(1, -1, 1, 1)).to(BF).float()
        yr = F.conv2d(act, ws[e].to(BF).float(), bp[e, :co].cpu())
        if shuf:
            got = out[e * ipe:(e + 1) * ipe, :, :, shuf:].float().cpu()                      # [ipe, 2H, 2W, shuf]
            yr = yr.view(ipe, 2, 2, shuf, H, W).permute(0, 4, 1, 5, 2, 3).reshape(ipe, 2 * H, 2 * W, shuf)
            close(got.permute(0, 3, 1, 2), yr.permute(0, 3, 1, 2), BF, f"1x1 + scatter over relu(bn(z)) e{e}")
        else:
            close(from_nhwc(out[e * ipe:(e + 1) * ipe], co), yr, BF, f"1x1 over relu(bn(z)) e{e}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("E,ipe,C,H,W,coff,ld", [(1, 3, 64, 32, 48, 0, 128), (2, 2, 128, 16, 16, 0, 256), (1, 2, 512, 8, 8, 0, 512)])
def test_bn_apply_with_fused_maxpool2(dtype, E, ipe, C, H, W, coff, ld):
    """pmoe_bn_apply_pool2 (round 4): BatchNorm + ReLU into the skip window of a concatenation buffer AND MaxPool2d(2, 2) of it in
    one pass: bit-identical to pmoe_bn_apply + pmoe_maxpool2s2_fwd, rest of the buffer untouched."""
    g = torch.Generator().manual_seed(C + H)
    N = E * ipe
    x = nhwc(rnd((N, C, H, W), g, dtype, 2.0), C, dtype)
    scale, shift, mean = (torch.randn(E, C, generator=g).to(DEV) for _ in range(3))
    ya = torch.full((N, H, W, ld), 5.0, dtype=dtype, device=DEV)
    pa = torch.empty(N, H // 2, W // 2, C, dtype=dtype, device=DEV)
    ops.bn_apply_pool2(x, ya, pa, scale, shift, mean, ipe, E, C, True, y_coff=coff)
    yb = torch.full((N, H, W, ld), 5.0, dtype=dtype, device=DEV)
    pb = torch.empty_like(pa)
    ops.bn_apply(x, None, yb, scale, shift, mean, ipe * H * W, E, C, True, y_coff=coff)
    ops.maxpool2_fwd(yb, pb, c=C, x_coff=coff)
    assert torch.equal(ya, yb) and torch.equal(pa, pb)
    assert ld == C or (ya[..., C:] == 5.0).all()
