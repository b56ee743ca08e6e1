"""TEST INFRASTRUCTURE (never imported by pmoe_amd): CPU statement of the fp8 policy of BASELINE config 5.

The reference has no fp8 path (all arithmetic is fp32), so there is no reference semantics to pin;
what IS pinned is that the HIP kernels implement exactly this policy (bit-exact quantiser tests in
tests/test_fp8_gpu.py) and that the network built from it stays within the measured bound of this emulation.

Policy (include/pmoe_hip.h ``pmoe_pack_conv_weights_fp8`` / ``pmoe_conv_desc.w_fp8``):
  * weights : one POWER-OF-TWO scale per output channel, s = 2^ceil(log2(max|W[co]| / 448)) (1 for an all-zero row),
              q = e4m3(W / s): OCP e4m3fn, round to nearest even, clamped to +-448.  q * s is exact in bf16.
  * inputs  : the conv's input (bf16 in HBM) is converted to e4m3(x * IN_SCALE) on the way into the matrix core; IN_SCALE
              is a fixed power of two (16: post-ReLU BatchNorm outputs are O(1), representable up to 28, resolved down
              to 1.2e-4).
  * forward : y = conv(Q(x), Q(W)) accumulated in f32.
  * backward: straight-through -- dx = conv^T(dy, Q(W)) on the bf16 matrix cores with the exactly dequantised weights;
              dW = dy (x) x with the UNQUANTISED bf16 input (the weight-gradient kernel reads the stored activation).
  * which   : (round 3) the dense 3x3 stride-1 convolutions with whole 128-channel chunks -- ResNet layer2-4, 9 convolutions,
              46 % of the forward MACs (model/blocks/backbone.py:57-70) -- i.e. the launches the block-scaled fp8 matrix
              instruction v_mfma_scale_f32_32x32x64_f8f6f4 serves at twice the bf16 rate.  Round 2 also quantised layer1, the
              stride-2 and the 1x1 convolutions, on kernels that were SLOWER than bf16; they are bf16 again.  The stem keeps
              bf16 (12 input channels; the ECA gate is folded into per-image weight packs).
  * where   : the activation is quantised ONCE, by the BatchNorm pass that produces it (e4m3(bf16(y) * IN_SCALE) written next
              to the bf16 tensor), not in the consumer's loader; same values either way.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

FP8_MAX = 448.0
IN_SCALE = 16.0


def row_scale(w):
    """[cout, ...] f32 -> [cout] power-of-two scales."""
    amax = w.detach().abs().flatten(1).amax(1)
    mant, ex = torch.frexp(amax / FP8_MAX)
    ex = torch.where(mant == 0.5, ex - 1, ex)
    s = torch.ldexp(torch.ones_like(amax), ex)
    return torch.where(amax > 0, s, torch.ones_like(amax))


def e4m3(x):
    """round to OCP e4m3fn (nearest even, saturating) and back to f32."""
    return x.clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).to(x.dtype)


def e4m3_bytes(x):
    return x.clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(torch.uint8)


def qdq_weight(w):
    s = row_scale(w).view(-1, *([1] * (w.dim() - 1)))
    return e4m3(w / s) * s


def qdq_act(x, in_scale=IN_SCALE):
    xb = x.to(torch.bfloat16).to(x.dtype)            # the stored activation is bf16
    return e4m3(xb * in_scale) / in_scale


class Fp8Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding, in_scale):
        wq = qdq_weight(w)
        ctx.save_for_backward(x, wq)
        ctx.conf = (stride, padding)
        return F.conv2d(qdq_act(x, in_scale), wq, None, stride, padding)

    @staticmethod
    def backward(ctx, dy):
        x, wq = ctx.saved_tensors
        stride, padding = ctx.conf
        dx = torch.nn.grad.conv2d_input(x.shape, wq, dy, stride, padding)
        dw = torch.nn.grad.conv2d_weight(x, wq.shape, dy, stride, padding)
        return dx, dw, None, None, None


def selected(name, mod):
    """round 3: the dense 3x3 stride-1 convolutions of ResNet layer2-4 (whole 128-channel chunks) -- the launches the block-scaled
    fp8 matrix instruction serves; the 64-channel layer1, the stride-2 and the 1x1 convolutions stay bf16"""
    return (isinstance(mod, nn.Conv2d) and ".backbone.layer" in "." + name and mod.kernel_size == (3, 3)
            and mod.stride == (1, 1) and mod.in_channels % 128 == 0)


def apply_fp8_policy(model, in_scale=IN_SCALE):
    """Route the selected convolutions of an oracle model through Fp8Conv (in place).  Returns their names."""
    names = []
    for name, mod in model.named_modules():
        if selected(name, mod):
            assert mod.bias is None and mod.groups == 1 and mod.dilation == (1, 1)
            mod.forward = (lambda x, m=mod: Fp8Conv.apply(x, m.weight, m.stride, m.padding, in_scale))
            names.append(name)
    return names
