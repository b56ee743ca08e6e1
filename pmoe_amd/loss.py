"""Drop-in for ``PMoE/trainer/loss.py:121-132`` (``moe_loss``) backed by the fused HIP loss kernel."""
import torch

from . import ops

F32 = torch.float32


class _MoeLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, mean, std, speeds, actions, target, c0, c1):
        B, E = probs.shape
        dev = probs.device
        loss = torch.empty(1, dtype=F32, device=dev)
        ll = torch.empty(B, dtype=F32, device=dev)
        dp = torch.empty(B, E, dtype=F32, device=dev)
        dm = torch.empty(B, E, 2, dtype=F32, device=dev)
        ds = torch.empty(B, E, 2, dtype=F32, device=dev)
        dsp = torch.empty(B, E, 1, dtype=F32, device=dev)
        ops.moe_loss(probs.contiguous(), mean.contiguous(), std.contiguous(), speeds.contiguous(),
                     actions.contiguous().float(), target.contiguous().float().view(B), c0, c1, loss, ll, dp, dm, ds,
                     dsp, B, E)
        ctx.save_for_backward(dp, dm, ds, dsp)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        dp, dm, ds, dsp = ctx.saved_tensors
        return g * dp, g * dm, g * ds, g * dsp, None, None, None, None


def moe_loss(action_dists, speed_pred, actions_gt, speed_gt, loss_coefs):
    """NLL of the Gaussian mixture + MSE(speeds, target)/E (loss.py:121-132), same argument order.
    ``action_dists`` is the distribution returned by MixtureOfExperts.forward (or any
    MixtureSameFamily(Categorical, Independent(Normal))); ``speed_pred`` is [B,E,1]."""
    hp = getattr(action_dists, "hip_params", None)
    if hp is None:
        hp = (action_dists.mixture_distribution.probs, action_dists.component_distribution.base_dist.loc,
              action_dists.component_distribution.base_dist.scale)
    if speed_pred.dim() != 3:
        raise NotImplementedError("moe_loss on a [B,1] speed prediction (moe_shared) is not on the HIP path yet")
    probs, mean, std = hp
    return _MoeLossFn.apply(probs, mean, std, speed_pred, actions_gt, speed_gt, float(loss_coefs[0]),
                            float(loss_coefs[1]))
