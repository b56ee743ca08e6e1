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
        shared_speed = speeds.dim() == 2          # MixtureOfExpertsShared: pred_speed [B,1] (loss.py:129-130)
        dsp = torch.empty_like(speeds, dtype=F32)
        ops.moe_loss(probs.contiguous(), mean.contiguous(), std.contiguous(), speeds.contiguous(),
                     actions.contiguous().float(), target.contiguous().float().view(B), c0, c1, loss, ll, dp, dm, ds,
                     dsp, B, E, shared_speed)
        ctx.save_for_backward(dp, dm, ds, dsp)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        dp, dm, ds, dsp = ctx.saved_tensors
        return g * dp, g * dm, g * ds, g * dsp, None, None, None, None


def moe_loss(action_dists, speed_pred, actions_gt, speed_gt, loss_coefs):
    """NLL of the Gaussian mixture + MSE(speeds, target)/E (loss.py:121-132), same argument order.
    ``action_dists`` is the distribution returned by MixtureOfExperts.forward (or any
    MixtureSameFamily(Categorical, Independent(Normal))); ``speed_pred`` is [B,E,1], or [B,1] from
    MixtureOfExpertsShared, in which case the speed term is plain mse(speed_pred, speed_gt)."""
    hp = getattr(action_dists, "hip_params", None)
    if hp is None:
        hp = (action_dists.mixture_distribution.probs, action_dists.component_distribution.base_dist.loc,
              action_dists.component_distribution.base_dist.scale)
    if speed_pred.dim() == 2 and (speed_pred.shape[1] != 1 or speed_gt.numel() != speed_pred.shape[0]):
        raise ValueError("moe_loss: a 2-D speed prediction must be [B,1] with one target speed per sample")
    probs, mean, std = hp
    return _MoeLossFn.apply(probs, mean, std, speed_pred, actions_gt, speed_gt, float(loss_coefs[0]),
                            float(loss_coefs[1]))
