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


class _ActionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, actions, speeds, actions_gt, speed_gt, c0, c1):
        B = actions.shape[0]
        dev = actions.device
        loss = torch.empty(1, dtype=F32, device=dev)
        da = torch.empty(B, 2, dtype=F32, device=dev)
        dsp = torch.empty(B, 1, dtype=F32, device=dev) if speeds is not None else None
        ops.action_loss(actions.contiguous().float(), speeds.contiguous().float() if speeds is not None else None,
                        actions_gt.contiguous().float(),
                        speed_gt.contiguous().float().view(B) if speeds is not None else None, c0, c1, loss, da, dsp, B)
        ctx.has_speed = speeds is not None
        ctx.save_for_backward(da, dsp if dsp is not None else da)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        da, dsp = ctx.saved_tensors
        return g * da, (g * dsp if ctx.has_speed else None), None, None, None, None


def punet_loss(actions, speed_pred, actions_gt, speed_gt, loss_coefs):
    """``loss.py:135-142``: ``c0 * L1(actions, gt) + c1 * MSE(speed_pred, speed_gt)`` (fused HIP kernel)."""
    if actions.shape != actions_gt.shape or actions.dim() != 2 or actions.shape[1] != 2:
        raise ValueError("punet_loss: actions and actions_gt must both be [B,2]")
    if speed_pred.numel() != actions.shape[0] or speed_gt.numel() != actions.shape[0]:
        raise ValueError("punet_loss: one predicted / target speed per sample expected")
    return _ActionLossFn.apply(actions, speed_pred, actions_gt, speed_gt, float(loss_coefs[0]), float(loss_coefs[1]))


def pmoe_loss(actions, speed_pred, actions_gt, speed_gt, loss_coefs):
    """``loss.py:145-151``: plain L1 imitation loss; the other arguments are dummies (interface consistency)."""
    if actions.shape != actions_gt.shape or actions.dim() != 2 or actions.shape[1] != 2:
        raise ValueError("pmoe_loss: actions and actions_gt must both be [B,2]")
    return _ActionLossFn.apply(actions, None, actions_gt, None, 1.0, 0.0)


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, mode):
        lg = logits.contiguous().float()
        tg = target.contiguous()
        loss, coefG, coefT = ops.seg_loss_fwd(lg, tg, mode)
        ctx.mode, ctx.saved = mode, (lg, tg, coefG, coefT)
        ctx.frame_losses = loss[1:]
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lg, tg, coefG, coefT = ctx.saved
        ctx.saved = None
        dl = torch.empty_like(lg)
        ops.seg_loss_bwd(lg, tg, coefG, coefT, g.contiguous().float().view(1), dl, ctx.mode)
        return dl, None, None


class AutoregressiveCriterion(torch.nn.Module):
    """Drop-in for ``trainer/loss.py:86-118`` (stage-1 PU-Net training, train_1.py:75-77,134): the per-frame loss summed
    over the predicted frames, in three HIP launches forward and one backward.  ``inputs`` [B,T,C,H,W] logits (f32, as
    ``PredictiveUnet.forward`` returns them), ``targets`` [B,T,H,W] int64 class indices.  ``'tversky'`` is
    ``0.5 * cross_entropy(weight = 1 - class dice) + 0.5 * tversky_loss`` exactly as the reference computes it,
    including its per-(class, image column) Tversky ratio (loss.py:40)."""

    def __init__(self, n_target_frames: int = 1, loss_type: str = "tversky"):
        super().__init__()
        if loss_type not in ops.SEG_MODES:
            raise ValueError(f"Unknown loss type {loss_type}, supported ones are L1, L2, and tversky")
        self.n_target_frames, self.loss_type = n_target_frames, loss_type

    def forward(self, inputs, targets):
        assert inputs.size(1) == self.n_target_frames
        assert targets.size(1) == self.n_target_frames
        if targets.dtype != torch.int64:
            raise ValueError("AutoregressiveCriterion: targets must be int64 class indices (data_loader.py segmentation masks)")
        return _SegLossFn.apply(inputs, targets, self.loss_type)
