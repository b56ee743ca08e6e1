"""Helpers shared by the PU-Net / PMoE tests: the reference constructors read checkpoint files (punet.py:40,
moe.py:278,335), so the tests write files with the right key sets first (contents are overwritten by
oracle.weights.fill_state_dict afterwards)."""
import torch

from pmoe_amd.model import blocks as B
from pmoe_amd.model.moe import MixtureOfExperts, get_model
from pmoe_amd.model.punet import PredictiveUnet
from pmoe_amd.utils import stage2_model_cfg


def write_checkpoints(tmp, model_type, n_experts, future_frames, with_moe=False):
    tmp.mkdir(parents=True, exist_ok=True)
    unet_path, punet_path, moe_dir = tmp / "unet.pth", tmp / "punet.pth", tmp / "moe.pth"
    torch.save({"unet": B.UNet().state_dict()}, unet_path)
    cfg = stage2_model_cfg(model_type, n_experts, dropout=0.0, future_frames=future_frames, unet_path=str(unet_path))
    pu = PredictiveUnet(**{**cfg.punet, "inter_repr": model_type == "punet_inter"})
    torch.save({"model": pu.state_dict()}, punet_path)
    if with_moe:
        torch.save(MixtureOfExperts(stage2_model_cfg("pmoe", n_experts, dropout=0.0)).state_dict(), moe_dir)
    return dict(unet_path=str(unet_path), punet_path=str(punet_path), moe_dir=str(moe_dir) if with_moe else "")


def build_product(tmp, meta, dropout=0.0, exclude_freeze=()):
    t = meta["type"]
    ck = write_checkpoints(tmp, "punet" if t.startswith("pmoe") else t, meta["n_experts"], meta["future_frames"],
                           with_moe=t.startswith("pmoe"))
    cfg = stage2_model_cfg(t, meta["n_experts"], dropout=dropout, future_frames=meta["future_frames"],
                           exclude_freeze=exclude_freeze, **ck)
    return get_model(cfg)
