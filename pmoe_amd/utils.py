"""Host utilities whose semantics are part of the drop-in boundary (``PMoE/utils/nn.py``, ``utils/utility.py``)."""
from typing import List

import torch.nn as nn
import yaml


def freeze(model: nn.Module, exclude: List = (), verbose: bool = False) -> nn.Module:
    """``utils/nn.py:22-58``: set requires_grad=False on every parameter whose NAME contains none of the
    substrings in ``exclude`` (an empty list freezes everything)."""
    exclude = list(exclude) if exclude is not None else []
    frozen = []
    for name, p in model.named_parameters():
        if not exclude or not any(tag in name for tag in exclude):
            p.requires_grad_(False)
            frozen.append(name)
    if verbose:
        total = sum(1 for _ in model.parameters())
        print(f"{len(frozen)} of {total} parameter tensors have been frozen.")
    return model


class AttrDict(dict):
    """Attribute-access mapping accepted wherever the reference takes an OmegaConf node: supports
    ``cfg.key``, ``**cfg.node`` and item assignment (``moe.py:55-66,274``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return [AttrDict.wrap(v) for v in obj]
        return obj


def get_conf(name: str):
    """``utils/utility.py:9-17`` without OmegaConf (not installed here): load a YAML file into AttrDicts."""
    name = name if name.split(".")[-1] == "yaml" else name + ".yaml"
    with open(name) as f:
        return AttrDict.wrap(yaml.safe_load(f))


def stage2_model_cfg(model_type="moe", n_experts=4, dropout=0.3, n_commands=6, n_frames=4, future_frames=6,
                     unet_path="", punet_path="", moe_dir="", punet_dir="", exclude_freeze=(), device="cpu"):
    """The ``model:`` node of ``conf/stage_2_moe.yaml:76-133`` / ``stage_2_pmoe.yaml:76-133`` (defaults as shipped by
    the reference; the checkpoint paths are the reference's ``punet.model_path``, ``punet_path``, ``pmoe.moe_dir`` and
    ``pmoe.punet_dir``)."""
    def mlp(dims, act, l_act=False):
        return dict(dims=list(dims), act=act, l_act=l_act, bn=False, dropout=dropout)

    return AttrDict.wrap(dict(
        verbose=False, type=model_type, n_experts=n_experts, loss_coefs=[0.7, 0.3], exclude_freeze=list(exclude_freeze),
        device=device, punet_path=punet_path,
        action_head=mlp([1536, 512, 512], "elu", True),
        speed_encoder=mlp([1, 512, 512], "relu"),
        command_encoder=mlp([n_commands, 512, 512], "relu"),
        speed_prediction=mlp([1536, 512, 512, 1], "relu"),
        backbone=dict(type="rgb", n_frames=n_frames, rgb=dict(arch="resnet18", pretrained=False, gamma=2, b=1)),
        punet=dict(past_frames=n_frames, future_frames=future_frames, in_features=3, num_classes=23, gamma=2, b=1,
                   unet_inter_repr=False, model_name="unet", model_path=unet_path),
        pmoe=dict(moe_dir=moe_dir, punet_dir=punet_dir),
    ))


def write_checkpoints(tmp, model_type, n_experts, future_frames, with_moe=False):
    """The reference constructors READ checkpoint files (``punet.py:40`` the stage-0 U-Net, ``moe.py:278`` the PU-Net,
    ``moe.py:335`` the frozen mixture): write random-init files with the right key sets into ``tmp`` and return the three
    paths in ``stage2_model_cfg``'s keyword form.  Used wherever a ``punet`` / ``pmoe`` model is built without a trained
    checkpoint at hand (bench.py's synthetic-weight configs, the tests -- which then overwrite every tensor)."""
    import torch

    from .model import blocks as B
    from .model.moe import MixtureOfExperts
    from .model.punet import PredictiveUnet
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
    """``get_model`` for the checkpoint-reading types (``punet``, ``punet_inter``, ``pmoe*``) with random-init checkpoint
    files written to ``tmp`` first; ``meta`` = dict(type, n_experts, future_frames)."""
    from .model.moe import get_model
    t = meta["type"]
    ck = write_checkpoints(tmp, "punet" if t.startswith("pmoe") else t, meta["n_experts"], meta["future_frames"],
                           with_moe=t.startswith("pmoe"))
    cfg = stage2_model_cfg(t, meta["n_experts"], dropout=dropout, future_frames=meta["future_frames"],
                           exclude_freeze=exclude_freeze, **ck)
    return get_model(cfg)
