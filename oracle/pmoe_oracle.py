"""CPU restatement (plain PyTorch fp32) of the reference's stage-2 MoE forward/backward path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker for the HIP path and the timed
``cpu_baseline``; never imported by ``pmoe_amd``.

Each class cites the reference code it restates.  Parameter / buffer names reproduce the
reference's ``state_dict`` layout, so one state_dict loads into the reference,
this oracle and ``pmoe_amd.model`` alike.  Pinned against the imported reference by
``oracle/make_golden.py`` + ``tests/test_oracle_golden.py``.
"""
from collections import OrderedDict
from math import log2

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.distributions as D

from . import resnet_topology


class Cfg(dict):
    """Attribute-access mapping standing in for an OmegaConf node (supports ``**`` and assignment,
    which ``model/moe.py:55-66,274`` rely on)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return Cfg({k: Cfg.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return [Cfg.wrap(v) for v in obj]
        return obj


def stage2_cfg(model_type="moe", n_experts=4, dropout=0.0, n_commands=6, n_frames=4):
    """The ``model:`` node of ``conf/stage_2_moe.yaml:76-133`` with the knobs the tests vary."""
    def mlp(dims, act, l_act=False):
        return dict(dims=list(dims), act=act, l_act=l_act, bn=False, dropout=dropout)

    return Cfg.wrap(dict(
        verbose=False, type=model_type, n_experts=n_experts, loss_coefs=[0.7, 0.3],
        exclude_freeze=[], device="cpu", punet_path="",
        action_head=mlp([1536, 512, 512], "elu", True),
        speed_encoder=mlp([1, 512, 512], "relu"),
        command_encoder=mlp([n_commands, 512, 512], "relu"),
        speed_prediction=mlp([1536, 512, 512, 1], "relu"),
        backbone=dict(type="rgb", n_frames=n_frames,
                      rgb=dict(arch="resnet18", pretrained=False, gamma=2, b=1)),
        pmoe=dict(moe_dir="", punet_dir=""),
    ))


def make_mlp(dims, act, l_act=False, bn=True, dropout=0.0):
    """``model/blocks/basics.py:10-44``: Linear(bias=not bn) [BN1d] act [Dropout] per hidden layer,
    bare Linear last, optional trailing activation; ONE shared activation module instance."""
    activation = {"relu": nn.ReLU(inplace=True), "tanh": nn.Tanh(),
                  "sigmoid": nn.Sigmoid(), "elu": nn.ELU()}[act.lower()]
    layers = []
    n = len(dims) - 1
    for i in range(n):
        layers.append(nn.Linear(dims[i], dims[i + 1], bias=not bn))
        if i != n - 1:
            if bn:
                layers.append(nn.BatchNorm1d(dims[i + 1]))
            layers.append(activation)
            if dropout > 0.0:
                layers.append(nn.Dropout(p=dropout))
    if l_act:
        layers.append(activation)
    return nn.Sequential(*layers)


def eca_kernel_size(channels, gamma=2, b=1):
    """``basics.py:66-67``: t = int(|log2(C)+b|/gamma), made odd upward."""
    t = int(abs((log2(channels) + b) / gamma))
    return t if t % 2 else t + 1


class EfficientBlock(nn.Module):
    """ECA channel attention, ``basics.py:61-76``: GAP -> Conv1d(k) across channels -> sigmoid -> scale."""

    def __init__(self, channels, gamma=2, b=1):
        super().__init__()
        k = eca_kernel_size(channels, gamma, b)
        self.conv = nn.Conv1d(1, 1, kernel_size=k, padding=k // 2, bias=False)

    def forward(self, x):
        g = x.mean(dim=(2, 3))                       # [B, C]
        s = torch.sigmoid(self.conv(g.unsqueeze(1)).squeeze(1))
        return x * s[:, :, None, None]


class EfficientConvBlock(nn.Module):
    """``basics.py:79-134``: eca1 -> conv3x3(in->64)+BN+ReLU -> eca2 -> conv3x3(64->out)+BN+ReLU."""

    def __init__(self, in_ch, out_ch, stride=1, gamma=2, b=1):
        super().__init__()
        def cbr(i, o):
            return nn.Sequential(nn.Conv2d(i, o, 3, stride=stride, padding=1, bias=False),
                                 nn.BatchNorm2d(o), nn.ReLU(inplace=True))
        self.layer1 = nn.Sequential(OrderedDict(
            [("eca1", EfficientBlock(in_ch, gamma, b)), ("conv1", cbr(in_ch, 64))]))
        self.layer2 = nn.Sequential(OrderedDict(
            [("eca2", EfficientBlock(64, gamma, b)), ("conv2", cbr(64, out_ch))]))

    def forward(self, x):
        return self.layer2(self.layer1(x))


def get_backbone(arch="resnet18", n_frames=4, pretrained=False, gamma=2, b=1, n_channels=3):
    """``blocks/backbone.py:48-72``: ResNet with conv1 := EfficientConvBlock (stride 1) and
    fc := Identity when fc.in_features == 512.  bn1/relu/maxpool of the ResNet stay in place."""
    if "resnet" not in arch:
        raise NotImplementedError(arch)
    ctor = {"resnet18": resnet_topology.resnet18, "resnet34": resnet_topology.resnet34}[arch.lower()]
    model = ctor(pretrained=False)
    model.conv1 = EfficientConvBlock(n_frames * n_channels, model.conv1.out_channels, gamma=gamma, b=b)
    model.fc = nn.Identity() if model.fc.in_features == 512 else nn.Linear(model.fc.in_features, 512)
    return model


class BaseExpert(nn.Module):
    """``model/moe.py:50-101``."""

    def __init__(self, params):
        super().__init__()
        self.speed_encoder = make_mlp(**params.speed_encoder)
        self.command_encoder = make_mlp(**params.command_encoder)
        assert params.backbone.type == "rgb"
        self.backbone = get_backbone(**{**params.backbone.rgb, "n_frames": params.backbone.n_frames})
        self.speed_pred = make_mlp(**params.speed_prediction)
        self.action_features = make_mlp(**params.action_head)
        width = params.action_head.dims[-1]
        self.alpha = nn.Linear(width, 1)
        self.action_pred = nn.Linear(width, 4)

    def _features(self, images, speed, command):
        s = self.speed_encoder(speed)
        c = self.command_encoder(command)
        x = images.reshape(images.shape[0], -1, images.shape[-2], images.shape[-1])
        return torch.cat([self.backbone(x), s, c], dim=-1)

    def forward(self, images, speed, command):
        feats = self._features(images, speed, command)
        pred_speed = self.speed_pred(feats)
        af = self.action_features(feats)
        mean, std = self.action_pred(af).split(2, dim=-1)
        std = F.elu(std) + 1
        alpha = torch.relu(self.alpha(af))
        return alpha, mean, std, pred_speed


class BaseExpertAlt(BaseExpert):
    """``model/moe.py:104-128``: alpha is an MLP on the 1536-d features, no ReLU."""

    def __init__(self, params):
        super().__init__(params)
        self.alpha = nn.Sequential(nn.Linear(1536, 512), nn.ReLU(inplace=True), nn.Linear(512, 1))

    def forward(self, images, speed, command):
        feats = self._features(images, speed, command)
        pred_speed = self.speed_pred(feats)
        af = self.action_features(feats)
        mean, std = self.action_pred(af).split(2, dim=-1)
        std = F.elu(std) + 1
        return self.alpha(feats), mean, std, pred_speed


class MixtureOfExperts(nn.Module):
    """``model/moe.py:131-177``."""

    def __init__(self, params):
        super().__init__()
        self.k = params.n_experts
        base = BaseExpert if params.type == "moe" else BaseExpertAlt
        self.moe = nn.ModuleList([base(params) for _ in range(self.k)])

    def mixture_params(self, images, speed, command):
        outs = [m(images, speed, command) for m in self.moe]
        probs = F.softmax(torch.cat([o[0] for o in outs], dim=1), dim=1)     # [B,E]
        mean = torch.stack([o[1] for o in outs], dim=1)                      # [B,E,2]
        std = torch.stack([o[2] for o in outs], dim=1)
        speeds = torch.stack([o[3] for o in outs], dim=1)                    # [B,E,1]
        return probs, mean, std, speeds

    def forward(self, images, speed, command):
        probs, mean, std, speeds = self.mixture_params(images, speed, command)
        dist = D.MixtureSameFamily(D.Categorical(probs), D.Independent(D.Normal(mean, std), 1))
        return dist, speeds

    def sample(self, images, speed, command):
        return self.forward(images, speed, command)[0].sample()


class MixtureOfExpertsShared(nn.Module):
    """``model/moe.py:180-265``: one trunk, ``n_experts`` Gaussian components from the last layer."""

    def __init__(self, params):
        super().__init__()
        self.speed_encoder = make_mlp(**params.speed_encoder)
        self.command_encoder = make_mlp(**params.command_encoder)
        assert params.backbone.type == "rgb"
        self.backbone = get_backbone(**{**params.backbone.rgb, "n_frames": params.backbone.n_frames})
        self.speed_pred = make_mlp(**params.speed_prediction)
        self.action_features = make_mlp(**params.action_head)
        width = params.action_head.dims[-1]
        self.n_experts = params.n_experts
        self.alpha = nn.Linear(width, params.n_experts)
        self.action_pred = nn.Linear(width, 4 * params.n_experts)

    def mixture_params(self, images, speed, command):
        s = self.speed_encoder(speed)
        c = self.command_encoder(command)
        x = images.reshape(images.shape[0], -1, images.shape[-2], images.shape[-1])
        feats = torch.cat([self.backbone(x), s, c], dim=-1)
        pred_speed = self.speed_pred(feats)                                  # [B,1]
        af = self.action_features(feats)
        mean, std = self.action_pred(af).view(images.shape[0], self.n_experts, -1).split(2, dim=-1)
        std = F.elu(std) + 1
        probs = F.softmax(self.alpha(af), dim=1)                             # no ReLU here (moe.py:226)
        return probs, mean, std, pred_speed

    def forward(self, images, speed, command):
        probs, mean, std, pred_speed = self.mixture_params(images, speed, command)
        dist = D.MixtureSameFamily(D.Categorical(probs), D.Independent(D.Normal(mean, std), 1))
        return dist, pred_speed

    def sample(self, images, speed, command):
        return self.forward(images, speed, command)[0].sample()


def get_model(cfg):
    """``model/moe.py:25-47`` (MoE families only in the oracle so far)."""
    if cfg.type in ("moe", "moe_alt"):
        return MixtureOfExperts(cfg)
    if cfg.type == "moe_shared":
        return MixtureOfExpertsShared(cfg)
    raise ValueError(f"{cfg.type} is UNKNOWN or not restated by the oracle yet")


def moe_loss(action_dists, speed_pred, actions_gt, speed_gt, loss_coefs):
    """``trainer/loss.py:121-132``: mixture NLL + MSE(speeds, target broadcast over experts)/E.
    (The reference unsqueezes ``speed_gt`` in place; the restatement leaves the caller's tensor alone.)"""
    nll = -action_dists.log_prob(actions_gt).mean(dim=0)
    if speed_pred.dim() > 2:
        tgt = speed_gt.unsqueeze(1).expand_as(speed_pred)
        speed_loss = F.mse_loss(speed_pred, tgt) / speed_pred.shape[1]
    else:
        speed_loss = F.mse_loss(speed_pred, speed_gt)
    return loss_coefs[0] * nll + loss_coefs[1] * speed_loss


def mixture_nll_explicit(probs, mean, std, actions):
    """MixtureSameFamily.log_prob written out (torch/distributions/mixture_same_family.py):
    logsumexp_k( log_softmax(log p)_k + sum_d logN(a_d; mu_kd, sigma_kd) ).  Returns per-sample log-lik."""
    a = actions.unsqueeze(1)
    comp = (-((a - mean) ** 2) / (2 * std ** 2) - std.log() - 0.5 * torch.log(torch.tensor(2 * torch.pi))).sum(-1)
    logp = torch.log_softmax(torch.log(probs), dim=-1)
    return torch.logsumexp(comp + logp, dim=-1)
