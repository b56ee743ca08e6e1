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


def stage2_cfg(model_type="moe", n_experts=4, dropout=0.0, n_commands=6, n_frames=4, future_frames=6,
               unet_path="", punet_path="", moe_dir="", punet_dir="", exclude_freeze=()):
    """The ``model:`` node of ``conf/stage_2_moe.yaml:76-133`` / ``stage_2_pmoe.yaml:76-133`` with the knobs the
    tests vary."""
    def mlp(dims, act, l_act=False):
        return dict(dims=list(dims), act=act, l_act=l_act, bn=False, dropout=dropout)

    return Cfg.wrap(dict(
        verbose=False, type=model_type, n_experts=n_experts, loss_coefs=[0.7, 0.3],
        exclude_freeze=list(exclude_freeze), device="cpu", punet_path=punet_path,
        action_head=mlp([1536, 512, 512], "elu", True),
        speed_encoder=mlp([1, 512, 512], "relu"),
        command_encoder=mlp([n_commands, 512, 512], "relu"),
        speed_prediction=mlp([1536, 512, 512, 1], "relu"),
        backbone=dict(type="rgb", n_frames=n_frames,
                      rgb=dict(arch="resnet18", pretrained=False, gamma=2, b=1)),
        punet=dict(past_frames=n_frames, future_frames=future_frames, in_features=3, num_classes=23, gamma=2, b=1,
                   unet_inter_repr=False, model_name="unet", model_path=unet_path),
        pmoe=dict(moe_dir=moe_dir, punet_dir=punet_dir),
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


def conv3(in_ch, out_ch, stride=1):
    """``blocks/basics.py:47-58``."""
    return nn.Sequential(
        nn.Conv2d(in_ch, out_ch, 3, stride, 1, bias=False), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True),
        nn.Conv2d(out_ch, out_ch, 3, stride, 1, bias=False), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True))


class UNet(nn.Module):
    """``blocks/unet.py:8-95``: 5 conv3 blocks down (MaxPool2d(2) between), 4x [ConvTranspose2d(k2,s2) -> cat skip ->
    conv3] up, 1x1 class conv.  ``Dropout2d(p=0)`` of the reference is the identity and is omitted."""

    def __init__(self, in_features=3, out_features=23, gamma=2, b=1, dropout=0.0, inter_repr=False):
        super().__init__()
        assert dropout == 0.0
        self.inter_repr = inter_repr
        self.dwn_1, self.dwn_2, self.dwn_3 = conv3(in_features, 64), conv3(64, 128), conv3(128, 256)
        self.dwn_4, self.dwn_5 = conv3(256, 512), conv3(512, 512)
        self.up_1, self.up_forw_1 = nn.ConvTranspose2d(512, 512, 2, 2), conv3(1024, 512)
        self.up_2, self.up_forw_2 = nn.ConvTranspose2d(512, 256, 2, 2), conv3(512, 256)
        self.up_3, self.up_forw_3 = nn.ConvTranspose2d(256, 128, 2, 2), conv3(256, 128)
        self.up_4, self.up_forw_4 = nn.ConvTranspose2d(128, 64, 2, 2), conv3(128, 64)
        self.out = nn.Conv2d(64, out_features, 1)

    def forward(self, image):
        skips, x = [], image
        for blk in (self.dwn_1, self.dwn_2, self.dwn_3, self.dwn_4):
            x = blk(x)
            skips.append(x)
            x = F.max_pool2d(x, 2, 2)
        x5 = x = self.dwn_5(x)
        for up, fw in ((self.up_1, self.up_forw_1), (self.up_2, self.up_forw_2), (self.up_3, self.up_forw_3),
                       (self.up_4, self.up_forw_4)):
            skip = skips.pop()
            x = fw(torch.cat([skip, up(x, output_size=skip.size())], 1))
        x = self.out(x)
        if self.inter_repr:
            return torch.flatten(F.adaptive_avg_pool2d(x5, 1), 1), x
        return x


class PredictiveUnet(nn.Module):
    """``model/punet.py:13-120``.  ``model_path=None`` skips the checkpoint read (the tests fill every weight from
    ``oracle.weights``); otherwise it is loaded exactly like the reference (punet.py:40-55)."""

    def __init__(self, past_frames=4, future_frames=4, in_features=3, num_classes=23, gamma=2, b=1, inter_repr=False,
                 unet_inter_repr=False, model_name="unet-swa", model_path="unet.pth"):
        super().__init__()
        self.n_past_frames, self.n_future_frames = past_frames, future_frames
        self.inter_repr, self.unet_inter_repr = inter_repr, unet_inter_repr
        self.unet = UNet(in_features, num_classes, gamma, b, inter_repr=unet_inter_repr)
        if model_path:
            self.unet.load_state_dict(torch.load(model_path)[model_name], strict=False)
        for p in self.unet.parameters():
            p.requires_grad = False
        self.unet.eval()
        self.entry_block = EfficientConvBlock(past_frames * num_classes, in_features, gamma=gamma, b=b)
        self.pred_unet = UNet(in_features, num_classes, gamma, b, inter_repr=inter_repr)

    def forward(self, img_list):
        assert img_list.shape[-4] == self.n_past_frames, "Number of images should match number of past frames"
        masks = [self.unet(img_list[:, i]) for i in range(self.n_past_frames)]
        assert not self.unet_inter_repr
        if self.n_future_frames == 0:          # punet.py:91-96: segmentation of the current frame
            return masks[-1]
        outs, inter = [], None
        for _ in range(self.n_future_frames):
            m = self.pred_unet(self.entry_block(torch.cat(masks[-self.n_past_frames:], dim=-3)))
            if self.inter_repr:
                inter, m = m
            masks.append(m)
            outs.append(m)
        return inter if self.inter_repr else torch.stack(outs, dim=1)


class PUNetExpert(nn.Module):
    """``model/moe.py:268-323``."""

    def __init__(self, params):
        super().__init__()
        self.return_inter = params.type == "punet_inter"
        params.punet.inter_repr = self.return_inter
        self.speed_encoder = make_mlp(**params.speed_encoder)
        self.command_encoder = make_mlp(**params.command_encoder)
        self.punet = PredictiveUnet(**params.punet)
        if params.punet_path:
            self.punet.load_state_dict(torch.load(params.punet_path, map_location=params.device)["model"])
        for p in self.punet.parameters():            # freeze(self.punet), utils/nn.py:22-58 with an empty exclude list
            p.requires_grad_(False)
        self.backbone = None if self.return_inter else get_backbone(
            **{**params.backbone.rgb, "n_frames": params.punet.future_frames, "n_channels": params.punet.num_classes})
        self.speed_pred = make_mlp(**params.speed_prediction)
        self.action_pred = nn.Sequential(make_mlp(**params.action_head), nn.Linear(params.action_head.dims[-1], 2))

    def forward(self, images, speed, command):
        s = self.speed_encoder(speed)
        c = self.command_encoder(command)
        if self.return_inter:
            img = self.punet(images)
        else:
            m = self.punet(images)
            img = self.backbone(m.view(m.shape[0], -1, m.shape[-2], m.shape[-1]))
        feats = torch.cat([img, s, c], dim=-1)
        return torch.tanh(self.action_pred(feats)), self.speed_pred(feats)

    def sample(self, images, speed, command):
        return self.forward(images, speed, command)[0]


def _freeze(model, exclude):
    """``utils/nn.py:22-58``."""
    for name, p in model.named_parameters():
        if not exclude or not any(tag in name for tag in exclude):
            p.requires_grad_(False)
    return model


class PMoE(nn.Module):
    """``model/moe.py:326-363``; ``blend`` is the deterministic tail of ``forward`` (moe.py:353-356)."""

    def __init__(self, params):
        super().__init__()
        self.moe = MixtureOfExperts(params)
        if params.pmoe.moe_dir:
            self.moe.load_state_dict(torch.load(params.pmoe.moe_dir), strict=False)
        self.moe = _freeze(self.moe, params.exclude_freeze)
        self.punet = PUNetExpert(params)
        if params.pmoe.punet_dir:
            self.punet.load_state_dict(torch.load(params.pmoe.punet_dir), strict=False)
            self.punet = _freeze(self.punet, params.exclude_freeze)
        self.lat_weights = nn.Linear(2, 1)
        self.long_weights = nn.Linear(2, 1)

    def blend(self, moe_actions, punet_actions):
        lat = self.lat_weights(torch.cat([moe_actions[:, 0:1], punet_actions[:, 0:1]], dim=-1))
        lon = self.long_weights(torch.cat([moe_actions[:, 1:], punet_actions[:, 1:]], dim=-1))
        return torch.tanh(torch.cat([lat, lon], dim=-1))

    def forward(self, images, speed, command):
        punet_actions, _ = self.punet(images.clone(), speed.clone(), command.clone())
        dists, _ = self.moe(images, speed, command)
        return self.blend(dists.sample(), punet_actions), -1

    def sample(self, images, speed, command):
        return self.forward(images, speed, command)[0]


def punet_loss(actions, speed_pred, actions_gt, speed_gt, loss_coefs):
    """``trainer/loss.py:135-142``."""
    return loss_coefs[0] * F.l1_loss(actions, actions_gt) + loss_coefs[1] * F.mse_loss(speed_pred, speed_gt)


def pmoe_loss(actions, speed_pred, actions_gt, speed_gt, loss_coefs):
    """``trainer/loss.py:145-151``."""
    return F.l1_loss(actions, actions_gt)


def class_dice_weights(pred, target, eps=1e-6):
    """``trainer/loss.py:6-17``: per-class ``1 - dice`` of the arg-max prediction over the WHOLE batch (no gradient)."""
    hard = pred.argmax(dim=1)
    w = torch.ones(pred.size(1), dtype=torch.float, device=pred.device)
    for c in range(pred.size(1)):
        p, t = hard == c, target == c
        w[c] = 1 - 2 * ((p & t).sum().float() + eps) / (p.sum() + t.sum() + eps)
    return w


def tversky_loss(pred, target, alpha=0.5, beta=0.5):
    """``trainer/loss.py:34-45``: 1 - mean TP / (TP + alpha FP + beta FN) on soft-max probabilities.  The reference
    derives the reduced axes from the TARGET's rank (loss.py:40: ``range(2, target.ndimension())`` with target
    [B,H,W]), i.e. it sums over batch and image ROWS only: the ratio is formed per (class, image column) and the mean
    runs over both -- restated as is."""
    onehot = F.one_hot(target, pred.size(1)).movedim(-1, 1).to(pred.dtype)
    probs = torch.softmax(pred, dim=1)
    dims = (0,) + tuple(range(2, target.dim()))
    tp = (probs * onehot).sum(dims)
    fp = (probs * (1 - onehot)).sum(dims)
    fn = ((1 - probs) * onehot).sum(dims)
    return 1 - (tp / (tp + alpha * fp + beta * fn)).mean()


def cross_entropy_tversky_weighted_loss(pred, target, cross_entropy_weight=0.5, tversky_weight=0.5):
    """``trainer/loss.py:48-57``."""
    if cross_entropy_weight + tversky_weight != 1:
        raise ValueError("Cross Entropy weight and Tversky weight should sum to 1")
    ce = F.cross_entropy(pred, target, weight=class_dice_weights(pred, target).to(pred.dtype))   # (cast: float64 checks)
    return cross_entropy_weight * ce + tversky_weight * tversky_loss(pred, target)


class AutoregressiveCriterion(nn.Module):
    """``trainer/loss.py:86-118`` (stage-1 PU-Net training, train_1.py:75-77,134): per-frame loss summed over the
    predicted frames.  inputs [B,T,C,H,W] logits, targets [B,T,H,W] class indices."""

    def __init__(self, n_target_frames=1, loss_type="tversky"):
        super().__init__()
        if loss_type not in ("l1", "l2", "tversky"):
            raise ValueError(f"Unknown loss type {loss_type}, supported ones are L1, L2, and tversky")
        self.n_target_frames, self.loss_type = n_target_frames, loss_type

    def forward(self, inputs, targets):
        assert inputs.size(1) == self.n_target_frames and targets.size(1) == self.n_target_frames
        total = 0
        for t in range(self.n_target_frames):
            x, y = inputs[:, t], targets[:, t]
            if self.loss_type == "tversky":
                total = total + cross_entropy_tversky_weighted_loss(x, y)
            else:
                onehot = F.one_hot(y, x.size(1)).movedim(-1, 1).to(x.dtype)
                total = total + (F.l1_loss(x, onehot) if self.loss_type == "l1" else F.mse_loss(x, onehot))
        return total


def get_model(cfg):
    """``model/moe.py:25-47``."""
    if cfg.type in ("moe", "moe_alt"):
        return MixtureOfExperts(cfg)
    if cfg.type == "moe_shared":
        return MixtureOfExpertsShared(cfg)
    if cfg.type in ("punet", "punet_inter"):
        return PUNetExpert(cfg)
    if cfg.type in ("pmoe", "pmoe+pretrained"):
        return PMoE(cfg)
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
