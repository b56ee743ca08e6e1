"""Parameter containers mirroring ``PMoE/model/blocks/{basics,backbone}.py``.

These modules hold parameters and buffers under exactly the reference's names (so reference
checkpoints load, ``utils.nn.freeze`` name matching works and ``state_dict()`` round-trips) but do
NOT compute: arithmetic for all experts of a mixture is issued layer by layer as grouped HIP
launches by ``pmoe_amd.engine`` -- one launch covers the same layer of every expert.  Calling a
container directly raises, there is no per-module PyTorch path.
"""
import math
from collections import OrderedDict
from math import log2

import torch
import torch.nn as nn


class _Held(nn.Module):
    """Base of the containers: forward is not available piecemeal."""

    def forward(self, *a, **k):
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; it is executed by the grouped HIP engine "
            "through its parent model (MixtureOfExperts / BaseExpert forward)")


class Linear(_Held):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        # nn.Linear's default init (reference uses nn.Linear untouched, basics.py:31)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features) if in_features > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"


class Activation(_Held):
    def __init__(self, kind):
        super().__init__()
        self.kind = kind

    def extra_repr(self):
        return self.kind


class Dropout(_Held):
    def __init__(self, p):
        super().__init__()
        self.p = p

    def extra_repr(self):
        return f"p={self.p}"


class BatchNorm1d(_Held):
    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class BatchNorm2d(BatchNorm1d):
    pass


class MLP(nn.Sequential):
    """``make_mlp`` result: an nn.Sequential so that the state_dict indices follow basics.py:30-42."""

    def forward(self, *a, **k):
        raise RuntimeError("MLP is a parameter container executed by the grouped HIP engine")


def make_mlp(dims, act, l_act=False, bn=True, dropout=0.0):
    """Layout of ``basics.py:10-44``: Linear(bias=not bn) [BN1d] act [Dropout] per hidden layer, bare
    Linear last, optional trailing activation."""
    act = act.lower()
    if act not in ("relu", "tanh", "sigmoid", "elu"):
        raise KeyError(act)
    layers = []
    n = len(dims) - 1
    for i in range(n):
        layers.append(Linear(dims[i], dims[i + 1], bias=not bn))
        if i != n - 1:
            if bn:
                layers.append(BatchNorm1d(dims[i + 1]))
            layers.append(Activation(act))
            if dropout > 0.0:
                layers.append(Dropout(dropout))
    if l_act:
        layers.append(Activation(act))
    m = MLP(*layers)
    m.spec = dict(dims=list(dims), act=act, l_act=bool(l_act), bn=bool(bn), dropout=float(dropout))
    return m


class Conv1d(_Held):
    def __init__(self, k):
        super().__init__()
        self.kernel_size = k
        self.weight = nn.Parameter(torch.empty(1, 1, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class Conv2d(_Held):
    def __init__(self, cin, cout, ks, stride=1, padding=0, init="default", bias=False):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = cin, cout, ks, stride, padding
        self.weight = nn.Parameter(torch.empty(cout, cin, ks, ks))
        if init == "fan_out":       # torchvision ResNet init
            nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        else:                       # nn.Conv2d default (EfficientConvBlock, basics.py:93,113)
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self.bias = None
        if bias:                    # nn.Conv2d default bias init (UNet.out, unet.py:47)
            self.bias = nn.Parameter(torch.empty(cout))
            bound = 1 / math.sqrt(cin * ks * ks)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size}, s={self.stride}, p={self.padding}"


class ConvTranspose2d(_Held):
    """nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2) (unet.py:34-44): weight [cin, cout, 2, 2] + bias [cout]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.weight = nn.Parameter(torch.empty(cin, cout, 2, 2))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(cout * 4)      # torch computes fan_in from dim 1 of the transposed-conv weight
        nn.init.uniform_(self.bias, -bound, bound)


def eca_kernel_size(channels, gamma=2, b=1):
    """basics.py:66-67."""
    t = int(abs((log2(channels) + b) / gamma))
    return t if t % 2 else t + 1


class EfficientBlock(_Held):
    """ECA (basics.py:61-76): holds ``conv.weight [1,1,k]``."""

    def __init__(self, channels, gamma=2, b=1):
        super().__init__()
        self.channels = channels
        self.conv = Conv1d(eca_kernel_size(channels, gamma, b))


class _Seq(nn.Sequential):
    def forward(self, *a, **k):
        raise RuntimeError("parameter container executed by the grouped HIP engine")


class EfficientConvBlock(_Held):
    """basics.py:79-134 -- keys layer1.eca1.conv.weight, layer1.conv1.{0,1}.*, layer2.eca2..., layer2.conv2.{0,1}.*"""

    def __init__(self, in_ch, out_ch, stride=1, gamma=2, b=1):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("EfficientConvBlock stride != 1 is never built by the reference")
        self.in_ch, self.out_ch = in_ch, out_ch
        self.layer1 = _Seq(OrderedDict([
            ("eca1", EfficientBlock(in_ch, gamma, b)),
            ("conv1", _Seq(Conv2d(in_ch, 64, 3, 1, 1), BatchNorm2d(64), Activation("relu")))]))
        self.layer2 = _Seq(OrderedDict([
            ("eca2", EfficientBlock(64, gamma, b)),
            ("conv2", _Seq(Conv2d(64, out_ch, 3, 1, 1), BatchNorm2d(out_ch), Activation("relu")))]))


def conv3(in_ch, out_ch):
    """basics.py:47-58: conv3x3(no bias) BN ReLU conv3x3(no bias) BN ReLU -> state_dict indices 0,1,3,4."""
    return _Seq(Conv2d(in_ch, out_ch, 3, 1, 1), BatchNorm2d(out_ch), Activation("relu"),
                Conv2d(out_ch, out_ch, 3, 1, 1), BatchNorm2d(out_ch), Activation("relu"))


class UNet(_Held):
    """blocks/unet.py:8-95 (parameter container; executed by pmoe_amd.engine_punet)."""

    def __init__(self, in_features=3, out_features=23, gamma=2, b=1, dropout=0.0, inter_repr=False):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("UNet Dropout2d(p>0) is never configured by the reference (punet.py:33-39,62-68)")
        self.inter_repr = inter_repr
        self.dwn_1 = conv3(in_features, 64)
        self.dwn_2 = conv3(64, 128)
        self.dwn_3 = conv3(128, 256)
        self.dwn_4 = conv3(256, 512)
        self.dwn_5 = conv3(512, 512)
        self.pool = Activation("maxpool2s2")
        self.avgpool = Activation("gap")
        self.dropout = Dropout(dropout)
        self.up_1 = ConvTranspose2d(512, 512)
        self.up_forw_1 = conv3(1024, 512)
        self.up_2 = ConvTranspose2d(512, 256)
        self.up_forw_2 = conv3(512, 256)
        self.up_3 = ConvTranspose2d(256, 128)
        self.up_forw_3 = conv3(256, 128)
        self.up_4 = ConvTranspose2d(128, 64)
        self.up_forw_4 = conv3(128, 64)
        self.out = Conv2d(64, out_features, 1, 1, 0, bias=True)


class BasicBlock(_Held):
    """torchvision BasicBlock (public definition; parity unpinned at this boundary)."""

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1, init="fan_out")
        self.bn1 = BatchNorm2d(planes)
        self.relu = Activation("relu")
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, init="fan_out")
        self.bn2 = BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = _Seq(Conv2d(inplanes, planes, 1, stride, 0, init="fan_out"), BatchNorm2d(planes))
        self.stride = stride


class ResNetBackbone(_Held):
    """``_get_resnet`` (backbone.py:48-72) for resnet18/34: conv1 := EfficientConvBlock(stride 1),
    bn1 / relu / maxpool kept, fc := Identity (512-d feature)."""

    def __init__(self, arch="resnet18", n_frames=4, pretrained=False, gamma=2, b=1, n_channels=3):
        super().__init__()
        arch = arch.lower()
        if arch not in ("resnet18", "resnet34"):
            raise NotImplementedError(
                f"backbone arch {arch!r}: only resnet18/resnet34 (BasicBlock, 512-d) run on the HIP engine")
        # `pretrained` would download ImageNet weights in the reference (backbone.py:61); weights come
        # from load_state_dict here, so the flag is accepted and ignored.
        depths = [2, 2, 2, 2] if arch == "resnet18" else [3, 4, 6, 3]
        self.conv1 = EfficientConvBlock(n_frames * n_channels, 64, gamma=gamma, b=b)
        self.bn1 = BatchNorm2d(64)
        self.relu = Activation("relu")
        self.maxpool = Activation("maxpool3s2")
        inpl = 64
        for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], depths), start=1):
            blocks = []
            for bi in range(nblk):
                stride = 2 if (bi == 0 and li > 1) else 1
                blocks.append(BasicBlock(inpl, planes, stride))
                inpl = planes
            setattr(self, f"layer{li}", _Seq(*blocks))
        self.avgpool = Activation("gap")
        self.fc = Activation("identity")


def get_backbone(arch="resnet18", n_frames=4, pretrained=False, gamma=2, b=1, n_channels=3):
    """backbone.py:13-25."""
    if "resnet" in arch:
        return ResNetBackbone(arch, n_frames, pretrained, gamma, b, n_channels)
    raise NotImplementedError(f"backbone arch {arch!r} is not on the HIP path (reference option, unused by stage-2 configs)")
