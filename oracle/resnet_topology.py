"""Restatement of the published torchvision (0.9.1) ResNet-18/34 topology.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torchvision is a pinned third-party dependency of
the reference (``requirements.txt:97``: torchvision==0.9.1+cu111; call site
``PMoE/model/blocks/backbone.py:57-61``) that is neither vendored in /root/reference nor
installed in this image, so its algorithm is restated here from its public definition:

    conv1(7x7/s2, replaced by the reference) -> bn1 -> relu -> maxpool(3, s2, p1)
    -> layer1..4 (BasicBlock x [2,2,2,2], planes 64/128/256/512, strides 1/2/2/2)
    -> avgpool(1x1) -> flatten -> fc

    BasicBlock: out = relu(bn2(conv2(relu(bn1(conv1(x))))) + (downsample(x) or x))
    downsample = Sequential(conv1x1(stride), BatchNorm2d) when stride != 1 or planes change.

Attribute names are torchvision's because ``backbone.py:63-70`` touches ``model.conv1.out_channels``
and ``model.fc.in_features`` and replaces ``model.conv1`` / ``model.fc``; they also generate the
state_dict keys listed in SURVEY.md section 8b.  "Parity unpinned" at this boundary: the reference
ships no tests or golden vectors for it.

This module doubles as the ``torchvision.models`` stand-in that ``oracle/make_golden.py`` injects
into ``sys.modules`` before importing the reference.
"""
import torch
import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class ResNet(nn.Module):
    def __init__(self, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        # torchvision's init: kaiming-normal(fan_out, relu) for convs, BN gamma=1 beta=0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes),
            )
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


def resnet18(pretrained=False, **kw):
    if pretrained:
        raise RuntimeError("no network: ImageNet weights cannot be downloaded; weights come from state_dict")
    return ResNet([2, 2, 2, 2])


def resnet34(pretrained=False, **kw):
    if pretrained:
        raise RuntimeError("no network: ImageNet weights cannot be downloaded; weights come from state_dict")
    return ResNet([3, 4, 6, 3])


def _unsupported(*a, **k):
    raise NotImplementedError("stand-in exposes resnet18/34 only")


resnet50 = mobilenet_v2 = mobilenet_v3_small = mobilenet_v3_large = _unsupported
