"""Backbone definitions with torchvision-compatible module / state-dict names.

The reference encoder (pytorch/bts.py:268-320) pulls its backbone from
``torchvision.models`` and depends on the *names* of the backbone's children:
``feat_names`` substring matching (bts.py:275-296, 309-319), the ``'fc'`` /
``'avgpool'`` skip (bts.py:310), and the freeze lists in bts_main.py:217-247
(``conv0``, ``norm``, ``denseblock1.denselayer1`` ... / ``base_model.conv1``,
``.bn``, ``base_model.layer1.0`` ...).  torchvision is not installed in the
target image and pretrained weights are unreachable offline, so this module
provides the same architectures, built from stock ``torch.nn`` layers, with the
same child names so that checkpoints keyed ``module.encoder.base_model.<name>``
load unchanged.  ``pretrained=True`` is accepted and ignored (random init).

The encoder stays stock PyTorch-ROCm by design (BASELINE.json north_star); no
HIP code here.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = [
    "densenet121", "densenet161", "densenet169", "densenet201",
    "resnet50", "resnet101", "resnext50_32x4d", "resnext101_32x8d",
    "mobilenet_v2",
]


# --------------------------------------------------------------------------
# DenseNet (Huang et al. 2017), torchvision naming
# --------------------------------------------------------------------------
class _DenseLayer(nn.Module):
    def __init__(self, c_in, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(c_in)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(c_in, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, feats):
        x = feats if torch.is_tensor(feats) else torch.cat(feats, 1)
        x = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(x)))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, n_layers, c_in, growth, bn_size):
        super().__init__()
        for i in range(n_layers):
            self["denselayer%d" % (i + 1)] = _DenseLayer(c_in + i * growth, growth, bn_size)

    def forward(self, x):
        feats = [x]
        for layer in self.values():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    def __init__(self, c_in, c_out):
        super().__init__(OrderedDict([
            ("norm", nn.BatchNorm2d(c_in)),
            ("relu", nn.ReLU(inplace=True)),
            ("conv", nn.Conv2d(c_in, c_out, 1, bias=False)),
            ("pool", nn.AvgPool2d(2, 2)),
        ]))


class DenseNet(nn.Module):
    def __init__(self, growth, blocks, c_init, bn_size=4, num_classes=1000):
        super().__init__()
        mods = OrderedDict([
            ("conv0", nn.Conv2d(3, c_init, 7, stride=2, padding=3, bias=False)),
            ("norm0", nn.BatchNorm2d(c_init)),
            ("relu0", nn.ReLU(inplace=True)),
            ("pool0", nn.MaxPool2d(3, stride=2, padding=1)),
        ])
        c = c_init
        for i, n in enumerate(blocks):
            mods["denseblock%d" % (i + 1)] = _DenseBlock(n, c, growth, bn_size)
            c += n * growth
            if i != len(blocks) - 1:
                mods["transition%d" % (i + 1)] = _Transition(c, c // 2)
                c //= 2
        mods["norm5"] = nn.BatchNorm2d(c)
        self.features = nn.Sequential(mods)
        self.classifier = nn.Linear(c, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = F.relu(self.features(x), inplace=True)
        x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        return self.classifier(x)


def densenet121(pretrained=False, **kw):
    return DenseNet(32, (6, 12, 24, 16), 64)


def densenet161(pretrained=False, **kw):
    return DenseNet(48, (6, 12, 36, 24), 96)


def densenet169(pretrained=False, **kw):
    return DenseNet(32, (6, 12, 32, 32), 64)


def densenet201(pretrained=False, **kw):
    return DenseNet(32, (6, 12, 48, 32), 64)


# --------------------------------------------------------------------------
# ResNet / ResNeXt bottleneck family, torchvision naming
# --------------------------------------------------------------------------
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, c_in, planes, stride, groups, base_width, downsample):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = nn.Conv2d(c_in, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, layers, groups=1, width_per_group=64, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        c = 64
        for i, (n, planes) in enumerate(zip(layers, (64, 128, 256, 512))):
            stride = 1 if i == 0 else 2
            blocks = []
            for j in range(n):
                s = stride if j == 0 else 1
                ds = None
                if j == 0 and (s != 1 or c != planes * 4):
                    ds = nn.Sequential(nn.Conv2d(c, planes * 4, 1, stride=s, bias=False),
                                       nn.BatchNorm2d(planes * 4))
                blocks.append(Bottleneck(c, planes, s, groups, width_per_group, ds))
                c = planes * 4
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(c, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet50(pretrained=False, **kw):
    return ResNet((3, 4, 6, 3))


def resnet101(pretrained=False, **kw):
    return ResNet((3, 4, 23, 3))


def resnext50_32x4d(pretrained=False, **kw):
    return ResNet((3, 4, 6, 3), groups=32, width_per_group=4)


def resnext101_32x8d(pretrained=False, **kw):
    return ResNet((3, 4, 23, 3), groups=32, width_per_group=8)


# --------------------------------------------------------------------------
# MobileNetV2 (Sandler et al. 2018), torchvision naming: features[0..18]
# --------------------------------------------------------------------------
class _ConvBNReLU6(nn.Sequential):
    def __init__(self, c_in, c_out, k=3, stride=1, groups=1):
        super().__init__(nn.Conv2d(c_in, c_out, k, stride, (k - 1) // 2, groups=groups, bias=False),
                         nn.BatchNorm2d(c_out), nn.ReLU6(inplace=True))


class _InvertedResidual(nn.Module):
    def __init__(self, c_in, c_out, stride, expand):
        super().__init__()
        hidden = int(round(c_in * expand))
        self.use_res = stride == 1 and c_in == c_out
        layers = []
        if expand != 1:
            layers.append(_ConvBNReLU6(c_in, hidden, k=1))
        layers += [_ConvBNReLU6(hidden, hidden, stride=stride, groups=hidden),
                   nn.Conv2d(hidden, c_out, 1, bias=False), nn.BatchNorm2d(c_out)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res else self.conv(x)


class MobileNetV2(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        cfg = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
               (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
        feats = [_ConvBNReLU6(3, 32, stride=2)]
        c = 32
        for t, co, n, s in cfg:
            for i in range(n):
                feats.append(_InvertedResidual(c, co, s if i == 0 else 1, t))
                c = co
        feats.append(_ConvBNReLU6(c, 1280, k=1))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_classes))

    def forward(self, x):
        x = self.features(x)
        return self.classifier(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


def mobilenet_v2(pretrained=False, **kw):
    return MobileNetV2()
