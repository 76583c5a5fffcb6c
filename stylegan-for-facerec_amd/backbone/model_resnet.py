"""``from backbone.model_resnet import ResNet_50, ResNet_101, ResNet_152`` (reference train.py:6, model_resnet.py:91-188).

Plain PyTorch, OFF the accelerated path: these bottleneck ResNets with the face-recognition output head are imported
by the reference driver but named by no shipped config and not by BASELINE.json ("must import;
plain PyTorch is fine").  Importable and state-dict compatible only: the classes carry the reference's keys, shapes and
initialisation (pinned by tests/golden/g12_resnet_structure.json) and run through ATen / MIOpen like any torch model,
but this repo's ``train.py`` does not wire them up (``build_backbone`` raises for ``BACKBONE_NAME = 'ResNet_*'``: its
loop is built around the frhip runner, ``separate_irse_bn_paras`` and the gradient arena).
"""
import torch.nn as nn

_STAGE_PLANES = (64, 128, 256, 512)
_EXPANSION = 4


class Bottleneck(nn.Module):
    """1x1 reduce -> 3x3 (stride) -> 1x1 expand, each followed by BN; ReLU after the first two and after the sum."""
    expansion = _EXPANSION

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        wide = planes * _EXPANSION
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, wide, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(wide)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNet(nn.Module):
    def __init__(self, input_size, block, layers, zero_init_residual=True):
        super().__init__()
        assert input_size[0] in [112, 224], "input_size should be [112, 112] or [224, 224]"
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        for i, (planes, n) in enumerate(zip(_STAGE_PLANES, layers)):
            setattr(self, "layer%d" % (i + 1), self._make_layer(block, planes, n, stride=1 if i == 0 else 2))
        side = input_size[0] // 28  # 112 -> 4, 224 -> 8 after the five stride-2 steps
        self.bn_o1 = nn.BatchNorm2d(512 * block.expansion)
        self.dropout = nn.Dropout()
        self.fc = nn.Linear(512 * block.expansion * side * side, 512)
        self.bn_o2 = nn.BatchNorm1d(512)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:  # every residual branch starts as zero, the unit as an identity
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        wide = planes * block.expansion
        down = None
        if stride != 1 or self.inplanes != wide:
            down = nn.Sequential(nn.Conv2d(self.inplanes, wide, 1, stride, bias=False), nn.BatchNorm2d(wide))
        units = [block(self.inplanes, planes, stride, down)] + [block(wide, planes) for _ in range(1, blocks)]
        self.inplanes = wide
        return nn.Sequential(*units)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.dropout(self.bn_o1(x))
        return self.bn_o2(self.fc(x.view(x.size(0), -1)))


def ResNet_50(input_size, **kwargs):
    return ResNet(input_size, Bottleneck, [3, 4, 6, 3], **kwargs)


def ResNet_101(input_size, **kwargs):
    return ResNet(input_size, Bottleneck, [3, 4, 23, 3], **kwargs)


def ResNet_152(input_size, **kwargs):
    return ResNet(input_size, Bottleneck, [3, 8, 36, 3], **kwargs)
