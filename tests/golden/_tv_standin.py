"""Minimal stand-in for the parts of torchvision 0.7.0 that the *reference*
imports (models/resnet_encoder.py:16, models/model_parts.py:3, models/utils.py:9).

Used ONLY by make_golden.py, in the build container, to let the reference's own
model-assembly code run so that golden vectors capture its wiring.  torchvision
is absent from /root/reference and from this image (requirements.txt:2), so the
block arithmetic below is OUR restatement of the published torchvision
algorithm -- vectors that flow through these classes pin the reference's wiring
(which layers, order, normalisation, dilation flags, key names), not
torchvision's arithmetic ("parity unpinned" at that boundary).
"""
import sys
import types

import torch
from torch import nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.bn2(self.conv2(o))
        return self.relu(o + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.relu(self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        return self.relu(o + idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, replace_stride_with_dilation=None):
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        rswd = replace_stride_with_dilation or [False, False, False]
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=rswd[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=rswd[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=rswd[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        prev = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, down, prev)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            seq.append(block(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*seq)


def _factory(block, layers):
    def make(pretrained=False, **kw):
        assert not pretrained, "no network in the build container"
        return ResNet(block, layers, **kw)
    return make


class ASPPConv(nn.Sequential):
    def __init__(self, cin, cout, dilation):
        super().__init__(nn.Conv2d(cin, cout, 3, padding=dilation, dilation=dilation, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU())


class ASPPPooling(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout),
                         nn.ReLU())

    def forward(self, x):
        size = x.shape[-2:]
        for m in self:
            x = m(x)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


def install():
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    resnet = types.ModuleType("torchvision.models.resnet")
    utils_ = types.ModuleType("torchvision.models._utils")
    seg = types.ModuleType("torchvision.models.segmentation")
    dl = types.ModuleType("torchvision.models.segmentation.deeplabv3")
    resnet.BasicBlock, resnet.Bottleneck, resnet.ResNet = BasicBlock, Bottleneck, ResNet
    resnet.model_urls = {}
    models.ResNet = ResNet
    models.resnet = resnet
    models.resnet18 = _factory(BasicBlock, [2, 2, 2, 2])
    models.resnet34 = _factory(BasicBlock, [3, 4, 6, 3])
    models.resnet50 = _factory(Bottleneck, [3, 4, 6, 3])
    models.resnet101 = _factory(Bottleneck, [3, 4, 23, 3])
    models.resnet152 = _factory(Bottleneck, [3, 8, 36, 3])
    utils_.IntermediateLayerGetter = object
    dl.ASPPConv, dl.ASPPPooling = ASPPConv, ASPPPooling
    seg.deeplabv3 = dl
    models._utils, models.segmentation = utils_, seg
    tv.models = models
    for m in (tv, models, resnet, utils_, seg, dl):
        sys.modules[m.__name__] = m
    sys.modules["kornia"] = types.ModuleType("kornia")
