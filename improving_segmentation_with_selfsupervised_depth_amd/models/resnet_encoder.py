"""Mirror of models/resnet_encoder.py with its own ResNet-v1.5 (the reference delegates to torchvision 0.7.0:
BasicBlock / Bottleneck / ResNet._make_layer -- restated here from the published algorithm; module names follow
torchvision so state_dict keys match: conv1, bn1, layerL.B.convK / bnK / downsample.0 / downsample.1)."""
import numpy as np
import torch
from torch import nn

from .. import functional as Fn
from .layers import Conv2d, BatchNorm2d


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x, x_ds=None, box=None):
        """x_ds / box: this block's input is a tensor several consumers read (an encoder feature: the skip source of the
        decoders): x and x_ds are two of its Fn.FanoutFn views -- for conv1 and for the downsample convolution -- and ``box``
        the box through which the consumers' data-gradients accumulate in their kernels"""
        if self.downsample is None and x.requires_grad:
            # identity skip: conv1's data-gradient lands on the skip-path gradient inside its kernel (Fn.SplitFn)
            box = {}
            xm, xs = Fn.SplitFn.apply(x, box)
            o = self.bn1(self.conv1(xm, grad_box=box), act="relu")
            return self.bn2(self.conv2(o), residual=xs, act="relu", grad_box=box)
        if self.downsample is not None and x_ds is None and x.requires_grad:
            (x, x_ds), box = Fn.fan_feature(x, 0, 2)       # conv1 and the downsample convolution read x: one summed gradient
        idt = x if self.downsample is None else self.downsample[1](self.downsample[0](x if x_ds is None else x_ds, grad_box=box))
        o = self.bn1(self.conv1(x, grad_box=box), act="relu")
        return self.bn2(self.conv2(o), residual=idt, act="relu")


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x, x_ds=None, box=None):
        """x_ds / box: see BasicBlock.forward"""
        if self.downsample is None and x.requires_grad:
            box = {}
            xm, xs = Fn.SplitFn.apply(x, box)
            o = self.bn1(self.conv1(xm, grad_box=box), act="relu")
            o = self.bn2(self.conv2(o), act="relu")
            return self.bn3(self.conv3(o), residual=xs, act="relu", grad_box=box)
        if self.downsample is not None and x_ds is None and x.requires_grad:
            (x, x_ds), box = Fn.fan_feature(x, 0, 2)
        idt = x if self.downsample is None else self.downsample[1](self.downsample[0](x if x_ds is None else x_ds, grad_box=box))
        o = self.bn1(self.conv1(x, grad_box=box), act="relu")
        o = self.bn2(self.conv2(o), act="relu")
        return self.bn3(self.conv3(o), residual=idt, act="relu")


class ResNet(nn.Module):
    def __init__(self, block, layers, num_input_images=1, replace_stride_with_dilation=None):
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        rswd = replace_stride_with_dilation or [False, False, False]
        self.conv1 = Conv2d(3 * num_input_images, 64, 7, 2, 3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)     # parameter-free placeholder; the HIP max-pool runs in forward
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=rswd[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=rswd[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=rswd[2])
        self.avgpool = nn.Identity()             # models/utils.py:46-47 replaces avgpool / fc by Identity
        self.fc = nn.Identity()
        for m in self.modules():                 # torchvision / resnet_encoder.py:36-41 initialisation
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
            down = nn.Sequential(Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 BatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, down, prev)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            seq.append(block(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*seq)


_SPECS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
          101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


def resnet_multiimage_input(num_layers, pretrained=False, num_input_images=1):
    """reference resnet_encoder.py:44-61"""
    assert num_layers in [18, 50], "Can only run with 18 or 50 layer resnet"
    if pretrained:
        raise RuntimeError("ImageNet weights need the torchvision model zoo (no network here); load a state_dict instead")
    block, layers = _SPECS[num_layers]
    return ResNet(block, layers, num_input_images=num_input_images)


class ResnetEncoder(nn.Module):
    """reference resnet_encoder.py:64-101.  forward(input_image NCHW in [0,1]) -> list of 5 NCHW-logical feature
    maps (channels-last memory)."""

    def __init__(self, num_layers, pretrained, num_input_images=1, **kwargs):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers not in _SPECS:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if pretrained:
            raise RuntimeError("ImageNet weights need the torchvision model zoo (no network here); load a state_dict instead")
        if num_input_images > 1:
            self.encoder = resnet_multiimage_input(num_layers, pretrained, num_input_images)
        else:
            block, layers = _SPECS[num_layers]
            self.encoder = ResNet(block, layers, 1, kwargs.get("replace_stride_with_dilation"))
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4

    def forward_nhwc(self, input_image):
        e = self.encoder
        # (x - 0.45) / 0.225 fused with the layout change; channels padded 3 -> 4 / 6 -> 8 with zeros so that the stem's
        # tile loads are 16-byte gathers (K = 49*4 instead of 49*3, but 2.5x faster than the scalar gather)
        # -- or, for the reference's own stem geometry, written with the zero border of the 7x7 window around it and
        # convolved by the dedicated stem kernel (Conv2d.forward_image)
        y0 = e.conv1.forward_image(input_image, 0.45, 0.225)
        if y0 is None:
            y0 = e.conv1(Fn.to_nhwc(input_image, 0.45, 0.225, pad_to=4))
        f0 = e.bn1(y0, act="relu")
        feats = [f0]
        # The features are read again as the skip sources of ``self.skip_consumers`` decoders (set by the model glue): every
        # feature becomes a set of Fn.FanoutFn views -- the ones for the next stage are used here, the decoders fetch theirs with
        # Fn.take_fan_view -- so that the consumers' data-gradients accumulate in their kernels instead of being
        # summed by autograd with one full-tensor pass per extra consumer.
        n_dec = int(getattr(self, "skip_consumers", 0))
        # ``defer_backward`` (set through JointSegmentationMonodepth.defer_trunk_backward): the decoders read the features behind a
        # gate that runs this encoder's backward once per forward, however many backward() calls the step makes (Fn.defer_trunk)
        defer = bool(getattr(self, "defer_backward", False)) and torch.is_grad_enabled()
        n = 1 if defer else n_dec          # outside consumers of every feature: the gate alone, or the decoders
        Fn.release_fans(id(self))          # views of the previous forward that no decoder came for
        (x0,), box0 = Fn.fan_feature(f0, n, 1, owner=id(self))
        x = Fn.MaxPoolFn.apply(x0, box0)
        layers = (e.layer1, e.layer2, e.layer3, e.layer4)
        x_ds, box = None, None
        for li, layer in enumerate(layers):
            for bi, blk in enumerate(layer):
                x = blk(x, x_ds, box) if (bi == 0 and x_ds is not None) else blk(x)
            feats.append(x)
            x_ds, box = None, None
            if li + 1 < len(layers) and n > 0 and layers[li + 1][0].downsample is not None:
                (x, x_ds), box = Fn.fan_feature(x, n, 2, owner=id(self))      # conv1 and the downsample convolution of the next stage + the decoders
                if box is None:
                    x_ds = None
        if defer:
            feats = Fn.defer_trunk(feats, id(self), n_dec)
        return feats

    def forward(self, input_image):
        self.features = [Fn.to_nchw(f) for f in self.forward_nhwc(input_image)]
        return self.features
