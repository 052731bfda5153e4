"""Mirror of models/model_parts.py (ASPP, SelfAttention).  ASPPConv / ASPPPooling are torchvision 0.7.0 classes in
the reference (model_parts.py:3); restated here from the published algorithm with the same Sequential indices so
the state_dict keys (convs.K.0.weight, convs.K.1.*, convs.4.1.weight, convs.4.2.*, project.0/1) match."""
import torch
from torch import nn

from .. import functional as Fn
from .layers import Conv2d, BatchNorm2d


class ASPPConv(nn.Sequential):
    def __init__(self, in_channels, out_channels, dilation):
        super().__init__(Conv2d(in_channels, out_channels, 3, padding=dilation, dilation=dilation, bias=False),
                         BatchNorm2d(out_channels), nn.ReLU())

    def forward(self, x, grad_box=None):
        return self[1](self[0](x, grad_box=grad_box), act="relu")


class ASPPPooling(nn.Sequential):
    def __init__(self, in_channels, out_channels):
        super().__init__(nn.AdaptiveAvgPool2d(1), Conv2d(in_channels, out_channels, 1, bias=False),
                         BatchNorm2d(out_channels), nn.ReLU())

    def forward(self, x):
        size = (x.shape[1], x.shape[2])
        g = Fn.GlobalAvgPoolFn.apply(x)
        g = self[2](self[1](g), act="relu")
        return Fn.ResizeFn.apply(g, size, False)


class ASPP(nn.Module):
    """reference model_parts.py:5-32"""

    def __init__(self, in_channels, atrous_rates, aspp_pooling=True, out_channels=256):
        super().__init__()
        in_channels, out_channels = int(in_channels), int(out_channels)
        modules = [nn.Sequential(Conv2d(in_channels, out_channels, 1, bias=False), BatchNorm2d(out_channels), nn.ReLU())]
        for r in atrous_rates:
            modules.append(ASPPConv(in_channels, out_channels, r))
        if aspp_pooling:
            modules.append(ASPPPooling(in_channels, out_channels))
        self.convs = nn.ModuleList(modules)
        self.project = nn.Sequential(
            Conv2d((1 + int(aspp_pooling) + len(atrous_rates)) * out_channels, out_channels, 1, bias=False),
            BatchNorm2d(out_channels), nn.ReLU(), nn.Dropout(0.5))

    def forward(self, x):
        branches = list(self.convs)
        if x.requires_grad:
            # the branches read the same tensor: their data-gradients are accumulated in the conv epilogues (Fn.FanoutFn)
            box = {}
            xs = Fn.FanoutFn.apply(x, box, len(branches))
        else:
            box, xs = None, [x] * len(branches)
        res = [self.convs[0][1](self.convs[0][0](xs[0], grad_box=box), act="relu")]
        for conv, xi in zip(branches[1:], xs[1:]):
            res.append(conv(xi, grad_box=box) if isinstance(conv, ASPPConv) else conv(xi))
        cat = Fn.ConcatFn.apply(*res)
        drop = self.project[3]
        p = drop.p if (drop.training and self.training) else 0.0
        return self.project[1](self.project[0](cat), act="relu", drop_p=p)


class SelfAttention(nn.Module):
    """reference model_parts.py:35-46: conv3x3(x) * sigmoid(conv3x3(x)), attention weights zero-initialised"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv2d(in_channels, out_channels, 3, padding=1, bias=False)
        self.attention = Conv2d(in_channels, out_channels, 3, padding=1, bias=False)
        with torch.no_grad():
            self.attention.weight.zero_()

    def forward(self, x):
        return Fn.GateFn.apply(self.conv(x), self.attention(x))
