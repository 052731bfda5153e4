"""Mirror of models/model_parts.py (ASPP, SelfAttention).  ASPPConv / ASPPPooling are torchvision 0.7.0 classes in
the reference (model_parts.py:3); restated here from the published algorithm with the same Sequential indices so
the state_dict keys (convs.K.0.weight, convs.K.1.*, convs.4.1.weight, convs.4.2.*, project.0/1) match."""
import torch
from torch import nn

import os

from .. import functional as Fn
from .. import hipops as H
from .layers import Conv2d, BatchNorm2d

_ASPP_STREAMS = os.environ.get("SEGSDE_ASPP_STREAMS", "0") == "1"
_SIDE_STREAMS = {}


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
        if _ASPP_STREAMS and x.is_cuda:
            res = self._branches_on_side_streams(branches, xs, box)
        else:
            res = [self.convs[0][1](self.convs[0][0](xs[0], grad_box=box), act="relu")]
            for conv, xi in zip(branches[1:], xs[1:]):
                res.append(conv(xi, grad_box=box) if isinstance(conv, ASPPConv) else conv(xi))
        cat = Fn.ConcatFn.apply(*res)
        drop = self.project[3]
        p = drop.p if (drop.training and self.training) else 0.0
        return self.project[1](self.project[0](cat), act="relu", drop_p=p)


    def _branches_on_side_streams(self, branches, xs, box):
        """EXPERIMENT (SEGSDE_ASPP_STREAMS=1): the branch convolutions' FORWARD launches go to one side stream each.  A dilated
        branch is 512 workgroups = one round of the chip whose tiles skip different numbers of dead tap rows (conv_igemm.hip,
        ConvP::tapskip): alone it takes as long as its longest tile pair; next to the other branches the CUs that finish early
        pick up their workgroups.  Only the convolution kernels move (their outputs are kept by autograd, nothing is freed
        before the join); BatchNorm / ReLU and the whole backward stay on the current stream."""
        dev = xs[0].device
        main = torch.cuda.current_stream(dev)
        side = _SIDE_STREAMS.setdefault(dev.index, [])
        while len(side) < len(branches):
            side.append(torch.cuda.Stream(dev))
        fork = torch.cuda.Event()
        fork.record(main)
        raw = []
        for i, (br, xi) in enumerate(zip(branches, xs)):
            if isinstance(br, ASPPPooling):
                raw.append(None)
                continue
            side[i].wait_event(fork)
            H.LAUNCH_STREAM = side[i]
            try:
                raw.append(br[0](xi, grad_box=box))
            finally:
                H.LAUNCH_STREAM = None
        for i, r in enumerate(raw):
            if r is not None:
                main.wait_stream(side[i])
        return [br(xi) if r is None else br[1](r, act="relu") for br, xi, r in zip(branches, xs, raw)]


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
