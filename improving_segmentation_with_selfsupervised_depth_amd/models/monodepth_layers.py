"""Mirror of the reference's models/monodepth_layers.py on the HIP ops (same names / constructor arguments).

Hot-path pieces (ConvBlock, Conv3x3, transformation_from_parameters) run on HIP kernels.  The geometry/SSIM
layers that the reference's MonodepthLoss composes out of many small ATen ops (BackprojectDepth, Project3D, SSIM,
get_smooth_loss -- monodepth_layers.py:145-254) are folded into this package's fused loss kernels
(loss/monodepth_loss.py); small API-parity helpers remain for the inference-side callers."""
import torch
from torch import nn

from .. import functional as Fn
from .layers import Conv2d, BatchNorm2d


def disp_to_depth(disp, min_depth, max_depth):
    """reference monodepth_layers.py:18-27 (inference-side helper; the training path computes this in-kernel)"""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return scaled_disp, 1 / scaled_disp


def transformation_from_parameters(axisangle, translation, invert=False):
    """reference monodepth_layers.py:30-46.  axisangle / translation: [B,1,3]."""
    B = axisangle.shape[0]
    aa = axisangle.reshape(B, 1, 1, 3)
    tr = translation.reshape(B, 1, 1, 3)
    return Fn.PoseMatrixFn.apply(aa, tr, bool(invert))


class Conv3x3(nn.Module):
    """reference monodepth_layers.py:127-142: reflection pad 1 + 3x3 conv (pad handled inside the conv kernel)"""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)   # kept for module-tree parity, no params
        self.conv = Conv2d(int(in_channels), int(out_channels), 3, padding=1, reflect=use_refl)

    def forward(self, x, skip=None, up=False, act="none"):
        return self.conv(x, skip, up, act)


class ConvBlock(nn.Module):
    """reference monodepth_layers.py:108-124: Conv3x3 -> [BN] -> ELU -> [Dropout2d]"""

    def __init__(self, in_channels, out_channels, bn=False, dropout=0.0):
        super().__init__()
        self.block = nn.Sequential(
            Conv3x3(in_channels, out_channels),
            BatchNorm2d(int(out_channels)) if bn else nn.Identity(),
            nn.ELU(inplace=True),
            # "Pay attention: 2d version of dropout is used" (reference :117-119)
            nn.Dropout2d(dropout) if dropout > 0 else nn.Identity(),
        )
        self.bn = bn

    def forward(self, x, skip=None, up=False):
        if self.bn:
            y = self.block[1](self.block[0](x, skip, up), act="elu")
        else:
            y = self.block[0](x, skip, up, act="elu")
        drop = self.block[3]
        if isinstance(drop, nn.Dropout2d) and drop.training and self.training and drop.p > 0:
            # whole channel maps are zeroed with probability p and the survivors scaled by 1 / (1 - p); the Bernoulli draw
            # is torch's (device RNG plumbing), the application and its adjoint are HIP
            B, C = y.shape[0], y.shape[3]
            keep = torch.bernoulli(torch.full((B, C), 1.0 - drop.p, device=y.device))
            y = Fn.ChannelDropFn.apply(y, keep / (1.0 - drop.p) if drop.p < 1 else keep)
        return y
