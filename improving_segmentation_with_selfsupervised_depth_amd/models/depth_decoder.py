"""Mirror of models/depth_decoder.py (same constructor, same positional ModuleList => same state_dict keys).

MI355X-first differences in *how* it runs: the nearest x2 upsample and the channel concat in front of every
("upconv", i, 1) ConvBlock (depth_decoder.py:93-101) are folded into that convolution's tile loader (two-source
implicit GEMM); ELU / sigmoid are conv epilogues; tensors are NHWC."""
from collections import OrderedDict

import numpy as np
from torch import nn

from .. import functional as Fn
from .layers import Conv2d, BatchNorm2d
from .model_parts import ASPP
from .monodepth_layers import ConvBlock, Conv3x3


class _SkipProj(nn.Sequential):
    def forward(self, x):
        return self[1](self[0](x), act="relu")


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales, max_scale_size, num_output_channels=1, use_skips=True,
                 intermediate_aspp=False, aspp_rates=[6, 12, 18], num_ch_dec=[16, 32, 64, 128, 256],
                 n_upconv=4, batch_norm=False, dropout=0.0, n_project_skip_ch=-1, aspp_pooling=True):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = "nearest"
        self.scales = scales
        self.enable_disparity = True
        self.max_scale_size = np.asarray(max_scale_size)   # only used by the reference's debug prints (:108)
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array(num_ch_dec)
        self.n_upconv = n_upconv
        self.convs = OrderedDict()
        for i in range(self.n_upconv, -1, -1):
            cin = self.num_ch_enc[-1] if i == self.n_upconv else self.num_ch_dec[i + 1]
            cout = self.num_ch_dec[i]
            if i == self.n_upconv and intermediate_aspp:
                self.convs[("upconv", i, 0)] = ASPP(cin, aspp_rates, aspp_pooling, cout)
            else:
                self.convs[("upconv", i, 0)] = ConvBlock(cin, cout, bn=batch_norm, dropout=dropout)
            cin = self.num_ch_dec[i]
            if self.use_skips and i > 0:
                if n_project_skip_ch == -1:
                    cin += self.num_ch_enc[i - 1]
                    self.convs[("skip_proj", i)] = nn.Identity()
                else:
                    cin += n_project_skip_ch
                    self.convs[("skip_proj", i)] = _SkipProj(Conv2d(int(self.num_ch_enc[i - 1]), n_project_skip_ch, 1),
                                                             BatchNorm2d(n_project_skip_ch), nn.ReLU(inplace=True))
            self.convs[("upconv", i, 1)] = ConvBlock(cin, self.num_ch_dec[i], bn=batch_norm, dropout=dropout)
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()

    def forward_nhwc(self, feats, x=None, exec_layer=None):
        """feats: list of NHWC encoder features; returns dict of NHWC tensors"""
        out = {}
        if x is None:
            x = feats[-1]
        for i in range(self.n_upconv, -1, -1):
            if exec_layer is not None and exec_layer != "all" and i not in exec_layer:
                continue
            x = self.convs[("upconv", i, 0)](x)
            up = x.shape[2] < feats[i - 1].shape[2] or i == 0
            skip, sbox = None, None
            if self.use_skips and i > 0:
                proj = self.convs[("skip_proj", i)]
                if isinstance(proj, nn.Identity):
                    # the encoder feature is read by several consumers: this decoder takes its own autograd view of it and the
                    # box through which the consumers' data-gradients accumulate in their kernels (Fn.fan_feature)
                    skip, sbox = Fn.take_fan_view(feats[i - 1])
                else:
                    skip = proj(feats[i - 1])
            x = self.convs[("upconv", i, 1)](x, skip, up, skip_box=sbox)
            out[("upconv", i)] = x
            if i in self.scales and self.enable_disparity:
                out[("disp", i)] = self.convs[("dispconv", i)](x, act="sigmoid")
        return out

    def takes_fan_views(self):
        """does forward_nhwc fetch the encoder features' gradient-collector views (identity skip projections do)"""
        return self.use_skips and any(isinstance(self.convs[("skip_proj", i)], nn.Identity) for i in range(1, self.n_upconv + 1)
                                      if ("skip_proj", i) in self.convs)

    def forward(self, input_features, x=None, exec_layer=None):
        feats = [Fn.to_nhwc(f) for f in input_features]
        xn = None if x is None else Fn.to_nhwc(x)
        self.outputs = {k: Fn.to_nchw(v) for k, v in self.forward_nhwc(feats, xn, exec_layer).items()}
        return self.outputs
