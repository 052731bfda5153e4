"""Mirror of models/pose_decoder.py (keys net.0..3)."""
from collections import OrderedDict

from torch import nn

from .. import functional as Fn
from .layers import Conv2d


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_input_features = num_input_features
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.convs = OrderedDict()
        self.convs[("squeeze")] = Conv2d(int(self.num_ch_enc[-1]), 256, 1)
        self.convs[("pose", 0)] = Conv2d(num_input_features * 256, 256, 3, stride, 1)
        self.convs[("pose", 1)] = Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 2)] = Conv2d(256, 6 * num_frames_to_predict_for, 1)
        self.relu = nn.ReLU()
        self.net = nn.ModuleList(list(self.convs.values()))

    def forward(self, input_features):
        """input_features: list (one per input) of feature lists (NCHW-logical); reference pose_decoder.py:41-58"""
        last = [Fn.to_nhwc(f[-1]) for f in input_features]
        cat = [self.convs["squeeze"](f, act="relu") for f in last]
        out = cat[0] if len(cat) == 1 else Fn.ConcatFn.apply(*cat)
        out = self.convs[("pose", 0)](out, act="relu")
        out = self.convs[("pose", 1)](out, act="relu")
        out = self.convs[("pose", 2)](out)
        out = Fn.GlobalAvgPoolFn.apply(out)                               # out.mean(3).mean(2)
        out = Fn.ScaleSliceFn.apply(out.reshape(-1, self.num_frames_to_predict_for, 1, 6), 0.01)
        return out[..., :3], out[..., 3:]
