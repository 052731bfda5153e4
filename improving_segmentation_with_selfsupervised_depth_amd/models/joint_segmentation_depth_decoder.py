"""Mirror of models/joint_segmentation_depth_decoder.py (JointSegDepthDecoder, PAD)."""
import numpy as np
import torch
from torch import nn

from .. import functional as Fn
from .layers import Conv2d, BatchNorm2d
from .model_parts import SelfAttention
from .utils import _get_layer, get_depth_decoder


class JointSegDepthDecoder(nn.Module):
    """reference :11-75.  Note the reference's private DepthDecoder also computes 4 disparity maps nobody reads
    (:29); they cannot influence any output or gradient, so this implementation skips them."""

    def __init__(self, num_ch_enc, num_ch_dec, num_classes, layers=None, head_inter_channels=64, weights="none",
                 head_dropout=0.1, layer_dropout=0, output_stride=1, layer_out_channels=64, depth_args=None,
                 head_inter=True):
        super().__init__()
        if layers is None:
            layers = [9]
        self.output_stride = output_stride
        self.num_ch_enc, self.num_ch_dec, self.num_classes, self.layers = num_ch_enc, num_ch_dec, num_classes, layers
        assert len(num_ch_enc) == 5 and len(num_ch_dec) == 5
        self.last_layer = len(num_ch_enc) + len(num_ch_dec) - 1
        self.unet_dec = get_depth_decoder(weights, num_ch_enc, **depth_args)
        accumulated_ch = 0
        project = {}
        for layer in layers:
            ch = num_ch_enc[layer] if layer <= 4 else num_ch_dec[self.last_layer - layer]
            accumulated_ch += layer_out_channels
            project["seg%d" % layer] = nn.Sequential(Conv2d(int(ch), layer_out_channels, 1, bias=False))
        self.project = nn.ModuleDict(project)
        self.head_inter = head_inter
        if head_inter:
            head_conv = [Conv2d(accumulated_ch, head_inter_channels, 3, padding=1, bias=False),
                         BatchNorm2d(head_inter_channels), nn.ReLU(), nn.Dropout(head_dropout)]
        else:
            head_conv = [nn.Identity()]
        # reference :50: nn.Dropout on the stacked features (the module keeps its place in the Sequential -- index 0 -- and its
        # train / eval switch; the mask itself is drawn and applied by the HIP dropout kernel, Fn.DropoutFn)
        self.head = nn.Sequential(nn.Dropout(layer_dropout) if layer_dropout > 0 else nn.Identity(), *head_conv,
                                  Conv2d(head_inter_channels, self.num_classes, 1))

    def forward(self, encoder_features):
        feats = [Fn.to_nhwc(f) for f in encoder_features]
        dec = self.unet_dec
        keep = dec.enable_disparity
        dec.enable_disparity = False                 # dead disparity heads (see class docstring)
        try:
            seg = dec.forward_nhwc(feats)
        finally:
            dec.enable_disparity = keep
        seg_size = tuple(_get_layer(feats, seg, self.last_layer).shape[1:3])
        last_size = tuple(int(s) // self.output_stride for s in seg_size)
        stacked = []
        for layer in self.layers:
            y = self.project["seg%d" % layer][0](_get_layer(feats, seg, layer))
            stacked.append(Fn.resize_bilinear(y, last_size, False))
        y = stacked[0] if len(stacked) == 1 else Fn.ConcatFn.apply(*stacked)
        d0 = self.head[0]
        if isinstance(d0, nn.Dropout) and d0.training and self.training and d0.p > 0:
            from .layers import _seed
            y = Fn.DropoutFn.apply(y, float(d0.p), _seed())
        if self.head_inter:
            drop = self.head[4]
            p = drop.p if (drop.training and self.training) else 0.0
            y = self.head[2](self.head[1](y), act="relu", drop_p=p)
            y = self.head[5](y)
        else:
            y = self.head[2](y)
        if last_size != seg_size:
            y = Fn.resize_bilinear(y, seg_size, False)
        return Fn.to_nchw(y)


class PAD(nn.Module):
    """reference :78-184"""

    def __init__(self, num_ch_enc, num_ch_dec, num_classes, final_layer=9, weights=None, output_stride=1,
                 depth_args=None, distillation_layer=7, side_output=True):
        super().__init__()
        self.output_stride = output_stride
        self.num_ch_enc, self.num_ch_dec, self.num_classes = num_ch_enc, num_ch_dec, num_classes
        self.side_output = side_output
        assert len(num_ch_enc) == 5 and len(num_ch_dec) == 5
        self.final_layer = final_layer
        self.last_layer = len(num_ch_enc) + len(num_ch_dec) - 1
        self.distillation_layer = distillation_layer
        self.dec_n_upconv = depth_args.get("n_upconv", 4)
        dch = int(self.layer_channels(distillation_layer))
        fch = int(self.layer_channels(final_layer))
        weights = "none" if weights is None else weights
        self.depth_dec = get_depth_decoder(weights, num_ch_enc, range(4), **depth_args)
        self.seg_dec = get_depth_decoder(weights, num_ch_enc, range(4), **depth_args)
        self.seg_dec.enable_disparity = False
        for s in range(4):
            self.seg_dec.convs[("dispconv", s)] = nn.Identity()   # dict only: the ModuleList keeps the params (:101-103)
        self.sa_depth = SelfAttention(dch, dch)
        self.sa_seg = SelfAttention(dch, dch)
        if self.side_output:
            self.seg_intermediate_head = nn.Sequential(Conv2d(dch, self.num_classes, 1))
        self.seg_final_head = nn.Sequential(Conv2d(fch, self.num_classes, 1))

    def layer_channels(self, layer):
        return self.num_ch_enc[layer] if layer <= 4 else self.num_ch_dec[self.last_layer - layer]

    def depth_params(self):
        return [*self.depth_dec.parameters(), *self.sa_seg.parameters()]

    def segmentation_params(self):
        params = [*self.seg_dec.parameters(), *self.sa_depth.parameters(), *self.seg_final_head.parameters()]
        if self.side_output:
            params.extend(self.seg_intermediate_head.parameters())
        return params

    def forward(self, encoder_features):
        feats = [Fn.to_nhwc(f) for f in encoder_features]
        seg_size = tuple(feats[0].shape[1:3])
        last_size = tuple(int(s) // self.output_stride for s in seg_size)
        di = self.last_layer - self.distillation_layer
        name = ("upconv", di)
        first = list(range(self.dec_n_upconv, di - 1, -1))
        second = list(range(di - 1, -1, -1))
        d = self.depth_dec.forward_nhwc(feats, exec_layer=first)
        s = self.seg_dec.forward_nhwc(feats, exec_layer=first)
        if getattr(self, "defer_backward", False) and torch.is_grad_enabled():
            # One backward per forward under the reference's call-per-loss step (functional.defer_gate).  The two
            # decoders exchange attention maps at the distillation layer, so the monodepth loss reaches BOTH first halves (through
            # `for_depth` and, with the low-resolution disparities, directly) and so does the segmentation loss: without a gate
            # here the bottleneck-side halves -- the two ASPP modules among them -- are back-propagated once per backward() call.
            # Everything the first halves hand on goes behind the gate; its parked backward then runs once, into the encoder's gate.
            dk = list(d)
            gated = Fn.defer_gate([d[k] for k in dk] + [s[name]], id(self))
            d = dict(zip(dk, gated[:-1]))
            s = dict(s)
            s[name] = gated[-1]
        inter = self.seg_intermediate_head[0](s[name]) if self.side_output else None
        fd = self.sa_depth(d[name])
        fs = self.sa_seg(s[name])
        for_seg = Fn.AddFn.apply(s[name], fd)
        for_depth = Fn.AddFn.apply(d[name], fs)
        d.update(self.depth_dec.forward_nhwc(feats, x=for_depth, exec_layer=second))
        s2 = self.seg_dec.forward_nhwc(feats, x=for_seg, exec_layer=second)
        final = self.seg_final_head[0](_get_layer(fd, s2, self.final_layer))
        if self.side_output and last_size != seg_size:
            inter = Fn.resize_bilinear(inter, seg_size, False)
        if last_size != seg_size:
            final = Fn.resize_bilinear(final, seg_size, False)
        out = {k: Fn.to_nchw(v) for k, v in d.items()}
        out["semantics"] = Fn.to_nchw(final)
        if self.side_output:
            out["intermediate_semantics"] = Fn.to_nchw(inter)
        return out
