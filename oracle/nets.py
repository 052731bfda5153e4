"""Oracle: encoder / decoders / pose nets as pure functions over a flat
``state_dict`` (TEST INFRASTRUCTURE, see oracle/__init__.py).

The key names are the reference's ``state_dict`` contract:
``models.encoder.encoder.layer1.0.conv1.weight`` ... .  Because every function
here reads its parameters *by reference key name*, running the oracle on a
state_dict taken from the reference (or from the product modules) is itself a
check of the naming contract.

Reference wiring followed:
  models/resnet_encoder.py:64-101       (encoder forward, input normalisation)
  models/depth_decoder.py:22-116        (U-Net ladder, positional ModuleList keys)
  models/model_parts.py:5-46            (ASPP, SelfAttention)
  models/joint_segmentation_depth_decoder.py:11-184 (JointSegDepthDecoder, PAD)
  models/pose_decoder.py:18-58          (PoseDecoder)
  models/joint_segmentation_depth.py:10-183 (assembly, predict_poses)
Third-party arithmetic restated from the published torchvision 0.7.0 algorithm
(requirements.txt:2; NOT under /root/reference => parity unpinned there):
  torchvision.models.resnet {BasicBlock, Bottleneck, ResNet._make_layer}
  torchvision.models.segmentation.deeplabv3 {ASPPConv, ASPPPooling}
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import geometry as G

RESNET_SPECS = {
    18: ("basic", [2, 2, 2, 2]),
    34: ("basic", [3, 4, 6, 3]),
    50: ("bottleneck", [3, 4, 6, 3]),
    101: ("bottleneck", [3, 4, 23, 3]),
    152: ("bottleneck", [3, 8, 36, 3]),
}


def num_ch_enc(num_layers):
    """resnet_encoder.py:71,87-88."""
    base = [64, 64, 128, 256, 512]
    if num_layers > 34:
        base = [base[0]] + [c * 4 for c in base[1:]]
    return base


# --------------------------------------------------------------------------
# architecture plans (shared by build_state_dict and the forward functions)
# --------------------------------------------------------------------------
def resnet_plan(num_layers, replace_stride_with_dilation=None):
    """torchvision ResNet._make_layer bookkeeping -> list of block dicts."""
    kind, counts = RESNET_SPECS[num_layers]
    if kind == "basic" and replace_stride_with_dilation and any(replace_stride_with_dilation):
        # torchvision BasicBlock: "Dilation > 1 not supported in BasicBlock" 
        raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
    rswd = replace_stride_with_dilation or [False, False, False]
    exp = 4 if kind == "bottleneck" else 1
    inplanes, dilation = 64, 1
    plan = []
    for li, (planes, n, stride, dilate) in enumerate(zip([64, 128, 256, 512], counts, [1, 2, 2, 2],
                                                         [False] + list(rswd))):
        prev_dil = dilation
        if dilate:
            dilation *= stride
            stride = 1
        for bi in range(n):
            s = stride if bi == 0 else 1
            d = prev_dil if bi == 0 else dilation
            down = bi == 0 and (s != 1 or inplanes != planes * exp)
            plan.append(dict(name="layer%d.%d" % (li + 1, bi), kind=kind, inplanes=inplanes, planes=planes,
                             stride=s, dilation=d, downsample=down, layer=li + 1))
            inplanes = planes * exp
    return plan


def decoder_plan(enc_ch, scales=range(4), num_output_channels=1, use_skips=True, intermediate_aspp=False,
                 aspp_rates=(6, 12, 18), num_ch_dec=(16, 32, 64, 128, 256), n_upconv=4, batch_norm=False,
                 dropout=0.0, n_project_skip_ch=-1, aspp_pooling=True, max_scale_size=None):
    """depth_decoder.py:42-72: ordered (key, idx, spec) entries of the positional ModuleList."""
    entries, idx = OrderedDict(), 0
    for i in range(n_upconv, -1, -1):
        cin = enc_ch[-1] if i == n_upconv else num_ch_dec[i + 1]
        cout = num_ch_dec[i]
        if i == n_upconv and intermediate_aspp:
            entries[("upconv", i, 0)] = dict(idx=idx, type="aspp", cin=cin, cout=cout, rates=list(aspp_rates),
                                             pooling=aspp_pooling)
        else:
            entries[("upconv", i, 0)] = dict(idx=idx, type="convblock", cin=cin, cout=cout, bn=batch_norm)
        idx += 1
        cin = num_ch_dec[i]
        if use_skips and i > 0:
            if n_project_skip_ch == -1:
                cin += enc_ch[i - 1]
                entries[("skip_proj", i)] = dict(idx=idx, type="identity")
            else:
                cin += n_project_skip_ch
                entries[("skip_proj", i)] = dict(idx=idx, type="skipproj", cin=enc_ch[i - 1], cout=n_project_skip_ch)
            idx += 1
        entries[("upconv", i, 1)] = dict(idx=idx, type="convblock", cin=cin, cout=num_ch_dec[i], bn=batch_norm)
        idx += 1
    for s in scales:
        entries[("dispconv", s)] = dict(idx=idx, type="conv3x3", cin=num_ch_dec[s], cout=num_output_channels)
        idx += 1
    return dict(entries=entries, n_upconv=n_upconv, use_skips=use_skips, scales=list(scales), dropout=dropout,
                num_ch_dec=list(num_ch_dec))


# --------------------------------------------------------------------------
# state_dict construction (key names + shapes = the contract)
# --------------------------------------------------------------------------
def _conv_w(sd, key, cout, cin, k, bias, gen, zero=False):
    fan_in = cin * k * k
    if zero:
        w = torch.zeros(cout, cin, k, k)
    else:
        w = torch.randn(cout, cin, k, k, generator=gen) * math.sqrt(2.0 / fan_in)
    sd[key + ".weight"] = w
    if bias:
        sd[key + ".bias"] = (torch.rand(cout, generator=gen) * 2 - 1) / math.sqrt(fan_in)


def _bn_p(sd, key, c, gen, randomize):
    if randomize:
        sd[key + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=gen)
        sd[key + ".bias"] = 0.1 * torch.randn(c, generator=gen)
        sd[key + ".running_mean"] = 0.1 * torch.randn(c, generator=gen)
        sd[key + ".running_var"] = 1.0 + 0.1 * torch.rand(c, generator=gen)
    else:
        sd[key + ".weight"] = torch.ones(c)
        sd[key + ".bias"] = torch.zeros(c)
        sd[key + ".running_mean"] = torch.zeros(c)
        sd[key + ".running_var"] = torch.ones(c)
    sd[key + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _resnet_sd(sd, p, num_layers, num_input_images, rswd, gen, rnd):
    _conv_w(sd, p + "conv1", 64, 3 * num_input_images, 7, False, gen)
    _bn_p(sd, p + "bn1", 64, gen, rnd)
    for b in resnet_plan(num_layers, rswd):
        q = p + b["name"] + "."
        if b["kind"] == "basic":
            _conv_w(sd, q + "conv1", b["planes"], b["inplanes"], 3, False, gen)
            _bn_p(sd, q + "bn1", b["planes"], gen, rnd)
            _conv_w(sd, q + "conv2", b["planes"], b["planes"], 3, False, gen)
            _bn_p(sd, q + "bn2", b["planes"], gen, rnd)
            out = b["planes"]
        else:
            w = b["planes"]
            _conv_w(sd, q + "conv1", w, b["inplanes"], 1, False, gen)
            _bn_p(sd, q + "bn1", w, gen, rnd)
            _conv_w(sd, q + "conv2", w, w, 3, False, gen)
            _bn_p(sd, q + "bn2", w, gen, rnd)
            _conv_w(sd, q + "conv3", w * 4, w, 1, False, gen)
            _bn_p(sd, q + "bn3", w * 4, gen, rnd)
            out = w * 4
        if b["downsample"]:
            _conv_w(sd, q + "downsample.0", out, b["inplanes"], 1, False, gen)
            _bn_p(sd, q + "downsample.1", out, gen, rnd)


def _decoder_sd(sd, p, plan, gen, rnd):
    for key, e in plan["entries"].items():
        q = "%sdecoder.%d." % (p, e["idx"])
        t = e["type"]
        if t == "convblock":
            _conv_w(sd, q + "block.0.conv", e["cout"], e["cin"], 3, True, gen)
            if e["bn"]:
                _bn_p(sd, q + "block.1", e["cout"], gen, rnd)
        elif t == "conv3x3":
            _conv_w(sd, q + "conv", e["cout"], e["cin"], 3, True, gen)
        elif t == "skipproj":
            _conv_w(sd, q + "0", e["cout"], e["cin"], 1, True, gen)
            _bn_p(sd, q + "1", e["cout"], gen, rnd)
        elif t == "aspp":
            _conv_w(sd, q + "convs.0.0", e["cout"], e["cin"], 1, False, gen)
            _bn_p(sd, q + "convs.0.1", e["cout"], gen, rnd)
            k = 1
            for _ in e["rates"]:
                _conv_w(sd, q + "convs.%d.0" % k, e["cout"], e["cin"], 3, False, gen)
                _bn_p(sd, q + "convs.%d.1" % k, e["cout"], gen, rnd)
                k += 1
            if e["pooling"]:
                _conv_w(sd, q + "convs.%d.1" % k, e["cout"], e["cin"], 1, False, gen)
                _bn_p(sd, q + "convs.%d.2" % k, e["cout"], gen, rnd)
                k += 1
            _conv_w(sd, q + "project.0", e["cout"], k * e["cout"], 1, False, gen)
            _bn_p(sd, q + "project.1", e["cout"], gen, rnd)


def _depth_args(model_cfg):
    da = dict(model_cfg.get("depth_args") or {})
    da.pop("max_scale_size", None)
    return da


def build_state_dict(model_cfg, n_classes, seed=0, randomize_bn=False, zero_attention=True):
    """Flat state_dict with the reference's key names/shapes for
    ``get_model(model_cfg, n_classes)`` (models/__init__.py:6, joint_segmentation_depth.py:116-183)."""
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    nl = int(model_cfg["backbone_name"].replace("resnet", ""))
    rswd = model_cfg.get("replace_stride_with_dilation")
    enc_ch = num_ch_enc(nl)
    _resnet_sd(sd, "models.encoder.encoder.", nl, 1, rswd, gen, randomize_bn)
    if model_cfg.get("enable_imnet_encoder"):
        r2 = rswd if model_cfg.get("imnet_encoder_dilation", True) else None
        _resnet_sd(sd, "models.imnet_encoder.encoder.", nl, 1, r2, gen, randomize_bn)
    frame_ids = tuple(model_cfg["frame_ids"])
    use_pose = not (frame_ids == (0, "s")) and not model_cfg.get("disable_pose")
    if use_pose and not model_cfg.get("disable_monodepth"):
        npf = 2 if model_cfg["pose_model_input"] == "pairs" else len(frame_ids)
        _resnet_sd(sd, "models.pose_encoder.encoder.", 18, npf, None, gen, randomize_bn)
        _conv_w(sd, "models.pose.net.0", 256, 512, 1, True, gen)
        _conv_w(sd, "models.pose.net.1", 256, 256, 3, True, gen)
        _conv_w(sd, "models.pose.net.2", 256, 256, 3, True, gen)
        _conv_w(sd, "models.pose.net.3", 12, 256, 1, True, gen)
    da = _depth_args(model_cfg)
    seg_name = model_cfg.get("segmentation_name")
    sa = dict(model_cfg.get("segmentation_args") or {})
    num_ch_dec = da.get("num_ch_dec", [16, 32, 64, 128, 256])
    if seg_name == "mtl_pad":
        p = "models.mtl_decoder."
        plan = decoder_plan(enc_ch, range(4), **da)
        _decoder_sd(sd, p + "depth_dec.", plan, gen, randomize_bn)
        _decoder_sd(sd, p + "seg_dec.", plan, gen, randomize_bn)
        dl, fl = sa.get("distillation_layer", 7), sa.get("final_layer", 9)
        dch = enc_ch[dl] if dl <= 4 else num_ch_dec[9 - dl]
        fch = enc_ch[fl] if fl <= 4 else num_ch_dec[9 - fl]
        for nm in ("sa_depth", "sa_seg"):
            _conv_w(sd, p + nm + ".conv", dch, dch, 3, False, gen)
            _conv_w(sd, p + nm + ".attention", dch, dch, 3, False, gen, zero=zero_attention)
        if sa.get("side_output", True):
            _conv_w(sd, p + "seg_intermediate_head.0", n_classes, dch, 1, True, gen)
        _conv_w(sd, p + "seg_final_head.0", n_classes, fch, 1, True, gen)
    else:
        if not model_cfg.get("disable_monodepth"):
            plan = decoder_plan(enc_ch, range(model_cfg["num_scales"]), **da)
            _decoder_sd(sd, "models.depth.", plan, gen, randomize_bn)
        if seg_name == "joint_seg_depth_dec":
            p = "models.segmentation."
            plan = decoder_plan(enc_ch, range(4), **da)
            _decoder_sd(sd, p + "unet_dec.", plan, gen, randomize_bn)
            layers = sa.get("layers") or [9]
            loc = sa.get("layer_out_channels", 64)
            hic = sa.get("head_inter_channels", 64)
            for layer in layers:
                ch = enc_ch[layer] if layer <= 4 else num_ch_dec[9 - layer]
                _conv_w(sd, p + "project.seg%d.0" % layer, loc, ch, 1, False, gen)
            if sa.get("head_inter", True):
                _conv_w(sd, p + "head.1", hic, loc * len(layers), 3, False, gen)
                _bn_p(sd, p + "head.2", hic, gen, randomize_bn)
                _conv_w(sd, p + "head.5", n_classes, hic, 1, True, gen)
            else:
                _conv_w(sd, p + "head.2", n_classes, hic, 1, True, gen)
        elif seg_name is not None:
            raise KeyError(seg_name)
    return sd


# --------------------------------------------------------------------------
# functional forward passes
# --------------------------------------------------------------------------
class Ctx:
    """train: BatchNorm uses batch statistics and updates running stats in ``sd``;
    dropout: apply nn.Dropout/Dropout2d layers (fresh torch RNG) -- parity runs keep it off."""

    def __init__(self, sd, train=True, dropout=False):
        self.sd, self.train, self.dropout = sd, train, dropout


def _conv(c, key, x, stride=1, padding=0, dilation=1):
    return F.conv2d(x, c.sd[key + ".weight"], c.sd.get(key + ".bias"), stride, padding, dilation)


def _bn(c, key, x):
    sd = c.sd
    if c.train:
        sd[key + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"],
                        sd[key + ".bias"], training=c.train, momentum=0.1, eps=1e-5)


def _drop(c, x, p, two_d=False):
    if not (c.dropout and c.train) or p <= 0:
        return x
    return F.dropout2d(x, p, True) if two_d else F.dropout(x, p, True)


def resnet_features(c, p, img, num_layers, rswd=None):
    """resnet_encoder.py:90-101 + torchvision ResNet blocks (v1.5: stride on the 3x3)."""
    x = (img - 0.45) / 0.225
    x = F.relu(_bn(c, p + "bn1", _conv(c, p + "conv1", x, 2, 3)))
    feats = [x]
    x = F.max_pool2d(x, 3, 2, 1)
    cur_layer = 1
    for b in resnet_plan(num_layers, rswd):
        if b["layer"] != cur_layer:
            feats.append(x)
            cur_layer = b["layer"]
        q = p + b["name"] + "."
        idt = x
        if b["kind"] == "basic":
            o = F.relu(_bn(c, q + "bn1", _conv(c, q + "conv1", x, b["stride"], b["dilation"], b["dilation"])))
            o = _bn(c, q + "bn2", _conv(c, q + "conv2", o, 1, b["dilation"], b["dilation"]))
        else:
            o = F.relu(_bn(c, q + "bn1", _conv(c, q + "conv1", x)))
            o = F.relu(_bn(c, q + "bn2", _conv(c, q + "conv2", o, b["stride"], b["dilation"], b["dilation"])))
            o = _bn(c, q + "bn3", _conv(c, q + "conv3", o))
        if b["downsample"]:
            idt = _bn(c, q + "downsample.1", _conv(c, q + "downsample.0", x, b["stride"]))
        x = F.relu(o + idt)
    feats.append(x)
    return feats


def _refl_conv3(c, key, x):
    """monodepth_layers.py:127-142 (Conv3x3: ReflectionPad2d(1) + 3x3 conv with bias)."""
    return _conv(c, key, F.pad(x, (1, 1, 1, 1), mode="reflect"))


def _convblock(c, q, e, x, dropout):
    """monodepth_layers.py:108-124."""
    x = _refl_conv3(c, q + "block.0.conv", x)
    if e["bn"]:
        x = _bn(c, q + "block.1", x)
    return _drop(c, F.elu(x), dropout, two_d=True)


def _aspp(c, q, e, x):
    """model_parts.py:5-32 + torchvision ASPPConv / ASPPPooling."""
    outs = [F.relu(_bn(c, q + "convs.0.1", _conv(c, q + "convs.0.0", x)))]
    k = 1
    for r in e["rates"]:
        outs.append(F.relu(_bn(c, q + "convs.%d.1" % k, _conv(c, q + "convs.%d.0" % k, x, 1, r, r))))
        k += 1
    if e["pooling"]:
        g = x.mean((2, 3), keepdim=True)
        g = F.relu(_bn(c, q + "convs.%d.2" % k, _conv(c, q + "convs.%d.1" % k, g)))
        outs.append(F.interpolate(g, size=x.shape[-2:], mode="bilinear", align_corners=False))
    y = F.relu(_bn(c, q + "project.1", _conv(c, q + "project.0", torch.cat(outs, 1))))
    return _drop(c, y, 0.5)


def decoder_forward(c, p, plan, feats, x=None, exec_layer=None, enable_disparity=True):
    """depth_decoder.py:75-116."""
    out, E = {}, plan["entries"]
    if x is None:
        x = feats[-1]
    for i in range(plan["n_upconv"], -1, -1):
        if exec_layer is not None and i not in exec_layer:
            continue
        e = E[("upconv", i, 0)]
        q = "%sdecoder.%d." % (p, e["idx"])
        x = _aspp(c, q, e, x) if e["type"] == "aspp" else _convblock(c, q, e, x, plan["dropout"])
        if x.shape[-1] < feats[i - 1].shape[-1] or i == 0:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        parts = [x]
        if plan["use_skips"] and i > 0:
            s = E[("skip_proj", i)]
            f = feats[i - 1]
            if s["type"] == "skipproj":
                sq = "%sdecoder.%d." % (p, s["idx"])
                f = F.relu(_bn(c, sq + "1", _conv(c, sq + "0", f)))
            parts.append(f)
        x = torch.cat(parts, 1)
        e = E[("upconv", i, 1)]
        x = _convblock(c, "%sdecoder.%d." % (p, e["idx"]), e, x, plan["dropout"])
        out[("upconv", i)] = x
        if i in plan["scales"] and enable_disparity:
            e = E[("dispconv", i)]
            out[("disp", i)] = torch.sigmoid(_refl_conv3(c, "%sdecoder.%d.conv" % (p, e["idx"]), x))
    return out


def _layer(feats, dec, layer):
    """models/utils.py:100-105."""
    return feats[layer] if layer <= 4 else dec[("upconv", 9 - layer)]


def jsd_forward(c, p, plan, feats, seg_args):
    """joint_segmentation_depth_decoder.py:55-75."""
    layers = seg_args.get("layers") or [9]
    os_ = seg_args.get("output_stride", 1)
    dec = decoder_forward(c, p + "unet_dec.", plan, feats)
    seg_size = tuple(_layer(feats, dec, 9).shape[2:])
    last_size = tuple(int(s) // os_ for s in seg_size)
    stacked = []
    for layer in layers:
        y = _conv(c, p + "project.seg%d.0" % layer, _layer(feats, dec, layer))
        stacked.append(F.interpolate(y, size=last_size, mode="bilinear", align_corners=False))
    y = _drop(c, torch.cat(stacked, 1), seg_args.get("layer_dropout", 0))
    if seg_args.get("head_inter", True):
        y = F.relu(_bn(c, p + "head.2", _conv(c, p + "head.1", y, 1, 1)))
        y = _drop(c, y, seg_args.get("head_dropout", 0.1))
        y = _conv(c, p + "head.5", y)
    else:
        y = _conv(c, p + "head.2", y)
    if last_size != seg_size:
        y = F.interpolate(y, size=seg_size, mode="bilinear", align_corners=False)
    return y


def _self_attention(c, q, x):
    """model_parts.py:35-46."""
    return _conv(c, q + ".conv", x, 1, 1) * torch.sigmoid(_conv(c, q + ".attention", x, 1, 1))


def pad_forward(c, p, plan, feats, seg_args):
    """joint_segmentation_depth_decoder.py:134-184."""
    os_ = seg_args.get("output_stride", 1)
    dl, fl = seg_args.get("distillation_layer", 7), seg_args.get("final_layer", 9)
    side = seg_args.get("side_output", True)
    seg_size = tuple(feats[0].shape[2:])
    last_size = tuple(int(s) // os_ for s in seg_size)
    di = 9 - dl
    first = list(range(plan["n_upconv"], di - 1, -1))
    second = list(range(di - 1, -1, -1))
    d = decoder_forward(c, p + "depth_dec.", plan, feats, exec_layer=first)
    s = decoder_forward(c, p + "seg_dec.", plan, feats, exec_layer=first, enable_disparity=False)
    name = ("upconv", di)
    inter = _conv(c, p + "seg_intermediate_head.0", s[name]) if side else None
    fd = _self_attention(c, p + "sa_depth", d[name])
    fs = _self_attention(c, p + "sa_seg", s[name])
    for_seg = s[name] + fd
    for_depth = d[name] + fs
    d.update(decoder_forward(c, p + "depth_dec.", plan, feats, x=for_depth, exec_layer=second))
    s2 = decoder_forward(c, p + "seg_dec.", plan, feats, x=for_seg, exec_layer=second, enable_disparity=False)
    final = _conv(c, p + "seg_final_head.0", _layer(fd, s2, fl))
    if side and last_size != seg_size:
        inter = F.interpolate(inter, size=seg_size, mode="bilinear", align_corners=False)
    if last_size != seg_size:
        final = F.interpolate(final, size=seg_size, mode="bilinear", align_corners=False)
    out = dict(d)
    out["semantics"] = final
    if side:
        out["intermediate_semantics"] = inter
    return out


def pose_decoder(c, p, feat, num_frames=2):
    """pose_decoder.py:41-58 (num_input_features=1)."""
    x = F.relu(_conv(c, p + "net.0", feat))
    x = F.relu(_conv(c, p + "net.1", x, 1, 1))
    x = F.relu(_conv(c, p + "net.2", x, 1, 1))
    x = _conv(c, p + "net.3", x)
    x = 0.01 * x.mean(3).mean(2).reshape(-1, num_frames, 1, 6)
    return x[..., :3], x[..., 3:]


def model_forward(sd, model_cfg, inputs, train=True, dropout=False, use_pose_net=None):
    """joint_segmentation_depth.py:77-100 (+ predict_poses :20-70, both pose_model_input modes)."""
    c = Ctx(sd, train, dropout)
    nl = int(model_cfg["backbone_name"].replace("resnet", ""))
    rswd = model_cfg.get("replace_stride_with_dilation")
    enc_ch = num_ch_enc(nl)
    da = _depth_args(model_cfg)
    sa = dict(model_cfg.get("segmentation_args") or {})
    sa.pop("weights", None)
    out = {}
    img = inputs[("color_aug", 0, 0)]
    feats = resnet_features(c, "models.encoder.encoder.", img, nl, rswd)
    out["bottleneck"] = feats[-1]
    seg_name = model_cfg.get("segmentation_name")
    if seg_name == "mtl_pad":
        out.update(pad_forward(c, "models.mtl_decoder.", decoder_plan(enc_ch, range(4), **da), feats, sa))
    else:
        if not model_cfg.get("disable_monodepth"):
            out.update(decoder_forward(c, "models.depth.", decoder_plan(enc_ch, range(model_cfg["num_scales"]), **da),
                                       feats))
        if seg_name is not None:
            out["semantics"] = jsd_forward(c, "models.segmentation.", decoder_plan(enc_ch, range(4), **da), feats, sa)
    if model_cfg.get("enable_imnet_encoder"):
        out["encoder_features"] = feats[-1]
        c2 = Ctx(sd, False, False)
        r2 = rswd if model_cfg.get("imnet_encoder_dilation", True) else None
        with torch.no_grad():
            out["imnet_features"] = resnet_features(c2, "models.imnet_encoder.encoder.", img, nl, r2)[-1].detach()
    frame_ids = tuple(model_cfg["frame_ids"])
    has_pose = not (frame_ids == (0, "s")) and not model_cfg.get("disable_pose") \
        and not model_cfg.get("disable_monodepth")
    if use_pose_net is None:
        use_pose_net = has_pose
    if use_pose_net and has_pose and model_cfg["pose_model_input"] != "pairs":
        # joint_segmentation_depth.py:52-68: every frame through the pose net at once, all poses together (no inversion)
        key = "color_full_aug" if model_cfg.get("provide_uncropped_for_pose") else "color_aug"
        x = torch.cat([inputs[(key, f, 0)] for f in frame_ids if f != "s"], 1)
        pf = resnet_features(c, "models.pose_encoder.encoder.", x, 18, None)[-1]
        aa, tr = pose_decoder(c, "models.pose.", pf)
        for i, f in enumerate(frame_ids[1:]):
            if f == "s":
                continue
            out[("axisangle", 0, f)], out[("translation", 0, f)] = aa, tr
            out[("cam_T_cam", 0, f)] = G.pose_matrix(aa[:, i], tr[:, i])
    elif use_pose_net and has_pose:
        key = "color_full_aug" if model_cfg.get("provide_uncropped_for_pose") else "color_aug"
        for f in frame_ids[1:]:
            if f == "s":
                continue
            pair = [inputs[(key, f, 0)], inputs[(key, 0, 0)]] if f < 0 else [inputs[(key, 0, 0)], inputs[(key, f, 0)]]
            pf = resnet_features(c, "models.pose_encoder.encoder.", torch.cat(pair, 1), 18, None)[-1]
            aa, tr = pose_decoder(c, "models.pose.", pf)
            out[("axisangle", 0, f)], out[("translation", 0, f)] = aa, tr
            out[("cam_T_cam", 0, f)] = G.pose_matrix(aa[:, 0], tr[:, 0], invert=(f < 0))
    return out


def predict_test_disp(sd, model_cfg, inputs):
    """joint_segmentation_depth.py:72-75 in eval mode (BatchNorm running statistics, no dropout), as
    loader/depth_estimator.py:63-81 calls it: depth decoder on the encoder features of ("color", 0, 0)."""
    c = Ctx(sd, False, False)
    nl = int(model_cfg["backbone_name"].replace("resnet", ""))
    feats = resnet_features(c, "models.encoder.encoder.", inputs[("color", 0, 0)], nl,
                            model_cfg.get("replace_stride_with_dilation"))
    return decoder_forward(c, "models.depth.", decoder_plan(num_ch_enc(nl), range(model_cfg["num_scales"]),
                                                            **_depth_args(model_cfg)), feats)
