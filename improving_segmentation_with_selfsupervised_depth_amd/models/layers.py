"""HIP-backed building blocks shared by the model mirror: Conv2d / BatchNorm2d subclasses that keep torch.nn's
parameter names, shapes and default initialisation (so reference checkpoints load and ``isinstance(m,
nn.BatchNorm2d)`` checks in the reference's train.py keep working) but run on NHWC tensors through the HIP ops."""
import os

import torch
from torch import nn

from .. import functional as Fn
from .. import hipops as H
from ..hipops import ConvGeom


def _seed():
    # CPU generator: no device sync; the dropout mask itself is generated in-kernel from this counter seed
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


_NO_FUSED_STATS = bool(int(os.environ.get("SEGSDE_NO_FUSED_STATS", "0")))   # debugging: BatchNorm statistics as a separate pass


GRAPH_SAFE_DROPOUT = [False]   # see BatchNorm2d.forward
_PACK_SCOPE = [0, 0]      # [id of the active weight-pack scope (0: none), last id handed out]


class weight_pack_scope:
    """``with weight_pack_scope():`` -- inside, a convolution packs its weight once and reuses the packs for every further
    forward of the scope (the DepthMix step runs the student three times between two optimizer steps).  The caller promises
    that weights are not modified inside the scope by means autograd cannot see: version counters catch optimizer steps and
    ``load_state_dict``, but not ``param.data[:] = ...`` (how the reference's own EMA update writes, train.py:353-357), which
    is why nothing is cached outside a scope."""

    def __init__(self, model=None):
        """model: optionally the module whose convolutions the scope is about to run -- their weights are then packed up
        front in ONE launch instead of one launch per convolution at its first forward (a training step re-packs every
        weight: ~160 launches of ~10 us for the ResNet-101 joint model)"""
        self._model = model

    def __enter__(self):
        self._outer = _PACK_SCOPE[0]
        if not self._outer:                # a nested scope joins the enclosing one
            _PACK_SCOPE[1] += 1
            _PACK_SCOPE[0] = _PACK_SCOPE[1]
            if self._model is not None:
                self._prepack(self._model, _PACK_SCOPE[0])
        return self

    @staticmethod
    def _prepack(model, scope_id):
        from .. import _lib
        # the walk over the module tree once per model (~2 000 generator steps per training step otherwise); a model whose tree
        # changes later drops the attribute (``del model._segsde_convs``) or simply packs the newcomers at their first forward
        allc = model.__dict__.get("_segsde_convs")
        if allc is None:
            allc = model.__dict__["_segsde_convs"] = [m for m in model.modules() if isinstance(m, Conv2d)]
        convs = [m for m in allc if (m.weight.is_cuda or _lib.HOST_POINTERS_OK) and m.weight.is_contiguous()
                 and m.weight.dtype == torch.float32 and m.in_channels % 4 == 0]   # (stems pad their weight per call)
        if not convs:
            return
        packs = H.pack_weights_multi([m.weight for m in convs])
        for m, pk in zip(convs, packs):
            w = m.weight
            d = m.__dict__        # (plain attributes, set ~250 times per step: nn.Module.__setattr__ costs 2 us each on a launch-bound step)
            d["_packs"], d["_pack_key"] = pk, (scope_id, w._version, w.data_ptr(), w.device)
        # the Winograd-transformed packs of the 3x3 / stride-1 convolutions with many channels, likewise in one launch
        # (the one-kernel route's packs have their own layout: two launches, one per layout)
        for kn in (True, False):
            wino = [m for m in convs if (H.winograd_fused_static_ok(m) if kn
                                         else (H.winograd_static_ok(m) and not H.winograd_fused_static_ok(m)))]
            if wino:
                for m, pk in zip(wino, H.winograd_packs_multi([m.weight for m in wino], kn=kn)):
                    w = m.weight
                    m._wino_cache["packs"], m._wino_cache["key"] = pk, (scope_id, w._version, w.data_ptr(), w.device)
                    m._wino_cache["kn"] = kn

    def __exit__(self, *exc):
        _PACK_SCOPE[0] = self._outer
        return False


class Conv2d(nn.Conv2d):
    """nn.Conv2d state, HIP implicit-GEMM compute.  ``forward(x_nhwc, skip=None, up=False, act="none")``:
    ``skip`` is an optional second NHWC source concatenated after x along channels, ``up`` nearest-upsamples x by 2
    inside the kernel's tile loader (models/depth_decoder.py:93-100 without materialising either)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True, reflect=False):
        super().__init__(int(in_channels), int(out_channels), kernel_size, stride, padding, dilation, bias=bias)
        self.reflect = reflect
        self._pack_key, self._packs = None, None   # weight packs of the current weight version (see _weight_packs)
        self._fold_cache = {}         # upsample-folded packs of the same weight version (filled by ConvFn inside a pack scope)
        self._wino_cache = {}         # Winograd-transformed packs, likewise
        self._stats_wanted = None     # None: unknown yet, True: a BatchNorm consumed the fused statistics, False: nobody did
        self._stats_offered = False
        k = self.kernel_size[0]
        assert self.kernel_size[0] == self.kernel_size[1] and self.stride[0] == self.stride[1]

    def _weight_packs(self, weight):
        """(forward pack, data-gradient pack) cached for the active weight_pack_scope, None outside one (ConvFn then packs
        per call, as the weights may have been rewritten behind autograd's back)"""
        d = self.__dict__
        if not _PACK_SCOPE[0]:
            d["_packs"] = None
            return None
        key = (_PACK_SCOPE[0], weight._version, weight.data_ptr(), weight.device)
        if key != d["_pack_key"] or d["_packs"] is None:
            d["_packs"], d["_pack_key"] = H.pack_weight_both(weight), key
        return d["_packs"]

    def forward_image(self, image, mean, std):
        """Network stem: self((image - mean) / std) for an NCHW image, as one layout pass + the dedicated 7x7 / stride-2
        kernel (Fn.StemFn); None when this module / image is not that case (the caller then takes to_nhwc + forward)."""
        if not H.stem_ok(image, self) or (torch.is_grad_enabled() and image.requires_grad):
            return None
        wstem = None
        if _PACK_SCOPE[0]:
            key = (_PACK_SCOPE[0], self.weight._version, self.weight.data_ptr(), self.weight.device)
            if key != getattr(self, "_stem_key", None):
                self._stem_pack, self._stem_key = H.stem_pack(self.weight), key
            wstem = self._stem_pack
        if self.training and not _NO_FUSED_STATS and self._stats_wanted is not False:
            if self._stats_wanted is None and self._stats_offered:
                self._stats_wanted = False
                return Fn.StemFn.apply(image, self.weight, mean, std, None, wstem)
            if not self._stats_offered:
                self._stats_offered = True
            holder = []
            y = Fn.StemFn.apply(image, self.weight, mean, std, holder, wstem)
            y._bn_partials = (holder[0], y._version, self) if holder and holder[0] is not None else None
            return y
        return Fn.StemFn.apply(image, self.weight, mean, std, None, wstem)

    def forward(self, x, skip=None, up=False, act="none", grad_box=None, skip_box=None):
        c0 = x.shape[3]
        c1 = 0 if skip is None else skip.shape[3]
        weight = self.weight
        if skip is None and c0 > self.in_channels:
            # input carries zero pad channels (network stem: 3 -> 4, 6 -> 8): matching zero weight planes; their gradient
            # is dropped by the adjoint of the pad
            weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, c0 - self.in_channels))
            packs = None                    # a fresh tensor every call: packed inside ConvFn
        else:
            assert c0 + c1 == self.in_channels, (c0, c1, self.in_channels)
            packs = self._weight_packs(weight)
        g = ConvGeom(c0, self.out_channels, self.kernel_size[0], self.stride[0], self.dilation[0], self.padding[0],
                     self.reflect, c1, up, cin_alg=self.in_channels if (skip is None and c0 > self.in_channels) else None)
        # x may be the activated output of another convolution of this package (ConvBlock -> ELU): then this conv's
        # data-gradient applies the activation's derivative itself (conv epilogue) and hands the pre-activation gradient
        # straight to the producer -- see Fn.ActGradFn
        x0_act = None
        pre = getattr(x, "_preact", None)
        if pre is not None and pre[2] == x._version and torch.is_grad_enabled() and pre[0].requires_grad:
            x, x0_act = pre[0], pre[1]
        wc = self._wino_cache
        wc["want"] = (_PACK_SCOPE[0], weight._version, weight.data_ptr(), weight.device) if _PACK_SCOPE[0] else None
        if self.bias is None and act == "none" and self.training and not _NO_FUSED_STATS and self._stats_wanted is not False:
            # bias-free, activation-free convolutions are the candidates for a following BatchNorm (torchvision ResNet /
            # ASPP convention): their epilogue also leaves the batch-statistics partials, picked up by
            # BatchNorm2d.forward.  A convolution whose partials nobody picked up (SelfAttention's convs, the
            # segmentation projections) stops producing them after its first training forward.
            if self._stats_wanted is None and self._stats_offered:
                self._stats_wanted = False
                return Fn.ConvFn.apply(x, skip, weight, self.bias, g, act, None, grad_box, packs, x0_act, None, None, wc)
            if not self._stats_offered:
                self._stats_offered = True
            holder = []
            y = Fn.ConvFn.apply(x, skip, weight, self.bias, g, act, holder, grad_box, packs, x0_act, None, None, wc)
            y._bn_partials = (holder[0], y._version, self) if holder and holder[0] is not None else None
            return y
        if act == "none":
            return Fn.ConvFn.apply(x, skip, weight, self.bias, g, act, None, grad_box, packs, x0_act, None, None, wc)
        box = {"need_dbias": self.bias is not None and self.bias.requires_grad}
        fc = self._fold_cache
        fc["want"] = (_PACK_SCOPE[0], weight._version, weight.data_ptr(), weight.device, c0) if (_PACK_SCOPE[0] and up) else None
        yz = Fn.ConvFn.apply(x, skip, weight, self.bias, g, act, None, grad_box, packs, x0_act, box, fc, wc, skip_box)
        if not (torch.is_grad_enabled() and yz.requires_grad):
            return yz
        y = Fn.ActGradFn.apply(yz, act, box)
        y._preact = (yz, act, y._version)
        return y


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d state; ``forward(x_nhwc, residual=None, act="none", drop_p=0.0)`` fuses the residual add,
    the activation and (for ASPP.project) the dropout into the normalisation pass."""

    def forward(self, x, residual=None, act="none", drop_p=0.0, grad_box=None):
        training = self.training or (self.running_mean is None)
        # num_batches_tracked is incremented by the statistics finalize kernel (no launch of its own)
        nbt = self.num_batches_tracked if (self.training and self.track_running_stats
                                           and self.num_batches_tracked is not None) else None
        if self.momentum is None:
            # torch.nn.BatchNorm2d: cumulative moving average, factor 1 / num_batches_tracked after this forward's
            # increment (reads the device counter: a host sync, on a path no reference config takes)
            momentum = 1.0 / float(int(self.num_batches_tracked) + 1) if nbt is not None else 0.0
        else:
            momentum = self.momentum
        if drop_p > 0 and GRAPH_SAFE_DROPOUT[0]:
            # hipGraph capture (bench.py --hip-graph): the fused dropout's seed is a host integer, i.e. a constant of the captured
            # launch -- every replay would draw the same mask.  Under capture the mask comes from torch's device generator instead
            # (Philox offsets advance per replay): BatchNorm (+ residual, activation) fused as always, dropout as one more
            # elementwise pass over the (small: ASPP projection / class head) tensor
            y = self.forward(x, residual, act, 0.0, grad_box)
            return torch.nn.functional.dropout(y, drop_p, True)
        seed = _seed() if drop_p > 0 else 0
        partials = getattr(x, "_bn_partials", None) if training else None
        if partials is not None:
            # valid only for the very tensor the convolution produced: same channel count and no in-place edit since
            partials, version, producer = partials
            if partials.shape[-1] != self.num_features or x._version != version:
                partials = None
            elif producer._stats_wanted is not True:
                producer._stats_wanted = True
        if training:
            Fn.fusion("bn_stats_from_conv_epilogue", partials is not None)
        return Fn.BNActFn.apply(x, self.weight, self.bias, residual, self.running_mean, self.running_var, training,
                                momentum, self.eps, act, drop_p, seed, partials, grad_box, nbt)
