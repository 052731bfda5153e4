"""``torch.ops.segsde.*``: the HIP operators of this package in PyTorch's operator registry.

SURVEY.md 8b puts the kernels "behind ``torch.ops.<ns>.*`` + ``autograd.Function``"; the package's own modules call the autograd
Functions of ``functional.py`` directly (one Python frame less per launch, which is what the small-batch step is bound by), and this
module registers the same operators under the ``segsde`` namespace for callers that want the registry: scripts that compose the
kernels without the reference's module tree, ``torch.ops`` introspection, schema checks.  Every operator is registered for
``CompositeImplicitAutograd`` -- its implementation is the package's autograd Function, so gradients flow exactly as they do inside
the models; operators marked "(forward only)" wrap kernels whose adjoint lives in a fused backward of the loss.

Importing the package registers the namespace (``import improving_segmentation_with_selfsupervised_depth_amd``); nothing here falls
back to ATen: without the HIP library every call raises like the modules do.

Activations / feature maps are NHWC tensors (``functional.to_nhwc``) unless a schema comment says otherwise; images, disparities
and loss inputs are NCHW like the reference's."""
import torch

from . import functional as Fn
from . import hipops as H

NAMESPACE = "segsde"
try:
    _lib = torch.library.Library(NAMESPACE, "DEF")
except RuntimeError:      # the namespace exists already in this process (the module imported a second time under another name)
    _lib = torch.library.Library(NAMESPACE, "FRAGMENT")
SCHEMAS = {}


def _op(schema):
    name = schema.split("(", 1)[0]

    def deco(fn):
        try:
            _lib.define(schema)
            _lib.impl(name, fn, "CompositeImplicitAutograd")
        except RuntimeError:
            if not hasattr(getattr(torch.ops, NAMESPACE), name):      # anything but "registered before": a real error
                raise
        SCHEMAS[name] = schema
        return fn
    return deco


# ----------------------------------------------------------------------------------------------- layout
@_op("to_nhwc(Tensor(a) x) -> Tensor(a)")
def to_nhwc(x):
    """NCHW image / feature map -> the package's channels-last activation (its adjoint is to_nchw)"""
    return Fn.to_nhwc(x)


@_op("to_nchw(Tensor(a) x) -> Tensor(a)")
def to_nchw(x):
    return Fn.to_nchw(x)


# ----------------------------------------------------------------------------------------------- encoder / decoder stages
@_op("conv2d(Tensor x, Tensor weight, Tensor? bias=None, Tensor? skip=None, int stride=1, int padding=0, int dilation=1, "
     "bool reflect=False, bool upsample=False) -> Tensor")
def conv2d(x, weight, bias=None, skip=None, stride=1, padding=0, dilation=1, reflect=False, upsample=False):
    """conv([nearest-2x?(x) | skip], weight) + bias: every convolution of models/resnet_encoder.py, depth_decoder.py and
    joint_segmentation_depth_decoder.py (weight OIHW like nn.Conv2d's; ``reflect``: Conv3x3's ReflectionPad2d(1),
    monodepth_layers.py:122-141; ``upsample`` + ``skip``: the decoder's ``cat([upsample(x), skip])``, depth_decoder.py:89-94)"""
    c1 = 0 if skip is None else skip.shape[3]
    g = H.ConvGeom(x.shape[3], weight.shape[0], weight.shape[2], stride, dilation, padding, reflect, c1, upsample)
    return Fn.ConvFn.apply(x, skip, weight, bias, g, "none")


@_op("batch_norm_act(Tensor x, Tensor weight, Tensor bias, Tensor? running_mean, Tensor? running_var, Tensor? residual=None, "
     "bool training=True, float momentum=0.1, float eps=1e-05, str act='none') -> Tensor")
def batch_norm_act(x, weight, bias, running_mean, running_var, residual=None, training=True, momentum=0.1, eps=1e-5, act="none"):
    """act(BatchNorm(x) + residual), act in none / relu / elu: the BatchNorm + ReLU (+ shortcut) of a torchvision bottleneck as one
    pass; training=True normalises with the batch statistics and updates the running buffers in place"""
    return Fn.BNActFn.apply(x, weight, bias, residual, running_mean, running_var, bool(training), float(momentum), float(eps), act,
                            0.0, 0)


@_op("max_pool_3x3_s2(Tensor x) -> Tensor")
def max_pool_3x3_s2(x):
    """the encoder stem's MaxPool2d(3, 2, 1)"""
    return Fn.MaxPoolFn.apply(x)


@_op("resize_bilinear(Tensor x, int[] size, bool align_corners=False) -> Tensor")
def resize_bilinear(x, size, align_corners=False):
    return Fn.resize_bilinear(x, (int(size[0]), int(size[1])), align_corners)


@_op("global_avg_pool(Tensor x) -> Tensor")
def global_avg_pool(x):
    return Fn.GlobalAvgPoolFn.apply(x)


@_op("pose_matrix(Tensor axisangle, Tensor translation, bool invert=False) -> Tensor")
def pose_matrix(axisangle, translation, invert=False):
    """transformation_from_parameters, monodepth_layers.py:30-45 -> [B,4,4]"""
    return Fn.PoseMatrixFn.apply(axisangle, translation, bool(invert))


# ----------------------------------------------------------------------------------------------- photometric loss stages
@_op("warp(Tensor disp, Tensor inv_K, Tensor K, Tensor T, Tensor src, float min_depth, float max_depth) -> (Tensor, Tensor, Tensor)")
def warp(disp, inv_K, K, T, src, min_depth, max_depth):
    """generate_images_pred for one (scale, frame), monodepth_loss.py:64-102: upsample -> depth -> backproject -> project ->
    grid_sample -> (warped frame [B,3,H,W], sampling grid [B,H,W,2], depth [B,1,H,W]).  (forward only: its adjoint is part of the
    fused photometric backward)"""
    return H.warp_forward(disp.detach().float().contiguous(), inv_K, K, T.detach().float().contiguous(), src, min_depth,
                          max_depth, want_grid=True, want_depth=True)


@_op("ssim(Tensor x, Tensor y) -> Tensor")
def ssim(x, y):
    """SSIM.forward, monodepth_layers.py:224-254"""
    from .models.monodepth_layers import _SSIMFn
    return _SSIMFn.apply(x.float().contiguous(), y.float().contiguous())


@_op("smooth_loss(Tensor disp, Tensor img) -> Tensor")
def smooth_loss(disp, img):
    """get_smooth_loss, monodepth_layers.py:208-221"""
    from .models.monodepth_layers import get_smooth_loss
    return get_smooth_loss(disp, img)


@_op("backproject_depth(Tensor depth, Tensor inv_K) -> Tensor")
def backproject_depth(depth, inv_K):
    """BackprojectDepth.forward, monodepth_layers.py:169-174 -> [B,4,H*W]"""
    from .models.monodepth_layers import _BackprojectFn
    return _BackprojectFn.apply(depth.float().contiguous(), inv_K.detach().float())


@_op("project3d(Tensor points, Tensor K, Tensor T, int height, int width, float eps=1e-07) -> Tensor")
def project3d(points, K, T, height, width, eps=1e-7):
    """Project3D.forward, monodepth_layers.py:188-199 -> [B,H,W,2] in [-1, 1]"""
    from .models.monodepth_layers import _Project3DFn
    return _Project3DFn.apply(points.float().contiguous(), K.detach().float(), T.float().contiguous(), int(height), int(width),
                              float(eps))


@_op("reprojection_error(Tensor pred, Tensor target, bool no_ssim=False) -> Tensor")
def reprojection_error(pred, target, no_ssim=False):
    """compute_reprojection_loss, monodepth_loss.py:104-116 -> [B,1,H,W] (forward only)"""
    B, _, Hh, W = pred.shape
    out = torch.empty((B, 1, Hh, W), dtype=torch.float32, device=pred.device)
    H.reprojection_error(pred.detach().float().contiguous(), target.detach().float().contiguous(), bool(no_ssim), out[:, 0])
    return out


@_op("automask_min(Tensor? ident, Tensor? noise, Tensor reproj, bool avg=False) -> (Tensor, Tensor, Tensor?)")
def automask_min(ident, noise, reproj, avg=False):
    """the per-pixel minimum of monodepth_loss.py:136-177 over [identity (+ 1e-5 noise) | reprojection] errors of 1..8 source frames
    -> (sum of the minima [1], argmin uint8 [B,H,W], identity_selection [B,H,W] or None) (forward only)"""
    return H.automask_min(None if ident is None else ident.float().contiguous(),
                          None if noise is None else noise.float().contiguous(), reproj.float().contiguous(), bool(avg))


# ----------------------------------------------------------------------------------------------- segmentation loss, DepthMix
@_op("cross_entropy2d(Tensor input, Tensor target, Tensor? class_weight=None, Tensor? pixel_weights=None) -> Tensor")
def cross_entropy2d(input, target, class_weight=None, pixel_weights=None):
    """loss/loss.py:17-37 (NCHW logits, int64 target, ignore_index 250, align_corners=True resize to the target's size)"""
    from .loss.loss import cross_entropy2d as ce
    return ce(input, target, class_weight, pixel_weights)


@_op("mix(Tensor mask, Tensor x) -> Tensor")
def mix(mask, x):
    """loader/transformsgpu.py:33-47: mask * x + (1 - mask) * roll(x) over the batch (paired halves when the mask has half the
    batch); bit-exact with the reference's op sequence (forward only)"""
    from .loader import transformsgpu
    if x.dtype == torch.int64:
        return transformsgpu.mix(mask, target=x)[1]
    return transformsgpu.mix(mask, data=x)[0]


@_op("depthcomp_mask(Tensor depths, float margin, float fg_threshold) -> Tensor")
def depthcomp_mask(depths, margin, fg_threshold):
    """the DepthMix mask of train.py:585-604: (d_i >= d_partner - margin) * (d_i >= fg_threshold) -> int64 [B,H,W]"""
    return H.depthcomp_mask(depths, float(margin), float(fg_threshold))


def names():
    """qualified names of everything registered above"""
    return ["%s::%s" % (NAMESPACE, n) for n in SCHEMAS]
