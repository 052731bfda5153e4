"""Mirror of the reference's models/monodepth_layers.py on the HIP ops (same names / constructor arguments).

Hot-path pieces (ConvBlock, Conv3x3, transformation_from_parameters) run on HIP kernels.  The geometry/SSIM
layers that the reference's MonodepthLoss composes out of many small ATen ops (BackprojectDepth, Project3D, SSIM,
get_smooth_loss -- monodepth_layers.py:145-254) are folded into this package's fused loss kernels
(loss/monodepth_loss.py); small API-parity helpers remain for the inference-side callers."""
import torch
from torch import nn

from .. import functional as Fn
from .. import hipops as H
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

    def forward(self, x, skip=None, up=False, act="none", skip_box=None):
        return self.conv(x, skip, up, act, skip_box=skip_box)


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

    def forward(self, x, skip=None, up=False, skip_box=None):
        if self.bn:
            y = self.block[1](self.block[0](x, skip, up), act="elu")
        else:
            y = self.block[0](x, skip, up, act="elu", skip_box=skip_box)
        drop = self.block[3]
        if isinstance(drop, nn.Dropout2d) and drop.training and self.training and drop.p > 0:
            # whole channel maps are zeroed with probability p and the survivors scaled by 1 / (1 - p); the Bernoulli draw
            # is torch's (device RNG plumbing), the application and its adjoint are HIP
            B, C = y.shape[0], y.shape[3]
            keep = torch.bernoulli(torch.full((B, C), 1.0 - drop.p, device=y.device))
            y = Fn.ChannelDropFn.apply(y, keep / (1.0 - drop.p) if drop.p < 1 else keep)
        return y


# ----------------------------------------------------------------------------------------------------------------------
# The remaining public names of the reference module (monodepth_layers.py:48-105, 145-254).  Only the reference's own
# loss/monodepth_loss.py imports them; this package's MonodepthLoss runs the same arithmetic inside its fused kernels.  They are
# kept as thin callables on device kernels so that a user script importing them by name still works (NCHW tensors in and
# out, like the reference).  All four are differentiable like the reference's (BackprojectDepth w.r.t. the depth, Project3D
# w.r.t. the points and the pose; the intrinsics are data and raise if they ask for a gradient).
# ----------------------------------------------------------------------------------------------------------------------
def get_translation_matrix(translation_vector):
    """reference :50-63: [B,1,1,3] / [B,3] translation -> [B,4,4]"""
    t = translation_vector.contiguous().view(-1, 1, 1, 3)
    return Fn.PoseMatrixFn.apply(torch.zeros_like(t), t, False)


def rot_from_axisangle(vec):
    """reference :66-105: axis-angle [B,1,3] -> [B,4,4] rotation"""
    v = vec.contiguous().view(-1, 1, 1, 3)
    return Fn.PoseMatrixFn.apply(v, torch.zeros_like(v), False)


def _intrinsics_are_data(*ts):
    if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
        raise NotImplementedError("the camera intrinsics are data here: no gradient is implemented for K / inv_K "
                                  "(the reference never trains them: loader/sequence_segmentation_loader.py:278-290)")


class _BackprojectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, inv_K):
        ctx.hw = (depth.shape[2], depth.shape[3])
        ctx.save_for_backward(inv_K)
        return H.backproject_depth(depth, inv_K)

    @staticmethod
    def backward(ctx, g):
        (inv_K,) = ctx.saved_tensors
        return H.backproject_depth_backward(g, inv_K, *ctx.hw), None


class BackprojectDepth(nn.Module):
    """reference :145-174: depth image -> homogeneous camera points [B,4,H*W]; differentiable w.r.t. the depth like the
    reference's torch code.  The reference registers its pixel grid as three frozen parameters (:156-167); they are kept (same
    names, shapes and values) so that a ``state_dict`` of a module that owns this layer has the same keys -- the kernel derives
    the pixel coordinates from the index and never reads them."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width
        ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
        self.id_coords = nn.Parameter(torch.stack([xs, ys], 0), requires_grad=False)
        self.ones = nn.Parameter(torch.ones(batch_size, 1, height * width), requires_grad=False)
        pix = torch.stack([xs.reshape(-1), ys.reshape(-1)], 0).unsqueeze(0).repeat(batch_size, 1, 1)
        self.pix_coords = nn.Parameter(torch.cat([pix, self.ones], 1), requires_grad=False)

    def forward(self, depth, inv_K):
        _intrinsics_are_data(inv_K)
        return _BackprojectFn.apply(depth.float().reshape(self.batch_size, 1, self.height, self.width).contiguous(),
                                    inv_K.detach().float())


class _Project3DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, K, T, height, width, eps):
        ctx.cfg = (height, width, eps)
        ctx.save_for_backward(points, K, T)
        return H.project3d(points, K, T, height, width, eps)

    @staticmethod
    def backward(ctx, g):
        points, K, T = ctx.saved_tensors
        gp, gT = H.project3d_backward(points, K, T, g, *ctx.cfg, need_points=ctx.needs_input_grad[0], need_T=ctx.needs_input_grad[2])
        return gp, None, gT, None, None, None


class Project3D(nn.Module):
    """reference :177-199: camera points -> normalised sampling grid [B,H,W,2] for intrinsics K at pose T; differentiable
    w.r.t. the points (-> depth) and T (-> pose network) like the reference's torch code"""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        _intrinsics_are_data(K)
        return _Project3DFn.apply(points.float().contiguous(), K.detach().float(), T.float().contiguous(), self.height, self.width,
                                  self.eps)


class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from .. import _lib
        B, Hh, W, C = x.shape
        y = torch.empty((B, 2 * Hh, 2 * W, C), dtype=torch.float32, device=x.device)
        H.check(_lib.lib().segsde_upsample2x_forward(H._p(H._f32(x)), H.nhwc_ld(x), B, Hh, W, C, H._p(y), C, H._stream(x)),
                "upsample2x_forward")
        return y

    @staticmethod
    def backward(ctx, g):
        g = Fn._c(g)
        B, H2, W2, C = g.shape
        dx = torch.empty((B, H2 // 2, W2 // 2, C), dtype=torch.float32, device=g.device)
        from .. import _lib
        H.check(_lib.lib().segsde_upsample2x_backward(H._p(g), H.nhwc_ld(g), B, H2 // 2, W2 // 2, C, H._p(dx), C, H._stream(g)),
                "upsample2x_backward")
        return dx


def upsample(x):
    """reference :202-205: nearest x2 of an NCHW-logical tensor.  (Inside the decoders the upsampling never materialises: the
    convolution's tile loader reads the low-resolution tensor, models/depth_decoder.py.)"""
    return Fn.to_nchw(_UpsampleFn.apply(Fn._c(Fn.to_nhwc(x))))


class _SmoothFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img):
        ctx.save_for_backward(disp, img)
        return H.smooth_loss_forward(disp, img).reshape(())

    @staticmethod
    def backward(ctx, g):
        disp, img = ctx.saved_tensors
        return H.smooth_loss_backward(disp, img, 1.0) * g, None


def get_smooth_loss(disp, img):
    """reference :208-221: edge-aware smoothness of ``disp`` [B,1,h,w] under ``img`` [B,3,h,w] (no gradient to ``img``:
    the reference's callers pass the input frame)"""
    return _SmoothFn.apply(disp.float().contiguous(), img.detach().float().contiguous())


class _SSIMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x, y)
        return H.ssim_map(x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        return H.ssim_map_backward(x, y, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])


class SSIM(nn.Module):
    """reference :224-254: per-channel SSIM loss map of two images (3x3 mean windows over the reflection-padded planes)"""

    def __init__(self):
        super().__init__()
        self.C1, self.C2 = 0.01 ** 2, 0.03 ** 2

    def forward(self, x, y):
        return _SSIMFn.apply(x.float().contiguous(), y.float().contiguous())
