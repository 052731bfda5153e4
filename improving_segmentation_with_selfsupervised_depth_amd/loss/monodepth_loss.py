"""Mirror of loss/monodepth_loss.py (same constructor kwargs and method protocol) on the HIP loss kernels.

``generate_images_pred`` fills ``outputs`` with the reference's API-visible tensors (("depth",0,s), ("sample",f,s),
("color",f,s), ("color_identity",f,s)); ``compute_losses`` evaluates the photometric + smoothness loss through ONE
autograd node whose backward is the hand-written kernel chain (SSIM/L1 -> auto-mask -> warp -> disparity upsample),
so nothing of the reference's ~2000-op graph (1 GB/img of saved intermediates) is recorded.
Gradients flow to ("disp", s) and ("cam_T_cam", 0, f) exactly as in the reference."""
import weakref

import torch

from .. import hipops as H
from ..functional import Function

_ONES = {}


class LazyOutputs(dict):
    """The model's ``outputs`` dict with entries that are computed on first access.  ``MonodepthLoss.generate_images_pred``
    registers the API-visible sampling grids ("sample", f, s) and depths ("depth", 0, s) here instead of materialising them
    every step: the reference's train step never reads them (train.py:479-514; SURVEY.md 8b "may be produced lazily"), only
    debug / evaluation code does.  Looks like a plain dict to everything else: ``in``, ``get``, iteration and ``len`` see the
    lazy keys (iteration materialises them)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._lazy = {}

    def set_lazy(self, key, thunk):
        dict.pop(self, key, None)
        self._lazy[key] = thunk

    def __missing__(self, key):
        thunk = self._lazy.pop(key, None)
        if thunk is None:
            raise KeyError(key)
        thunk()                                   # fills this key (and its siblings of the same launch)
        return dict.__getitem__(self, key)

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)
        dict.__setitem__(self, key, value)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default

    def materialize(self):
        for key in list(self._lazy):
            if key in self._lazy:
                self.__missing__(key)
        return self

    def keys(self):
        return dict.keys(self.materialize())

    def items(self):
        return dict.items(self.materialize())

    def values(self):
        return dict.values(self.materialize())

    def __iter__(self):
        return dict.__iter__(self.materialize())

    def __len__(self):
        return dict.__len__(self) + len(self._lazy)

    def update(self, *a, **k):
        for key, v in dict(*a, **k).items():
            self[key] = v

    def pop(self, key, *default):
        if key in self._lazy:
            self.__missing__(key)
        return dict.pop(self, key, *default)

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        self[key] = default
        return default

    def copy(self):
        """a plain dict of everything (the lazy entries are computed)"""
        return dict(dict.items(self.materialize()))

    def __eq__(self, other):
        return dict.__eq__(self.materialize(), other.materialize() if isinstance(other, LazyOutputs) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __reduce__(self):
        return (dict, (self.copy(),))               # pickles as the plain dict it stands for


def _one(dev):
    if dev not in _ONES:
        _ONES[dev] = torch.ones(1, dtype=torch.float32, device=dev)
    return _ONES[dev]


class _MonoLossFn(Function):
    @staticmethod
    def forward(ctx, obj, inputs, cache, outputs, *tensors):
        S, frames = obj.num_scales, list(obj.frame_ids[1:])
        disps = [t.detach().contiguous() for t in tensors[:S]]
        Ts = [t.detach().contiguous() for t in tensors[S:]]
        target = inputs[("color", 0, 0)].contiguous()
        B, _, Hh, W = target.shape
        dev = target.device
        automask = not obj.disable_automasking
        avg = bool(obj.avg_reprojection)
        nf = len(frames)
        if nf > 2:
            return _MonoLossFn._forward_staged(ctx, obj, inputs, cache, outputs, disps, Ts, frames)
        # one source frame (the stereo-only set (0, "s")): the two-frame kernels run on that frame twice.  The minimum / mean over
        # two identical candidates is the candidate, a tie goes to the first index like torch.min's, the tie-break noise of the one
        # identity channel (reference :163-164, shape [B, 1, H, W]) sits in both slots, and the two pose-gradient slots add up
        dup = nf == 1
        if dup:
            frames, Ts = frames * 2, Ts * 2
        srcs = [inputs[("color", f, 0)].contiguous() for f in frames]
        inv_K, K = inputs[("inv_K", 0)].contiguous(), inputs[("K", 0)].contiguous()
        ident = None
        if automask:   # identical for every scale (reference recomputes it 4x, monodepth_loss.py:139-147)
            ident = H.photometric_identity(srcs[0], srcs[1], target, obj.no_ssim)
        losses, saved = [], []
        for s in range(S):
            colors = []
            for j, f in enumerate(frames):
                if dup and j == 1:
                    colors.append(colors[0])
                    continue
                col = cache.get(("color", f, s)) if cache is not None else None
                if col is None:
                    col, _, _ = H.warp_forward(disps[s], inv_K, K, Ts[j], srcs[j], obj.min_depth, obj.max_depth)
                colors.append(col)
            noise = None
            if automask:
                if obj.tiebreak_noise is not None:
                    noise = obj.tiebreak_noise[s].to(dev).float().contiguous()
                else:
                    noise = torch.randn((B, 1 if avg else nf, Hh, W), device=dev)
                if dup and not avg:
                    noise = noise.expand(B, 2, Hh, W).contiguous()
            ssum, sel, isel = H.photometric_forward(colors[0], colors[1], target, ident, noise, obj.no_ssim, avg)
            if automask:
                outputs["identity_selection/{}".format(s)] = isel
            color_s = inputs[("color", 0, s)].contiguous()
            smooth, mean_disp = H.smoothness_forward(disps[s], color_s)
            loss_s = ssum[0] / float(B * Hh * W) + smooth[0] * (obj.disparity_smoothness / (2 ** s))
            losses.append(loss_s)
            saved.append((sel, colors, mean_disp, color_s))
        total = losses[0]
        for l in losses[1:]:
            total = total + l
        total = total / S
        ctx.obj, ctx.saved, ctx.disps, ctx.Ts, ctx.srcs, ctx.target = obj, saved, disps, Ts, srcs, target
        ctx.geo = (inv_K, K)
        ctx.automask, ctx.avg, ctx.dup = automask, avg, dup
        return (total,) + tuple(losses)

    @staticmethod
    def _forward_staged(ctx, obj, inputs, cache, outputs, disps, Ts, frames):
        """Three or more source frames (monodepth2's (0, -1, 1, "s"); reference :136-177 loops over ``frame_ids[1:]`` whatever their
        number).  The packed kernels above are written for the two frames of every shipped configuration; a larger set runs the
        per-stage entry points frame by frame: warp, SSIM + L1 error into a channel of [B, nf, H, W], one n-way minimum."""
        S, nf = obj.num_scales, len(frames)
        target = inputs[("color", 0, 0)].contiguous()
        B, _, Hh, W = target.shape
        dev = target.device
        automask, avg = not obj.disable_automasking, bool(obj.avg_reprojection)
        srcs = [inputs[("color", f, 0)].contiguous() for f in frames]
        inv_K, K = inputs[("inv_K", 0)].contiguous(), inputs[("K", 0)].contiguous()
        ident = None
        if automask:                                   # the same for every scale
            ident = torch.empty((B, nf, Hh, W), dtype=torch.float32, device=dev)
            for j in range(nf):
                H.reprojection_error(srcs[j], target, obj.no_ssim, ident[:, j])
        losses, saved = [], []
        for s in range(S):
            colors = []
            reproj = torch.empty((B, nf, Hh, W), dtype=torch.float32, device=dev)
            for j, f in enumerate(frames):
                col = cache.get(("color", f, s)) if cache is not None else None
                if col is None:
                    col, _, _ = H.warp_forward(disps[s], inv_K, K, Ts[j], srcs[j], obj.min_depth, obj.max_depth)
                colors.append(col)
                H.reprojection_error(col, target, obj.no_ssim, reproj[:, j])
            noise = None
            if automask:
                if obj.tiebreak_noise is not None:
                    noise = obj.tiebreak_noise[s].to(dev).float().contiguous()
                else:
                    noise = torch.randn((B, 1 if avg else nf, Hh, W), device=dev)
            ssum, sel, isel = H.automask_min(ident, noise, reproj, avg)
            if automask:
                outputs["identity_selection/{}".format(s)] = isel
            color_s = inputs[("color", 0, s)].contiguous()
            smooth, mean_disp = H.smoothness_forward(disps[s], color_s)
            losses.append(ssum[0] / float(B * Hh * W) + smooth[0] * (obj.disparity_smoothness / (2 ** s)))
            saved.append((sel, colors, mean_disp, color_s))
        total = losses[0]
        for l in losses[1:]:
            total = total + l
        total = total / S
        ctx.obj, ctx.saved, ctx.disps, ctx.Ts, ctx.srcs, ctx.target = obj, saved, disps, Ts, srcs, target
        ctx.geo = (inv_K, K)
        ctx.automask, ctx.avg, ctx.dup, ctx.staged = automask, avg, False, True
        return (total,) + tuple(losses)

    @staticmethod
    def _backward_staged(ctx, g_total, g_scales):
        obj, S, nf = ctx.obj, ctx.obj.num_scales, len(ctx.srcs)
        inv_K, K = ctx.geo
        target = ctx.target
        B, _, Hh, W = target.shape
        dev = target.device
        gd_out = []
        gT_acc = [torch.zeros((B, 4, 4), dtype=torch.float32, device=dev) for _ in range(nf)]
        for s in range(S):
            sel, colors, mean_disp, color_s = ctx.saved[s]
            w_s = (g_total / S + g_scales[s]).reshape(1).contiguous()
            greproj = H.automask_min_backward(sel, ctx.automask, nf, ctx.avg, 1.0 / float(B * Hh * W))
            gup = torch.zeros((B, Hh, W), dtype=torch.float32, device=dev)
            for j in range(nf):
                gpred = H.reprojection_error_backward(colors[j], target, greproj[:, j], obj.no_ssim)
                gT = torch.zeros((B, 4, 4), dtype=torch.float32, device=dev)
                H.warp_backward(gpred, ctx.disps[s], inv_K, K, ctx.Ts[j], ctx.srcs[j], obj.min_depth, obj.max_depth, gup, gT)
                gT_acc[j].add_(gT * w_s)
            hs, ws = ctx.disps[s].shape[-2:]
            gdisp = H.resize_bilinear_backward(gup.reshape(B, Hh, W, 1), (hs, ws), False).reshape(B, 1, hs, ws)
            H.smoothness_backward(ctx.disps[s], color_s, mean_disp, obj.disparity_smoothness / (2 ** s), gdisp)
            gd_out.append(H.axpby_dev(w_s, gdisp))
        return (None, None, None, None) + tuple(gd_out) + tuple(gT_acc)

    @staticmethod
    def backward(ctx, g_total, *g_scales):
        if getattr(ctx, "staged", False):
            return _MonoLossFn._backward_staged(ctx, g_total, g_scales)
        obj, S = ctx.obj, ctx.obj.num_scales
        inv_K, K = ctx.geo
        target = ctx.target
        B, _, Hh, W = target.shape
        dev = target.device
        gd_out = []
        gT_acc = [torch.zeros((B, 4, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        for s in range(S):
            sel, colors, mean_disp, color_s = ctx.saved[s]
            w_s = (g_total / S + g_scales[s]).reshape(1).contiguous()    # upstream weight of loss/s (device scalar)
            gup = H.photometric_backward(colors[0], colors[1], target, sel, ctx.automask, ctx.disps[s], inv_K, K, ctx.Ts[0],
                                         ctx.Ts[1], ctx.srcs[0], ctx.srcs[1], obj.min_depth, obj.max_depth, obj.no_ssim,
                                         ctx.avg, 1.0 / float(B * Hh * W), w_s, gT_acc[0], gT_acc[1])
            hs, ws = ctx.disps[s].shape[-2:]
            gdisp = H.resize_bilinear_backward(gup.reshape(B, Hh, W, 1), (hs, ws), False).reshape(B, 1, hs, ws)
            H.smoothness_backward(ctx.disps[s], color_s, mean_disp, obj.disparity_smoothness / (2 ** s), gdisp)
            gd_out.append(H.axpby_dev(w_s, gdisp))
        if ctx.dup:
            gT_acc = [gT_acc[0] + gT_acc[1]]
        return (None, None, None, None) + tuple(gd_out) + tuple(gT_acc)


class MonodepthLoss:
    def __init__(self, num_scales, frame_ids, height, width, batch_size, min_depth, max_depth, test_min_depth,
                 test_max_depth, disparity_smoothness, no_ssim, avg_reprojection, disable_automasking, crop_h=None,
                 crop_w=None, is_train=True):
        self.num_scales = num_scales
        self.scales = list(range(self.num_scales))
        self.height = height if crop_h is None or not is_train else crop_h      # reference :22-23
        self.width = width if crop_w is None or not is_train else crop_w
        self.batch_size = batch_size
        self.frame_ids = list(frame_ids)
        self.min_depth, self.max_depth = min_depth, max_depth
        self.test_min_depth, self.test_max_depth = test_min_depth, test_max_depth
        self.disparity_smoothness = disparity_smoothness
        self.no_ssim = no_ssim
        self.avg_reprojection = avg_reprojection
        self.disable_automasking = disable_automasking
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.depth_metric_names = ["abs_rel", "sq_rel", "rms", "log_rms", "a1", "a2", "a3"]
        # source frames: two (the shipped configs' temporal pair, or one temporal frame + the stereo frame "s") run the packed
        # photometric kernels, one (the stereo-only set (0, "s")) runs them as a pair of itself, three or more (monodepth2's
        # (0, -1, 1, "s")) the per-stage kernels frame by frame (_MonoLossFn._forward_staged)
        if not 2 <= len(self.frame_ids) <= 9:
            raise NotImplementedError("MonodepthLoss: one to eight source frames expected, got frame_ids = %r" % (self.frame_ids,))
        self.tiebreak_noise = None    # tests: dict scale -> tensor replacing the fresh randn of reference :163-164
        self._cache = None

    @staticmethod
    def _pose(inputs, outputs, f):
        """reference :82-85: the stereo frame "s" is warped with the fixed baseline transform of the batch, every other frame with
        the pose network's prediction"""
        return inputs["stereo_T"] if f == "s" else outputs[("cam_T_cam", 0, f)]

    def generate_depth_test_pred(self, outputs):
        """reference :54-62 (eval only)"""
        assert tuple(outputs[("disp", 0)].shape[-2:]) == (self.height, self.width), outputs[("disp", 0)].shape[-2:]
        for s in self.scales:
            outputs[("depth", 0, s)] = H.disp_to_depth_upsampled(outputs[("disp", s)].detach(), (self.height, self.width),
                                                                 self.test_min_depth, self.test_max_depth)

    def generate_images_pred(self, inputs, outputs):
        """reference :64-102; tensors written to ``outputs`` are detached (the differentiable path is compute_losses)"""
        assert tuple(outputs[("disp", 0)].shape[-2:]) == (self.height, self.width), \
            f'{outputs[("disp", 0)].shape[-2:]} should be {(self.height, self.width)} '
        cache = {}
        lazy = isinstance(outputs, LazyOutputs)    # what this package's models return; a plain dict gets everything eagerly
        for s in self.scales:
            disp = outputs[("disp", s)].detach().contiguous()
            for i, f in enumerate(self.frame_ids[1:]):
                T = self._pose(inputs, outputs, f).detach().float().contiguous()
                color, grid, depth = H.warp_forward(disp, inputs[("inv_K", 0)], inputs[("K", 0)], T,
                                                    inputs[("color", f, 0)], self.min_depth, self.max_depth,
                                                    want_grid=not lazy, want_depth=(i == 0 and not lazy))
                if lazy:
                    # the thunk holds the very tensors this launch read (not the `inputs` dict: a caller that re-fills the
                    # dict for the next batch must not change what a later access computes); they stay alive with `outputs`
                    # (the dict itself through a weak reference: a thunk stored IN the dict that also holds the dict is a
                    # reference cycle -- the step's whole autograd graph hangs off `outputs`, and a cycle is only freed when
                    # Python's cyclic collector happens to run: every few steps two steps' activations were alive at once,
                    # 110 GB peak instead of 75 at cfg3)
                    def fill(disp=disp, T=T, f=f, s=s, i=i, invK=inputs[("inv_K", 0)], K=inputs[("K", 0)],
                             src=inputs[("color", f, 0)], out_ref=weakref.ref(outputs), lo=self.min_depth, hi=self.max_depth):
                        out = out_ref()
                        if out is None:
                            return
                        _, g_, d_ = H.warp_forward(disp, invK, K, T, src, lo, hi, want_grid=True, want_depth=(i == 0))
                        dict.__setitem__(out, ("sample", f, s), g_)
                        out._lazy.pop(("sample", f, s), None)
                        if i == 0:
                            dict.__setitem__(out, ("depth", 0, s), d_)
                            out._lazy.pop(("depth", 0, s), None)
                    outputs.set_lazy(("sample", f, s), fill)
                    if i == 0:
                        outputs.set_lazy(("depth", 0, s), fill)
                else:
                    if i == 0:
                        outputs[("depth", 0, s)] = depth
                    outputs[("sample", f, s)] = grid
                outputs[("color", f, s)] = color
                cache[("color", f, s)] = color
                if not self.disable_automasking:
                    outputs[("color_identity", f, s)] = inputs[("color", f, 0)]
        self._cache = (cache, [outputs[("disp", s)] for s in self.scales])

    def compute_losses(self, inputs, outputs):
        """reference :118-192 -> {"loss/0".."loss/3", "loss"}"""
        disps = [outputs[("disp", s)] for s in self.scales]
        Ts = [self._pose(inputs, outputs, f).float() for f in self.frame_ids[1:]]
        cache = None
        if self._cache is not None and all(a is b for a, b in zip(self._cache[1], disps)):
            cache = self._cache[0]           # warped frames from generate_images_pred for these very tensors
        self._cache = None
        res = _MonoLossFn.apply(self, inputs, cache, outputs, *disps, *Ts)
        losses = {"loss/{}".format(s): res[1 + s] for s in self.scales}
        losses["loss"] = res[0]
        return losses
