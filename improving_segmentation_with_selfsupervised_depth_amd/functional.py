"""autograd glue: each class wraps one fused HIP op pair (forward + hand-written backward) from hipops.

All activations flowing between these Functions are NHWC fp32 tensors.  Module-level APIs of the reference speak
NCHW; ``to_nhwc`` / ``to_nchw`` convert at that edge for free when the tensor is already channels-last in memory
(which is what every module of this package returns), and with a HIP layout kernel otherwise.
"""
import collections
import weakref

import torch
from torch.autograd import Function as _TorchFunction

from . import hipops as H


class Function(_TorchFunction):
    """torch.autograd.Function whose ``apply`` goes straight to the C++ binding.  The stock Python ``apply`` first looks for a
    ``setup_context``, asks functorch whether a transform is active and walks the arguments for dead functorch wrappers -- 3-4 us
    for a 14-argument ConvFn, x 211 Functions per cfg1 step whose whole budget is ~20 us per launch.  Nothing here uses functorch
    transforms or ``setup_context``; arguments are positional."""

    @classmethod
    def apply(cls, *args):
        return super(_TorchFunction, cls).apply(*args)


# Fusion hand-offs travel as attributes / shared boxes next to the tensors (conv -> BatchNorm statistics partials, conv -> fused
# activation backward, residual / fan-out gradient accumulation) and fall back to the unfused kernels when a view or an
# in-place edit got in between.  The fallbacks are correct but slower, so every hand-off is COUNTED here: ``taken`` vs
# ``missed`` per kind (bench.py prints the per-step numbers, tests pin them for the models of the path) -- a maintainer who
# inserts an op between a convolution and its BatchNorm sees ``bn_stats.missed`` go up instead of silently losing 8 % of a step.
FUSIONS = {}


def _grad_dest(param):
    """the parameter's slice of a data-parallel gradient bucket, if a GradAllReducer handed it out as the place to write this
    step's gradient (ddp.grad_destination: zero-copy buckets); None in single-process runs"""
    from . import ddp
    return ddp.grad_destination(param) if ddp._GRAD_DEST else None


def fp32_region(fn):
    """Entry points the reference calls under ``torch.cuda.amp.autocast`` when ``amp: True`` (train.py:468, 502: the model's forward
    and the segmentation loss).  This package computes in fp32 -- at least the reference's precision -- so autocast is switched off
    for the duration of the call (ATen helpers inside, e.g. the pose networks' ``torch.cat``, then stay fp32 whatever their
    autocast policy) and floating-point tensor arguments arrive as fp32.  The reference's GradScaler protocol around it works
    unchanged: every backward kernel is linear in the incoming gradient, and scaling by a power of two is exact."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        if torch.is_autocast_enabled("cuda"):
            old = H.COMPUTE_F16[0]
            H.COMPUTE_F16[0] = AMP_COMPUTE[0] == "f16"
            try:
                with torch.autocast(device_type="cuda", enabled=False):
                    return fn(*args, **kwargs)
            finally:
                H.COMPUTE_F16[0] = old
        return fn(*args, **kwargs)
    return wrapped


# What the convolutions compute in under autocast.  "f32" (default): the fp32 kernels -- an `amp: True` run is then a run at the
# reference's `amp: False` precision.  "f16" (SEGSDE_AMP_COMPUTE=f16): the reference's reduced-precision arithmetic -- every
# convolution created inside an autocast region rounds its operands to fp16 in the kernel and multiplies them on
# v_mfma_f32_32x32x16_f16 with fp32 accumulation (hipops.ConvGeom.compute; forward, data-gradient and weight gradient of that
# convolution alike, like autocast's own backward), the Winograd routes are not taken, BatchNorm / losses / optimizer stay fp32
# like under autocast.  Activations and weights stay fp32 in memory.
AMP_COMPUTE = [__import__("os").environ.get("SEGSDE_AMP_COMPUTE", "f32").lower()]


def fusion(kind, taken):
    c = FUSIONS.setdefault(kind, [0, 0])
    c[0 if taken else 1] += 1


def fusion_report(reset=False):
    out = {k: {"taken": v[0], "missed": v[1]} for k, v in sorted(FUSIONS.items())}
    if reset:
        FUSIONS.clear()
    return out


def _c(t):
    """dense NHWC tensor (grad tensors produced by our kernels already are)"""
    if t.is_contiguous():
        return t
    if t.dim() == 4 and t.stride(3) == 1:
        try:
            H.nhwc_ld(t)     # a channel slice of a wider NHWC buffer: kernels take its pixel pitch
            return t
        except ValueError:
            pass
    if t.dim() == 4 and t.permute(0, 3, 1, 2).is_contiguous():   # an NCHW-contiguous grad seen through an NHWC view
        return H.nchw_to_nhwc(t.permute(0, 3, 1, 2))
    return t.contiguous()


class _ToNHWC(Function):
    @staticmethod
    def forward(ctx, x, mean, std, pad_to):
        ctx.std, ctx.C = std, x.shape[1]
        return H.nchw_to_nhwc(x, mean, std, pad_to)

    @staticmethod
    def backward(ctx, g):
        gx = H.nhwc_to_nchw(_c(g[..., :ctx.C]))
        if ctx.std != 1.0:
            gx = H.axpby(1.0 / ctx.std, gx)
        return gx, None, None, None


class _ToNCHWDense(Function):
    @staticmethod
    def forward(ctx, x):
        return H.nhwc_to_nchw(x)

    @staticmethod
    def backward(ctx, g):
        return H.nchw_to_nhwc(g.contiguous())


def to_nhwc(x, mean=0.0, std=1.0, pad_to=1):
    """NCHW-logical tensor -> NHWC tensor (zero-copy when x is channels-last in memory and no normalisation);
    pad_to > 1 appends zero channels up to a multiple of pad_to."""
    if mean == 0.0 and std == 1.0 and x.shape[1] % pad_to == 0:
        v = x.permute(0, 2, 3, 1)
        if v.is_contiguous():
            return v
    return _ToNHWC.apply(x, mean, std, pad_to)


def to_nchw(x_nhwc):
    """NHWC tensor -> NCHW-logical view (channels-last memory format; what the package's modules return)."""
    return x_nhwc.permute(0, 3, 1, 2)


def to_nchw_dense(x_nhwc):
    return _ToNCHWDense.apply(x_nhwc)


class StemFn(Function):
    """y = conv7x7/2((image - mean) / std, weight) for the network stems (models/resnet_encoder.py:90-93: the normalisation
    followed by encoder.conv1).  The image is data: no gradient flows to it (callers route an image that requires one
    through to_nhwc + ConvFn instead)."""

    @staticmethod
    def forward(ctx, image, weight, mean, std, stats_out=None, wstem=None):
        xpad = H.stem_input(image, mean, std)
        if wstem is None:
            wstem = H.stem_pack(weight)
        C = weight.shape[1]
        if stats_out is not None:
            y, part = H.stem_forward(xpad, wstem, C, want_stats=True)
            stats_out.append(part)
        else:
            y = H.stem_forward(xpad, wstem, C)
        ctx.C = C
        ctx.save_for_backward(xpad)
        return y

    @staticmethod
    def backward(ctx, dy):
        (xpad,) = ctx.saved_tensors
        dw = H.stem_wgrad(xpad, _c(dy), ctx.C) if ctx.needs_input_grad[1] else None
        return None, dw, None, None, None, None


class ConvFn(Function):
    """y = act(conv([up2x?(x0) | x1], weight) + bias); geometry in ``g`` (hipops.ConvGeom)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, g, act, stats_out=None, grad_box=None, packs=None, x0_act=None, act_box=None,
                fold_cache=None, wino_cache=None, skip_box=None):
        """stats_out: optional list; receives the BatchNorm statistics partials of y (or None) -- see Conv2d.forward.
        grad_box: optional dict shared with the BNActFn that adds this conv's input as a residual (see SplitFn): when its
        backward has already left the residual-path gradient there, this conv's data-gradient is accumulated onto it.
        x0_act: x0 is the activated output of a producing ConvFn (its raw output tensor, see ActGradFn): the data-gradient
        w.r.t. x0 is returned already multiplied by that activation's derivative.
        NOTE for act != "none": the gradient arriving at this Function's output is taken to be w.r.t. the PRE-activation --
        consumers reach the output either through ActGradFn (which applies the derivative) or as a fused x0_act consumer."""
        ctx.grad_box = grad_box
        ctx.skip_box = skip_box          # like grad_box, for the second source x1 (an encoder feature several decoders read)
        ctx.x0_act, ctx.act_box = x0_act, act_box
        x0 = _c(x0) if x0.stride(-1) != 1 else x0
        ctx.wd = None
        ctx.wd_gen = None
        if packs is not None:
            wp, ctx.wd = packs                          # the owning module's cache (valid for this version of the weight)
            ctx.wd_gen = getattr(ctx.wd, "_segsde_gen", None)   # multi-pack buffers are rewritten by the next prepack
        elif ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
            wp, ctx.wd = H.pack_weight_both(weight)     # the data-gradient pack is needed by backward: one launch for both
        else:
            wp = H.pack_weight(weight, False)
        ctx.wp = wp if g.reflect else None      # the mirrored convolution's data-gradient border kernel reads the forward pack
        ctx.fold = None
        # Winograd route (3x3 / stride 1 / many channels): the transformed weight packs, cached by the owning module for the
        # active weight_pack_scope like the plain packs
        ctx.wino = None
        # kn: the one-kernel Winograd route (64 .. 256 channels, one source) and its [16][K][N] packs; otherwise the grouped-GEMM
        # route from 256 channels on (two sources, dilation, 512 channels)
        # (round 5: the one-kernel route also takes the decoder's [upsample(x0) | x1] layers, forward only)
        up_f = 2 if g.up0 else 1
        kn = H.winograd_fused_ok(g, x0.shape[0], x0.shape[1] * up_f, x0.shape[2] * up_f)
        # (and, whatever route the forward takes, the skip-source data-gradient of the two-source layers: the flipped pack)
        kn_d2 = (x1 is not None and g.up0 and (ctx.needs_input_grad[1])
                 and H.winograd_fused_dgrad2_ok(g, x0.shape[0], x0.shape[1] * up_f, x0.shape[2] * up_f))
        if kn or kn_d2 or (not g.up0 and H.winograd_ok(g, x0.shape[0], x0.shape[1], x0.shape[2])):
            knl = kn or kn_d2                              # pack layout: [16][K][N] of the one-kernel route
            if (wino_cache is not None and wino_cache.get("key") is not None and wino_cache.get("key") == wino_cache.get("want")
                    and bool(wino_cache.get("kn")) == knl):
                ctx.wino = wino_cache["packs"]
            else:
                ctx.wino = H.winograd_pack(weight, kn=knl)
                if wino_cache is not None and wino_cache.get("want") is not None:
                    wino_cache["packs"], wino_cache["key"], wino_cache["kn"] = ctx.wino, wino_cache["want"], knl
        wino_f = None if (ctx.wino is None or (kn_d2 and not kn)) else ctx.wino[0]
        keep_v = wino_f is not None and not kn and ctx.needs_input_grad[2]   # the weight gradient reuses the forward's transformed input
        H.WINO_V[0] = None
        if stats_out is not None:
            y, part = H.conv_forward(g, x0, x1, wp, bias, act, want_stats=True, wino=wino_f, keep_v=keep_v)
            stats_out.append(part)
        else:
            if H.upfold_ok(g, 4 * x0.shape[0] * x0.shape[1] * x0.shape[2]) and x0.is_contiguous() and (x1 is None or x1.is_contiguous()):
                # decoder Conv3x3 on [upsample(x0) | x1]: the upsample-folded route (4 taps instead of 9 on the upsampled channels)
                # (the owning module caches the folded packs next to the plain ones for the active weight_pack_scope)
                if fold_cache is not None and fold_cache.get("key") is not None and fold_cache.get("key") == fold_cache.get("want"):
                    ctx.fold = fold_cache["fold"]
                else:
                    ctx.fold = H.upfold_pack(weight, g.C0)
                    if fold_cache is not None and fold_cache.get("want") is not None:
                        fold_cache["fold"], fold_cache["key"] = ctx.fold, fold_cache["want"]
            y = H.conv_forward(g, x0, x1, wp, bias, act, wfold=None if ctx.fold is None else ctx.fold[0], wino=wino_f, keep_v=keep_v)
        ctx.wino_v, H.WINO_V[0] = H.WINO_V[0], None
        ctx.g, ctx.act = g, act
        ctx.in_hw = (x0.shape[1] * (2 if g.up0 else 1), x0.shape[2] * (2 if g.up0 else 1))
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x0, x1, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, weight = ctx.saved_tensors
        g = ctx.g
        dz = _c(dy)                      # act != "none": already the pre-activation gradient (ActGradFn / fused consumers)
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        dbias = None
        if need_b:
            stash = ctx.act_box.pop("dbias", None) if ctx.act_box is not None else None
            # the bias gradient ActGradFn reduced in its pass is valid only if this IS its dz and nothing was added to it since
            # (autograd may accumulate a second contribution into the same buffer in place: the version counter moves)
            same = stash is not None and stash[0].data_ptr() == dz.data_ptr() and stash[0]._version == stash[1] == dz._version
            dbias = stash[2] if same else H.colsum(dz)
            fusion("bias_grad_from_activation_pass", bool(same))
        actgrad = (x0, ctx.x0_act) if ctx.x0_act is not None else None
        dx0 = dx1 = dw = None
        if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
            wd = ctx.wd if ctx.wd is not None else H.pack_weight(weight, True)
            if ctx.wd_gen is not None and getattr(wd, "_segsde_gen", None) != ctx.wd_gen:
                raise RuntimeError("the weight packs of this graph were rebuilt by a later weight_pack_scope(model): a graph "
                                   "must be consumed before the next prepack of the same weights")
            dx0 = None
            box = ctx.grad_box
            if box is not None and box.get("g") is not None and x1 is None:
                dx0, dx1 = H.conv_dgrad(g, dz, wd, weight.detach(), ctx.in_hw, accumulate_into=box["g"], actgrad=actgrad,
                                        wino=None if ctx.wino is None else ctx.wino[1], wfpack=ctx.wp)
                if dx0 is not None:
                    box["fused"] = True      # dx0 IS the residual-path gradient, now holding the sum
            if dx0 is None:
                sbox = ctx.skip_box if (x1 is not None and ctx.needs_input_grad[1]) else None
                dx0, dx1 = H.conv_dgrad(g, dz, wd, weight.detach(), ctx.in_hw, actgrad=actgrad, fold=ctx.fold,
                                        need0=ctx.needs_input_grad[0], need1=x1 is not None and ctx.needs_input_grad[1],
                                        wino=None if ctx.wino is None else ctx.wino[1],
                                        accumulate_skip_into=None if sbox is None else sbox.get("g"), wfpack=ctx.wp)
                if sbox is not None and dx1 is not None:
                    if H.SKIP_ACCUMULATED[0]:
                        sbox["fused"] = True      # dx1 IS the shared tensor, now holding the sum
                    elif sbox.get("publish") and sbox.get("g") is None and dx1.is_contiguous():
                        sbox["g"] = dx1           # the next consumer's skip gradient is accumulated onto this tensor
                        sbox["fused"] = True
                if box is not None and box.get("publish") and box.get("g") is None and x1 is None and not g.up0 \
                        and dx0.is_contiguous():
                    box["g"] = dx0           # FanoutFn: the next consumer's data-gradient is accumulated onto this tensor
                    box["fused"] = True
            if actgrad is not None and ctx.needs_input_grad[0]:
                # counted from what the kernel did: a shape whose epilogue cannot take the derivative ran it as a separate pass
                fusion("activation_backward_in_dgrad_epilogue", bool(H.ACTGRAD_FUSED[0]))
            if not ctx.needs_input_grad[0]:
                dx0 = None
            if x1 is None or not ctx.needs_input_grad[1]:
                dx1 = None
        if ctx.needs_input_grad[2]:
            dw = H.conv_wgrad(g, x0, x1, dz, wino_v=ctx.wino_v, out=_grad_dest(weight))
            ctx.wino_v = None
        return dx0, dx1, dw, dbias, None, None, None, None, None, None, None, None, None, None


class ActGradFn(Function):
    """The activation of a ConvFn(act=...) seen from autograd: forward hands the already activated tensor on (the
    activation itself ran in the conv epilogue); backward is the activation backward dz = dy * act'(y) as its own pass
    (plus, in the same pass, the bias gradient, left in ``box`` for the ConvFn).  Consumers that can apply act'(y) in their
    own data-gradient epilogue bypass this node: they take the ConvFn's raw output (``y._preact``) and return the
    pre-activation gradient directly -- autograd adds both kinds of contribution at the ConvFn's output."""

    passes = 0               # diagnostics / tests: how many separate activation-backward passes ran

    @staticmethod
    def forward(ctx, y, act, box):
        ctx.act, ctx.box = act, box
        ctx.save_for_backward(y)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        ActGradFn.passes += 1
        fusion("activation_backward_in_dgrad_epilogue", False)
        dz, dbias = H.act_backward(_c(dy), y, ctx.act, need_dbias=bool(ctx.box.get("need_dbias")))
        if dbias is not None:
            ctx.box["dbias"] = (dz, dz._version, dbias)
        return dz, None, None


class BNActFn(Function):
    """y = dropout(act(BN(x) + residual)); training=True uses batch statistics and updates the running buffers."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, training, momentum, eps, act, drop_p, seed,
                partials=None, grad_box=None, nbt=None):
        ctx.grad_box = grad_box
        x = _c(x)
        if training and partials is not None:
            # the producing convolution already summed x and x^2 per tile in its epilogue
            mean, invstd = H.bn_stats_from_partials(partials, x.numel() // x.shape[-1], running_mean, running_var, momentum,
                                                    eps, update_running=running_mean is not None, num_batches_tracked=nbt)
        elif training:
            mean, invstd = H.bn_stats(x, running_mean, running_var, momentum, eps, update_running=running_mean is not None,
                                      num_batches_tracked=nbt)
        else:
            mean, invstd = H.bn_eval_stats(running_mean, running_var, eps)
        y = H.bn_apply(x, mean, invstd, gamma, beta, residual, act, drop_p, seed)
        # the backward pass reads the saved output only where the activation mask is not a function of x alone
        remask = act in ("none", "relu") and residual is None and not drop_p
        ctx.cfg = (act, drop_p, seed, training, residual is not None, remask)
        ctx.beta_param = beta          # (the parameter object: its gradient-bucket destination is looked up in backward)
        ctx.save_for_backward(x, gamma, beta if remask else None, mean, invstd, None if remask else y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd, y = ctx.saved_tensors
        act, drop_p, seed, training, has_res, remask = ctx.cfg
        want_g = gamma is not None and ctx.needs_input_grad[1]
        bp = ctx.beta_param
        dx, dres, dgamma, dbeta = H.bn_backward(_c(dy), y, x, mean, invstd, gamma, act, drop_p, seed, batch_stats=training,
                                                need_dx=ctx.needs_input_grad[0],
                                                need_dres=has_res and ctx.needs_input_grad[3], beta=beta,
                                                dgamma_out=_grad_dest(gamma) if want_g else None,
                                                dbeta_out=_grad_dest(bp) if (want_g and bp is not None and ctx.needs_input_grad[2]) else None)
        ctx.beta_param = None
        if gamma is None or not ctx.needs_input_grad[1]:
            dgamma = dbeta = None
        if ctx.grad_box is not None and dres is not None:
            ctx.grad_box["g"] = dres         # the block's first conv may add its data-gradient onto this tensor
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None, None


class SplitFn(Function):
    """x -> (x, x) for a residual block: one copy feeds the block's first convolution, the other the skip connection.
    Backward combines the two gradients; when the convolution's data-gradient was accumulated in its kernel epilogue
    directly onto the skip-path gradient (``box["fused"]``, both incoming gradients are then the same tensor) there is
    nothing left to add -- the elementwise gradient-accumulation pass of a plain autograd graph (12 B per element, 41
    times per ResNet-101 + ResNet-18 step) disappears."""

    fused_count = 0          # diagnostics / tests: how many backward passes found the sum already formed

    @staticmethod
    def forward(ctx, x, box):
        ctx.box = box
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g_main, g_skip):
        box = ctx.box
        fused = bool(box.get("fused")) and g_main is not None and g_skip is not None and g_main.data_ptr() == g_skip.data_ptr()
        box.clear()
        if fused:
            SplitFn.fused_count += 1
            fusion("residual_grad_accumulate", True)
            return g_main, None
        if g_main is None:
            return g_skip, None
        if g_skip is None:
            return g_main, None
        fusion("residual_grad_accumulate", False)
        return g_main + g_skip, None


class FanoutFn(Function):
    """x -> n views of x for n consumers (ASPP: the four convolution branches and the pooling branch read the same
    tensor).  The consumers that are convolutions share ``box``: the first data-gradient that is computed becomes the
    accumulation target, every later one is added onto it inside its kernel epilogue (conv descriptor ``accumulate``);
    backward sums what is left -- the shared tensor once plus the gradients of the non-convolution consumers -- instead of
    n-1 elementwise adds over the full tensor."""

    shared_count = 0         # diagnostics / tests: gradients that arrived already summed in the shared tensor

    @staticmethod
    def forward(ctx, x, box, n):
        ctx.box = box
        box["publish"] = True
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        box = ctx.box
        shared = box.get("g") if box.get("fused") else None
        box.clear()
        total, used_shared = None, False
        for gi in grads:
            if gi is None:
                continue
            if shared is not None and gi.data_ptr() == shared.data_ptr():
                if used_shared:
                    FanoutFn.shared_count += 1
                    fusion("fanout_grad_accumulate", True)
                    continue                 # the same accumulated tensor handed back by another consumer
                used_shared = True
            if total is not None:
                fusion("fanout_grad_accumulate", False)
            total = gi if total is None else H.axpby(1.0, _c(total), 1.0, _c(gi))
        return total, None, None


# Encoder features are read by the next encoder stage AND as the skip source of every decoder (models/depth_decoder.py:93-101):
# plain autograd sums those gradients with one 12-byte-per-element pass per extra consumer (104 such adds per ResNet-101 joint
# step, 3.4 ms).  ``fan_feature`` turns a feature into n views of a FanoutFn whose convolution consumers accumulate inside
# their data-gradient epilogues; the views meant for the decoders wait in a registry keyed by the feature's storage, because
# the feature crosses the module boundary as an NCHW-logical view (a new tensor object) -- ``take_fan_view`` hands a decoder
# its own view plus the shared box, or the tensor it was given when there is nothing registered for it.
_FANS = collections.OrderedDict()      # at most _FANS_MAX entries: parked views keep their feature alive, old ones are dropped
_FANS_MAX = 16


def release_fans(owner):
    """drop the views an earlier forward of ``owner`` parked and nobody fetched (a decoder that was not run in that pass): they
    would keep that forward's features -- and, under retain_graph, its graph -- alive until _FANS_MAX newer entries push them out"""
    for k in [k for k, ent in _FANS.items() if ent[3] == owner]:
        del _FANS[k]


def fan_feature(x, n_spare, n_main=1, owner=None):
    """-> (the n_main views the producer's own module continues with, the shared box or None).  n_spare more views are parked for
    ``take_fan_view``.  Without a gradient (or with nobody to share with) the views are x itself.  owner: an id the producer
    passes to ``release_fans`` at the start of its next forward."""
    if n_spare + n_main <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n_main, None
    box = {}
    views = FanoutFn.apply(x, box, n_main + n_spare)
    if n_spare > 0:
        base = views[0]._base if views[0]._base is not None else views[0]
        _FANS.pop(id(base), None)
        _FANS[id(base)] = (weakref.ref(base), box, list(views[n_main:]), owner)
        while len(_FANS) > _FANS_MAX:          # a forward whose decoders never came for their views (or a dead feature)
            _FANS.popitem(last=False)
    return tuple(views[:n_main]), box


def take_fan_view(t):
    """t: an NHWC tensor as a consumer received it.  -> (the view to read instead, shared box) or (t, None)"""
    base = t._base if t._base is not None else t
    ent = _FANS.get(id(base))
    if ent is None or ent[0]() is not base or not ent[2]:
        return t, None
    v = ent[2].pop()
    if v.shape != t.shape or v.data_ptr() != t.data_ptr() or v.stride() != t.stride():
        ent[2].append(v)
        return t, None
    if not ent[2]:
        del _FANS[id(base)]                    # every parked view has been handed out
    return v, ent[1]


# ---------------------------------------------------------------------------------------------------------------------------
# Deferred trunk backward.  The reference's Trainer.train_step (train.py:486, 499, 510) back-propagates every loss with its own
# ``backward()`` call -- ``mono_total_loss.backward(retain_graph=True)``, then ``segmentation_total_loss.backward()`` -- so the
# shared encoder is walked once per loss (two ResNet-101 backward passes per step, +17 % of the step's multiply-adds).
# Gradients are linear in the loss: the encoder backward of the sum equals the sum of the encoder backwards.  ``defer_trunk``
# cuts the graph at the encoder's features: the decoders read detached leaves behind a gate; a backward pass that KEEPS its
# graph (``retain_graph=True``: by the reference's own convention more losses of this forward are to come) only parks the
# feature gradients at the gate, and the pass that RELEASES its graph runs the encoder backward once on the sum (a re-entrant
# ``torch.autograd.backward`` from inside the gate's node, i.e. still inside the caller's last ``backward()``: parameter
# hooks, the gradient all-reduce of ddp.GradAllReducer and ``clip_grad_norm_`` see complete gradients afterwards).
#
# How the gate knows which kind of pass it is in: the engine releases a node's saved variables right after running it unless
# the pass keeps the graph.  A sentinel node sits between the gate and the decoders -- it runs, and is released, before the
# gate's node runs; reading ``saved_tensors`` of a released node raises, and that is the test.
#
# A forward whose LAST backward keeps the graph (a monodepth-only step: the reference passes retain_graph=True there too)
# would park its encoder gradient for ever; deferral is therefore opt-in per model (``model.defer_trunk_backward``, set by
# ``trainer.train_step`` / the INTEGRATION.md shim only for configurations that end on a releasing call), parked gradients
# can be flushed by hand (``flush_deferred_trunks``), and a gradient that is still parked at the next forward of the same
# encoder or at ANY ``optimizer.step()`` raises instead of training on half a gradient.
class _TrunkState(object):
    __slots__ = ("roots", "pending", "eboxes", "sentinel", "owner", "passes", "group", "serial", "__weakref__")

    def has_pending(self):
        return any(p is not None for p in self.pending)

    def flush(self):
        roots, grads = [], []
        for i, (v, p) in enumerate(zip(self.roots, self.pending)):
            if p is None or not v.requires_grad:
                continue
            roots.append(v)
            grads.append(p)
            box = self.eboxes[i]
            if box is not None and box.get("publish") and box.get("g") is None and p.is_contiguous():
                # the next stage's first convolutions (and the stem's max-pooling) add their data-gradients onto the parked
                # tensor inside their kernels, like they do onto a decoder's skip gradient without the gate
                box["g"], box["fused"] = p, True
        self.pending = [None] * len(self.pending)
        self.roots = None                    # the graph behind the gate is consumed by this call
        if _TRUNKS.get(self.owner) is not None and _TRUNKS[self.owner]() is self:
            del _TRUNKS[self.owner]
        if roots:
            TrunkGateFn.trunk_backwards += 1
            torch.autograd.backward(roots, grads)


_TRUNKS = {}                 # owner id -> weak reference to the state of that module's latest deferred forward (the graph owns it)
_TRUNK_HOOK = [None]
_TRUNK_SERIAL = [0]          # creation order of the gates
_TRUNK_GROUP = [0]           # one group per model forward: the encoder's gate opens it, gates further up the model join it
_TASK_GROUPS = {}            # autograd graph-task id -> groups whose gates fired in that (releasing) task


def _trunk_states():
    out = []
    for k, r in list(_TRUNKS.items()):
        st = r()
        if st is None:
            del _TRUNKS[k]
        else:
            out.append(st)
    return sorted(out, key=lambda st: st.serial)


def _trunk_guard(*_a, **_k):
    for st in _trunk_states():
        if st.has_pending():
            raise RuntimeError(
                "deferred trunk backward: %d backward pass(es) of the last forward kept their graph (retain_graph=True) and none "
                "released it, so the shared encoder has not been back-propagated -- its gradients are incomplete.  End the "
                "forward's losses on a plain backward(), call functional.flush_deferred_trunks() before the gradients are "
                "read, or switch model.defer_trunk_backward off for this configuration." % st.passes)


def flush_deferred_trunks(groups=None):
    """run the parked backward of every gate that still holds gradients (see ``defer_trunk``), the gate created LAST first: a
    gate further up the model (PAD's, between the two halves of its decoders) back-propagates into the encoder's gate, never
    the other way round"""
    for st in reversed(_trunk_states()):
        if st.has_pending() and (groups is None or st.group in groups):
            st.flush()


def pending_deferred_trunks():
    return sum(1 for st in _trunk_states() if st.has_pending())


def _flush_after_task(st):
    """called from a gate's node in a pass that releases its graph: the gates of this forward are flushed ONCE, after the whole
    graph task has run (an engine callback), so that each of them holds every contribution of the pass -- the encoder's gate is
    reached both by the decoders' skip connections in the outer task and, later, by the nested backward of a gate above it"""
    tid = torch._C._current_graph_task_id()
    groups = _TASK_GROUPS.get(tid)
    if groups is not None:
        groups.add(st.group)
        return
    _TASK_GROUPS[tid] = {st.group}

    def run():
        flush_deferred_trunks(_TASK_GROUPS.pop(tid, None))
    torch.autograd.Variable._execution_engine.queue_callback(run)


class TrunkSentinelFn(Function):
    """identity on the consumers' side of the gate; its only job is to be released by the engine before the gate's node runs"""

    @staticmethod
    def forward(ctx, st, *xs):
        ctx.set_materialize_grads(False)
        st.sentinel = ctx
        return tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *grads):
        return (None,) + grads


class TrunkGateFn(Function):
    trunk_backwards = 0      # diagnostics / tests: parked backward passes started by a gate
    parked_passes = 0        # ... and passes that only parked their gradients

    @staticmethod
    def forward(ctx, st, *leaves):
        ctx.set_materialize_grads(False)
        ctx.st = st
        return tuple(x.view_as(x) for x in leaves)

    @staticmethod
    def backward(ctx, *grads):
        st = ctx.st
        if st.roots is None:
            raise RuntimeError("deferred trunk backward: the graph behind this gate has already been back-propagated (a pass "
                               "that released its graph came before this one)")
        for i, g in enumerate(grads):
            if g is None:
                continue
            g, p = _c(g), st.pending[i]
            st.pending[i] = g if p is None else H.axpby(1.0, p, 1.0, g)
        st.passes += 1
        try:
            st.sentinel.saved_tensors        # raises once the engine has released the sentinel: this pass frees its graph
            final = False
        except RuntimeError:
            final = True
        if final:
            _flush_after_task(st)
        else:
            TrunkGateFn.parked_passes += 1
        return (None,) * (1 + len(grads))


def defer_gate(tensors, owner, eboxes=None, new_group=False):
    """tensors (graph attached) -> the tensors their consumers read instead: detached leaves behind a TrunkGateFn whose backward
    parks the incoming gradients until a pass releases its graph (see the comment above).  owner: an id of the calling module
    (one live gate per owner); new_group: this gate opens a model forward (the encoder's), later gates join its group."""
    if not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors)):
        return list(tensors)
    old = _TRUNKS.get(owner)
    old = old() if old is not None else None
    if old is not None and old.has_pending():
        # the previous forward of this module never got its releasing backward: say so ONCE (its graph is being replaced anyway;
        # a training loop that catches the error and goes on must not meet it again at every forward)
        passes = old.passes
        for st in _trunk_states():           # every gate of that forward (the PAD decoder's sits above the encoder's)
            if st.group == old.group:
                st.pending, st.roots = [None] * len(st.pending), None
                _TRUNKS.pop(st.owner, None)
        raise RuntimeError(
            "deferred trunk backward: %d backward pass(es) of the previous forward kept their graph (retain_graph=True) and none "
            "released it, so the graph behind the gate was never back-propagated -- those gradients are lost.  End the forward's "
            "losses on a plain backward(), call functional.flush_deferred_trunks() after the last one, or switch "
            "model.defer_trunk_backward off for this configuration." % passes)
    if _TRUNK_HOOK[0] is None:
        from torch.optim.optimizer import register_optimizer_step_pre_hook
        _TRUNK_HOOK[0] = register_optimizer_step_pre_hook(_trunk_guard)
    if new_group:
        _TRUNK_GROUP[0] += 1
    _TRUNK_SERIAL[0] += 1
    st = _TrunkState()
    st.owner, st.passes, st.sentinel, st.group, st.serial = owner, 0, None, _TRUNK_GROUP[0], _TRUNK_SERIAL[0]
    st.roots = list(tensors)
    st.eboxes = list(eboxes) if eboxes is not None else [None] * len(tensors)
    st.pending = [None] * len(tensors)
    _TRUNKS[owner] = weakref.ref(st)
    leaves = [v.detach().requires_grad_(v.requires_grad) for v in st.roots]
    return list(TrunkSentinelFn.apply(st, *TrunkGateFn.apply(st, *leaves)))


def defer_trunk(feats, owner, n_consumers=0):
    """feats: the encoder's NHWC features (graph attached).  -> the tensors the decoders read instead (``defer_gate``).
    n_consumers: decoders that fetch a view of every feature with ``take_fan_view`` (the gradient collector moves to the
    decoders' side of the gate; on the encoder's side the gate is the one outside consumer)."""
    if not (torch.is_grad_enabled() and any(f.requires_grad for f in feats)):
        return feats
    roots, eboxes = [], []
    for f in feats:
        v, box = take_fan_view(f)            # the view the encoder parked for the gate (its gradient collector's box)
        roots.append(v)
        eboxes.append(box)
    outs = defer_gate(roots, owner, eboxes, new_group=True)
    for o, box in zip(outs, eboxes):
        if box is not None and n_consumers > 1:
            fan_feature(o, n_consumers, 0, owner=owner)      # parks the decoders' views; they share one in-kernel accumulation
    return outs


class MaxPoolFn(Function):
    """box: the shared box of the pooled tensor's gradient collector (Fn.fan_feature), if it has one: the pooling gradient is
    added onto the gradient another consumer already left there"""

    @staticmethod
    def forward(ctx, x, box=None):
        y, idx = H.maxpool_forward(_c(x))
        ctx.shape = tuple(x.shape)
        ctx.box = box
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        box = ctx.box
        acc = box.get("g") if (box is not None and box.get("fused")) else None
        dx = H.maxpool_backward(_c(dy), idx, ctx.shape, accumulate_into=acc)
        if box is not None and acc is None and box.get("publish") and box.get("g") is None:
            box["g"], box["fused"] = dx, True
        return dx, None


class ResizeFn(Function):
    @staticmethod
    def forward(ctx, x, out_hw, align_corners):
        ctx.in_hw, ctx.ac = (x.shape[1], x.shape[2]), align_corners
        return H.resize_bilinear(x, out_hw, align_corners)

    @staticmethod
    def backward(ctx, dy):
        return H.resize_bilinear_backward(_c(dy), ctx.in_hw, ctx.ac), None, None


def resize_bilinear(x_nhwc, out_hw, align_corners=False):
    out_hw = (int(out_hw[0]), int(out_hw[1]))
    if not align_corners and out_hw == (x_nhwc.shape[1], x_nhwc.shape[2]):
        return x_nhwc          # F.interpolate to the same size with align_corners=False is the identity
    return ResizeFn.apply(x_nhwc, out_hw, align_corners)


class GlobalAvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return H.global_avgpool(x)

    @staticmethod
    def backward(ctx, dy):
        return H.global_avgpool_backward(_c(dy), ctx.shape)


class GateFn(Function):
    """SelfAttention gate: f * sigmoid(a)"""

    @staticmethod
    def forward(ctx, f, a):
        f, a = _c(f), _c(a)
        ctx.save_for_backward(f, a)
        return H.gate_forward(f, a)

    @staticmethod
    def backward(ctx, dy):
        f, a = ctx.saved_tensors
        return H.gate_backward(_c(dy), f, a)


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return H.axpby(1.0, _c(a), 1.0, _c(b))

    @staticmethod
    def backward(ctx, g):
        return g, g


class ConcatFn(Function):
    """channel concat of NHWC tensors (ASPP branches, models/model_parts.py:31)"""

    @staticmethod
    def forward(ctx, *xs):
        B, Hh, W, _ = xs[0].shape
        ctx.sizes = [x.shape[3] for x in xs]
        out = torch.empty((B, Hh, W, sum(ctx.sizes)), dtype=torch.float32, device=xs[0].device)
        o = 0
        for x, c in zip(xs, ctx.sizes):
            H.copy_channels(x, out[..., o:o + c])
            o += c
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        outs, o = [], 0
        for c in ctx.sizes:
            outs.append(g[..., o:o + c])   # pitch-aware consumers read the slice in place
            o += c
        return tuple(outs)


class ChannelDropFn(Function):
    """nn.Dropout2d on an NHWC tensor: y = x * scale[b, c], scale in {0, 1 / (1 - p)} (models/monodepth_layers.py:117-119)"""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.save_for_backward(scale)
        return H.scale_channels(_c(x), scale)

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        return H.scale_channels(_c(dy), scale), None


class DropoutFn(Function):
    """nn.Dropout on an NHWC tensor: the mask is a function of (seed, element index), so backward regenerates it"""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        if p >= 1.0:                         # nn.Dropout(1.0) is legal in the reference: everything is dropped
            return torch.zeros_like(x, memory_format=torch.contiguous_format)
        return H.dropout(_c(x), p, seed)

    @staticmethod
    def backward(ctx, dy):
        if ctx.p >= 1.0:
            return torch.zeros_like(dy, memory_format=torch.contiguous_format), None, None
        return H.dropout(_c(dy), ctx.p, ctx.seed), None, None


class PoseMatrixFn(Function):
    """transformation_from_parameters(axisangle[:, 0], translation[:, 0], invert) on the [B,F,1,3] network outputs"""

    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        axisangle, translation = axisangle.contiguous(), translation.contiguous()
        ctx.invert = invert
        ctx.save_for_backward(axisangle, translation)
        return H.pose_matrix(axisangle, translation, invert)

    @staticmethod
    def backward(ctx, dM):
        aa, tr = ctx.saved_tensors
        daa, dtr = H.pose_matrix_backward(aa, tr, dM, ctx.invert)
        return daa, dtr, None


class ScaleSliceFn(Function):
    """out = alpha * x  (PoseDecoder's 0.01 factor, pose_decoder.py:51)"""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return H.axpby(alpha, x)

    @staticmethod
    def backward(ctx, g):
        return H.axpby(ctx.alpha, g), None
