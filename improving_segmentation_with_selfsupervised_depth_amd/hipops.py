"""Thin tensor-level wrappers over the C ABI (include/segsde_hip.h).

PyTorch is used here for device memory (``torch.empty``), the current HIP stream and dtype/shape checks only;
every arithmetic step is a call into libsegsde_hip.so.  Activations are NHWC fp32 tensors ``[B, H, W, C]`` whose
last dimension is contiguous (a channel slice of a wider buffer is fine: its pixel pitch ``ld`` is passed on).
"""
import ctypes
import os
import threading

import torch

from . import _lib
from ._lib import ConvDesc, check

ACT = {"none": 0, "relu": 1, "elu": 2, "sigmoid": 3}


class _Slot(threading.local):
    """A one-element, per-thread side channel (``slot[0]``): conv_forward / conv_dgrad leave a route's by-product here (the
    transformed input a Winograd forward kept, whether an epilogue fusion was applied) and the calling autograd Function
    picks it up right after the call.  Per thread because autograd runs backward nodes on its own engine threads (one per
    device): two devices or threads in one process must not see each other's values."""

    def __init__(self, default=None):
        self.default = default
        self.v = default

    def __getattr__(self, name):          # first touch from another thread: threading.local re-runs __init__ only with args
        if name == "v":
            return self.__dict__.setdefault("v", self.__dict__.get("default"))
        raise AttributeError(name)

    def __getitem__(self, i):
        return self.v

    def __setitem__(self, i, value):
        self.v = value

# Optional live kernel timing (bench.py): when set to a list, every implicit-GEMM launch -- and every launch (group) of the
# HBM-bound kernel families, kind "hbm_*" -- is bracketed by HIP events recorded on the stream the kernel is launched on;
# entries are (kind, algorithmic flops (conv_*) or algorithmic bytes (hbm_*: the operands once each), start, end, tag).
PROFILE = None


# Sampled timing: a HIP event pair around a launch drains the stream on both sides (the next kernel's launch latency is
# no longer hidden behind the previous kernel: ~5 us per event, ~2 000 events = 9 ms of a 320 ms cfg3 step when EVERY launch
# is bracketed).  With PROFILE_PERIOD = P the launch with sequence number q of step i is bracketed iff (q + i) % P == 0:
# every launch site is timed in one step out of P, every step carries 1/P of the events; the launches that are not
# bracketed are still recorded (events = None) so that the consumer knows the population.  bench.py sets these.
PROFILE_PERIOD = 1
_profile_state = [0, 0]      # [sequence number inside the step, step index]


def profile_step(i):
    _profile_state[0], _profile_state[1] = 0, int(i)


def _timed(kind, flops, like, launch, tag="", executed=None):
    """executed: the multiply-add work the launch really issues when that is less than the algorithmic figure (the
    upsample-folded convolutions); defaults to ``flops``"""
    if PROFILE is None or not like.is_cuda:
        return launch()
    q = _profile_state[0]
    _profile_state[0] = q + 1
    if PROFILE_PERIOD > 1 and (q + _profile_state[1]) % PROFILE_PERIOD:
        PROFILE.append((kind, flops, None, None, tag, flops if executed is None else executed))
        return launch()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = launch()
    e.record()
    PROFILE.append((kind, flops, s, e, tag, flops if executed is None else executed))
    return r


# Upsample-folded route of the decoder's Conv3x3 on [upsample(x) | skip] (csrc/conv_igemm.hip, "Upsample-folded 3x3
# convolutions"): SEGSDE_UPFOLD=0 keeps the plain 9-tap route (A/B measurements, debugging)
UPFOLD = os.environ.get("SEGSDE_UPFOLD", "1") != "0"
UPFOLD_TAKEN = {"fwd": 0, "dgrad": 0, "wgrad": 0}    # diagnostics / tests: launches that took the folded route


UPFOLD_MIN_SAVED_MACS = float(os.environ.get("SEGSDE_UPFOLD_MIN_MACS", "1e10"))


def upfold_ok(g, pixels=None):
    """pixels = B * H * W of the output (None: shape test only).  The folded route issues ~10 more launches per direction than
    the plain one (pack, class launches, border launches): it is taken when the multiply-adds it saves -- 5 of 9 taps on the
    C0 upsampled channels -- outweigh ~100 us of launches at ~130 TFLOP/s (cfg1's 2 x 256 x 512 layers stay on the plain route:
    measured 28.9 vs 31.7 ms/step)."""
    if not (UPFOLD and g.up0 and g.reflect and g.k == 3 and g.stride == 1 and g.dil == 1 and g.pad == 1 and g.C0 % 32 == 0
            and g.C1 % 32 == 0 and g.Cout % 32 == 0):
        return False
    return pixels is None or 5.0 * pixels * g.C0 * g.Cout >= UPFOLD_MIN_SAVED_MACS


def upfold_pack(w_oihw, C0):
    """(wfold [4][Cout][2][2][C0], wdfold [C0][4][4][Cout]) of an OIHW 3x3 weight whose first C0 input channels see a
    nearest-upsampled source: the pre-summed 2x2 taps of the four output parity classes / the 4x4 stride-2 kernel of the
    low-resolution data-gradient"""
    w = _f32(w_oihw.detach()).contiguous()
    Cout, Ctot, KH, KW = w.shape
    assert KH == 3 and KW == 3 and 0 < C0 <= Ctot
    wf = torch.empty((4, Cout, 2, 2, C0), dtype=torch.float32, device=w.device)
    wd = torch.empty((C0, 4, 4, Cout), dtype=torch.float32, device=w.device)
    check(_lib.lib().segsde_upfold_pack(_p(w), Cout, C0, Ctot, _p(wf), _p(wd), _stream(w)), "upfold_pack")
    return wf, wd


# Winograd F(2x2,3x3) route of the stride-1 3x3 convolutions with many channels (csrc/winograd.hip, segsde_conv2d_winograd):
# SEGSDE_WINOGRAD=0 keeps the direct implicit GEMM (A/B measurements; the exactness tests of the direct kernels use it)
WINOGRAD = os.environ.get("SEGSDE_WINOGRAD", "1") != "0"
WINOGRAD_MIN_CH = int(os.environ.get("SEGSDE_WINOGRAD_MIN_CH", "256"))
WINOGRAD_MIN_MACS = float(os.environ.get("SEGSDE_WINOGRAD_MIN_MACS", "1e8"))
WINOGRAD_TAKEN = {"fwd": 0, "dgrad": 0, "wgrad": 0}


def winograd_ok(g, B=None, H=None, W=None, dgrad=False):
    """Shape gate of the Winograd route.  Measured (profiles/probe_r04_winograd_gate.log, probe_r04_winograd_route.log): with the
    transforms as separate passes the route wins from 256 channels on (1.4x at 256 channels, 1.8x at 512); at 128 channels the
    transforms' traffic eats the 2.25x fewer multiply-adds.  The multiply-add floor keeps tiny launches (and the small-shape
    golden tests of the direct kernels) on the direct route.  Forward: one or two sources, zero / mirrored padding (mirrored:
    dilation 1), any dilation that divides the map into even sub-lattices.  Data-gradient: one source, zero padding."""
    if g.compute:
        return False
    cin, cout = (g.Cout, g.C0) if dgrad else (g.Cin, g.Cout)      # the data-gradient is the convolution Cout -> C0
    if not (WINOGRAD and g.k == 3 and g.stride == 1 and g.pad == g.dil and not g.up0 and g.cin_alg is None and g.C0 % 4 == 0
            and cin % 32 == 0 and cout % 64 == 0 and min(g.Cin, g.Cout) >= WINOGRAD_MIN_CH and not (g.reflect and g.dil != 1)):
        return False
    if dgrad and (g.C1 or g.reflect):
        return False
    if B is None:
        return True
    if H % (2 * g.dil) or W % (2 * g.dil) or H < 4 * g.dil or W < 4 * g.dil:
        return False
    return 9.0 * B * H * W * g.Cin * g.Cout >= WINOGRAD_MIN_MACS


def winograd_static_ok(conv):
    """could this nn.Conv2d ever take the route (weight_pack_scope packs those up front)"""
    return (WINOGRAD and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == conv.dilation
            and conv.in_channels % 32 == 0 and conv.out_channels % 64 == 0
            and min(conv.in_channels, conv.out_channels) >= WINOGRAD_MIN_CH)


def winograd_pack(w_oihw, kn=False):
    """(forward pack [16][Cout][Cin], data-gradient pack [16][Cin][Cout]) = G g G^T of an OIHW 3x3 weight; kn: the [16][K][N]
    packs of the one-kernel route instead ([16][Cin][Cout], [16][Cout][Cin])"""
    if kn:
        return winograd_fused_pack(w_oihw, False), winograd_fused_pack(w_oihw, True)
    w = _f32(w_oihw.detach()).contiguous()
    O, I, KH, KW = w.shape
    assert KH == 3 and KW == 3
    uf = torch.empty((16, O, I), dtype=torch.float32, device=w.device)
    ud = torch.empty((16, I, O), dtype=torch.float32, device=w.device)
    check(_lib.lib().segsde_winograd_pack(_p(w), O, I, _p(uf), _p(ud), _stream(w)), "winograd_pack")
    return uf, ud


_WINO_MULTI = {}


def winograd_packs_multi(weights, kn=False):
    """[(forward pack, data-gradient pack)] of many 3x3 weights in ONE launch; the packs are views into a flat buffer that the
    next call with the same weights rewrites (what a training step needs, like pack_weights_multi)"""
    ws = [_f32(w.detach()) for w in weights]
    assert ws and all(w.is_contiguous() and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3) for w in ws)
    dev = ws[0].device
    key = (dev, bool(kn), tuple((w.data_ptr(), tuple(w.shape)) for w in ws))
    hit = _WINO_MULTI.get(key)
    if hit is None:
        if len(_WINO_MULTI) > 8:
            _WINO_MULTI.clear()
        total = sum(16 * w.shape[0] * w.shape[1] for w in ws)
        flat = torch.empty(2 * total, dtype=torch.float32, device=dev)
        jobs = (_lib.WinoJob * len(ws))()
        views, off, blk = [], 0, 0
        for i, w in enumerate(ws):
            O, I = w.shape[0], w.shape[1]
            n = 16 * O * I
            if kn:
                uf, ud = _kn(flat[off:off + n].view(16, I, O)), _kn(flat[off + n:off + 2 * n].view(16, O, I))
            else:
                uf, ud = flat[off:off + n].view(16, O, I), flat[off + n:off + 2 * n].view(16, I, O)
            off += 2 * n
            jobs[i] = _lib.WinoJob(w.data_ptr(), uf.data_ptr(), ud.data_ptr(), O, I, blk, (3 if WINO_FUSED_UBLK else 1) if kn else 0)
            blk += 2 * ((O * I + 255) // 256)
            views.append((uf, ud))
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        hit = _WINO_MULTI[key] = (raw, flat, views, len(ws), blk)
    raw, flat, views, n, blk = hit
    check(_lib.lib().segsde_winograd_pack_multi(_p(raw), n, blk, _stream(flat)), "winograd_pack_multi")
    return views


WINOGRAD_KEEP_V = os.environ.get("SEGSDE_WINOGRAD_KEEP_V", "1") != "0"   # a training forward keeps its transformed input for the weight gradient
WINO_V = _Slot(None)     # the transformed input of the last _winograd(..., keep_v=True) call (ConvFn picks it up)


def _winograd(kind, g, x0, x1, upack, Cout, want_stats, flops, tag, bias=None, act="none", reflect=False, keep_v=False):
    """one Winograd convolution [x0 | x1] [B,H,W,C] -> [B,H,W,Cout]; returns (y, partials or None), or None when the kernel
    declines"""
    B, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    L = _lib.lib()
    d = ConvDesc(B=B, H=H, W=W, C0=C0, C1=C1, ld0=nhwc_ld(x0), ld1=nhwc_ld(x1) if x1 is not None else 0, up0=0, Ho=H, Wo=W,
                 Cout=Cout, ldy=Cout, ldy2=0, nsplit=0, KH=3, KW=3, stride=1, dil=g.dil, pad=g.dil,
                 pad_mode=PAD_REFLECT if reflect else PAD_ZERO, in_div=1, act=ACT[act], sum2x2=0)
    nbytes = L.segsde_conv2d_winograd_workspace(ctypes.byref(d))
    if not nbytes:
        return None
    y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x0.device)
    part = None
    if want_stats:
        part = torch.empty((int(L.segsde_conv2d_winograd_stats_rows(ctypes.byref(d))), 2, Cout), dtype=torch.float64, device=x0.device)
    ws = _ws(nbytes, x0)
    Tp = (B * (H // 2) * (W // 2) + 127) // 128 * 128          # rows per position plane: whole 128-row GEMM tiles
    vk = torch.empty((16, Tp, C0 + C1), dtype=torch.float32, device=x0.device) if keep_v else None
    rc = _timed(kind, flops, x0, lambda: L.segsde_conv2d_winograd(ctypes.byref(d), _p(_f32(x0)), _p(x1), _p(upack), _p(bias), _p(y),
                                                                   _p(part), _p(vk), _p(ws), nbytes, _stream(x0)), tag + " wino",
                executed=flops * 16.0 / 36.0)
    if rc == -4:
        return None
    check(rc, "conv2d_winograd")
    WINO_V[0] = vk
    WINOGRAD_TAKEN["fwd" if kind == "conv_fwd" else "dgrad"] += 1
    return y, part


# ---- Winograd with the transforms inside the kernel (csrc/winograd_fused.hip): 3x3 / stride 1 / padding 1, one source,
# 64 .. 256 channels.  Measured (profiles/probe_r04_winograd_fused.log, B = 16): 1.6-1.7x the direct kernel forward, 1.55x
# data-gradient from 64 channels on, 1.3x the grouped-GEMM route of winograd.hip at 256 channels (168 against 217 us), equal to
# it at 512 (where the grouped route also hands its transformed input to the weight gradient).  SEGSDE_WINO_FUSED=0: off.
WINO_FUSED = os.environ.get("SEGSDE_WINO_FUSED", "1") != "0"
WINO_FUSED_MAX_CH = int(os.environ.get("SEGSDE_WINO_FUSED_MAX_CH", "256"))
# round 5: the decoder's Conv3x3 on [upsample(x) | skip] (mirrored padding) through the same kernel -- upsampling and concat are
# index arithmetic of its patch loader -- instead of the upsample-folded direct route, when the layer's folded multiply-add share
# (4/9 on the upsampled channels, 9/9 on the skip) is above WINO_FUSED2_MIN_FOLD: the route runs all channels at 16/36 but at
# ~0.75 of the direct kernel's multiply-add rate (profiles/probe_r05_*.log); SEGSDE_WINO_FUSED2=0: off
WINO_FUSED2 = os.environ.get("SEGSDE_WINO_FUSED2", "1") != "0"
# the kernels' zero-position skipping on nearest-upsampled sources (SEGSDE_WINO_UP_SKIP=0 switches it off in both csrc launchers;
# here only the reported `executed` multiply-add counts depend on it)
WINO_UP_SKIP = os.environ.get("SEGSDE_WINO_UP_SKIP", "1") != "0"
WINO_FUSED2_MAX_CIN = int(os.environ.get("SEGSDE_WINO_FUSED2_MAX_CIN", "768"))
WINO_FUSED2_MIN_FOLD = float(os.environ.get("SEGSDE_WINO_FUSED2_MIN_FOLD", "0.6"))
# round 5: data-gradients of the mirrored-padding Conv3x3 (zero-padded one-kernel launch + border launches) and data-gradients
# with the activation-derivative epilogue on the one-kernel route; SEGSDE_WINO_FUSED_DGRAD_EXT=0 keeps them on the direct kernel
WINO_FUSED_DGRAD_EXT = os.environ.get("SEGSDE_WINO_FUSED_DGRAD_EXT", "1") != "0"
# the mirrored convolution's data-gradient takes the route from this many pixels on (measured, profiles/probe_r05_winograd_dgrad_*.log:
# 1.35-1.6x from 16 x 64 x 128 pixels; at 16 x 32 x 64 the two border launches -- 64 workgroups with long reductions -- cost more
# than the Winograd launch saves: 395 vs 338 us)
WINO_FUSED_REFLECT_DGRAD_MIN_PIX = int(os.environ.get("SEGSDE_WINO_FUSED_REFLECT_DGRAD_MIN_PIX", str(1 << 17)))
# the skip-source gradient of the decoder's two-source layers (dy -> the C1 encoder channels, mirrored padding, accumulated onto the
# feature's gradient collector) on the one-kernel route -- a column slice of the flipped pack -- while the upsampled source's
# low-resolution gradient stays on the folded route's 4x4 / stride-2 launch.  SEGSDE_WINO_FUSED_DGRAD2=0: off
WINO_FUSED_DGRAD2 = os.environ.get("SEGSDE_WINO_FUSED_DGRAD2", "1") != "0"
BORDERS2 = os.environ.get("SEGSDE_BORDERS2", "1") != "0"     # 0: the mirrored-padding terms as implicit-GEMM border launches (first version)
# layout of the route's transformed weights: blocked (the eight B operands of a lane and step contiguous: two 16-byte requests
# instead of eight 4-byte ones) unless SEGSDE_WINO_FUSED_UBLK=0 -- the library reads the same variable (segsde_wino_ublk)
WINO_FUSED_UBLK = os.environ.get("SEGSDE_WINO_FUSED_UBLK", "1") != "0"
WINO_FUSED_TAKEN = {"fwd": 0, "dgrad": 0, "fwd2": 0, "dgrad_refl": 0, "dgrad_actgrad": 0, "dgrad2": 0, "wgrad": 0}


_FUSED_SHAPE_OK = {}


def _fused_shape_ok(B, H, W, cin, cout):
    """segsde_winograd_fused_ok, remembered per shape (the routers ask three times per convolution call; on the launch-bound
    small workloads a foreign-function call each time shows)"""
    key = (B, H, W, cin, cout)
    r = _FUSED_SHAPE_OK.get(key)
    if r is None:
        r = _FUSED_SHAPE_OK[key] = bool(_lib.lib().segsde_winograd_fused_ok(B, H, W, cin, cout))
    return r


def winograd_fused_ok(g, B=None, H=None, W=None, dgrad=False):
    """the shapes csrc/winograd_fused.hip takes: 3x3 / stride 1 / padding 1 / dilation 1, channel counts multiples of 64.
    One source at output resolution: up to WINO_FUSED_MAX_CH channels, zero or mirrored padding, forward and data-gradient
    (mirrored: + segsde_reflect_adjoint_borders).  Forward on [upsample(x0) | x1] (g.up0): C0 % 64 == 0, up to
    WINO_FUSED2_MAX_CIN input channels, when the folded route's multiply-add share is above WINO_FUSED2_MIN_FOLD."""
    if g.compute:
        return False
    cin, cout = (g.Cout, g.C0) if dgrad else (g.Cin, g.Cout)
    if not (WINOGRAD and WINO_FUSED and g.k == 3 and g.stride == 1 and g.dil == 1 and g.pad == 1 and g.cin_alg is None
            and cin % 64 == 0 and cout % 64 == 0):
        return False
    if g.up0 or g.C1:
        # two sources without upsampling (the bottleneck-resolution layer) stay on the grouped route, which takes them
        if dgrad or not (WINO_FUSED2 and g.up0 and g.C0 % 64 == 0 and cin <= WINO_FUSED2_MAX_CIN and cout <= WINO_FUSED_MAX_CH
                         and _fold_frac(g) >= WINO_FUSED2_MIN_FOLD):
            return False
    else:
        if max(cin, cout) > WINO_FUSED_MAX_CH or (dgrad and g.reflect and not WINO_FUSED_DGRAD_EXT):
            return False
    if B is None:
        return True
    return _fused_shape_ok(B, H, W, cin, cout) and 9.0 * B * H * W * cin * cout >= WINOGRAD_MIN_MACS


def winograd_fused_dgrad2_ok(g, B=None, H=None, W=None):
    """the skip-source data-gradient of a [upsample(x0) | x1] -> Cout mirrored 3x3 convolution as the one-kernel Winograd
    convolution Cout -> C1 (+ border kernel)"""
    if g.compute:
        return False
    if not (WINOGRAD and WINO_FUSED and WINO_FUSED_DGRAD_EXT and WINO_FUSED_DGRAD2 and BORDERS2 and g.k == 3 and g.stride == 1 and g.dil == 1
            and g.pad == 1 and g.reflect and g.up0 and g.C1 and g.cin_alg is None and g.C1 % 64 == 0 and g.Cout % 64 == 0
            and g.C0 % 64 == 0       # (the slice starts on a 64-filter block of the blocked pack layout)
            and g.Cout <= WINO_FUSED_MAX_CH and g.Cin <= WINO_FUSED2_MAX_CIN):
        return False
    if B is None:
        return True
    return (B * H * W >= WINO_FUSED_REFLECT_DGRAD_MIN_PIX and _fused_shape_ok(B, H, W, g.Cout, g.C1)
            and 9.0 * B * H * W * g.C1 * g.Cout >= WINOGRAD_MIN_MACS)


def winograd_fused_static_ok(conv):
    """could this nn.Conv2d ever take the one-kernel route (weight_pack_scope packs those in the [16][K][N] layout up front)"""
    if not (WINOGRAD and WINO_FUSED and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0):
        return False
    if max(conv.in_channels, conv.out_channels) <= WINO_FUSED_MAX_CH:
        return True
    # the decoder's two-source layers ([upsample(x) | skip], mirrored padding)
    return bool(WINO_FUSED2 and getattr(conv, "reflect", False) and conv.out_channels <= WINO_FUSED_MAX_CH
                and conv.in_channels <= WINO_FUSED2_MAX_CIN)


class _KnPack(torch.Tensor):
    """a transformed weight pack of the one-kernel route: 16 K N floats, logical shape [16][K][N], stored in the blocked order the
    kernel reads (WINO_FUSED_UBLK; the sixteen [K][N] planes with SEGSDE_WINO_FUSED_UBLK=0) -- a plain tensor with a type the
    dispatch can see; only the library's pack and convolution kernels interpret its contents"""


def _kn(t):
    return t.as_subclass(_KnPack)


def winograd_fused_pack(w_oihw, flip):
    """OIHW 3x3 -> U (logical [16][K][N], see _KnPack for the storage order): flip False = forward pack (K = Cin, N = Cout), True =
    data-gradient pack"""
    O, I = w_oihw.shape[0], w_oihw.shape[1]
    w = _f32(w_oihw.detach()).contiguous()
    u = torch.empty((16, O, I) if flip else (16, I, O), dtype=torch.float32, device=w.device)
    check(_lib.lib().segsde_winograd_fused_pack(_p(w), O, I, 1 if flip else 0, _p(u), _stream(w)), "winograd_fused_pack")
    return _kn(u)


def winograd_fused(kind, x, u_kn, bias=None, act="none", want_stats=False, tag=None, reflect=False, accumulate_into=None, x1=None,
                   up0=False, actgrad=None, adjoint=None, cols=None):
    """[up2x?(x) | x1] [B,H,W,C] NHWC -> act(conv3x3(.) + bias) [B,H,W,N] with u_kn [16][C][N]; (y, statistics partials or None).
    accumulate_into: a dense [B,H,W,N] tensor the result is added onto (and which is returned).
    Data-gradient extras: actgrad = (saved activation output [B,H,W,N], kind): the result is multiplied by the activation's
    derivative; adjoint = (ConvGeom, wdpack): the convolution was mirror-padded -- the gradient of the padding cells is added by
    segsde_reflect_adjoint_borders after the zero-padded launch."""
    B, H0, W0, C0 = x.shape
    H, W = (2 * H0, 2 * W0) if up0 else (H0, W0)
    C1 = 0 if x1 is None else x1.shape[3]
    C = C0 + C1
    ldu = u_kn.shape[2]
    n_off, N = (0, ldu) if cols is None else cols       # cols = (first column, count): the gradient of ONE source of a two-source convolution
    assert u_kn.shape[1] == C and n_off + N <= ldu and (cols is None or (actgrad is not None or adjoint is not None or kind == "conv_dgrad"))
    L = _lib.lib()
    if accumulate_into is not None:
        y = accumulate_into
        if tuple(y.shape) != (B, H, W, N) or not y.is_contiguous() or bias is not None or act != "none" or want_stats:
            return None
    else:
        y = torch.empty((B, H, W, N), dtype=torch.float32, device=x.device)
    part = None
    if want_stats:
        part = torch.empty((int(L.segsde_winograd_fused_stats_rows(B, H, W)), 2, N), dtype=torch.float64, device=x.device)
    flops = 2.0 * B * H * W * C * N * 9
    tagf = (tag or "%d+%d->%d k3 s1 d1 %dx%d" % (C0, C1, N, H, W)) + " wino-fused"
    if x1 is not None or up0:
        assert accumulate_into is None and actgrad is None and adjoint is None
        rc = _timed(kind, flops, x, lambda: L.segsde_conv2d_winograd_fused2(
            _p(_f32(x)), nhwc_ld(x), C0, 1 if up0 else 0, _p(x1), nhwc_ld(x1) if x1 is not None else 0, C1, B, H, W, 1 if reflect else 0,
            _p(u_kn), N, _p(bias), ACT[act], _p(y), N, _p(part), _stream(x)), tagf,
            # issued multiply-adds: 16 / 36 of the algorithmic count, 9 / 36 on the channels of a nearest-upsampled source (seven of
            # the sixteen positions are identically zero there and skipped: csrc/winograd_fused.hip, UPSKIP)
            executed=flops * ((9.0 * C0 + 16.0 * C1) / (36.0 * (C0 + C1)) if (up0 and WINO_UP_SKIP) else 16.0 / 36.0))
        if rc == -4:
            return None
        check(rc, "conv2d_winograd_fused2")
        WINO_FUSED_TAKEN["fwd2"] += 1
        return y, part
    if actgrad is not None or adjoint is not None or cols is not None:
        assert bias is None and act == "none" and not want_stats and not reflect
        ag_y, ag_kind = (actgrad[0], ACT[actgrad[1]]) if actgrad is not None else (None, 0)
        ag_ld = nhwc_ld(ag_y) if ag_y is not None else 0
        acc = 1 if accumulate_into is not None else 0
        d = wdpack = wfpack = None
        if adjoint is not None:
            g, wdpack, wfpack = adjoint
            if wfpack is None or not BORDERS2:
                assert cols is None
                d = _borders_desc(B, H, W, C, N, nhwc_ld(x))

        def launch():
            # a column slice starts n_off floats into a row of the planar layout, n_off / 64 blocks of 256 floats into the blocked one
            u_off = 4 * ((n_off // 64) * 256 if WINO_FUSED_UBLK else n_off)
            rc = L.segsde_conv2d_winograd_fused_dgrad(_p(_f32(x)), nhwc_ld(x), B, H, W, C, ctypes.c_void_p(u_kn.data_ptr() + u_off), ldu,
                                                      N, _p(y), N, acc, _p(ag_y), ag_ld, ag_kind, _stream(x))
            if rc == 0 and adjoint is not None:
                if d is None:      # the border kernel of csrc/winograd_fused.hip (two launches), forward pack (its channel slice)
                    rc = L.segsde_reflect_adjoint_borders2(_p(x), nhwc_ld(x), ctypes.c_void_p(wfpack.data_ptr() + 4 * n_off), wfpack.shape[3],
                                                           _p(y), N, _p(ag_y), ag_ld, ag_kind, B, H, W, N, C, _stream(x))
                else:              # four border launches of the implicit-GEMM kernel + corner kernel, data-gradient pack
                    rc = L.segsde_reflect_adjoint_borders(ctypes.byref(d), _p(x), _p(wdpack), _p(y), _p(ag_y), ag_ld, ag_kind, _stream(x))
                if rc == -4:
                    raise RuntimeError("the mirrored-padding launches declined a shape the one-kernel data-gradient took: "
                                       "the zero-padded part is already in dx (hipops.conv_dgrad must gate this)")
            return rc
        rc = _timed(kind, flops, x, launch, tagf, executed=flops * 16.0 / 36.0)
        if rc == -4:
            return None
        check(rc, "conv2d_winograd_fused_dgrad")
        WINO_FUSED_TAKEN["dgrad"] += 1
        WINO_FUSED_TAKEN["dgrad_refl"] += adjoint is not None
        WINO_FUSED_TAKEN["dgrad_actgrad"] += actgrad is not None
        return y, None
    rc = _timed(kind, flops, x, lambda: L.segsde_conv2d_winograd_fused(_p(_f32(x)), nhwc_ld(x), B, H, W, C, 1 if reflect else 0, _p(u_kn), N, _p(bias),
                                                                       ACT[act], _p(y), N, 1 if accumulate_into is not None else 0, _p(part), _stream(x)),
                tagf, executed=flops * 16.0 / 36.0)
    if rc == -4:
        return None
    check(rc, "conv2d_winograd_fused")
    WINO_FUSED_TAKEN["fwd" if kind == "conv_fwd" else "dgrad"] += 1
    return y, part


def _borders_desc(B, H, W, Cout, Cin, lddy):
    return ConvDesc(B=B, H=H, W=W, C0=Cout, C1=0, ld0=lddy, ld1=0, up0=0, Ho=H, Wo=W, Cout=Cin, ldy=Cin, ldy2=0, nsplit=Cin, KH=3, KW=3,
                    stride=1, dil=1, pad=1, pad_mode=PAD_REFLECT_ADJOINT, in_div=1, act=0, sum2x2=0, accumulate=1)


def reflect_borders_ok(B, H, W, Cout, Cin, lddy, ag_ld=0):
    """does segsde_reflect_adjoint_borders add the mirrored-padding gradient of this [B,H,W,Cout] -> [B,H,W,Cin] data-gradient"""
    return bool(_lib.lib().segsde_reflect_adjoint_borders_ok(ctypes.byref(_borders_desc(B, H, W, Cout, Cin, lddy)), int(ag_ld)))


# round 5: the weight gradient on the one-kernel Winograd scheme (csrc/winograd_wgrad.hip): every 3x3 / stride-1 layer below 512
# channels whose forward leaves no transformed input behind, incl. the decoder's two-source / upsampled layers when the folded
# route's multiply-add share is above WINO_FUSED_WGRAD_MIN_FOLD.  SEGSDE_WINO_FUSED_WGRAD=0: off
WINO_FUSED_WGRAD = os.environ.get("SEGSDE_WINO_FUSED_WGRAD", "1") != "0"
WINO_FUSED_WGRAD_MAX_CH = int(os.environ.get("SEGSDE_WINO_FUSED_WGRAD_MAX_CH", "256"))
WINO_FUSED_WGRAD_MAX_CIN2 = int(os.environ.get("SEGSDE_WINO_FUSED_WGRAD_MAX_CIN2", "768"))
# (round 6: 0.6 -> 0.4 -- with the zero positions of an upsampled source skipped, the pure upsampled layer 64 -> 64 @512x1024, share
# 4/9, runs 1.23x the folded route: 2272 -> 1849 us, profiles/probe_r06_upskip_wgrad.log)
WINO_FUSED_WGRAD_MIN_FOLD = float(os.environ.get("SEGSDE_WINO_FUSED_WGRAD_MIN_FOLD", "0.4"))


def winograd_fused_wgrad_ok(g, B=None, H=None, W=None):
    if g.compute:            # half-precision operand mode: the direct / folded kernels (ConvGeom.compute)
        return False
    if not (WINOGRAD and WINO_FUSED and WINO_FUSED_WGRAD and g.k == 3 and g.stride == 1 and g.dil == 1 and g.pad == 1
            and g.cin_alg is None and g.C0 % 32 == 0 and g.C1 % 32 == 0 and g.Cout % 64 == 0 and g.Cout <= WINO_FUSED_WGRAD_MAX_CH):
        return False
    if g.up0 or g.C1:
        if not (g.up0 and g.Cin <= WINO_FUSED_WGRAD_MAX_CIN2 and _fold_frac(g) >= WINO_FUSED_WGRAD_MIN_FOLD):
            return False
    elif g.Cin > WINO_FUSED_WGRAD_MAX_CH:
        return False
    if B is None:
        return True
    return H % 2 == 0 and W % 2 == 0 and H >= 4 and W >= 4 and 9.0 * B * H * W * g.Cin * g.Cout >= WINOGRAD_MIN_MACS


def _fold_frac(g):
    return (4.0 * g.C0 + 9.0 * g.C1) / (9.0 * (g.C0 + g.C1))


def _live_tap_frac(g, H, W, wgrad=False):
    """Executed share of a dilated, zero-padded window (csrc/conv_igemm.hip, ConvP::tapskip): tap rows that only see padding
    for a whole 128-pixel tile (forward / data-gradient) or a whole output row (weight gradient, 128-column reduction tiles
    inside one tap row) are not multiplied.  Mirrors the kernels' tile-uniform rule for the reported 'executed' FLOPs; 1.0
    where the rule does not apply."""
    if g.reflect or g.dil <= 1 or g.k <= 1 or g.stride != 1 or g.up0 or os.environ.get("SEGSDE_TUNE", "").find("tskip=0") >= 0:
        return 1.0
    if wgrad:
        if g.C1 or g.C0 % 128 or W % 32:
            return 1.0
        live = sum(max(0, min(H - 1, H - 1 + g.pad - kh * g.dil) - max(0, g.pad - kh * g.dil) + 1) for kh in range(g.k))
        return live / float(g.k * H)
    if (H * W) % 128 or (128 % W and W % 128):
        return 1.0
    rows = max(1, 128 // W)
    live = total = 0
    for r0 in range(0, H, rows):
        r1 = r0 + rows - 1
        live += sum(1 for kh in range(g.k) if r1 + kh * g.dil - g.pad >= 0 and r0 + kh * g.dil - g.pad <= H - 1)
        total += g.k
    return live / float(total)


def _tag(g, H, W):
    return "%d+%d->%d k%d s%d d%d %dx%d%s%s" % (g.C0, g.C1, g.Cout, g.k, g.stride, g.dil, H, W, " up" if g.up0 else "",
                                                " refl" if g.reflect else "")
PAD_ZERO, PAD_REFLECT, PAD_REFLECT_ADJOINT = 0, 1, 2
COMPUTE_F16 = [False]      # see ConvGeom.compute


# the current stream's handle straight from the C binding (torch.cuda.current_stream(device).cuda_stream builds a Stream object
# per call: 5.4 us x ~600 calls per step showed as 1.4 ms of cfg1's 20 ms step in each of the two host threads)
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    if t.is_cuda:
        if _RAW_STREAM is not None:
            return ctypes.c_void_p(_RAW_STREAM(t.device.index))
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    if not _lib.HOST_POINTERS_OK:
        raise RuntimeError("segsde HIP kernels need tensors on a ROCm device (got %s); there is no CPU path" % t.device)
    return ctypes.c_void_p(0)


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _f32(t, name="tensor"):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t


def nhwc_ld(t):
    """pixel pitch of an NHWC tensor (validates that it is a dense-row channel slice)"""
    B, H, W, C = t.shape
    sb, sh, sw, sc = t.stride()
    if C > 1 and sc != 1:
        raise ValueError("NHWC tensor must be channel-contiguous")
    ld = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else C))
    if W > 1 and H > 1 and sh != W * ld or (B > 1 and H * W > 1 and sb != H * W * ld) or ld < C:
        raise ValueError("unsupported NHWC strides %s for shape %s" % (t.stride(), tuple(t.shape)))
    return int(ld)


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


# ----------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------
class ConvGeom:
    """Static description of one convolution of the reference model (shared by fwd / dgrad / wgrad)."""

    def __init__(self, C0, Cout, k, stride=1, dil=1, pad=0, reflect=False, C1=0, up0=False, cin_alg=None):
        """cin_alg: input channels of the convolution as the REFERENCE defines it when the launch carries zero pad channels
        (network stems: 3 -> 4, 6 -> 8): the algorithmic FLOP count of the bench uses it, the executed count the padded width"""
        self.C0, self.C1, self.Cout, self.k = int(C0), int(C1), int(Cout), int(k)
        self.cin_alg = int(cin_alg) if cin_alg else None
        self.stride, self.dil, self.pad, self.reflect, self.up0 = int(stride), int(dil), int(pad), bool(reflect), bool(up0)
        # segsde_conv_desc.compute of every launch of this convolution (forward, data-gradient, weight gradient): 1 under the
        # half-precision operand mode of `amp: True` (COMPUTE_F16, set by functional.fp32_region), where the Winograd routes --
        # fp32 kernels, and 16 / 36 of the multiply-adds at 1 / 16 of the fp16 matrix rate -- are not taken
        self.compute = 1 if COMPUTE_F16[0] else 0
        if reflect and (self.k != 3 or self.stride != 1 or self.dil != 1 or self.pad != 1):
            raise NotImplementedError("reflection padding is implemented for the reference's 3x3/s1/p1 Conv3x3 only")

    @property
    def Cin(self):
        return self.C0 + self.C1

    @property
    def CinAlg(self):
        return self.cin_alg if self.cin_alg else self.C0 + self.C1

    def out_hw(self, H, W):
        e = self.dil * (self.k - 1) + 1
        return (H + 2 * self.pad - e) // self.stride + 1, (W + 2 * self.pad - e) // self.stride + 1


def pack_weight(w_oihw, for_dgrad=False):
    """OIHW -> [O][KH][KW][I] (forward) or [I][KH][KW][O] flipped (data-gradient)."""
    w = _f32(w_oihw.detach()).contiguous()
    O, I, KH, KW = w.shape
    out = torch.empty((I, KH, KW, O) if for_dgrad else (O, KH, KW, I), dtype=torch.float32, device=w.device)
    check(_lib.lib().segsde_pack_weight(_p(w), _p(out), O, I, KH, KW, int(for_dgrad), _stream(w)), "pack_weight")
    return out


def pack_weight_both(w_oihw):
    """(forward pack, data-gradient pack) of one OIHW weight in a single launch"""
    w = _f32(w_oihw.detach()).contiguous()
    O, I, KH, KW = w.shape
    outf = torch.empty((O, KH, KW, I), dtype=torch.float32, device=w.device)
    outd = torch.empty((I, KH, KW, O), dtype=torch.float32, device=w.device)
    check(_lib.lib().segsde_pack_weight_both(_p(w), _p(outf), _p(outd), O, I, KH, KW, _stream(w)), "pack_weight_both")
    return outf, outd


_MULTI_PACK = {}     # (device, weight pointers / shapes) -> (job table on the device, flat pack buffer, [(fwd view, dgrad view)])


def pack_weights_multi(weights):
    """[(forward pack, data-gradient pack)] of many OIHW weights in ONE launch.  The packs are views into one flat buffer
    that is reused by the next call with the same weights (same storage, same shapes): valid until then -- what a training
    step needs (models/layers.weight_pack_scope), not something to keep."""
    ws = [_f32(w.detach()) for w in weights]
    assert ws and all(w.is_contiguous() and w.dim() == 4 for w in ws)
    dev = ws[0].device
    key = (dev, tuple((w.data_ptr(), tuple(w.shape)) for w in ws))
    hit = _MULTI_PACK.get(key)
    if hit is None:
        if len(_MULTI_PACK) > 8:            # weights were re-allocated (load_state_dict into new storage ...): start over
            _MULTI_PACK.clear()
        total = sum(w.numel() for w in ws)
        flat = torch.empty(2 * total, dtype=torch.float32, device=dev)
        jobs = (_lib.PackJob * (len(ws) + 1))()
        views, off, blk = [], 0, 0
        for i, w in enumerate(ws):
            O, I, KH, KW = w.shape
            n = w.numel()
            f, d = flat[off:off + n].view(O, KH, KW, I), flat[off + n:off + 2 * n].view(I, KH, KW, O)
            off += 2 * n
            tiled = KH * KW <= 9                              # one block per 32 x 32 channel block, transposed through LDS
            jobs[i] = _lib.PackJob(w.data_ptr(), f.data_ptr(), d.data_ptr(), O, I, KH, KW, blk, 1 if tiled else 0)
            blk += ((O + 31) // 32) * ((I + 31) // 32) if tiled else max(1, min(2048, (n + 1023) // 1024))
            views.append((f, d))
        jobs[len(ws)] = _lib.PackJob(None, None, None, 0, 0, 0, 0, blk, 0)
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        hit = _MULTI_PACK[key] = [raw, flat, views, len(ws), blk, 0]
    raw, flat, views, n, blk, gen = hit
    hit[5] = gen = gen + 1
    for f, d in views:       # a graph that still holds the previous contents must not run its data-gradients (ConvFn.backward checks)
        d._segsde_gen = gen
    check(_lib.lib().segsde_pack_weight_both_multi(_p(raw), n, blk, _stream(flat)), "pack_weight_both_multi")
    return views


def conv_forward(g, x0, x1, wpack, bias, act="none", want_stats=False, wfold=None, wino=None, keep_v=False):
    """y = act(conv(cat[up?(x0), x1]) + bias).  x0: [B,H0,W0,C0] (H0 = H/2 if g.up0), x1: [B,H,W,C1] or None.
    want_stats: also return the per-tile statistics partials of y for the BatchNorm that follows ([rows,2,Cout] doubles,
    or None when this shape cannot fuse them) -> (y, partials)."""
    B, H0, W0, C0 = x0.shape
    H, W = (2 * H0, 2 * W0) if g.up0 else (H0, W0)
    assert C0 == g.C0 and (g.C1 == 0) == (x1 is None)
    if x1 is not None:
        assert tuple(x1.shape) == (B, H, W, g.C1), (tuple(x1.shape), (B, H, W, g.C1))
    Ho, Wo = g.out_hw(H, W)
    y = torch.empty((B, Ho, Wo, g.Cout), dtype=torch.float32, device=x0.device)
    d = ConvDesc(B=B, H=H, W=W, C0=g.C0, C1=g.C1, ld0=nhwc_ld(x0), ld1=nhwc_ld(x1) if x1 is not None else 0,
                 up0=int(g.up0), Ho=Ho, Wo=Wo, Cout=g.Cout, ldy=g.Cout, ldy2=0, nsplit=0, KH=g.k, KW=g.k,
                 stride=g.stride, dil=g.dil, pad=g.pad, pad_mode=PAD_REFLECT if g.reflect else PAD_ZERO, in_div=1,
                 act=ACT[act], sum2x2=0, compute=g.compute)
    flops = 2.0 * B * Ho * Wo * g.Cout * g.CinAlg * g.k * g.k
    flops_x = flops * g.Cin / g.CinAlg * _live_tap_frac(g, H, W)   # executed: zero pad channels of a stem are multiplied too, dead tap rows are not
    if isinstance(wino, _KnPack):
        if (not want_stats or (bias is None and act == "none")) and winograd_fused_ok(g, B, H, W):
            r = winograd_fused("conv_fwd", x0, wino, bias=bias, act=act, want_stats=want_stats, tag=_tag(g, H, W), reflect=g.reflect,
                               x1=x1, up0=g.up0)
            if r is not None:
                return r if want_stats else r[0]
    elif wino is not None and not g.up0 and (not want_stats or (bias is None and act == "none")) and winograd_ok(g, B, H, W):
        WINO_V[0] = None
        r = _winograd("conv_fwd", g, x0, x1, wino, g.Cout, want_stats, flops, _tag(g, H, W), bias=bias, act=act, reflect=g.reflect,
                      keep_v=keep_v and WINOGRAD_KEEP_V)
        if r is not None:
            return r if want_stats else r[0]
    if wfold is not None and not want_stats:
        rc = _timed("conv_fwd", flops, x0, lambda: _lib.lib().segsde_conv2d_forward_upfold(
            ctypes.byref(d), _p(_f32(x0)), _p(x1), _p(wpack), _p(wfold), _p(bias), _p(y), _stream(x0)), _tag(g, H, W) + " fold",
            executed=flops * _fold_frac(g))
        if rc == 0:
            UPFOLD_TAKEN["fwd"] += 1
            return y
        if rc != -4:
            check(rc, "conv2d_forward_upfold")
    part = None
    if want_stats and bias is None and act == "none":
        rows = int(_lib.lib().segsde_conv2d_stats_rows(ctypes.byref(d)))
        if rows > 0:
            part = torch.empty((rows, 2, g.Cout), dtype=torch.float64, device=x0.device)
    _timed("conv_fwd", flops, x0, lambda: check(_lib.lib().segsde_conv2d_forward_stats(
        ctypes.byref(d), _p(_f32(x0)), _p(x1), _p(wpack), _p(bias), _p(y), None, _p(part), _stream(x0)), "conv2d_forward"),
        _tag(g, H, W), executed=flops_x)
    return (y, part) if want_stats else y


ACTGRAD_FUSED = _Slot(False)      # did the last conv_dgrad(actgrad=...) apply the derivative in its epilogue (functional.py counts from this)


SKIP_ACCUMULATED = _Slot(False)   # did the last conv_dgrad(accumulate_skip_into=...) add its skip gradient onto that tensor


def conv_dgrad(g, dy, wdpack, w_oihw, in_hw, need0=True, need1=True, accumulate_into=None, actgrad=None, fold=None, wino=None,
               accumulate_skip_into=None, wfpack=None):
    """Data gradient(s) of conv_forward w.r.t. (x0, x1).  dy: [B,Ho,Wo,Cout]; in_hw = (H, W) of the virtual input.
    wfpack: the FORWARD pack, if the caller has it (the mirrored-padding border kernel of the one-kernel Winograd route reads it).
    Returns (dx0, dx1); dx0 is at the *stored* resolution of x0 (2x2-summed when g.up0).
    accumulate_into: a dense [B,H,W,C0] tensor that already holds another gradient of x0 (single-source, non-upsampled
    convs): the result is ADDED to it in the kernel epilogue and that tensor is returned as dx0 (None is returned instead
    when this shape cannot accumulate in place -- the caller then adds).
    actgrad = (x0_saved, kind): x0 is the OUTPUT of an activation ("relu" / "elu" / "sigmoid"); dx0 is returned already
    multiplied by its derivative, i.e. as the gradient w.r.t. the PRE-activation (fused into the kernel epilogue where the
    shape allows, otherwise by a separate pass)."""
    B, Ho, Wo, Cout = dy.shape
    H, W = in_hw
    assert Cout == g.Cout
    ACTGRAD_FUSED[0] = False
    SKIP_ACCUMULATED[0] = False
    dx1 = torch.empty((B, H, W, g.C1), dtype=torch.float32, device=dy.device) if g.C1 else None
    L = _lib.lib()
    flops = 2.0 * B * Ho * Wo * Cout * g.CinAlg * g.k * g.k
    ag_y, ag_ld, ag_kind = None, 0, 0
    if actgrad is not None:
        ag_y, ag_kind = actgrad[0], ACT[actgrad[1]]
        ag_ld = nhwc_ld(ag_y)

    if isinstance(wino, _KnPack):
        ext = actgrad is not None or g.reflect
        if (winograd_fused_ok(g, B, H, W, dgrad=True) and (Ho, Wo) == (H, W) and (not ext or WINO_FUSED_DGRAD_EXT)
                and (not g.reflect or B * H * W >= WINO_FUSED_REFLECT_DGRAD_MIN_PIX)
                and (not g.reflect or (wfpack is not None and BORDERS2 and H >= 4 and W >= 4 and g.C0 % 32 == 0 and Cout % 32 == 0 and Cout <= 1024)
                     or reflect_borders_ok(B, H, W, Cout, g.C0, nhwc_ld(dy), ag_ld))
                and (actgrad is None or (ag_ld % 4 == 0 and ag_y.data_ptr() % 16 == 0))):
            r = winograd_fused("conv_dgrad", dy, wino, tag=_tag(g, H, W), accumulate_into=accumulate_into, actgrad=actgrad,
                               adjoint=(g, wdpack, wfpack) if g.reflect else None)
            if r is not None:
                ACTGRAD_FUSED[0] = actgrad is not None
                return r[0], None
    elif (wino is not None and accumulate_into is None and actgrad is None and not g.reflect and g.C1 == 0
            and winograd_ok(g, B, H, W, dgrad=True) and (Ho, Wo) == (H, W)):
        # zero-padded 3x3 / stride 1: dX is the same convolution of dY with the flipped, transposed kernel
        r = _winograd("conv_dgrad", g, dy, None, wino, g.C0, False, flops, _tag(g, H, W))
        if r is not None:
            return r[0], None

    def desc(sum2x2, accumulate=0):
        return ConvDesc(B=B, H=Ho, W=Wo, C0=Cout, C1=0, ld0=nhwc_ld(dy), ld1=0, up0=0, Ho=H, Wo=W, Cout=g.Cin, ldy=g.C0,
                        ldy2=g.C1, nsplit=g.C0, KH=g.k, KW=g.k, stride=1, dil=g.dil, pad=(g.k - 1) * g.dil - g.pad,
                        pad_mode=PAD_REFLECT_ADJOINT if g.reflect else PAD_ZERO, in_div=g.stride, act=0, sum2x2=sum2x2,
                        accumulate=accumulate, compute=g.compute)

    def launch(d, y, y2, fuse):
        return L.segsde_conv2d_dgrad_actgrad(ctypes.byref(d), _p(_f32(dy)), None, _p(wdpack), None, _p(y), _p(y2), None,
                                             _p(ag_y) if fuse else None, ag_ld if fuse else 0, ag_kind if fuse else 0,
                                             _stream(dy))

    def finish_unfused(dx0):
        """the activation derivative as its own pass (shapes whose epilogue cannot take it)"""
        if actgrad is None or dx0 is None:
            return dx0
        dz, _ = act_backward(dx0, ag_y, actgrad[1])
        return dz

    if accumulate_into is not None:
        acc = accumulate_into
        if g.up0 or g.C1 or tuple(acc.shape) != (B, H, W, g.C0) or not acc.is_contiguous():
            return None, None
        d = desc(0, 1)
        rc = _timed("conv_dgrad", flops, dy, lambda: launch(d, acc, None, actgrad is not None), _tag(g, H, W),
                    executed=flops * _live_tap_frac(g, H, W))
        if rc == -4:
            return None, None
        check(rc, "conv2d dgrad (accumulate)")
        ACTGRAD_FUSED[0] = actgrad is not None
        return acc, None
    if g.up0 and fold is not None and accumulate_into is None:
        # upsample-folded route: low-resolution gradient as a 4x4 stride-2 convolution of dy (+ clamp adjoint on the border),
        # skip-channel gradient as the ordinary reflection adjoint over its own channels
        wfold, wdfold = fold
        dx0 = torch.empty((B, H // 2, W // 2, g.C0), dtype=torch.float32, device=dy.device) if need0 else None
        dx1f = dx1 if need1 else None
        # accumulate_skip_into: a dense [B,H,W,C1] tensor that already holds another consumer's gradient of the skip source (the
        # encoder feature both decoders read): the skip launch adds onto it in its epilogue instead of producing a tensor that
        # autograd would add with a 12-byte-per-element pass
        acc1 = accumulate_skip_into
        if dx1f is not None and acc1 is not None and tuple(acc1.shape) == (B, H, W, g.C1) and acc1.is_contiguous():
            dx1f = acc1
        else:
            acc1 = None
        if dx0 is None and dx1f is None:
            return None, None
        # round 5: the skip-source gradient on the one-kernel Winograd route (flipped pack's columns [C0, C0 + C1), border kernel on
        # the forward pack's same channels); the folded call below then computes the upsampled source's gradient only
        w1 = None
        if (dx1f is not None and isinstance(wino, _KnPack) and wfpack is not None and winograd_fused_dgrad2_ok(g, B, H, W)
                and (Ho, Wo) == (H, W)):
            r = winograd_fused("conv_dgrad", dy, wino, tag="%d+0->%d k3 s1 d1 %dx%d refl skip-of-%d+%d" % (Cout, g.C1, H, W, g.C0, g.C1),
                               accumulate_into=acc1, adjoint=(g, wdpack, wfpack), cols=(g.C0, g.C1))
            if r is not None:
                w1 = r[0]
                WINO_FUSED_TAKEN["dgrad2"] += 1
        if w1 is not None:
            if dx0 is None:
                SKIP_ACCUMULATED[0] = acc1 is not None
                return None, w1
            dx1f = None
        df = ConvDesc(B=B, H=H, W=W, C0=g.C0, C1=g.C1, ld0=g.C0, ld1=g.C1, up0=1, Ho=Ho, Wo=Wo, Cout=Cout, ldy=Cout, ldy2=0,
                      nsplit=0, KH=3, KW=3, stride=1, dil=1, pad=1, pad_mode=PAD_REFLECT, in_div=1, act=0, sum2x2=0, compute=g.compute)
        fr = ((4.0 * g.C0 if dx0 is not None else 0.0) + (9.0 * g.C1 if dx1f is not None else 0.0)) / (9.0 * g.Cin)
        if w1 is not None:
            flops = flops * g.C0 / g.Cin          # the skip channels' multiply-adds were counted (and timed) with the Winograd launch
            fr = 4.0 / 9.0
        rc = _timed("conv_dgrad", flops, dy, lambda: L.segsde_conv2d_dgrad_upfold(
            ctypes.byref(df), _p(_f32(dy)), nhwc_ld(dy), _p(wdpack), _p(wfold), _p(wdfold), _p(dx0), _p(dx1f), 1 if acc1 is not None else 0,
            _p(ag_y) if dx0 is not None else None, ag_ld, ag_kind, _stream(dy)), _tag(g, H, W) + " fold", executed=flops * fr)
        if rc == 0:
            UPFOLD_TAKEN["dgrad"] += 1
            SKIP_ACCUMULATED[0] = acc1 is not None
            ACTGRAD_FUSED[0] = actgrad is not None and dx0 is not None
            return dx0, (w1 if w1 is not None else dx1f)
        if w1 is not None:
            check(rc, "conv2d_dgrad_upfold (upsampled source only)")
        if rc != -4:
            check(rc, "conv2d_dgrad_upfold")
    if g.up0:
        # fused: the 2x2 sum of the upsample adjoint happens in the GEMM epilogue (no full-resolution gradient tensor)
        dx0 = torch.empty((B, H // 2, W // 2, g.C0), dtype=torch.float32, device=dy.device)
        d = desc(1)
        rc = _timed("conv_dgrad", flops, dy, lambda: launch(d, dx0, dx1, actgrad is not None), _tag(g, H, W))
        if rc == 0:
            ACTGRAD_FUSED[0] = actgrad is not None
            return dx0, dx1
        if rc == -4 and actgrad is not None:
            rc = launch(d, dx0, dx1, False)
            if rc == 0:
                return finish_unfused(dx0), dx1
        if rc != -4:
            check(rc, "conv2d dgrad (fused upsample adjoint)")
    full0 = torch.empty((B, H, W, g.C0), dtype=torch.float32, device=dy.device)
    d = desc(0)
    fused = actgrad is not None and not g.up0
    rc = _timed("conv_dgrad", flops, dy, lambda: launch(d, full0, dx1, fused), _tag(g, H, W), executed=flops * _live_tap_frac(g, H, W))
    if rc == -4 and fused:
        fused = False
        rc = launch(d, full0, dx1, False)
    check(rc, "conv2d dgrad")
    if g.up0:
        dx0 = torch.empty((B, H // 2, W // 2, g.C0), dtype=torch.float32, device=dy.device)
        check(L.segsde_upsample2x_backward(_p(full0), g.C0, B, H // 2, W // 2, g.C0, _p(dx0), g.C0, _stream(dy)),
              "upsample2x_backward")
    else:
        dx0 = full0
    ACTGRAD_FUSED[0] = fused
    return (dx0 if fused else finish_unfused(dx0)), dx1


def conv_wgrad(g, x0, x1, dy, wino_v=None, out=None):
    """dW in OIHW layout.  wino_v: the transformed input the forward's Winograd call kept (hipops.WINO_V), if any.
    out: a dense [Cout, Cin, k, k] tensor to write into (the parameter's slice of a gradient bucket, ddp.grad_destination)."""
    B, H0, W0, _ = x0.shape
    H, W = (2 * H0, 2 * W0) if g.up0 else (H0, W0)
    _, Ho, Wo, Cout = dy.shape
    d = ConvDesc(B=B, H=H, W=W, C0=g.C0, C1=g.C1, ld0=nhwc_ld(x0), ld1=nhwc_ld(x1) if x1 is not None else 0,
                 up0=int(g.up0), Ho=Ho, Wo=Wo, Cout=Cout, ldy=Cout, ldy2=0, nsplit=0, KH=g.k, KW=g.k, stride=g.stride,
                 dil=g.dil, pad=g.pad, pad_mode=PAD_REFLECT if g.reflect else PAD_ZERO, in_div=1, act=0, sum2x2=0, compute=g.compute)
    L = _lib.lib()
    if out is not None and tuple(out.shape) == (Cout, g.Cin, g.k, g.k) and out.is_contiguous() and out.dtype == torch.float32:
        dw = out
    else:
        dw = torch.empty((Cout, g.Cin, g.k, g.k), dtype=torch.float32, device=dy.device)
    flops = 2.0 * B * Ho * Wo * Cout * g.CinAlg * g.k * g.k
    flops_x = flops * g.Cin / g.CinAlg * _live_tap_frac(g, H, W, wgrad=True)
    if wino_v is None and (Ho, Wo) == (H, W) and winograd_fused_wgrad_ok(g, B, H, W):
        nbytes = L.segsde_conv2d_wgrad_winograd_fused_workspace(ctypes.byref(d))
        if nbytes:
            ws = _ws(nbytes, dy)
            rc = _timed("conv_wgrad", flops, dy, lambda: L.segsde_conv2d_wgrad_winograd_fused(
                ctypes.byref(d), _p(_f32(x0)), _p(x1), _p(_f32(dy)), nhwc_ld(dy), _p(dw), _p(ws), nbytes, _stream(dy)),
                _tag(g, H, W) + " wino-fused",
                # (9 / 36 on the upsampled source's channels where the launch skips its zero positions: that source carries at
                # least half of the channels, csrc/winograd_wgrad.hip)
                executed=flops * ((9.0 * g.C0 + 16.0 * g.C1) / (36.0 * g.Cin)
                                  if (g.up0 and WINO_UP_SKIP and 2 * g.C0 >= g.Cin) else 16.0 / 36.0))
            if rc == 0:
                WINO_FUSED_TAKEN["wgrad"] += 1
                return dw
            if rc != -4:
                check(rc, "conv2d_wgrad_winograd_fused")
    if not g.up0 and winograd_ok(g, B, H, W) and (Ho, Wo) == (H, W):
        nbytes = L.segsde_conv2d_wgrad_winograd_workspace(ctypes.byref(d))
        if nbytes:
            ws = _ws(nbytes, dy)
            rc = _timed("conv_wgrad", flops, dy, lambda: L.segsde_conv2d_wgrad_winograd(
                ctypes.byref(d), _p(x0), _p(x1), _p(_f32(dy)), nhwc_ld(dy), _p(wino_v), _p(dw), _p(ws), nbytes, _stream(dy)),
                _tag(g, H, W) + " wino", executed=flops * 16.0 / 36.0)
            if rc == 0:
                WINOGRAD_TAKEN["wgrad"] += 1
                return dw
            if rc != -4:
                check(rc, "conv2d_wgrad_winograd")
    if upfold_ok(g, B * Ho * Wo):
        nbytes = L.segsde_conv2d_wgrad_upfold_workspace(ctypes.byref(d))
        if nbytes:
            ws = _ws(nbytes, dy)
            rc = _timed("conv_wgrad", flops, dy, lambda: L.segsde_conv2d_wgrad_upfold(
                ctypes.byref(d), _p(x0), _p(x1), _p(_f32(dy)), nhwc_ld(dy), _p(dw), _p(ws), nbytes, _stream(dy)),
                _tag(g, H, W) + " fold", executed=flops * _fold_frac(g))
            if rc == 0:
                UPFOLD_TAKEN["wgrad"] += 1
                return dw
            if rc != -4:
                check(rc, "conv2d_wgrad_upfold")
    nbytes = L.segsde_conv2d_wgrad_workspace(ctypes.byref(d))
    ws = _ws(nbytes, dy)
    _timed("conv_wgrad", flops, dy, lambda: check(L.segsde_conv2d_wgrad(
        ctypes.byref(d), _p(x0), _p(x1), _p(_f32(dy)), nhwc_ld(dy), _p(dw), _p(ws), nbytes, _stream(dy)), "conv2d_wgrad"),
        _tag(g, H, W), executed=flops_x)
    return dw


# ----------------------------------------------------------------------------------------------
# BatchNorm / activations / pooling / resampling
# ----------------------------------------------------------------------------------------------
def _rows(t):
    """(M, C, ld) of an NHWC (or [M, C]) tensor"""
    if t.dim() == 2:
        return t.shape[0], t.shape[1], int(t.stride(0)) if t.shape[0] > 1 else t.shape[1]
    B, H, W, C = t.shape
    return B * H * W, C, nhwc_ld(t)


def bn_stats(x, running_mean, running_var, momentum, eps, update_running=True, num_batches_tracked=None):
    M, C, ld = _rows(x)
    L = _lib.lib()
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    nb = L.segsde_bn_stats_workspace(M, C)
    ws = _ws(nb, x)
    _timed('hbm_bn_stats', 4.0 * M * C, x, lambda: check(L.segsde_bn_stats(_p(_f32(x)), ld, M, C, _p(mean), _p(invstd), _p(running_mean if update_running else None),
                            _p(running_var if update_running else None), float(momentum), float(eps),
                            _p(num_batches_tracked), _p(ws), nb, _stream(x)), "bn_stats"))
    return mean, invstd


_BN_PARTIALS_WS = {}          # channel count -> workspace bytes (a pure function of C)


def bn_stats_from_partials(part, M, running_mean, running_var, momentum, eps, update_running=True, num_batches_tracked=None):
    """batch statistics from the partial sums a conv_forward(want_stats=True) launch produced (same outputs / running
    update as bn_stats, without reading the activation tensor again)"""
    rows, _, C = part.shape
    L = _lib.lib()
    mi = torch.empty((2, C), dtype=torch.float32, device=part.device)      # one allocation: ~70 calls per step of the small workloads
    mean, invstd = mi[0], mi[1]
    nb = _BN_PARTIALS_WS.get(C)
    if nb is None:
        nb = _BN_PARTIALS_WS[C] = L.segsde_bn_stats_from_partials_workspace(C)
    ws = _ws(nb, part)
    check(L.segsde_bn_stats_from_partials(_p(part), rows, M, C, _p(mean), _p(invstd),
                                          _p(running_mean if update_running else None),
                                          _p(running_var if update_running else None), float(momentum), float(eps),
                                          _p(num_batches_tracked), _p(ws), nb, _stream(part)), "bn_stats_from_partials")
    return mean, invstd


def bn_eval_stats(running_mean, running_var, eps):
    C = running_mean.numel()
    mean = torch.empty(C, dtype=torch.float32, device=running_mean.device)
    invstd = torch.empty_like(mean)
    check(_lib.lib().segsde_bn_eval_stats(_p(running_mean), _p(running_var), C, float(eps), _p(mean), _p(invstd),
                                          _stream(running_mean)), "bn_eval_stats")
    return mean, invstd


def bn_apply(x, mean, invstd, gamma, beta, residual=None, act="none", drop_p=0.0, seed=0, out=None):
    M, C, ld = _rows(x)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device) if out is None else out
    ldr = _rows(residual)[2] if residual is not None else 0
    _timed('hbm_bn_apply', (8.0 + (4.0 if residual is not None else 0.0)) * M * C, x, lambda: check(_lib.lib().segsde_bn_apply(_p(x), ld, M, C, _p(mean), _p(invstd), _p(gamma), _p(beta), _p(residual), ldr, _p(y),
                                     _rows(y)[2], ACT[act], float(drop_p), int(seed), _stream(x)), "bn_apply"))
    return y


def bn_backward(dy, y, x, mean, invstd, gamma, act="none", drop_p=0.0, seed=0, batch_stats=True, need_dx=True,
                need_dres=False, beta=None, dgamma_out=None, dbeta_out=None):
    """y=None selects the remask mode of segsde_bn_backward (act none / plain ReLU: mask recomputed from x, beta needed).
    dgamma_out / dbeta_out: dense [C] tensors to write into (gradient-bucket slices, ddp.grad_destination)."""
    M, C, ldx = _rows(x)
    L = _lib.lib()

    def _dst(t):
        ok = t is not None and tuple(t.shape) == (C,) and t.is_contiguous() and t.dtype == torch.float32
        return t if ok else torch.empty(C, dtype=torch.float32, device=x.device)
    dgamma, dbeta = _dst(dgamma_out), _dst(dbeta_out)
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device) if need_dx else None
    dres = torch.empty(x.shape, dtype=torch.float32, device=x.device) if need_dres else None
    nb = L.segsde_bn_backward_workspace(M, C)
    ws = _ws(nb, x)
    _timed('hbm_bn_backward', (8.0 + (4.0 if y is not None else 0.0) + (4.0 if need_dx else 0.0) + (4.0 if need_dres else 0.0)) * M * C + 8.0 * M * C, x, lambda: check(L.segsde_bn_backward(_p(_f32(dy)), _rows(dy)[2], _p(y), _rows(y)[2] if y is not None else ldx, _p(x), ldx, M, C,
                               _p(mean), _p(invstd), _p(gamma), _p(beta), ACT[act], float(drop_p), int(seed), int(batch_stats), _p(dgamma), _p(dbeta),
                               _p(dx), C, _p(dres), C, _p(ws), nb, _stream(x)), "bn_backward"))
    return dx, dres, dgamma, dbeta


def act_backward(dy, y, act, need_dbias=False, need_dz=True):
    M, C, ldy = _rows(y)
    L = _lib.lib()
    dz = torch.empty(y.shape, dtype=torch.float32, device=y.device) if need_dz else None
    dbias = torch.empty(C, dtype=torch.float32, device=y.device) if need_dbias else None
    nb = L.segsde_colsum_workspace(M, C)
    ws = _ws(nb, y)
    _timed('hbm_act_backward', (8.0 + (4.0 if need_dz else 0.0)) * M * C, y, lambda: check(L.segsde_act_backward(_p(_f32(dy)), _rows(dy)[2], _p(y), ldy, M, C, ACT[act], _p(dz), C, _p(dbias), _p(ws), nb,
                                _stream(y)), "act_backward"))
    return dz, dbias


def colsum(x):
    M, C, ld = _rows(x)
    if C == 1 and ld == 1 and M >= 4096 and M % 64 == 0:
        # a single column (the bias gradient of a disparity head: 8 M values) gives the column kernel one float per row
        # to chew on -- 0.85 ms for 33 MB; as 64 columns of M / 64 rows it is an ordinary coalesced reduction (+ a 64-value sum)
        return colsum(x.reshape(M // 64, 64)).sum().reshape(1)
    L = _lib.lib()
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    nb = L.segsde_colsum_workspace(M, C)
    ws = _ws(nb, x)
    check(L.segsde_colsum(_p(_f32(x)), ld, M, C, _p(out), _p(ws), nb, _stream(x)), "colsum")
    return out


def maxpool_forward(x):
    B, H, W, C = x.shape
    x = x.contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    _timed('hbm_maxpool_fwd', 4.0 * x.numel() * 1.25 + 0.25 * x.numel(), x, lambda: check(_lib.lib().segsde_maxpool3x3s2_forward(_p(_f32(x)), B, H, W, C, _p(y), _p(idx), _stream(x)), "maxpool_fwd"))
    return y, idx


def maxpool_backward(dy, idx, in_shape, accumulate_into=None):
    """accumulate_into: a dense tensor of in_shape that already holds another consumer's gradient of the pooled tensor: the
    result is added to it in the kernel and that tensor is returned"""
    B, H, W, C = in_shape
    dy = dy.contiguous()
    acc = accumulate_into if (accumulate_into is not None and tuple(accumulate_into.shape) == tuple(in_shape)
                              and accumulate_into.is_contiguous()) else None
    dx = acc if acc is not None else torch.empty(in_shape, dtype=torch.float32, device=dy.device)
    check(_lib.lib().segsde_maxpool3x3s2_backward(_p(_f32(dy)), _p(idx), B, H, W, C, _p(dx), 1 if acc is not None else 0,
                                                  _stream(dy)), "maxpool_bwd")
    return dx


def resize_bilinear(x, out_hw, align_corners=False):
    B, Hi, Wi, C = x.shape
    Ho, Wo = out_hw
    y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
    check(_lib.lib().segsde_resize_bilinear_forward(_p(_f32(x)), nhwc_ld(x), B, Hi, Wi, C, _p(y), C, Ho, Wo,
                                                    int(align_corners), _stream(x)), "resize_fwd")
    return y


def resize_bilinear_backward(dy, in_hw, align_corners=False):
    B, Ho, Wo, C = dy.shape
    Hi, Wi = in_hw
    dx = torch.empty((B, Hi, Wi, C), dtype=torch.float32, device=dy.device)
    check(_lib.lib().segsde_resize_bilinear_backward(_p(_f32(dy)), nhwc_ld(dy), B, Hi, Wi, C, _p(dx), C, Ho, Wo,
                                                     int(align_corners), _stream(dy)), "resize_bwd")
    return dx


def global_avgpool(x):
    B, H, W, C = x.shape
    y = torch.empty((B, 1, 1, C), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    nws = int(L.segsde_global_avgpool_workspace(B, H * W, C))   # > 0: few images x few channels, the pixels are sliced
    ws = _ws(nws, x) if nws else None
    check(L.segsde_global_avgpool_forward(_p(_f32(x)), nhwc_ld(x), B, H * W, C, _p(y), _p(ws) if nws else None, nws,
                                          _stream(x)), "gap_fwd")
    return y


def global_avgpool_backward(dy, in_shape):
    B, H, W, C = in_shape
    dy = dy.contiguous()
    dx = torch.empty(in_shape, dtype=torch.float32, device=dy.device)
    check(_lib.lib().segsde_global_avgpool_backward(_p(_f32(dy)), B, H * W, C, _p(dx), C, _stream(dy)), "gap_bwd")
    return dx


def gate_forward(f, a):
    f, a = f.contiguous(), a.contiguous()
    y = torch.empty_like(f)
    check(_lib.lib().segsde_gate_forward(_p(_f32(f)), _p(_f32(a)), f.numel(), _p(y), _stream(f)), "gate_fwd")
    return y


def gate_backward(dy, f, a):
    dy = dy.contiguous()
    df, da = torch.empty_like(f), torch.empty_like(a)
    check(_lib.lib().segsde_gate_backward(_p(_f32(dy)), _p(f), _p(a), f.numel(), _p(df), _p(da), _stream(f)), "gate_bwd")
    return df, da


def axpby(alpha, x, beta=0.0, y=None, out=None):
    x = x.contiguous()
    if y is not None:
        y = y.contiguous()
    out = torch.empty_like(x) if out is None else out
    check(_lib.lib().segsde_axpby(x.numel(), float(alpha), _p(_f32(x)), float(beta), _p(y), _p(out), _stream(x)), "axpby")
    return out


def axpby_dev(alpha_t, x, beta_t=None, y=None, out=None):
    """out = alpha_t[0]*x (+ beta_t[0]*y) with 1-element DEVICE tensors as scalars (no host sync)"""
    x = x.contiguous()
    if y is not None:
        y = y.contiguous()
    out = torch.empty_like(x) if out is None else out
    check(_lib.lib().segsde_axpby_dev(x.numel(), _p(_f32(alpha_t)), _p(_f32(x)), _p(beta_t), _p(y), _p(out), _stream(x)),
          "axpby_dev")
    return out


def copy_channels(src, dst):
    M, C, lds = _rows(src)
    M2, C2, ldd = _rows(dst)
    assert M == M2 and C == C2
    check(_lib.lib().segsde_copy_channels(_p(_f32(src)), lds, _p(dst), ldd, M, C, _stream(src)), "copy_channels")
    return dst


def scale_channels(x, scale):
    """x: NHWC [B,H,W,C]; scale [B,C] -> x * scale[b, c] (nn.Dropout2d mask application and its adjoint)"""
    B, Hh, W, C = x.shape
    y = torch.empty((B, Hh, W, C), dtype=torch.float32, device=x.device)
    check(_lib.lib().segsde_scale_channels(_p(_f32(x)), nhwc_ld(x), B, Hh * W, C, _p(_f32(scale.contiguous())), _p(y), C,
                                           _stream(x)), "scale_channels")
    return y


def dropout(x, p, seed):
    """nn.Dropout(p) on an NHWC tensor with the counter-based mask of ``seed`` (its own adjoint with the same seed)"""
    M, C, ld = _rows(x)
    y = torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)
    check(_lib.lib().segsde_dropout(_p(_f32(x)), ld, M, C, float(p), int(seed), _p(y), C, _stream(x)), "dropout")
    return y


def nchw_to_nhwc(x, mean=0.0, std=1.0, pad_to=1):
    """pad_to > 1 rounds the channel count of the result up to a multiple of pad_to; the extra channels are zero
    (the 3 / 6-channel network input becomes 4 / 8 so that the stem convolution takes the float4 gather)."""
    x = _f32(x).contiguous()
    B, C, H, W = x.shape
    Cp = -(-C // pad_to) * pad_to
    y = torch.empty((B, H, W, Cp), dtype=torch.float32, device=x.device)     # the kernel writes the zero channels too
    check(_lib.lib().segsde_nchw_to_nhwc(_p(x), B, C, H, W, float(mean), float(std), _p(y), Cp, _stream(x)), "nchw_to_nhwc")
    return y


# ----------------------------------------------------------------------------------------------
# network stems (7x7, stride 2, padding 3 on the normalised image): see csrc/conv_igemm.hip "Network stems"
# ----------------------------------------------------------------------------------------------
STEM = os.environ.get("SEGSDE_STEM", "1") != "0"
STEM_TAKEN = {"fwd": 0, "wgrad": 0}


def stem_ok(image, conv):
    """conv: the module holding the stem's state (nn.Conv2d attributes).  The dedicated path needs the reference's stem geometry."""
    return (STEM and image.dim() == 4 and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
            and conv.dilation == (1, 1) and conv.bias is None and conv.groups == 1 and image.shape[1] == conv.in_channels
            and conv.in_channels <= 8 and conv.out_channels % 4 == 0 and image.shape[2] >= 2 and image.shape[3] >= 2
            and image.shape[0] * (image.shape[2] + 6) * (image.shape[3] + 8) * 8 < 2 ** 31)


def stem_input(image, mean, std):
    """(image - mean) / std as 4 / 8-channel pixels inside a zero border of 3 rows above / below, 3 columns left, 5 right."""
    x = _f32(image).contiguous()
    B, C, H, W = x.shape
    cp = 4 if C <= 4 else 8
    y = torch.empty((B, H + 6, W + 8, cp), dtype=torch.float32, device=x.device)   # the kernel writes the border too
    check(_lib.lib().segsde_nchw_to_nhwc_bordered(_p(x), B, C, H, W, float(mean), float(std), _p(y), cp, 3, 3, H + 6, W + 8,
                                                  _stream(x)), "nchw_to_nhwc_bordered")
    return y


def stem_pack(weight):
    Cout, C = weight.shape[0], weight.shape[1]
    cp = 4 if C <= 4 else 8
    w = _f32(weight.detach()).contiguous()
    out = torch.empty((Cout, 7, 8 * cp), dtype=torch.float32, device=w.device)
    check(_lib.lib().segsde_stem_pack(_p(w), Cout, C, cp, _p(out), _stream(w)), "stem_pack")
    return out


def stem_forward(xpad, wstem, C, want_stats=False):
    """xpad: stem_input(); wstem: stem_pack(); C: real input planes (FLOP accounting) -> y [B, Ho, Wo, Cout] (, partials)"""
    B, Hp, Wp, cp = xpad.shape
    Cout = wstem.shape[0]
    H, W = Hp - 6, Wp - 8
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=xpad.device)
    part = None
    if want_stats:
        rows = int(_lib.lib().segsde_stem7x7_stats_rows(B, Hp, Wp, cp, Cout))
        if rows > 0:
            part = torch.empty((rows, 2, Cout), dtype=torch.float64, device=xpad.device)
    flops = 2.0 * B * Ho * Wo * Cout * C * 49
    _timed("conv_fwd", flops, xpad, lambda: check(_lib.lib().segsde_stem7x7_forward(
        _p(xpad), B, Hp, Wp, cp, _p(wstem), Cout, _p(y), _p(part), _stream(xpad)), "stem7x7_forward"),
        "stem c%d k7 s2 %dx%d" % (C, H, W), executed=2.0 * B * Ho * Wo * Cout * 56 * cp)
    STEM_TAKEN["fwd"] += 1
    return (y, part) if want_stats else y


def stem_wgrad(xpad, dy, C):
    B, Hp, Wp, cp = xpad.shape
    _, Ho, Wo, Cout = dy.shape
    dy = _f32(dy)
    dw = torch.empty((Cout, C, 7, 7), dtype=torch.float32, device=dy.device)
    nbytes = int(_lib.lib().segsde_stem7x7_wgrad_workspace(B, Hp, Wp, cp, Cout))
    ws = _ws(nbytes, dy)
    flops = 2.0 * B * Ho * Wo * Cout * C * 49
    _timed("conv_wgrad", flops, dy, lambda: check(_lib.lib().segsde_stem7x7_wgrad(
        _p(xpad), B, Hp, Wp, cp, _p(dy), nhwc_ld(dy), Cout, C, _p(dw), _p(ws), nbytes, _stream(dy)), "stem7x7_wgrad"),
        "stem c%d k7 s2 %dx%d" % (C, Hp - 6, Wp - 8), executed=2.0 * B * Ho * Wo * Cout * 56 * cp)
    STEM_TAKEN["wgrad"] += 1
    return dw


def nhwc_to_nchw(x):
    B, H, W, C = x.shape
    y = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    check(_lib.lib().segsde_nhwc_to_nchw(_p(_f32(x)), nhwc_ld(x), B, C, H, W, _p(y), _stream(x)), "nhwc_to_nchw")
    return y


# ----------------------------------------------------------------------------------------------
# pose
# ----------------------------------------------------------------------------------------------
def pose_matrix(axisangle, translation, invert):
    """axisangle / translation: [B, F, 1, 3] network outputs; frame 0 is used (joint_segmentation_depth.py:48-49)."""
    assert axisangle.is_contiguous() and translation.is_contiguous()
    B = axisangle.shape[0]
    stride = axisangle.stride(0) if B > 1 else axisangle[0].numel()
    M = torch.empty((B, 4, 4), dtype=torch.float32, device=axisangle.device)
    check(_lib.lib().segsde_pose_matrix_forward(_p(_f32(axisangle)), _p(_f32(translation)), B, int(stride), int(invert),
                                                _p(M), _stream(axisangle)), "pose_fwd")
    return M


def pose_matrix_backward(axisangle, translation, dM, invert):
    B = axisangle.shape[0]
    stride = axisangle.stride(0) if B > 1 else axisangle[0].numel()
    daa, dtr = torch.zeros_like(axisangle), torch.zeros_like(translation)
    check(_lib.lib().segsde_pose_matrix_backward(_p(axisangle), _p(translation), _p(_f32(dM.contiguous())), B, int(stride),
                                                 int(invert), _p(daa), _p(dtr), _stream(axisangle)), "pose_bwd")
    return daa, dtr


# ----------------------------------------------------------------------------------------------
# monodepth loss
# ----------------------------------------------------------------------------------------------
def warp_forward(disp, inv_K, K, T, src, min_depth, max_depth, want_grid=False, want_depth=False):
    B, _, hs, ws = disp.shape
    _, _, H, W = src.shape
    color = torch.empty((B, 3, H, W), dtype=torch.float32, device=src.device)
    grid = torch.empty((B, H, W, 2), dtype=torch.float32, device=src.device) if want_grid else None
    depth = torch.empty((B, 1, H, W), dtype=torch.float32, device=src.device) if want_depth else None
    _timed('hbm_warp_fwd', 4.0 * B * (hs * ws + (3 + 3 + (2 if want_grid else 0) + (1 if want_depth else 0)) * H * W), src, lambda: check(_lib.lib().segsde_warp_forward(_p(_f32(disp.contiguous())), hs, ws, _p(_f32(inv_K.contiguous())),
                                         _p(_f32(K.contiguous())), _p(_f32(T.contiguous())), _p(_f32(src.contiguous())),
                                         B, H, W, float(min_depth), float(max_depth), _p(color), _p(grid), _p(depth),
                                         _stream(src)), "warp_fwd"))
    return color, grid, depth


def warp_backward(gcolor, disp, inv_K, K, T, src, min_depth, max_depth, g_disp_up, gT):
    """accumulates into g_disp_up [B,H,W] and gT [B,4,4]"""
    B, _, hs, ws = disp.shape
    _, _, H, W = src.shape
    L = _lib.lib()
    nb = L.segsde_warp_backward_workspace(B, H, W)
    ws_ = _ws(nb, src)
    check(L.segsde_warp_backward(_p(_f32(gcolor.contiguous())), _p(disp.contiguous()), hs, ws, _p(inv_K.contiguous()),
                                 _p(K.contiguous()), _p(T.contiguous()), _p(src.contiguous()), B, H, W, float(min_depth),
                                 float(max_depth), _p(g_disp_up), _p(gT), _p(ws_), nb, _stream(src)), "warp_bwd")


def reprojection_error(pred, target, no_ssim, out_plane):
    """out_plane: a [B,H,W] view (channel of a [B,n,H,W] tensor)"""
    B, _, H, W = pred.shape
    assert out_plane.stride(2) == 1 and out_plane.stride(1) == W
    check(_lib.lib().segsde_reprojection_error_forward(_p(_f32(pred.contiguous())), _p(_f32(target.contiguous())), B, H, W,
                                                       int(no_ssim), _p(out_plane), int(out_plane.stride(0)) if B > 1 else H * W,
                                                       _stream(pred)), "reproj_err_fwd")


def reprojection_error_backward(pred, target, gerr_plane, no_ssim):
    B, _, H, W = pred.shape
    L = _lib.lib()
    nb = 0 if no_ssim else L.segsde_reprojection_error_backward_workspace(B, H, W)
    ws_ = _ws(nb, pred) if nb else None
    gpred = torch.empty_like(pred)
    check(L.segsde_reprojection_error_backward(_p(pred.contiguous()), _p(target.contiguous()), _p(gerr_plane),
                                               int(gerr_plane.stride(0)) if B > 1 else H * W, B, H, W, int(no_ssim),
                                               _p(gpred), _p(ws_), nb, _stream(pred)), "reproj_err_bwd")
    return gpred


def photometric_identity(src0, src1, target, no_ssim):
    """err(src_f, target), f = 0, 1 -> [B,2,H,W] (the auto-mask's identity terms, the same for every scale)"""
    B, _, Hh, W = target.shape
    ident = torch.empty((B, 2, Hh, W), dtype=torch.float32, device=target.device)
    _timed('hbm_photometric_identity', 4.0 * B * Hh * W * (9 + 2), target, lambda: check(_lib.lib().segsde_photometric_identity(_p(_f32(src0.contiguous())), _p(_f32(src1.contiguous())),
                                                 _p(_f32(target.contiguous())), B, Hh, W, int(no_ssim), _p(ident),
                                                 _stream(target)), "photometric_identity"))
    return ident


def photometric_forward(pred0, pred1, target, ident, noise, no_ssim, avg, want_selection=True):
    """fused SSIM+L1 errors of both warped frames + auto-mask minimum -> (sum [1], sel uint8 [B,H,W], identity_selection)"""
    B, _, Hh, W = target.shape
    L = _lib.lib()
    sel = torch.empty((B, Hh, W), dtype=torch.uint8, device=target.device)
    isel = torch.empty((B, Hh, W), dtype=torch.float32, device=target.device) if (want_selection and ident is not None) else None
    out = torch.empty(1, dtype=torch.float32, device=target.device)
    nb = L.segsde_photometric_workspace(B, Hh, W)
    ws_ = _ws(nb, target)
    _timed('hbm_photometric_fwd', B * Hh * W * (4.0 * (9 + (2 if ident is not None else 0) + (2 if noise is not None else 0) + (1 if isel is not None else 0)) + 1.0), target, lambda: check(L.segsde_photometric_forward(_p(_f32(pred0.contiguous())), _p(_f32(pred1.contiguous())), _p(_f32(target.contiguous())),
                                       _p(ident), _p(noise), B, Hh, W, int(no_ssim), int(avg), _p(sel), _p(isel), _p(out),
                                       _p(ws_), nb, _stream(target)), "photometric_forward"))
    return out, sel, isel


def photometric_backward(pred0, pred1, target, sel, has_ident, disp, inv_K, K, T0, T1, src0, src1, min_depth, max_depth,
                         no_ssim, avg, scale, weight, gT0, gT1):
    """-> g_disp_up [B,H,W] (d loss / d upsampled disparity, both frames); gT0 / gT1 [B,4,4] are accumulated into with the
    device scalar ``weight`` (upstream gradient of this scale's loss)"""
    B, _, Hh, W = target.shape
    _, _, hs, ws = disp.shape
    L = _lib.lib()
    gup = torch.empty((B, Hh, W), dtype=torch.float32, device=target.device)
    nb = L.segsde_photometric_workspace(B, Hh, W)
    ws_ = _ws(nb, target)
    _timed('hbm_photometric_bwd', B * (Hh * W * (4.0 * (9 + 6 + 1) + 1.0) + 4.0 * hs * ws), target, lambda: check(L.segsde_photometric_backward(_p(pred0.contiguous()), _p(pred1.contiguous()), _p(target.contiguous()), _p(sel),
                                        2 if has_ident else 0, _p(disp.contiguous()), hs, ws, _p(inv_K.contiguous()),
                                        _p(K.contiguous()), _p(T0.contiguous()), _p(T1.contiguous()), _p(src0.contiguous()),
                                        _p(src1.contiguous()), B, Hh, W, float(min_depth), float(max_depth), int(no_ssim),
                                        int(avg), float(scale), _p(weight), _p(gup), _p(gT0), _p(gT1), _p(ws_), nb,
                                        _stream(target)), "photometric_backward"))
    return gup


def automask_min(ident, noise, reproj, avg, want_selection=True):
    B, nr, H, W = reproj.shape
    L = _lib.lib()
    sel = torch.empty((B, H, W), dtype=torch.uint8, device=reproj.device)
    isel = torch.empty((B, H, W), dtype=torch.float32, device=reproj.device) if (want_selection and ident is not None) else None
    out = torch.empty(1, dtype=torch.float32, device=reproj.device)
    nb = L.segsde_automask_workspace(B, H, W)
    ws_ = _ws(nb, reproj)
    check(L.segsde_automask_min_forward(_p(ident), _p(noise), _p(_f32(reproj)), nr, int(avg), B, H, W, _p(sel), _p(isel),
                                        _p(out), _p(ws_), nb, _stream(reproj)), "automask_fwd")
    return out, sel, isel


def automask_min_backward(sel, has_ident, n_reproj, avg, scale):
    B, H, W = sel.shape
    g = torch.empty((B, n_reproj, H, W), dtype=torch.float32, device=sel.device)
    check(_lib.lib().segsde_automask_min_backward(_p(sel), 2 if has_ident else 0, n_reproj, int(avg), B, H, W, float(scale),
                                                  _p(g), _stream(sel)), "automask_bwd")
    return g


def smoothness_forward(disp, img):
    B, _, h, w = disp.shape
    L = _lib.lib()
    mean = torch.empty(B, dtype=torch.float32, device=disp.device)
    out = torch.empty(1, dtype=torch.float32, device=disp.device)
    nb = L.segsde_smoothness_workspace(B, h, w)
    ws_ = _ws(nb, disp)
    _timed('hbm_smooth_fwd', 16.0 * B * h * w, disp, lambda: check(L.segsde_smoothness_forward(_p(_f32(disp.contiguous())), _p(_f32(img.contiguous())), B, h, w, _p(mean), _p(out),
                                      _p(ws_), nb, _stream(disp)), "smooth_fwd"))
    return out, mean


def smoothness_backward(disp, img, mean, scale, gdisp):
    """accumulates into gdisp [B,1,h,w]"""
    B, _, h, w = disp.shape
    L = _lib.lib()
    nb = L.segsde_smoothness_workspace(B, h, w)
    ws_ = _ws(nb, disp)
    _timed('hbm_smooth_bwd', 24.0 * B * h * w, disp, lambda: check(L.segsde_smoothness_backward(_p(disp.contiguous()), _p(img.contiguous()), _p(mean), B, h, w, float(scale),
                                       _p(gdisp), _p(ws_), nb, _stream(disp)), "smooth_bwd"))


def smooth_loss_forward(disp, img):
    """get_smooth_loss(disp, img) without the mean normalisation (models/monodepth_layers.py:208-221) -> [1]"""
    B, _, h, w = disp.shape
    L = _lib.lib()
    out = torch.empty(1, dtype=torch.float32, device=disp.device)
    nb = L.segsde_smooth_loss_workspace(B, h, w)
    ws_ = _ws(nb, disp)
    check(L.segsde_smooth_loss_forward(_p(_f32(disp.contiguous())), _p(_f32(img.contiguous())), B, h, w, _p(out), _p(ws_), nb,
                                       _stream(disp)), "smooth_loss_fwd")
    return out


def smooth_loss_backward(disp, img, scale):
    B, _, h, w = disp.shape
    L = _lib.lib()
    g = torch.zeros((B, 1, h, w), dtype=torch.float32, device=disp.device)
    nb = L.segsde_smooth_loss_workspace(B, h, w)
    ws_ = _ws(nb, disp)
    check(L.segsde_smooth_loss_backward(_p(disp.contiguous()), _p(img.contiguous()), B, h, w, float(scale), _p(g), _p(ws_), nb,
                                        _stream(disp)), "smooth_loss_bwd")
    return g


def ssim_map(x, y):
    """SSIM.forward (models/monodepth_layers.py:240-254): per-channel clamp((1 - SSIM) / 2, 0, 1) of two [B,C,H,W] images"""
    B, C, Hh, W = x.shape
    assert tuple(y.shape) == tuple(x.shape)
    out = torch.empty((B, C, Hh, W), dtype=torch.float32, device=x.device)
    check(_lib.lib().segsde_ssim_map_forward(_p(_f32(x.contiguous())), _p(_f32(y.contiguous())), B, C, Hh, W, _p(out),
                                             _stream(x)), "ssim_map_fwd")
    return out


def ssim_map_backward(x, y, gout, need_x=True, need_y=True):
    B, C, Hh, W = x.shape
    gx = torch.empty((B, C, Hh, W), dtype=torch.float32, device=x.device) if need_x else None
    gy = torch.empty((B, C, Hh, W), dtype=torch.float32, device=x.device) if need_y else None
    check(_lib.lib().segsde_ssim_map_backward(_p(x.contiguous()), _p(y.contiguous()), _p(_f32(gout.contiguous())), B, C, Hh, W,
                                              _p(gx), _p(gy), _stream(x)), "ssim_map_bwd")
    return gx, gy


def backproject_depth(depth, inv_K):
    """BackprojectDepth.forward (models/monodepth_layers.py:169-174): depth [B,1,H,W], inv_K [B,4,4] -> cam_points [B,4,H*W]"""
    B, _, Hh, W = depth.shape
    out = torch.empty((B, 4, Hh * W), dtype=torch.float32, device=depth.device)
    check(_lib.lib().segsde_backproject_depth(_p(_f32(depth.contiguous())), _p(_f32(inv_K.contiguous())), B, Hh, W, _p(out),
                                              _stream(depth)), "backproject_depth")
    return out


def project3d(points, K, T, Hh, W, eps=1e-7):
    """Project3D.forward (models/monodepth_layers.py:188-199): points [B,4,H*W] -> sampling grid [B,H,W,2]"""
    B = points.shape[0]
    assert tuple(points.shape) == (B, 4, Hh * W)
    out = torch.empty((B, Hh, W, 2), dtype=torch.float32, device=points.device)
    check(_lib.lib().segsde_project3d(_p(_f32(points.contiguous())), _p(_f32(K.contiguous())), _p(_f32(T.contiguous())), B, Hh, W,
                                      float(eps), _p(out), _stream(points)), "project3d")
    return out


def backproject_depth_backward(g_cam, inv_K, Hh, W):
    """adjoint of backproject_depth w.r.t. the depth: g_cam [B,4,H*W] -> [B,1,H,W]"""
    B = g_cam.shape[0]
    out = torch.empty((B, 1, Hh, W), dtype=torch.float32, device=g_cam.device)
    check(_lib.lib().segsde_backproject_depth_backward(_p(_f32(g_cam.contiguous())), _p(_f32(inv_K.contiguous())), B, Hh, W, _p(out),
                                                       _stream(g_cam)), "backproject_depth_backward")
    return out


def project3d_backward(points, K, T, g_pix, Hh, W, eps=1e-7, need_points=True, need_T=True):
    """adjoint of project3d: g_pix [B,H,W,2] -> (d points [B,4,H*W] or None, d T [B,4,4] or None)"""
    B = points.shape[0]
    L = _lib.lib()
    gp = torch.empty_like(points, memory_format=torch.contiguous_format) if need_points else None
    gT = torch.empty((B, 4, 4), dtype=torch.float32, device=points.device) if need_T else None
    nbytes = L.segsde_project3d_backward_workspace(B, Hh, W) if need_T else 0
    ws = _ws(nbytes, points) if need_T else None
    check(L.segsde_project3d_backward(_p(_f32(points.contiguous())), _p(_f32(K.contiguous())), _p(_f32(T.contiguous())),
                                      _p(_f32(g_pix.contiguous())), B, Hh, W, float(eps), _p(gp), _p(gT), _p(ws), nbytes,
                                      _stream(points)), "project3d_backward")
    return gp, gT


# ----------------------------------------------------------------------------------------------
# segmentation loss, mix, masks
# ----------------------------------------------------------------------------------------------
def cross_entropy_forward(logits_nhwc, target, ignore_index, class_weight=None, pixel_weights=None):
    M, C, ld = _rows(logits_nhwc)
    L = _lib.lib()
    out = torch.empty(2, dtype=torch.float32, device=logits_nhwc.device)
    nb = L.segsde_cross_entropy_workspace(M)
    ws_ = _ws(nb, logits_nhwc)
    assert target.dtype == torch.int64 and target.is_contiguous() and target.numel() == M
    _timed('hbm_ce_fwd', M * (4.0 * C + 8.0), logits_nhwc, lambda: check(L.segsde_cross_entropy_forward(_p(_f32(logits_nhwc)), ld, M, C, _p(target), int(ignore_index), _p(class_weight),
                                         _p(pixel_weights), _p(out), _p(ws_), nb, _stream(logits_nhwc)), "ce_fwd"))
    return out


def cross_entropy_backward(logits_nhwc, target, ignore_index, scale, class_weight=None, pixel_weights=None):
    M, C, ld = _rows(logits_nhwc)
    dl = torch.empty(logits_nhwc.shape, dtype=torch.float32, device=logits_nhwc.device)
    _timed('hbm_ce_bwd', M * (8.0 * C + 8.0), logits_nhwc, lambda: check(_lib.lib().segsde_cross_entropy_backward(_p(logits_nhwc), ld, M, C, _p(target), int(ignore_index),
                                                   _p(class_weight), _p(pixel_weights), _p(_f32(scale)), _p(dl), C,
                                                   _stream(logits_nhwc)), "ce_bwd"))
    return dl


def mix(mask, x):
    """x: [B,C,H,W] fp32, NCHW-contiguous or channels-last; mask [Bm,H,W] int64 / float32"""
    B, C, H, W = x.shape
    if not (x.stride(3) == 1 or x.stride(1) == 1):
        x = x.contiguous()
    out = torch.empty_strided(x.shape, x.stride(), dtype=torch.float32, device=x.device)
    if mask.dtype == torch.int64:
        is64 = 1
    elif mask.dtype == torch.float32:
        is64 = 0
    else:
        raise TypeError("mask must be int64 or float32")
    mask = mask.contiguous()
    sb = x.stride(0) if B > 1 else C * H * W
    _timed('hbm_mix', B * H * W * (12.0 * C + (8.0 if is64 else 4.0)), x, lambda: check(_lib.lib().segsde_mix(_p(mask), is64, mask.shape[0], _p(_f32(x)), B, C, H, W, int(sb), int(x.stride(1)),
                                int(x.stride(2)), int(x.stride(3)), _p(out), _stream(x)), "mix"))
    return out


def mix_labels(mask, target):
    B, H, W = target.shape
    out = torch.empty_like(target)
    check(_lib.lib().segsde_mix_labels(_p(mask.contiguous()), _p(target.contiguous()), B, H, W, _p(out), _stream(target)),
          "mix_labels")
    return out


def depthcomp_mask(depths, margin, fg_threshold):
    """fg_threshold: a float for every image, or a DEVICE tensor [B] (one threshold per image, train.py:592-599)"""
    B = depths.shape[0]
    HW = depths[0].numel()
    mask = torch.empty((B,) + tuple(depths.shape[-2:]), dtype=torch.int64, device=depths.device)
    per = None
    if torch.is_tensor(fg_threshold):
        per = _f32(fg_threshold.reshape(-1)).to(depths.device).contiguous()
        if per.numel() != B:
            raise ValueError("one foreground threshold per image: got %d for a batch of %d" % (per.numel(), B))
    check(_lib.lib().segsde_depthcomp_mask(_p(_f32(depths.contiguous())), B, HW, float(margin),
                                           0.0 if per is not None else float(fg_threshold), _p(per), _p(mask),
                                           _stream(depths)), "depthcomp_mask")
    return mask


def depth_threshold_mask(depth, t1, t2=0.0, two=False):
    depth = depth.contiguous()
    mask = torch.empty_like(depth)
    check(_lib.lib().segsde_depth_threshold_mask(_p(_f32(depth)), depth.numel(), float(t1), float(t2), int(two), _p(mask),
                                                 _stream(depth)), "depth_threshold_mask")
    return mask


def class_mask(pred, classes):
    pred, classes = pred.contiguous(), classes.contiguous().to(torch.int64)
    mask = torch.empty_like(pred)
    check(_lib.lib().segsde_class_mask(_p(pred), pred.numel(), _p(classes), classes.numel(), _p(mask), _stream(pred)),
          "class_mask")
    return mask


# ----------------------------------------------------------------------------------------------
# trainer-side callers
# ----------------------------------------------------------------------------------------------
MT_CHUNK = 65536


def multi_tensor_table(dsts, srcs):
    """device table of segsde_mt_chunk {dst, src, n} covering every (dst, src) tensor pair in chunks of <= 65536 floats"""
    rows = []
    for d, s in zip(dsts, srcs):
        assert d.dtype == torch.float32 and s.dtype == torch.float32 and d.is_contiguous() and s.is_contiguous()
        assert d.numel() == s.numel() and d.device == s.device
        n, dp, sp = d.numel(), d.data_ptr(), s.data_ptr()
        for o in range(0, n, MT_CHUNK):
            rows.append((dp + 4 * o, sp + 4 * o, min(MT_CHUNK, n - o)))
    if not rows:
        return None
    return torch.tensor(rows, dtype=torch.int64).to(dsts[0].device)


def multi_tensor_lerp(table, alpha, one_minus_alpha, like):
    """dst = alpha*dst + one_minus_alpha*src for every chunk of the table (one launch)"""
    check(_lib.lib().segsde_multi_tensor_lerp(_p(table), int(table.shape[0]), float(alpha), float(one_minus_alpha),
                                              _stream(like)), "multi_tensor_lerp")


def pseudo_label(prob, threshold, ignore_index, want_max=False, want_weight=True):
    """prob: [B,C,H,W] (NCHW, dense) -> (label int64 [B,H,W], count uint64-as-int64 [1], max_prob or None,
    pixel_weight [B,H,W] or None)"""
    prob = _f32(prob).contiguous()
    B, C, H, W = prob.shape
    label = torch.empty((B, H, W), dtype=torch.int64, device=prob.device)
    count = torch.empty(1, dtype=torch.int64, device=prob.device)
    maxp = torch.empty((B, H, W), dtype=torch.float32, device=prob.device) if want_max else None
    pw = torch.empty((B, H, W), dtype=torch.float32, device=prob.device) if want_weight else None
    check(_lib.lib().segsde_pseudo_label(_p(prob), B, C, H * W, float(threshold), int(ignore_index), _p(label), _p(maxp),
                                         _p(count), _p(pw), _stream(prob)), "pseudo_label")
    if pw is not None:
        pw._segsde_finite = True      # count / total: cannot be NaN (cross_entropy2d skips its host-side NaN check)
    return label, count, maxp, pw


def color_jitter(x, params, order):
    """x [B,3,H,W] dense NCHW; params [B,4] device tensor of (brightness, contrast, saturation, hue) factors; order: a
    permutation of (0, 1, 2, 3) -- see segsde_color_jitter"""
    x = _f32(x).contiguous()
    B, C, Hh, W = x.shape
    assert C == 3 and tuple(params.shape) == (B, 4)
    y = torch.empty_like(x)
    arr = (ctypes.c_int * 4)(*[int(o) for o in order])
    check(_lib.lib().segsde_color_jitter(_p(x), B, Hh * W, _p(_f32(params.contiguous())), arr, _p(y), _stream(x)), "color_jitter")
    return y


def gaussian_blur(x, wy, wx):
    """x [B,C,H,W] dense NCHW; wy / wx: 1-D device tensors of (odd many) normalised taps for the column / row pass"""
    x = _f32(x).contiguous()
    B, C, Hh, W = x.shape
    tmp, y = torch.empty_like(x), torch.empty_like(x)
    _timed('hbm_blur', 16.0 * x.numel(), x, lambda: check(_lib.lib().segsde_gaussian_blur(_p(x), B * C, Hh, W, _p(_f32(wy.contiguous())), wy.numel(), _p(_f32(wx.contiguous())),
                                          wx.numel(), _p(tmp), _p(y), _stream(x)), "gaussian_blur"))
    return y


def softmax_to_nchw(logits_nhwc):
    """class softmax of NHWC logits [B,H,W,C] -> dense NCHW probabilities [B,C,H,W] (train.py:666)"""
    B, Hh, W, C = logits_nhwc.shape
    out = torch.empty((B, C, Hh, W), dtype=torch.float32, device=logits_nhwc.device)
    _timed('hbm_softmax', 8.0 * B * Hh * W * C, logits_nhwc, lambda: check(_lib.lib().segsde_softmax_nhwc_to_nchw(_p(_f32(logits_nhwc)), nhwc_ld(logits_nhwc), B, Hh * W, C, _p(out),
                                                 _stream(logits_nhwc)), "softmax_nhwc_to_nchw"))
    return out


def onehot_select_(prob_nchw, onehot, is_labeled):
    """mix_use_gt (train.py:667-672), in place: ``prob[i] = onehot[i]`` for every sample with ``is_labeled[i]``.
    prob: [B,C,H,W] fp32 dense; onehot: [B,C,H,W] int64 / float32 / uint8 (the loader's planes); is_labeled: [B]
    bool / integer tensor (moved to the device as uint8, never read back)."""
    B, C, Hh, W = prob_nchw.shape
    if tuple(onehot.shape) != (B, C, Hh, W):
        raise ValueError("onehot_lbl %s does not match the teacher softmax %s" % (tuple(onehot.shape), tuple(prob_nchw.shape)))
    if not prob_nchw.is_contiguous() or prob_nchw.dtype != torch.float32:
        raise ValueError("onehot_select_ works in place on a dense fp32 NCHW tensor")
    code = {torch.float32: 0, torch.int64: 1, torch.uint8: 2, torch.bool: 2}.get(onehot.dtype)
    if code is None:
        onehot, code = onehot.to(torch.int64), 1
    onehot = onehot.to(prob_nchw.device).contiguous()
    flags = torch.as_tensor(is_labeled).reshape(-1).to(device=prob_nchw.device).ne(0).to(torch.uint8).contiguous()
    if flags.numel() != B:
        raise ValueError("is_labeled has %d entries for a batch of %d" % (flags.numel(), B))
    check(_lib.lib().segsde_onehot_select(_p(prob_nchw), _p(onehot), code, _p(flags), B, C, Hh * W, _stream(prob_nchw)),
          "onehot_select")
    return prob_nchw


def minmax_normalize(x, as_uint8=False):
    """x: [B, ...] -> (x - min_b) / (max_b - min_b) per sample b (train.py:690-697); as_uint8: the 8-bit image
    loader/depth_estimator.py:83-91 stores (mul(255).byte() of the normalised map) instead"""
    x = _f32(x).contiguous()
    B = x.shape[0]
    HW = x[0].numel()
    out = torch.empty(x.shape, dtype=torch.uint8 if as_uint8 else torch.float32, device=x.device)
    L = _lib.lib()
    nb = L.segsde_minmax_normalize_workspace(B, HW)
    ws_ = _ws(nb, x)
    check(L.segsde_minmax_normalize(_p(x), B, HW, None if as_uint8 else _p(out), None, _p(out) if as_uint8 else None,
                                    _p(ws_), nb, _stream(x)), "minmax_normalize")
    return out


def disp_to_depth_upsampled(disp, out_hw, min_depth, max_depth):
    """[B,1,hs,ws] disparity -> [B,1,H,W] depth (loss/monodepth_loss.py:54-62)"""
    disp = _f32(disp).contiguous()
    B, _, hs, ws = disp.shape
    Hh, W = out_hw
    depth = torch.empty((B, 1, Hh, W), dtype=torch.float32, device=disp.device)
    check(_lib.lib().segsde_disp_to_depth(_p(disp), hs, ws, B, Hh, W, float(min_depth), float(max_depth), _p(depth),
                                          _stream(disp)), "disp_to_depth")
    return depth


def confusion_update(hist, gt, pred=None, logits=None):
    """hist: int64 [C*C] on the device (accumulated in place); gt: int64 [B,H,W]; pred int64 [B,H,W] or logits [B,C,H,W]
    (NCHW-logical, any of dense NCHW / channels-last memory)."""
    gt = gt.contiguous()
    B = gt.shape[0]
    HW = gt.numel() // B
    C = int(round(hist.numel() ** 0.5))
    if logits is not None:
        lg = _f32(logits)
        assert lg.shape[1] == C and lg.shape[0] == B and lg.shape[2] * lg.shape[3] == HW
        sb, sc, sh, sw = lg.stride()
        if sh != lg.shape[3] * sw:      # rows must follow each other with the pixel stride
            lg = lg.contiguous()
            sb, sc, sh, sw = lg.stride()
        check(_lib.lib().segsde_confusion_update(_p(lg), sb, sc, sw, None, _p(gt), B, HW, C, _p(hist), _stream(gt)),
              "confusion_update")
    else:
        pred = pred.contiguous().to(torch.int64)
        check(_lib.lib().segsde_confusion_update(None, 0, 0, 0, _p(pred), _p(gt), B, HW, C, _p(hist), _stream(gt)),
              "confusion_update")
    return hist
