"""The Winograd route as built (segsde_conv2d_winograd: transforms + grouped position GEMMs) against the direct implicit GEMM
on the layers that take it, forward (with BatchNorm statistics, as the encoder runs it) and data-gradient; error of both against
a float64 convolution on one image.  B = 16, HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402

dev, B = "cuda", 16
H.WINOGRAD_MIN_CH, H.WINOGRAD_MIN_MACS = 64, 0.0


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, Hh, W, C in (("256->256 @32x64 (R101 layer3 conv2)", 32, 64, 256), ("512->512 @32x64 (layer4.0 conv2)", 32, 64, 512),
                       ("512->512 @16x32 (pose R18 layer4)", 16, 32, 512), ("256->256 @16x32 (pose R18 layer3)", 16, 32, 256),
                       ("128->128 @64x128 (layer2 conv2)", 64, 128, 128), ("128->128 @128x256 (decoder)", 128, 256, 128),
                       ("64->64 @128x256 (layer1 conv2)", 128, 256, 64)):
    g = H.ConvGeom(C, C, 3, 1, 1, 1, False, 0, False)
    x = torch.randn(B, Hh, W, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
    dy = torch.randn(B, Hh, W, C, device=dev)
    wp, wd = H.pack_weight_both(w)
    uf, ud = H.winograd_pack(w)
    t_pack = timeit(lambda: H.winograd_pack(w))
    t_fd = timeit(lambda: H.conv_forward(g, x, None, wp, None, want_stats=True))
    t_fw = timeit(lambda: H.conv_forward(g, x, None, wp, None, want_stats=True, wino=uf))
    t_dd = timeit(lambda: H.conv_dgrad(g, dy, wd, w, (Hh, W)))
    t_dw = timeit(lambda: H.conv_dgrad(g, dy, wd, w, (Hh, W), wino=ud))
    yd = H.conv_forward(g, x, None, wp, None)
    yw = H.conv_forward(g, x, None, wp, None, wino=uf)
    want = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), padding=1).permute(0, 2, 3, 1)
    sc = float(want.abs().max())
    e_d, e_w = float((yd[:1].double().cpu() - want).abs().max()) / sc, float((yw[:1].double().cpu() - want).abs().max()) / sc
    r_d = float((yd[:1].double().cpu() - want).pow(2).mean().sqrt()) / sc
    r_w = float((yw[:1].double().cpu() - want).pow(2).mean().sqrt()) / sc
    print("%-38s fwd direct %7.1f us  winograd %7.1f us (%4.2fx) | dgrad direct %7.1f us  winograd %7.1f us (%4.2fx) | weight transform %5.1f us | "
          "error vs float64 / max: direct max %.1e rms %.1e, winograd max %.1e rms %.1e (x%.2f, x%.2f)" % (
              name, t_fd, t_fw, t_fd / t_fw, t_dd, t_dw, t_dd / t_dw, t_pack, e_d, r_d, e_w, r_w, e_w / e_d, r_w / r_d), flush=True)
