"""Winograd F(2x2,3x3) with the transforms inside the kernel (csrc/winograd_fused.hip) against the direct implicit GEMM on the
64 / 128-channel 3x3 layers (B = 16): forward with the BatchNorm statistics, data-gradient; error against float64 on a crop."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402

B, dev = 16, "cuda"
shapes = [("64->64 @128x256 (layer1 conv2)", 128, 256, 64, 64), ("128->128 @64x128 (layer2 conv2)", 64, 128, 128, 128),
          ("128->128 @128x256", 128, 256, 128, 128), ("64->64 @256x512", 256, 512, 64, 64), ("128->64 @256x512", 256, 512, 128, 64),
          ("256->256 @32x64 (layer3 conv2)", 32, 64, 256, 256), ("512->512 @32x64 (layer4.0 conv2)", 32, 64, 512, 512),
          ("256->128 @64x128", 64, 128, 256, 128), ("64->64 @512x1024", 512, 1024, 64, 64)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


for name, Hh, W, C, Co in shapes:
    g = H.ConvGeom(C, Co, 3, 1, 1, 1, False, 0, False)
    x = torch.randn(B, Hh, W, C, device=dev)
    dy = torch.randn(B, Hh, W, Co, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
    wp, wdp = H.pack_weight_both(w)
    uf, ud = H.winograd_fused_pack(w, False), H.winograd_fused_pack(w, True)
    t_d = timed(lambda: H.conv_forward(g, x, None, wp, None, want_stats=True))
    t_f = timed(lambda: H.winograd_fused("conv_fwd", x, uf, want_stats=True))
    t_dd = timed(lambda: H.conv_dgrad(g, dy, wdp, w, (Hh, W)))
    t_fd = timed(lambda: H.winograd_fused("conv_dgrad", dy, ud))
    yd, _ = H.conv_forward(g, x, None, wp, None, want_stats=True)
    yf, _ = H.winograd_fused("conv_fwd", x, uf, want_stats=True)
    want = torch.nn.functional.conv2d(x[:2].permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    e_d, e_f = float((yd[:2].double() - want).abs().max()), float((yf[:2].double() - want).abs().max())
    gf = 2.0 * B * Hh * W * C * Co * 9 / 1e6
    print("%-34s fwd direct %7.1f us (%5.1f TF)  fused %7.1f us (%5.1f TF alg, %.2fx) | dgrad direct %7.1f us  fused %7.1f us (%.2fx) | max error vs float64: direct %.1e fused %.1e"
          % (name, t_d, gf / t_d, t_f, gf / t_f, t_d / t_f, t_dd, t_fd, t_dd / t_fd, e_d, e_f), flush=True)
