"""Half-precision operand mode of the implicit-GEMM convolution (segsde_conv_desc.compute = 1: v_mfma_f32_32x32x16_f16, fp32
accumulation) against the fp32 kernel and the routes the fp32 step takes: time per launch and error against float64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402

B, dev = 16, "cuda"
shapes = [("256->1024 k1 @32x64", 32, 64, 256, 1024, 1, 1, 0, False), ("1024->256 k1 @32x64", 32, 64, 1024, 256, 1, 1, 0, False),
          ("256->256 k3 @32x64", 32, 64, 256, 256, 3, 1, 1, False), ("2048->256 k3 d6 @32x64", 32, 64, 2048, 256, 3, 6, 6, False),
          ("128->64 k3 refl @256x512", 256, 512, 128, 64, 3, 1, 1, True), ("64->64 k3 @128x256", 128, 256, 64, 64, 3, 1, 1, False),
          ("64->256 k1 @128x256", 128, 256, 64, 256, 1, 1, 0, False)]


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


for name, Hh, W, C, Co, k, dil, pad, refl in shapes:
    x = torch.relu(torch.randn(B, Hh, W, C, device=dev))
    dy = torch.randn(B, Hh, W, Co, device=dev)
    w = torch.randn(Co, C, k, k, device=dev) * (2.0 / (k * k * C)) ** 0.5
    wp, wdp = H.pack_weight_both(w)
    res = {}
    for mode in (0, 1):
        H.COMPUTE_F16[0] = bool(mode)
        g = H.ConvGeom(C, Co, k, 1, dil, pad, refl, 0, False)
        H.COMPUTE_F16[0] = False
        y = H.conv_forward(g, x, None, wp, None)
        dx, _ = H.conv_dgrad(g, dy, wdp, w, (Hh, W))
        dw = H.conv_wgrad(g, x, None, dy)
        t_f = timed(lambda: H.conv_forward(g, x, None, wp, None))
        t_d = timed(lambda: H.conv_dgrad(g, dy, wdp, w, (Hh, W)))
        t_w = timed(lambda: H.conv_wgrad(g, x, None, dy))
        res[mode] = (y, dx, dw, t_f, t_d, t_w)
    nb = 2
    xr = x[:nb].permute(0, 3, 1, 2).double()
    if refl:
        xr = torch.nn.functional.pad(xr, (1, 1, 1, 1), mode="reflect")
        want = torch.nn.functional.conv2d(xr, w.double()).permute(0, 2, 3, 1)
    else:
        want = torch.nn.functional.conv2d(xr, w.double(), padding=pad, dilation=dil).permute(0, 2, 3, 1)
    sc = float(want.abs().max())
    e0 = float((res[0][0][:nb].double() - want).abs().max()) / sc
    e1 = float((res[1][0][:nb].double() - want).abs().max()) / sc
    rel_dx = float((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max())
    rel_dw = float((res[1][2] - res[0][2]).abs().max() / res[0][2].abs().max())
    gf = 2.0 * B * Hh * W * C * Co * k * k / 1e6
    print("%-26s fwd fp32 %7.1f us  f16 %7.1f us (%.2fx, %5.0f TF)  dgrad %7.1f -> %7.1f (%.2fx)  wgrad %7.1f -> %7.1f | fwd err vs f64 (of max): fp32 %.1e f16 %.1e; dgrad f16 vs fp32 %.1e, wgrad %.1e"
          % (name, res[0][3], res[1][3], res[0][3] / res[1][3], gf / res[1][3], res[0][4], res[1][4], res[0][4] / res[1][4], res[0][5], res[1][5], e0, e1, rel_dx, rel_dw), flush=True)
