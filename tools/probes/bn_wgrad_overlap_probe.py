"""Does an HBM-bound kernel (BatchNorm backward) overlap with an MFMA-bound one (a convolution's weight gradient) when the two
are issued on different streams?  probe_r02_two_stream_wgrad.log measured dgrad || wgrad (matrix pipe against matrix pipe:
+1..2 %); the backward pass of a residual block is (bn_backward -> dgrad -> wgrad) per convolution, and only dgrad and
wgrad want the matrix pipe.  Chain of n such triples of one layer shape, (i) all on one stream, (ii) the weight gradients on
a side stream that waits for its dy (an event after the bn_backward that produces it), as the real backward would.
MI355X, HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402

B, dev = 16, "cuda"
shapes = [("layer3 conv2 256->256 k3 @32x64", 32, 64, 256, 256, 3, 1, 1, False),
          ("layer3 conv1 1024->256 k1 @32x64", 32, 64, 1024, 256, 1, 1, 0, False),
          ("layer3 conv3 256->1024 k1 @32x64", 32, 64, 256, 1024, 1, 1, 0, False),
          ("layer1 conv2 64->64 k3 @128x256", 128, 256, 64, 64, 3, 1, 1, False),
          ("layer2 conv2 128->128 k3 @64x128", 64, 128, 128, 128, 3, 1, 1, False)]
side = torch.cuda.Stream()
for name, Hh, W, C, Co, k, dil, pad, refl in shapes:
    g = H.ConvGeom(C, Co, k, 1, dil, pad, refl, 0, False)
    x = torch.randn(B, Hh, W, C, device=dev)
    w = torch.randn(Co, C, k, k, device=dev) * 0.05
    wp, wd = H.pack_weight_both(w)
    z = torch.randn(B, Hh, W, Co, device=dev)          # conv output = BatchNorm input
    gy = torch.randn(B, Hh, W, Co, device=dev)         # gradient arriving at the BatchNorm output
    mean, invstd = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    gamma, beta = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    n = 20

    def bn():
        return H.bn_backward(gy, None, z, mean, invstd, gamma, act="relu", beta=beta)[0]

    def one_stream():
        for _ in range(n):
            dy = bn()
            H.conv_dgrad(g, dy, wd, w, (Hh, W))
            H.conv_wgrad(g, x, None, dy)

    def two_streams():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        for _ in range(n):
            dy = bn()
            ev = torch.cuda.Event()
            ev.record(main)
            H.conv_dgrad(g, dy, wd, w, (Hh, W))
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dy.record_stream(side)
                H.conv_wgrad(g, x, None, dy)
        main.wait_stream(side)

    def parts():
        out = []
        for fn in (bn, lambda: H.conv_dgrad(g, gy, wd, w, (Hh, W)), lambda: H.conv_wgrad(g, x, None, gy)):
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                fn()
            e.record()
            torch.cuda.synchronize()
            out.append(s.elapsed_time(e) / n)
        return out

    res = []
    for fn in (one_stream, two_streams, one_stream, two_streams):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / n)
    p = parts()
    print("%-36s bn %.3f dgrad %.3f wgrad %.3f | one stream %.3f %.3f ms per triple | two streams %.3f %.3f ms  (%.1f %%)" % (
        name, p[0], p[1], p[2], res[0], res[2], res[1], res[3], 100.0 * (1 - min(res[1], res[3]) / min(res[0], res[2]))),
        flush=True)
