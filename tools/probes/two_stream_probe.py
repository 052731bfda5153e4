"""Would running a convolution's weight gradient on a second stream next to its data gradient buy anything?
A chain of (dgrad, wgrad) pairs of one layer shape, timed (i) all on one stream, (ii) wgrads on a side stream that waits
for nothing but its own predecessor (independent inputs: an upper bound for the overlap).  MI355X, HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402

B, dev = 16, "cuda"
shapes = [("layer3 conv2 256->256 @32x64", 32, 64, 256, 256, 3, 1, 1, False),
          ("layer3 conv1 1x1 1024->256 @32x64", 32, 64, 1024, 256, 1, 1, 0, False),
          ("layer3 conv3 1x1 256->1024 @32x64", 32, 64, 256, 1024, 1, 1, 0, False),
          ("layer2 conv2 128->128 @64x128", 64, 128, 128, 128, 3, 1, 1, False),
          ("dec 128->128 refl @128x256", 128, 256, 128, 128, 3, 1, 1, True),
          ("dec 64->64 refl @512x1024", 512, 1024, 64, 64, 3, 1, 1, True)]
side = torch.cuda.Stream()
for name, Hh, W, C, Co, k, dil, pad, refl in shapes:
    g = H.ConvGeom(C, Co, k, 1, dil, pad, refl, 0, False)
    x = torch.randn(B, Hh, W, C, device=dev)
    w = torch.randn(Co, C, k, k, device=dev) * 0.05
    wp, wd = H.pack_weight_both(w)
    dy = torch.randn(B, Hh, W, Co, device=dev)
    n = 20

    def one_stream():
        for _ in range(n):
            H.conv_dgrad(g, dy, wd, w, (Hh, W))
            H.conv_wgrad(g, x, None, dy)

    def two_streams():
        side.wait_stream(torch.cuda.current_stream())
        for _ in range(n):
            H.conv_dgrad(g, dy, wd, w, (Hh, W))
            with torch.cuda.stream(side):
                H.conv_wgrad(g, x, None, dy)
        torch.cuda.current_stream().wait_stream(side)

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
    print("%-40s one stream %.3f %.3f ms per pair | two streams %.3f %.3f ms  (%.1f %%)" % (
        name, res[0], res[2], res[1], res[3], 100.0 * (1 - min(res[1], res[3]) / min(res[0], res[2]))), flush=True)
