"""What does the BatchNorm-statistics epilogue (double-precision column sums of the output tile) cost a forward launch?
conv_forward with and without want_stats on the encoder's layer shapes, B = 16, HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402

B, dev = 16, "cuda"
shapes = [("layer3 conv3 256->1024 k1 @32x64", 32, 64, 256, 1024, 1, 0, 23),
          ("layer3 conv1 1024->256 k1 @32x64", 32, 64, 1024, 256, 1, 0, 22),
          ("layer3 conv2 256->256 k3 @32x64", 32, 64, 256, 256, 3, 1, 23),
          ("layer1 conv2 64->64 k3 @128x256", 128, 256, 64, 64, 3, 1, 11),
          ("layer1 conv3 64->256 k1 @128x256", 128, 256, 64, 256, 1, 0, 4),
          ("layer1 conv1 256->64 k1 @128x256", 128, 256, 256, 64, 1, 0, 2),
          ("layer2 conv2 128->128 k3 @64x128", 64, 128, 128, 128, 3, 1, 9),
          ("layer2 conv3 128->512 k1 @64x128", 64, 128, 128, 512, 1, 0, 4),
          ("layer4 conv2 512->512 k3 d2 @32x64", 32, 64, 512, 512, 3, 2, 2),
          ("aspp 2048->256 k1 @32x64", 32, 64, 2048, 256, 1, 0, 2)]
tot = [0.0, 0.0]
for name, Hh, W, C, Co, k, pad, per_step in shapes:
    dil = 2 if "d2" in name else 1
    g = H.ConvGeom(C, Co, k, 1, dil, pad * dil if k == 3 else 0, False, 0, False)
    x = torch.randn(B, Hh, W, C, device=dev)
    w = torch.randn(Co, C, k, k, device=dev) * 0.05
    wp = H.pack_weight(w)
    res = []
    for stats in (False, True, False, True):
        for _ in range(3):
            H.conv_forward(g, x, None, wp, None, want_stats=stats)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        s.record()
        for _ in range(n):
            H.conv_forward(g, x, None, wp, None, want_stats=stats)
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / n * 1e3)
    plain, st = min(res[0], res[2]), min(res[1], res[3])
    tot[0] += plain * per_step
    tot[1] += st * per_step
    print("%-38s plain %7.1f us   with statistics %7.1f us   (+%4.1f %%)  x %d per step" % (name, plain, st, 100 * (st / plain - 1), per_step), flush=True)
print("per step over these layers: plain %.2f ms, with statistics %.2f ms" % (tot[0] / 1e3, tot[1] / 1e3))
