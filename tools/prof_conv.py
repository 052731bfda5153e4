"""A handful of representative conv launches for rocprofv3 --pmc runs (one launch each after a warm-up)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402

B = int(os.environ.get("BENCH_B", "8"))
shapes = [
    ("up0_1", 512, 1024, 64, 0, True, 64, 3, 1, 1, 1, True),
    ("up1_1", 256, 512, 128, 64, True, 128, 3, 1, 1, 1, True),
    ("l3c2", 32, 64, 256, 0, False, 256, 3, 1, 1, 1, False),
    ("aspp", 32, 64, 2048, 0, False, 256, 3, 1, 12, 12, False),
]
for (name, Hh, W, C0, C1, up0, Cout, k, stride, dil, pad, refl) in shapes:
    g = H.ConvGeom(C0, Cout, k, stride, dil, pad, refl, C1, up0)
    H0, W0 = (Hh // 2, W // 2) if up0 else (Hh, W)
    x0 = torch.randn(B, H0, W0, C0, device="cuda")
    x1 = torch.randn(B, Hh, W, C1, device="cuda") if C1 else None
    w = torch.randn(Cout, C0 + C1, k, k, device="cuda") * 0.05
    wp, wd = H.pack_weight(w), H.pack_weight(w, True)
    for _ in range(2):
        y = H.conv_forward(g, x0, x1, wp, None)
        dy = torch.randn_like(y)
        H.conv_dgrad(g, dy, wd, w, (Hh, W))
        H.conv_wgrad(g, x0, x1, dy)
    torch.cuda.synchronize()
print("done")
