"""phase costs of wino_fused_kernel: the same launches with parts of the kernel compiled out (tools/build_variant.sh variants, chosen
with SEGSDE_LIB); prints microseconds per launch.  Results of the variants are garbage by construction -- timing only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402

B, dev = 16, "cuda"
shapes = [("64->64 @128x256", 128, 256, 64, 64), ("128->128 @64x128", 64, 128, 128, 128), ("256->256 @32x64", 32, 64, 256, 256),
          ("128->64 @256x512", 256, 512, 128, 64), ("128->128 @128x256", 128, 256, 128, 128)]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


print("# lib:", os.environ.get("SEGSDE_LIB", "shipped"))
for name, Hh, W, C, Co in shapes:
    x = torch.randn(B, Hh, W, C, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
    uf = H.winograd_fused_pack(w, False)
    t_s = timed(lambda: H.winograd_fused("conv_fwd", x, uf, want_stats=True))
    t_n = timed(lambda: H.winograd_fused("conv_fwd", x, uf, want_stats=False))
    gf = 2.0 * B * Hh * W * C * Co * 16 / 4 / 1e6       # executed: 16 multiply-adds per 2x2 outputs
    mfma_us = gf / 157.3
    print("%-22s fwd+stats %7.1f us  fwd %7.1f us   (pure MFMA time at peak %6.1f us -> %.3f of the pipe)" % (name, t_s, t_n, mfma_us, mfma_us / t_n), flush=True)
