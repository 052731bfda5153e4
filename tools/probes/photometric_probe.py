"""The monodepth loss chain at the headline size (B = 16, 512x1024, the four disparity scales): warp, identity errors,
photometric forward (errors + auto-mask minimum), photometric backward (SSIM / L1 adjoint + warp adjoint), HIP events.
Run once per knob setting (SEGSDE_PHOTO_PACKED=0/1, SEGSDE_WARP_BLOCKS=512/2048): the knobs are read once per process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402
from improving_segmentation_with_selfsupervised_depth_amd.models.monodepth_layers import transformation_from_parameters  # noqa: E402

B, Hh, W, dev = 16, 512, 1024, "cuda"
gen = torch.Generator().manual_seed(5)
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * Hh, 0.5 * Hh, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]]).repeat(B, 1, 1)
iK, K = torch.linalg.pinv(K).to(dev), K.to(dev)
Ts = [transformation_from_parameters(0.01 * torch.randn(B, 1, 3, generator=gen).to(dev), 0.05 * torch.randn(B, 1, 3, generator=gen).to(dev),
                                     invert=(j == 0)).contiguous() for j in range(2)]


def frames(kind):
    if kind == "noise":      # bench.py's synthetic batch: independent uniform noise per frame
        return [torch.rand(B, 3, Hh, W, generator=gen).to(dev) for _ in range(3)]
    base = torch.rand(B, 3, Hh // 8, W // 8, generator=gen)   # smooth frames, the sources shifted by two pixels
    smooth = torch.nn.functional.interpolate(base, size=(Hh, W), mode="bilinear", align_corners=False)
    return [(0.8 * smooth + 0.2 * torch.rand(B, 3, Hh, W, generator=gen)).to(dev)] + \
           [(0.8 * torch.roll(smooth, (j * 4 - 2), 3) + 0.2 * torch.rand(B, 3, Hh, W, generator=gen)).to(dev) for j in range(2)]


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print("knobs: SEGSDE_PHOTO_PACKED=%s SEGSDE_PHOTO_SPLIT=%s SEGSDE_WARP_BLOCKS=%s" % (
    os.environ.get("SEGSDE_PHOTO_PACKED", "1 (default)"), os.environ.get("SEGSDE_PHOTO_SPLIT", "1 (default)"),
    os.environ.get("SEGSDE_WARP_BLOCKS", "2048 (default)")))
for kind, automask in (("noise", True), ("smooth", True), ("noise", False)):
    tgt, s0, s1 = frames(kind)
    srcs = [s0, s1]
    t_ident = timed(lambda: H.photometric_identity(srcs[0], srcs[1], tgt, False))
    ident = H.photometric_identity(srcs[0], srcs[1], tgt, False) if automask else None
    for s in (0, 2):
        hs, ws = Hh >> s, W >> s
        lo = torch.rand(B, 1, max(hs // 16, 1), max(ws // 16, 1), generator=gen)   # a smooth disparity map, as a network emits
        disp = (0.3 + 0.4 * torch.nn.functional.interpolate(lo, size=(hs, ws), mode="bilinear", align_corners=False)).to(dev)
        t_warp = timed(lambda: H.warp_forward(disp, iK, K, Ts[0], srcs[0], 0.1, 100.0))
        cols = [H.warp_forward(disp, iK, K, Ts[j], srcs[j], 0.1, 100.0)[0] for j in range(2)]
        noise = torch.randn(B, 2, Hh, W, device=dev) if automask else None
        t_fwd = timed(lambda: H.photometric_forward(cols[0], cols[1], tgt, ident, noise, False, False))
        ssum, sel, isel = H.photometric_forward(cols[0], cols[1], tgt, ident, noise, False, False)
        gT = [torch.zeros(B, 4, 4, device=dev) for _ in range(2)]
        t_bwd = timed(lambda: H.photometric_backward(cols[0], cols[1], tgt, sel, automask, disp, iK, K, Ts[0], Ts[1], srcs[0], srcs[1],
                                                     0.1, 100.0, False, False, 1.0 / (B * Hh * W), None, gT[0], gT[1]))
        frac = [float((sel == k).float().mean()) for k in range(4 if automask else 2)]
        print("%-6s frames, auto-mask %-5s scale %d: warp %6.1f us   identity %6.1f us   forward %6.1f us   backward %6.1f us   selected %s"
              % (kind, automask, s, t_warp, t_ident, t_fwd, t_bwd, " ".join("%.2f" % f for f in frac)), flush=True)
