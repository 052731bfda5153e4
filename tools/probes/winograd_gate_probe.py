"""Gate experiment for Winograd on the fp32 matrix pipe: an UPPER BOUND of what F(2,3) / F(2x2,3x3) can
reach for stride-1 3x3 convolutions with >= 128 channels, measured with the kernels that exist.

Winograd turns the 3x3 convolution into independent GEMMs over the channels, one per transform position, between an input
transform (B^T d B) and an output transform (A^T m A):
  * 1-D F(2,3) along W: 4 positions, each a 3x1 convolution (the three tap rows stay a direct sum) over W/2 pixel pairs:
    12 C multiply-adds per pixel pair and output channel instead of 18 C -> 1.5x fewer;
  * 2-D F(2x2,3x3): 16 positions, each a 1x1 convolution over (H/2)(W/2) blocks: 16 C instead of 36 C -> 2.25x fewer.
The position GEMMs are timed here as ONE launch of conv_igemm_kernel with the positions stacked along the batch axis (same
tile count and reduction length as a grouped launch would have; identical weights for all positions do not change the timing),
the transforms as the HBM traffic they cannot avoid when they run as separate passes (measured copy kernels of the same
byte counts).  A fused kernel could hide the transforms but not the position GEMMs' shorter reductions (K = 3C or C instead
of 9C), which is where this matrix pipe loses its efficiency (~20 us of fixed cost per tile round)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from improving_segmentation_with_selfsupervised_depth_amd import _lib, hipops as H  # noqa: E402
from improving_segmentation_with_selfsupervised_depth_amd._lib import ConvDesc  # noqa: E402

dev = "cuda"


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


def raw_conv(x, wpack, Cout, KH, KW, pad):
    B, Hh, W, C = x.shape
    y = torch.empty(B, Hh, W, Cout, device=dev)
    d = ConvDesc(B=B, H=Hh, W=W, C0=C, C1=0, ld0=C, ld1=0, up0=0, Ho=Hh, Wo=W, Cout=Cout, ldy=Cout, ldy2=0, nsplit=0, KH=KH, KW=KW,
                 stride=1, dil=1, pad=pad, pad_mode=0, in_div=1, act=0, sum2x2=0)
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    return lambda: H.check(L.segsde_conv2d_forward(ctypes.byref(d), H._p(x), None, H._p(wpack), None, H._p(y), None, st), "conv")


B = 16
print("%-34s %9s | %-30s | %-30s" % ("layer (B = 16)", "direct", "1-D F(2,3): 4 x (3x1, K = 3C)", "2-D F(2x2,3x3): 16 x (1x1, K = C)"))
for name, Hh, W, C in (("256->256 @32x64 (layer3 conv2)", 32, 64, 256), ("128->128 @64x128 (layer2 conv2)", 64, 128, 128),
                       ("128->128 @128x256 (decoder)", 128, 256, 128), ("512->512 @32x64 (layer4 conv2, d1)", 32, 64, 512)):
    g = H.ConvGeom(C, C, 3, 1, 1, 1, False, 0, False)
    x = torch.randn(B, Hh, W, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    wp = H.pack_weight(w)
    t_direct = timeit(lambda: H.conv_forward(g, x, None, wp, None))
    # 1-D: 4 positions stacked along the batch, W/2 pixel pairs, 3x1 taps
    x1 = torch.randn(4 * B, Hh, W // 2, C, device=dev)
    w1 = torch.randn(C, 3, 1, C, device=dev) * 0.05            # packed [O][KH][KW][I]
    t_g1 = timeit(raw_conv(x1, w1, C, 3, 1, 1))
    # 2-D: 16 positions stacked along the batch, (H/2)(W/2) blocks, 1x1
    x2 = torch.randn(16 * B, Hh // 2, W // 2, C, device=dev)
    w2 = torch.randn(C, 1, 1, C, device=dev) * 0.05
    t_g2 = timeit(raw_conv(x2, w2, C, 1, 1, 0))
    # transforms as separate passes: input x -> 2x (1-D) / 4x (2-D) its size, output 2x / 4x -> 1x; copy kernels of those byte counts
    nb = x.numel()
    src1, dst1 = torch.empty(3 * nb, device=dev), torch.empty(3 * nb, device=dev)
    t_tr1 = timeit(lambda: dst1[:nb * 3 // 2].copy_(src1[:nb * 3 // 2]))         # read 1 + write 2 = 3 units -> copy of 1.5 units
    src2, dst2 = torch.empty(5 * nb, device=dev), torch.empty(5 * nb, device=dev)
    t_tr2 = timeit(lambda: dst2[:nb * 5 // 2].copy_(src2[:nb * 5 // 2]))         # read 1 + write 4 = 5 units -> copy of 2.5 units
    print("%-34s %7.1f us | GEMMs %6.1f us, in/out transforms 2 x %5.1f us: fused %4.2fx, unfused %4.2fx | "
          "GEMMs %6.1f us, 2 x %5.1f us: fused %4.2fx, unfused %4.2fx" % (
              name, t_direct, t_g1, t_tr1, t_direct / t_g1, t_direct / (t_g1 + 2 * t_tr1),
              t_g2, t_tr2, t_direct / t_g2, t_direct / (t_g2 + 2 * t_tr2)), flush=True)
    del x, x1, x2, src1, dst1, src2, dst2
