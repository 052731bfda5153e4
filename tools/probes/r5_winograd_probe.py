"""Round 5 probe (B = 16, cfg3's decoder / encoder geometries): the one-kernel Winograd route's new directions against the routes
they replace -- (a) data-gradient of the mirrored Conv3x3 with the ELU derivative in the epilogue (direct reflection-adjoint
kernel), (b) forward on [upsample(x0) | x1] (upsample-folded direct route), (c) the weight gradient (direct / folded / grouped
Winograd).  `python r5_winograd_probe.py [dgrad] [fwd2] [wgrad]`; the weight gradient's staging variant is an environment knob
read once per process (SEGSDE_WGRAD_FUSED_VAR = 0 / 1 / 2): run once per variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402

B, dev = int(os.environ.get("PROBE_B", "16")), "cuda"
what = sys.argv[1:] or ["dgrad", "fwd2", "wgrad"]
H.WINOGRAD_MIN_MACS = 0.0


def timed(fn, n=10):
    for _ in range(2):
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


def rnd(*shape):
    return torch.randn(*shape, device=dev)


if "dgrad" in what:
    print("# (a) data-gradient, mirrored padding, ELU derivative in the epilogue: direct reflection-adjoint kernel vs one-kernel Winograd + border launches")
    for name, Hh, W, C, Co in [("128->64 @256x512", 256, 512, 128, 64), ("128->128 @128x256", 128, 256, 128, 128),
                               ("256->128 @64x128", 64, 128, 256, 128), ("256->256 @32x64", 32, 64, 256, 256),
                               ("64->64 @128x256 zero-pad (layer1)", 128, 256, 64, 64)]:
        refl = "zero-pad" not in name
        g = H.ConvGeom(C, Co, 3, 1, 1, 1, refl, 0, False)
        dy, a = rnd(B, Hh, W, Co), torch.nn.functional.elu(rnd(B, Hh, W, C))
        w = rnd(Co, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
        wfp, wdp = H.pack_weight_both(w)
        ud = H.winograd_fused_pack(w, True)
        ag = (a, "elu")
        t_d = timed(lambda: H.conv_dgrad(g, dy, wdp, w, (Hh, W), actgrad=ag))
        t_f1 = timed(lambda: H.conv_dgrad(g, dy, wdp, w, (Hh, W), actgrad=ag, wino=ud))      # border terms: implicit-GEMM border launches
        t_f = timed(lambda: H.conv_dgrad(g, dy, wdp, w, (Hh, W), actgrad=ag, wino=ud, wfpack=wfp))   # border kernel (two launches)
        H.WINO_FUSED_DGRAD_EXT = False
        t_fz = timed(lambda: H.winograd_fused("conv_dgrad", dy, ud))      # the zero-padded launch alone: what the borders + epilogue cost
        H.WINO_FUSED_DGRAD_EXT = True
        d1, _ = H.conv_dgrad(g, dy, wdp, w, (Hh, W), actgrad=ag)
        d2, _ = H.conv_dgrad(g, dy, wdp, w, (Hh, W), actgrad=ag, wino=ud, wfpack=wfp)
        err = float((d1 - d2).abs().max()) / float(d1.abs().max())
        gf = 2.0 * B * Hh * W * C * Co * 9 / 1e6
        print("%-36s direct %7.1f us (%5.1f TF)  fused %7.1f us (%5.1f TF alg, %.2fx; with implicit-GEMM border launches %7.1f us; zero-padded launch alone, no derivative %7.1f us)  rel. diff %.1e"
              % (name, t_d, gf / t_d, t_f, gf / t_f, t_d / t_f, t_f1, t_fz, err), flush=True)

if "fwd2" in what:
    print("# (b) forward on [upsample(x0) | x1], mirrored padding, bias + ELU: upsample-folded direct route vs one-kernel Winograd")
    H.WINO_FUSED2_MIN_FOLD = 0.0
    for name, Hh, W, C0, C1, Co in [("128+64->128 @256x512", 256, 512, 128, 64, 128), ("128+256->128 @128x256", 128, 256, 128, 256, 128),
                                    ("256+512->256 @64x128", 64, 128, 256, 512, 256), ("64+0->64 @512x1024", 512, 1024, 64, 0, 64)]:
        g = H.ConvGeom(C0, Co, 3, 1, 1, 1, True, C1, True)
        x0, x1 = rnd(B, Hh // 2, W // 2, C0), (rnd(B, Hh, W, C1) if C1 else None)
        w = rnd(Co, C0 + C1, 3, 3) * (2.0 / (9 * (C0 + C1))) ** 0.5
        bias = rnd(Co)
        wp = H.pack_weight(w)
        wf, _ = H.upfold_pack(w, C0)
        uf = H.winograd_fused_pack(w, False)
        t_d = timed(lambda: H.conv_forward(g, x0, x1, wp, bias, act="elu", wfold=wf))
        t_f = timed(lambda: H.conv_forward(g, x0, x1, wp, bias, act="elu", wino=uf))
        y1 = H.conv_forward(g, x0, x1, wp, bias, act="elu", wfold=wf)
        y2 = H.conv_forward(g, x0, x1, wp, bias, act="elu", wino=uf)
        err = float((y1 - y2).abs().max()) / float(y1.abs().max())
        gf = 2.0 * B * Hh * W * (C0 + C1) * Co * 9 / 1e6
        print("%-36s folded %7.1f us (%5.1f TF alg)  fused %7.1f us (%5.1f TF alg, %.2fx)  rel. diff %.1e"
              % (name, t_d, gf / t_d, t_f, gf / t_f, t_d / t_f, err), flush=True)

if "wgrad" in what:
    print("# (c) weight gradient: the route of round 4 (direct / folded / grouped Winograd) vs the one-kernel Winograd scheme, variant %s, %s workgroups"
          % (os.environ.get("SEGSDE_WGRAD_FUSED_VAR", "1"), os.environ.get("SEGSDE_WGRAD_FUSED_WGS", "512")))
    H.WINO_FUSED_WGRAD_MIN_FOLD = 0.0
    for name, Hh, W, C0, C1, Co, up, refl in [
            ("64->64 @128x256 (layer1)", 128, 256, 64, 0, 64, False, False), ("128->128 @64x128 (layer2)", 64, 128, 128, 0, 128, False, False),
            ("256->256 @32x64 (layer3)", 32, 64, 256, 0, 256, False, False), ("128->64 @256x512 refl", 256, 512, 128, 0, 64, False, True),
            ("128->128 @128x256 refl", 128, 256, 128, 0, 128, False, True), ("256->128 @64x128 refl", 64, 128, 256, 0, 128, False, True),
            ("128+64->128 @256x512 up refl", 256, 512, 128, 64, 128, True, True), ("128+256->128 @128x256 up refl", 128, 256, 128, 256, 128, True, True),
            ("256+512->256 @64x128 up refl", 64, 128, 256, 512, 256, True, True), ("64+0->64 @512x1024 up refl", 512, 1024, 64, 0, 64, True, True)]:
        g = H.ConvGeom(C0, Co, 3, 1, 1, 1, refl, C1, up)
        x0 = rnd(B, Hh // 2 if up else Hh, W // 2 if up else W, C0)
        x1 = rnd(B, Hh, W, C1) if C1 else None
        dy = rnd(B, Hh, W, Co)
        H.WINO_FUSED_WGRAD = False
        t_d = timed(lambda: H.conv_wgrad(g, x0, x1, dy))
        d1 = H.conv_wgrad(g, x0, x1, dy)
        H.WINO_FUSED_WGRAD = True
        n0 = H.WINO_FUSED_TAKEN["wgrad"]
        t_f = timed(lambda: H.conv_wgrad(g, x0, x1, dy))
        d2 = H.conv_wgrad(g, x0, x1, dy)
        took = H.WINO_FUSED_TAKEN["wgrad"] > n0
        err = float((d1 - d2).abs().max()) / float(d1.abs().max())
        gf = 2.0 * B * Hh * W * (C0 + C1) * Co * 9 / 1e6
        print("%-36s round-4 route %7.1f us (%5.1f TF alg)  fused %7.1f us (%5.1f TF alg, %5.1f executed, %.2fx)%s  rel. diff %.1e"
              % (name, t_d, gf / t_d, t_f, gf / t_f, gf / t_f * 16 / 36, t_d / t_f, "" if took else "  [DECLINED]", err), flush=True)
