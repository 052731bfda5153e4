"""Micro-benchmark of the hot kernels on the GPU box (HIP-event timed on the launch stream)."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = int(os.environ.get("BENCH_B", "4"))
    dev = "cuda"
    shapes = [  # name, H, W, C0, C1, up0, Cout, k, stride, dil, pad, reflect
        ("dec_up0_1 refl 64->64 @512x1024 (up)", 512, 1024, 64, 0, True, 64, 3, 1, 1, 1, True),
        ("dec_up1_1 refl 128up+64->128 @256x512", 256, 512, 128, 64, True, 128, 3, 1, 1, 1, True),
        ("dec_up1_0 refl 128->128 @128x256", 128, 256, 128, 0, False, 128, 3, 1, 1, 1, True),
        ("dec_up2_1 refl 128up+256->128 @128x256", 128, 256, 128, 256, True, 128, 3, 1, 1, 1, True),
        ("dec_up3_1 refl 256up+512->256 @64x128", 64, 128, 256, 512, True, 256, 3, 1, 1, 1, True),
        ("dec_up4_1 refl 256+1024->256 @32x64", 32, 64, 256, 1024, False, 256, 3, 1, 1, 1, True),
        ("aspp d12 2048->256 @32x64", 32, 64, 2048, 0, False, 256, 3, 1, 12, 12, False),
        ("layer3 conv2 256->256 @32x64", 32, 64, 256, 0, False, 256, 3, 1, 1, 1, False),
        ("layer1 conv2 64->64 @128x256", 128, 256, 64, 0, False, 64, 3, 1, 1, 1, False),
        ("layer3 conv3 1x1 256->1024 @32x64", 32, 64, 256, 0, False, 1024, 1, 1, 1, 0, False),
        ("layer3 conv1 1x1 1024->256 @32x64", 32, 64, 1024, 0, False, 256, 1, 1, 1, 0, False),
        ("layer1 conv1 1x1 256->64 @128x256", 128, 256, 256, 0, False, 64, 1, 1, 1, 0, False),
        ("probe: 3x3 512->64 @128x256 (N=64 tile, long K)", 128, 256, 512, 0, False, 64, 3, 1, 1, 1, False),
        ("probe: 3x3 64->64 @512x1024 zero pad no up", 512, 1024, 64, 0, False, 64, 3, 1, 1, 1, False),
        ("probe: 3x3 128->128 @128x256 zero pad (dec_up1_0 without reflection)", 128, 256, 128, 0, False, 128, 3, 1, 1, 1, False),
        ("probe: 1x1 4096->128 @128x256 (long K)", 128, 256, 4096, 0, False, 128, 1, 1, 1, 0, False),
        ("probe: 1x1 32->128 @512x1024 (short K)", 512, 1024, 32, 0, False, 128, 1, 1, 1, 0, False),
        ("layer2 conv2 3x3s2 128->128 @128x256", 128, 256, 128, 0, False, 128, 3, 2, 1, 1, False),
        ("layer2 down 1x1s2 256->512 @128x256", 128, 256, 256, 0, False, 512, 1, 2, 1, 0, False),
        ("stem4 7x7s2 4->64 @512x1024", 512, 1024, 4, 0, False, 64, 7, 2, 1, 3, False),
        ("stem8 7x7s2 8->64 @512x1024", 512, 1024, 8, 0, False, 64, 7, 2, 1, 3, False),
        ("stem 7x7s2 3->64 @512x1024", 512, 1024, 3, 0, False, 64, 7, 2, 1, 3, False),
        ("seg head 1x1 64->19 @512x1024", 512, 1024, 64, 0, False, 19, 1, 1, 1, 0, False),
        ("dispconv refl 64->1 @512x1024", 512, 1024, 64, 0, False, 1, 3, 1, 1, 1, True),
    ]
    rows = []
    flt = [f for f in os.environ.get("BENCH_FILTER", "").split(",") if f]
    if flt:
        shapes = [sh for sh in shapes if any(f in sh[0] for f in flt)]
    for (name, Hh, W, C0, C1, up0, Cout, k, stride, dil, pad, refl) in shapes:
        g = H.ConvGeom(C0, Cout, k, stride, dil, pad, refl, C1, up0)
        H0, W0 = (Hh // 2, W // 2) if up0 else (Hh, W)
        x0 = torch.randn(B, H0, W0, C0, device=dev)
        x1 = torch.randn(B, Hh, W, C1, device=dev) if C1 else None
        w = torch.randn(Cout, C0 + C1, k, k, device=dev) * 0.05
        wp, wd = H.pack_weight(w), H.pack_weight(w, True)
        y = H.conv_forward(g, x0, x1, wp, None)
        dy = torch.randn_like(y)
        Ho, Wo = y.shape[1:3]
        flop = 2.0 * B * Ho * Wo * Cout * (C0 + C1) * k * k
        t_f = timeit(lambda: H.conv_forward(g, x0, x1, wp, None))
        t_d = timeit(lambda: H.conv_dgrad(g, dy, wd, w, (Hh, W))) if C0 > 3 else float("nan")
        t_w = timeit(lambda: H.conv_wgrad(g, x0, x1, dy))
        row = dict(name=name, B=B, gflop=flop / 1e9, fwd_ms=t_f, dgrad_ms=t_d, wgrad_ms=t_w,
                   fwd_tflops=flop / t_f / 1e9, dgrad_tflops=flop / t_d / 1e9, wgrad_tflops=flop / t_w / 1e9)
        rows.append(row)
        print("%-44s %8.1f GF  fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF" % (
            name, row["gflop"], t_f, row["fwd_tflops"], t_d, row["dgrad_tflops"], t_w, row["wgrad_tflops"]), flush=True)
        del x0, x1, y, dy
    if os.environ.get("BENCH_ONLY_CONV"):
        return
    # HBM-bound kernels
    x = torch.randn(B, 128, 256, 256, device=dev)
    rm, rv = torch.zeros(256, device=dev), torch.ones(256, device=dev)
    gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    nbytes = x.numel() * 4
    t = timeit(lambda: H.bn_stats(x, rm, rv, 0.1, 1e-5))
    print("bn_stats   %.3f ms  %.0f GB/s" % (t, nbytes / t / 1e6))
    mean, invstd = H.bn_stats(x, rm, rv, 0.1, 1e-5)
    t = timeit(lambda: H.bn_apply(x, mean, invstd, gam, bet, None, "relu"))
    print("bn_apply   %.3f ms  %.0f GB/s" % (t, 2 * nbytes / t / 1e6))
    y = H.bn_apply(x, mean, invstd, gam, bet, None, "relu")
    t = timeit(lambda: H.bn_backward(x, y, x, mean, invstd, gam, "relu"))
    print("bn_backward %.3f ms  %.0f GB/s (7 passes)" % (t, 7 * nbytes / t / 1e6))
    rows.append(dict(name="bn_backward", ms=t))
    # trainer-side rows: EMA multi-tensor update over a ResNet-101-sized parameter set, pseudo labels
    from improving_segmentation_with_selfsupervised_depth_amd import trainer as T
    sizes = [64 * 3 * 49, 64, 64] + [256 * 64, 64 * 64 * 9, 256 * 64, 256, 256] * 60 + [2048 * 512 * 9] * 8 + [19 * 64, 19]
    mp = [torch.randn(n, device=dev) for n in sizes]
    ep = [torch.randn(n, device=dev) for n in sizes]
    up = T.EmaUpdater(mp, ep)
    nel = sum(sizes)
    t = timeit(lambda: up.step(0.99, 1000))
    print("ema update  %.3f ms  %d tensors %.1f M floats  %.0f GB/s (12 B/elt)" % (t, len(sizes), nel / 1e6, 12 * nel / t / 1e6))
    def ref_loop():
        for e_, p_ in zip(ep, mp):
            e_.data[:] = 0.99 * e_.data + 0.01 * p_.data
    t2 = timeit(ref_loop)
    print("ema update, reference-style Python loop of torch ops: %.3f ms" % t2)
    soft = torch.softmax(torch.randn(B, 19, 512, 1024, device=dev) * 3, 1)
    t = timeit(lambda: H.pseudo_label(soft, 0.968, 250))
    print("pseudo_label %.3f ms  %.0f GB/s (76+8+4 B/px)" % (t, 88 * B * 512 * 1024 / t / 1e6))
    out = os.environ.get("BENCH_OUT")
    if out:
        json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
