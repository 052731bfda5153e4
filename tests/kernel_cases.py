"""Kernel parity cases shared by the CPU run (kernel sources interpreted by tests/hipemu) and the GPU run
(`-m gpu`, real libsegsde_hip.so through the C ABI).  The checker is plain PyTorch fp32 on CPU / the oracle."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from improving_segmentation_with_selfsupervised_depth_amd import hipops as H


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def assert_close(a, b, rtol=1e-3, atol=1e-5, what=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = (a - b).abs()
    ok = err <= atol * scale + rtol * b.abs()
    assert bool(ok.all()), "%s: max abs err %.3e (scale %.3e), %d/%d bad" % (what, float(err.max()), scale,
                                                                             int((~ok).sum()), ok.numel())


# each: (name, B, H, W, C0, C1, up0, Cout, k, stride, dil, pad, reflect, bias, act)
CONV_CASES = [
    ("1x1", 2, 9, 11, 16, 0, False, 24, 1, 1, 1, 0, False, False, "none"),
    ("3x3_zero", 2, 10, 13, 8, 0, False, 40, 3, 1, 1, 1, False, True, "relu"),
    ("3x3_dil2", 1, 12, 14, 12, 0, False, 16, 3, 1, 2, 2, False, False, "none"),
    ("3x3_s2", 2, 11, 14, 8, 0, False, 16, 3, 2, 1, 1, False, False, "none"),
    ("1x1_s2", 2, 10, 12, 8, 0, False, 16, 1, 2, 1, 0, False, False, "none"),
    ("7x7_s2_c3", 2, 18, 22, 3, 0, False, 16, 7, 2, 1, 3, False, False, "none"),
    ("refl", 2, 9, 12, 8, 0, False, 12, 3, 1, 1, 1, True, True, "elu"),
    ("refl_up_cat", 2, 8, 12, 8, 12, True, 16, 3, 1, 1, 1, True, True, "elu"),
    ("refl_cat_noup", 1, 6, 7, 4, 8, False, 8, 3, 1, 1, 1, True, True, "none"),
    ("refl_cout1", 2, 8, 10, 8, 0, False, 1, 3, 1, 1, 1, True, True, "sigmoid"),
    ("1x1_cout19", 1, 7, 9, 16, 0, False, 19, 1, 1, 1, 0, False, True, "none"),
    ("3x3_big_n", 1, 6, 6, 8, 0, False, 136, 3, 1, 1, 1, False, False, "none"),
    ("refl_tiny_h3", 1, 3, 4, 4, 0, False, 4, 3, 1, 1, 1, True, False, "none"),
    # channel counts that are multiples of 32 take the FAST (uniform-tap, branch-free) loader path
    ("fast_3x3", 2, 9, 11, 32, 0, False, 64, 3, 1, 1, 1, False, True, "relu"),
    ("fast_refl_up_cat", 2, 8, 12, 32, 64, True, 32, 3, 1, 1, 1, True, True, "elu"),
    ("fast_refl_cat", 1, 5, 6, 64, 32, False, 64, 3, 1, 1, 1, True, False, "none"),
    ("fast_dil3", 1, 12, 10, 64, 0, False, 32, 3, 1, 3, 3, False, False, "none"),
    ("fast_s2", 2, 11, 14, 32, 0, False, 64, 3, 2, 1, 1, False, False, "none"),
    ("fast_s2_even", 1, 8, 12, 64, 0, False, 32, 3, 2, 1, 1, False, True, "none"),
    # stride 2 with dilation 3: row and column offsets of the mixed parity classes differ -> generic data-gradient
    ("fast_s2_dil3", 2, 11, 10, 32, 0, False, 64, 3, 2, 3, 3, False, False, "none"),
    ("fast_1x1", 2, 7, 9, 96, 0, False, 160, 1, 1, 1, 0, False, True, "none"),
    ("fast_1x1_s2", 1, 8, 10, 64, 0, False, 32, 1, 2, 1, 0, False, False, "none"),
    ("fast64_refl_up_cat", 1, 8, 12, 64, 128, True, 64, 3, 1, 1, 1, True, True, "elu"),
    # output rows that are multiples of 32 pixels wide take the table-driven weight-gradient loader
    ("w32_refl_up_cat", 2, 4, 32, 8, 12, True, 16, 3, 1, 1, 1, True, True, "elu"),
    ("w64_s2_zero", 1, 6, 64, 8, 0, False, 16, 3, 2, 1, 1, False, False, "none"),
    ("w32_dil2", 2, 5, 32, 16, 0, False, 8, 3, 1, 2, 2, False, False, "relu"),
    ("w32_fast_cat", 1, 3, 32, 32, 32, False, 32, 3, 1, 1, 1, True, True, "none"),
    ("w32_1x1", 1, 2, 32, 20, 0, False, 36, 1, 1, 1, 0, False, False, "none"),
    # wide rows: some waves of a reflection-adjoint tile own no border pixel (LDS-DMA loop) while others do (register loop)
    ("fast_refl_w64", 1, 6, 64, 32, 0, False, 32, 3, 1, 1, 1, True, True, "elu"),
    ("fast_refl_w64_up_cat", 1, 4, 64, 32, 32, True, 64, 3, 1, 1, 1, True, True, "none"),
    # full tiles whose columns lie entirely on one side of the concat split (buffer-store epilogue, both destinations)
    ("fast_cat_sides", 1, 8, 16, 128, 128, False, 32, 3, 1, 1, 1, True, True, "none"),
    # 1x1 with several small images per tile and a second tile that starts inside a later image (linear row path)
    ("lin_1x1_multi", 4, 8, 8, 64, 0, False, 64, 1, 1, 1, 0, False, False, "none"),
    # adjoint data-gradient on 128-wide tiles of an image 128 pixels wide: border rows' extras come from the second table bank
    ("xtab_w128", 1, 4, 128, 128, 0, False, 128, 3, 1, 1, 1, True, False, "none"),
    # disparity heads: single output channel -> dedicated stencil kernels (C = 64 / 128 / 256)
    ("disp_c64", 2, 9, 11, 64, 0, False, 1, 3, 1, 1, 1, True, True, "sigmoid"),
    ("disp_c128_tiny", 1, 3, 5, 128, 0, False, 1, 3, 1, 1, 1, True, True, "sigmoid"),
    ("disp_c256_zero", 1, 4, 6, 256, 0, False, 1, 3, 1, 1, 1, False, False, "none"),
    # segmentation-head shaped 1x1 convs (narrow class side): register kernels for the data- and weight-gradient
    ("head_c64_19", 2, 5, 7, 64, 0, False, 19, 1, 1, 1, 0, False, True, "none"),
    ("head_c128_22", 1, 4, 6, 128, 0, False, 22, 1, 1, 1, 0, False, True, "none"),
    ("head_c256_3", 1, 3, 5, 256, 0, False, 3, 1, 1, 1, 0, False, False, "none"),
    # dilated, zero-padded windows (ASPP rates, dilated layer4): tap rows that are dead for a whole tile / pixel chunk are skipped
    # -- dilation below / at / above the map height (only the centre tap row survives), 128-pixel tiles inside one image, tiles
    # that span images (no skipping there), reduction tiles inside one tap row (C = 128) and across tap rows (C = 32)
    ("aspp_d6_h16", 2, 16, 64, 128, 0, False, 64, 3, 1, 6, 6, False, False, "none"),
    ("aspp_d18_h16", 1, 16, 64, 128, 0, False, 64, 3, 1, 18, 18, False, False, "none"),
    ("aspp_d6_h16_n128", 1, 16, 64, 64, 0, False, 128, 3, 1, 6, 6, False, False, "none"),
    ("aspp_d12_h20_w32", 2, 20, 32, 64, 0, False, 128, 3, 1, 12, 12, False, True, "relu"),
    ("aspp_d3_h5_c32", 3, 5, 32, 32, 0, False, 32, 3, 1, 3, 3, False, False, "none"),
    ("aspp_d4_h4_multi", 5, 4, 32, 32, 0, False, 32, 3, 1, 4, 4, False, False, "none"),
    ("aspp_d2_h9_odd", 2, 9, 37, 64, 0, False, 32, 3, 1, 2, 2, False, True, "none"),
    # widths that are multiples of 8 take the strip (sliding-window) stencil kernels
    ("disp_strip_c64", 2, 6, 24, 64, 0, False, 1, 3, 1, 1, 1, True, True, "sigmoid"),
    ("disp_strip_c128_zero", 1, 5, 16, 128, 0, False, 1, 3, 1, 1, 1, False, True, "none"),
    ("disp_strip_c256", 1, 4, 16, 256, 0, False, 1, 3, 1, 1, 1, True, False, "sigmoid"),
]


def random_conv_case(rng, i):
    """a random convolution geometry (numpy RandomState): odd sizes, partial tiles, several images per tile, both padding
    modes, strides, dilations, upsample + concat -- the shapes the fixed list above does not hold"""
    k = int(rng.choice([1, 3, 3, 3]))
    reflect = bool(k == 3 and rng.rand() < 0.5)
    stride = 1 if reflect else int(rng.choice([1, 1, 1, 2]))
    dil = 1 if (reflect or k == 1) else int(rng.choice([1, 1, 2, 3]))
    pad = 0 if k == 1 else dil
    up0 = bool(stride == 1 and rng.rand() < 0.35)
    C0 = int(rng.choice([32, 64, 64, 96, 128, 256]))
    C1 = int(rng.choice([0, 0, 32, 64, 128])) if (stride == 1 and k == 3) else 0
    Cout = int(rng.choice([32, 64, 64, 128, 128, 160, 192, 256, 19, 48]))
    big = 3 if os.environ.get("STRESS_BIG") else 1
    Hh, W = int(rng.randint(6, 70 * big)), int(rng.randint(6, 160 * big))
    if rng.rand() < 0.3:
        W = int(rng.choice([32, 64, 128, 160]))          # table-driven weight gradient, second table bank of the adjoint
    if up0:
        Hh, W = Hh + (Hh & 1), W + (W & 1)
    B = int(rng.choice([1, 2, 3, 5]))
    r = rng.rand()
    if r < 0.08:        # network stems: 7x7 stride 2 on 3 / 4 / 6 / 8 channels
        k, stride, dil, pad, reflect, up0, C1 = 7, 2, 1, 3, False, False, 0
        C0, Cout = int(rng.choice([3, 4, 6, 8])), 64
    elif r < 0.16:      # disparity heads (Cout = 1, stencil kernels) and class heads (skinny kernels)
        C0, C1, up0, stride, dil = int(rng.choice([64, 128, 256])), 0, False, 1, 1
        if rng.rand() < 0.5:
            k, pad, reflect, Cout = 3, 1, bool(rng.rand() < 0.7), 1
        else:
            k, pad, reflect, Cout = 1, 0, False, int(rng.choice([19, 3, 22]))
    elif r < 0.24:      # channel counts that are no multiple of 32 (float4 gather) or of 4 (scalar gather)
        C0, C1, up0 = int(rng.choice([19, 20, 36, 12, 5])), 0, False
        Cout = int(rng.choice([8, 19, 24, 40]))
    bias = bool(rng.rand() < 0.5)
    act = str(rng.choice(["none", "none", "elu"]))     # (no ReLU: a sign flip of a pre-activation next to zero between two
    #                                                       correct implementations changes the mask the gradient is checked with)
    return ("stress%03d" % i, B, Hh, W, C0, C1, up0, Cout, k, stride, dil, pad, reflect, bias, act)



def conv_reference(case, x0, x1, w, b):
    name, B, Hh, W, C0, C1, up0, Cout, k, stride, dil, pad, reflect, bias, act = case
    a = nchw(x0)
    if up0:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    if x1 is not None:
        a = torch.cat([a, nchw(x1)], 1)
    if reflect:
        a = F.pad(a, (1, 1, 1, 1), mode="reflect")
        y = F.conv2d(a, w, b, stride, 0, dil)
    else:
        y = F.conv2d(a, w, b, stride, pad, dil)
    y = {"none": lambda t: t, "relu": F.relu, "elu": F.elu, "sigmoid": torch.sigmoid}[act](y)
    return y


def run_conv_case(case, device, seed=0):
    name, B, Hh, W, C0, C1, up0, Cout, k, stride, dil, pad, reflect, bias, act = case
    gen = torch.Generator().manual_seed(seed)
    H0, W0 = (Hh // 2, W // 2) if up0 else (Hh, W)
    x0 = torch.randn(B, H0, W0, C0, generator=gen)
    x1 = torch.randn(B, Hh, W, C1, generator=gen) if C1 else None
    w = torch.randn(Cout, C0 + C1, k, k, generator=gen) * 0.2
    b = torch.randn(Cout, generator=gen) if bias else None
    x0r = x0.clone().requires_grad_(True)
    x1r = x1.clone().requires_grad_(True) if x1 is not None else None
    wr = w.clone().requires_grad_(True)
    ref = conv_reference(case, x0r, x1r, wr, b)
    gy = torch.randn(ref.shape, generator=gen)
    ref.backward(gy)

    g = H.ConvGeom(C0, Cout, k, stride, dil, pad, reflect, C1, up0)
    d = lambda t: None if t is None else t.to(device)
    wp = H.pack_weight(d(w), False)
    y = H.conv_forward(g, d(x0), d(x1), wp, d(b), act)
    assert_close(nchw(y), ref, what=name + " fwd")
    if not bias and act == "none":
        # BatchNorm statistics fused into the conv epilogue: same y, and (where the shape can fuse) the same mean / invstd /
        # running statistics as a separate pass over y
        y_s, part = H.conv_forward(g, d(x0), d(x1), wp, None, "none", want_stats=True)
        assert torch.equal(y_s, y), name + " fwd with stats"
        if part is not None:
            M = y.numel() // Cout
            rm1, rv1 = torch.zeros(Cout, device=device), torch.ones(Cout, device=device)
            rm2, rv2 = rm1.clone(), rv1.clone()
            m1, i1 = H.bn_stats_from_partials(part, M, rm1, rv1, 0.1, 1e-5)
            m2, i2 = H.bn_stats(y, rm2, rv2, 0.1, 1e-5)
            assert_close(m1, m2, rtol=1e-5, atol=1e-6, what=name + " fused bn mean")
            assert_close(i1, i2, rtol=1e-4, what=name + " fused bn invstd")
            assert_close(rm1, rm2, rtol=1e-5, atol=1e-6, what=name + " fused running_mean")
            assert_close(rv1, rv2, rtol=1e-4, what=name + " fused running_var")
    # backward: pre-activation gradient as the autograd Functions will feed it
    dy_nhwc = nhwc(gy).to(device)
    if act != "none":
        dz, dbias = H.act_backward(dy_nhwc, y, act, need_dbias=bias)
    else:
        dz, dbias = dy_nhwc, (H.colsum(dy_nhwc) if bias else None)
    wd = H.pack_weight(d(w), True)
    dx0, dx1 = H.conv_dgrad(g, dz, wd, d(w), (Hh, W))
    assert_close(dx0, x0r.grad, what=name + " dgrad0")
    if x1 is not None:
        assert_close(dx1, x1r.grad, what=name + " dgrad1")
    dw = H.conv_wgrad(g, d(x0), d(x1), dz)
    assert_close(dw, wr.grad, rtol=1e-3, what=name + " wgrad")
    if bias:
        # bias gradient of the reference
        bref = torch.autograd.grad(conv_reference(case, x0, x1, w, b.clone().requires_grad_(True)), [], allow_unused=True) \
            if False else None
        bb = b.clone().requires_grad_(True)
        conv_reference(case, x0, x1, w, bb).backward(gy)
        assert_close(dbias, bb.grad, rtol=1e-3, what=name + " dbias")


# ---------------------------------------------------------------------------------------------
# BatchNorm / activation / pooling / resize
# ---------------------------------------------------------------------------------------------
def run_dgrad_epilogue_variants(case, device, seed=0):
    """the data-gradient's epilogue variants against its own plain result: accumulate onto an existing gradient
    (single-source, non-upsampled convolutions) and the derivative of the activation whose output the convolution read"""
    name, B, Hh, W, C0, C1, up0, Cout, k, stride, dil, pad, reflect, bias, act = case
    if C0 < 4 or Cout == 1:
        return
    gen = torch.Generator().manual_seed(1000 + seed)
    g = H.ConvGeom(C0, Cout, k, stride, dil, pad, reflect, C1, up0)
    w = (torch.randn(Cout, C0 + C1, k, k, generator=gen) * 0.2).to(device)
    wd = H.pack_weight(w, True)
    Ho, Wo = g.out_hw(Hh, W)
    dz = torch.randn(B, Ho, Wo, Cout, generator=gen).to(device)
    dx0, dx1 = H.conv_dgrad(g, dz, wd, w, (Hh, W))
    h0, w0 = (Hh // 2, W // 2) if up0 else (Hh, W)
    ysaved = F.elu(torch.randn(B, h0, w0, C0, generator=gen)).to(device)
    der = torch.where(ysaved > 0, torch.ones_like(ysaved), ysaved + 1.0)
    ag0, ag1 = H.conv_dgrad(g, dz, wd, w, (Hh, W), actgrad=(ysaved, "elu"))
    assert_close(ag0, dx0 * der, rtol=1e-5, atol=1e-6, what=name + " dgrad x ELU'")
    if dx1 is not None:
        assert torch.equal(ag1, dx1), name + " dgrad of the skip source is not touched by the activation derivative"
    if not up0 and not C1:
        base = torch.randn(B, Hh, W, C0, generator=gen).to(device)
        want = base + dx0
        acc, _ = H.conv_dgrad(g, dz, wd, w, (Hh, W), accumulate_into=base)
        if acc is None:          # this shape cannot accumulate in the epilogue: the caller adds
            acc = base + dx0
        assert_close(acc, want, rtol=1e-6, atol=1e-6, what=name + " dgrad accumulate")


def run_bn_case(device, C=24, act="relu", residual=True, train=True, drop_p=0.0, seed=0, shape=(3, 5, 7)):
    gen = torch.Generator().manual_seed(seed)
    B, Hh, W = shape
    x = torch.randn(B, C, Hh, W, generator=gen) * 2 + 0.5
    res = torch.randn(B, C, Hh, W, generator=gen) if residual else None
    gamma = torch.rand(C, generator=gen) + 0.5
    beta = torch.randn(C, generator=gen)
    rm, rv = torch.randn(C, generator=gen) * 0.1, torch.rand(C, generator=gen) + 0.5
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if residual else None
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(xr, rm_ref, rv_ref, gr, br, training=train, momentum=0.1, eps=1e-5)
    if residual:
        y = y + rr
    y = {"none": lambda t: t, "relu": F.relu, "elu": F.elu}[act](y)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)

    d = lambda t: None if t is None else t.to(device)
    xn, rn = d(nhwc(x)), d(nhwc(res)) if residual else None
    rm_d, rv_d = d(rm.clone()), d(rv.clone())
    if train:
        mean, invstd = H.bn_stats(xn, rm_d, rv_d, 0.1, 1e-5)
        assert_close(rm_d, rm_ref, what="running_mean")
        assert_close(rv_d, rv_ref, what="running_var")
    else:
        mean, invstd = H.bn_eval_stats(rm_d, rv_d, 1e-5)
    out = H.bn_apply(xn, mean, invstd, d(gamma), d(beta), rn, act)
    assert_close(nchw(out), y, what="bn fwd")
    dx, dres, dgamma, dbeta = H.bn_backward(d(nhwc(gy)), out, xn, mean, invstd, d(gamma), act, batch_stats=train,
                                            need_dres=residual)
    assert_close(nchw(dx), xr.grad, what="bn dx", rtol=1e-3, atol=2e-5)
    assert_close(dgamma, gr.grad, rtol=1e-3, what="bn dgamma")
    assert_close(dbeta, br.grad, rtol=1e-3, what="bn dbeta")
    if residual:
        assert_close(nchw(dres), rr.grad, what="bn dres")
    if act in ("none", "relu") and not residual:
        # remask mode (saved output not read): the mask recomputed from x must reproduce the same gradients bit for bit
        dx2, _, dgamma2, dbeta2 = H.bn_backward(d(nhwc(gy)), None, xn, mean, invstd, d(gamma), act, batch_stats=train,
                                                beta=d(beta))
        assert torch.equal(dx2, dx) and torch.equal(dgamma2, dgamma) and torch.equal(dbeta2, dbeta), "bn remask"


def run_dropout_case(device):
    """dropout inside bn_apply: keep fraction, scaling, and backward consistency with the regenerated mask"""
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(4, 16, 16, 32, generator=gen).to(device)
    mean, invstd = torch.zeros(32).to(device), torch.ones(32).to(device)
    y0 = H.bn_apply(x, mean, invstd, None, None, None, "relu", 0.0)
    y = H.bn_apply(x, mean, invstd, None, None, None, "relu", 0.5, seed=1234)
    kept = (y != 0) | (y0 == 0)
    frac = float(((y != 0) & (y0 != 0)).sum()) / float((y0 != 0).sum())
    assert 0.45 < frac < 0.55, frac
    assert_close(y[kept], (2.0 * y0)[kept], what="dropout scale")
    gy = torch.ones_like(y)
    dx, _, _, _ = H.bn_backward(gy, y, x, mean, invstd, None, "relu", 0.5, seed=1234, batch_stats=False)
    assert_close(dx, 2.0 * (y != 0).float(), what="dropout bwd")
    y2 = H.bn_apply(x, mean, invstd, None, None, None, "relu", 0.5, seed=99)
    assert not torch.equal(y, y2)


def run_misc_cases(device):
    gen = torch.Generator().manual_seed(5)
    d = lambda t: t.to(device)
    # max pool (ties: post-ReLU zeros)
    x = F.relu(torch.randn(2, 12, 9, 11, generator=gen)).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    yo, idx = H.maxpool_forward(d(nhwc(x.detach())))
    assert torch.equal(nchw(yo).cpu(), y.detach())
    dx = H.maxpool_backward(d(nhwc(gy)), idx, (2, 9, 11, 12))
    assert_close(nchw(dx), x.grad, what="maxpool bwd")
    base = torch.randn(2, 9, 11, 12, generator=gen)
    acc = d(base.clone())
    out = H.maxpool_backward(d(nhwc(gy)), idx, (2, 9, 11, 12), accumulate_into=acc)
    assert out.data_ptr() == acc.data_ptr()
    assert_close(out, base + dx.cpu(), rtol=1e-6, atol=1e-6, what="maxpool bwd accumulated onto another consumer's gradient")
    x6 = F.relu(torch.randn(1, 6, 8, 7, generator=gen))          # channel count off the four-channel kernels
    yo6, idx6 = H.maxpool_forward(d(nhwc(x6)))
    assert torch.equal(nchw(yo6).cpu(), F.max_pool2d(x6, 3, 2, 1)), "maxpool, scalar kernel"
    # bilinear resize, both conventions, up and down, and 1x1 -> HxW (ASPP pooling)
    for (hi, wi, ho, wo, ac) in [(5, 7, 10, 14, False), (5, 7, 11, 13, True), (8, 12, 4, 6, False), (1, 1, 6, 9, False),
                                 (4, 8, 32, 64, False), (6, 6, 6, 6, False)]:
        x = torch.randn(2, 5, hi, wi, generator=gen).requires_grad_(True)
        y = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=ac)
        gy = torch.randn(y.shape, generator=gen)
        y.backward(gy)
        yo = H.resize_bilinear(d(nhwc(x.detach())), (ho, wo), ac)
        assert_close(nchw(yo), y, what="resize fwd %s" % ((hi, wi, ho, wo, ac),))
        dx = H.resize_bilinear_backward(d(nhwc(gy)), (hi, wi), ac)
        assert_close(nchw(dx), x.grad, what="resize bwd %s" % ((hi, wi, ho, wo, ac),))
    gy = torch.randn(2, 8, 6, 9, generator=gen)      # 1x1 source with a channel count of the four-channel kernel
    assert_close(nchw(H.resize_bilinear_backward(d(nhwc(gy)), (1, 1), False)), gy.sum((2, 3), keepdim=True), rtol=1e-5, atol=1e-6,
                 what="resize bwd to 1x1, 8 channels")
    # global average pool
    x = torch.randn(3, 70, 5, 6, generator=gen).requires_grad_(True)
    y = x.mean((2, 3), keepdim=True)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    yo = H.global_avgpool(d(nhwc(x.detach())))
    assert_close(nchw(yo), y, what="gap fwd")
    assert_close(nchw(H.global_avgpool_backward(d(nhwc(gy)), (3, 5, 6, 70))), x.grad, what="gap bwd")
    # few images x few channels over many pixels: the pixel range is sliced over blocks, partial sums in a workspace
    x = torch.randn(1, 20, 37, 41, generator=gen) + 3.0
    assert_close(nchw(H.global_avgpool(d(nhwc(x)))), x.mean((2, 3), keepdim=True), rtol=1e-6, what="gap fwd sliced")
    x = torch.randn(3, 72, 5, 6, generator=gen) - 1.0          # four channels per thread, one slice
    assert_close(nchw(H.global_avgpool(d(nhwc(x)))), x.mean((2, 3), keepdim=True), rtol=1e-6, atol=1e-7, what="gap fwd, 4 channels per thread")
    # gate
    f = torch.randn(2, 4, 5, 8, generator=gen).requires_grad_(True)
    a = torch.randn(2, 4, 5, 8, generator=gen).requires_grad_(True)
    y = f * torch.sigmoid(a)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    assert_close(H.gate_forward(d(f.detach()), d(a.detach())), y, what="gate fwd")
    df, da = H.gate_backward(d(gy), d(f.detach()), d(a.detach()))
    assert_close(df, f.grad, what="gate df")
    assert_close(da, a.grad, what="gate da")
    # axpby / copy_channels / layout edges / upsample adjoint
    x, y = torch.randn(1000, generator=gen), torch.randn(1000, generator=gen)
    assert_close(H.axpby(0.99, d(x), 0.01, d(y)), 0.99 * x + 0.01 * y, what="axpby")
    buf = torch.zeros(2, 3, 4, 20).to(device)
    src = torch.randn(2, 3, 4, 8, generator=gen)
    H.copy_channels(d(src), buf[..., 4:12])
    assert torch.equal(buf[..., 4:12].cpu(), src) and float(buf[..., :4].abs().sum()) == 0
    img = torch.rand(2, 3, 6, 10, generator=gen)
    assert_close(nchw(H.nchw_to_nhwc(d(img), 0.45, 0.225)), (img - 0.45) / 0.225, rtol=1e-6, atol=1e-6, what="img norm")
    for _ in range(2):      # padded to 4 channels: the kernel itself writes the zero channel (the output is torch.empty)
        padded = H.nchw_to_nhwc(d(img), 0.45, 0.225, pad_to=4)
        assert padded.shape[-1] == 4 and float(padded[..., 3].abs().max()) == 0.0
        assert_close(nchw(padded[..., :3]), (img - 0.45) / 0.225, rtol=1e-6, atol=1e-6, what="img norm, padded")
        padded.fill_(7.0)   # the next call probably gets this very block back from the caching allocator
        del padded
    img6 = torch.rand(2, 6, 5, 7, generator=gen)       # a pose-network pair: 6 planes -> 8 channels
    p8 = H.nchw_to_nhwc(d(img6), 0.45, 0.225, pad_to=8)
    assert p8.shape[-1] == 8 and float(p8[..., 6:].abs().max()) == 0.0
    assert_close(nchw(p8[..., :6]), (img6 - 0.45) / 0.225, rtol=1e-6, atol=1e-6, what="img norm, 6 -> 8 channels")
    t = torch.randn(2, 6, 10, 7, generator=gen)
    assert torch.equal(H.nhwc_to_nchw(d(t)).cpu(), nchw(t))
    assert torch.equal(H.nchw_to_nhwc(d(nchw(t))).cpu(), t)
    # colsum with a pitch
    wide = torch.randn(50, 40, generator=gen).to(device)
    assert_close(H.colsum(wide[:, 8:24]), wide[:, 8:24].cpu().sum(0), what="colsum")
    one = torch.randn(2, 64, 64, 1, generator=gen) + 0.3     # a single column (bias gradient of a disparity head)
    assert_close(H.colsum(d(one)), one.double().sum().float().reshape(1), rtol=1e-5, what="colsum of one column")


# ---------------------------------------------------------------------------------------------
# pose
# ---------------------------------------------------------------------------------------------
def run_pose_case(device, golden):
    g = golden("geom")
    from oracle import geometry as G
    for inv, tag in ((False, "fwd"), (True, "inv")):
        aa4 = torch.zeros(3, 2, 1, 3)
        tr4 = torch.zeros(3, 2, 1, 3)
        aa4[:, 0] = g["axisangle"]
        tr4[:, 0] = g["translation"]
        aa4[:, 1] = 7.0  # frame 1 must be ignored
        M = H.pose_matrix(aa4.to(device), tr4.to(device), inv)
        assert_close(M, g["M_" + tag], rtol=1e-5, atol=1e-6, what="pose M")
        daa, dtr = H.pose_matrix_backward(aa4.to(device), tr4.to(device), g["M_weight"].to(device), inv)
        assert_close(daa[:, 0], g["grad_aa_" + tag], rtol=1e-4, atol=1e-6, what="pose daa")
        assert_close(dtr[:, 0], g["grad_tr_" + tag], rtol=1e-4, atol=1e-6, what="pose dtr")
        assert float(daa[:, 1].abs().sum()) == 0
    M0 = H.pose_matrix(torch.zeros(1, 2, 1, 3).to(device), torch.ones(1, 2, 1, 3).to(device), False)
    assert_close(M0, g["M_zero"], what="pose zero angle")
    daa, _ = H.pose_matrix_backward(torch.zeros(1, 2, 1, 3).to(device), torch.ones(1, 2, 1, 3).to(device),
                                    torch.ones(1, 4, 4).to(device), False)
    assert bool(torch.isfinite(daa).all())


# ---------------------------------------------------------------------------------------------
# segmentation loss / mix / masks  (bit-exact where the reference is integer / mask arithmetic)
# ---------------------------------------------------------------------------------------------
def run_segmix_cases(device, golden):
    g = golden("segmix")
    d = lambda t: t.to(device)
    logits = d(nhwc(g["ce_logits"]))
    tgt = d(g["ce_target"])
    M = tgt.numel()
    out = H.cross_entropy_forward(logits, tgt, 250)
    assert_close(out[0] / out[1], g["ce_loss"], rtol=1e-5, what="ce loss")
    scale = (1.0 / out[1]).reshape(1)
    dl = H.cross_entropy_backward(logits, tgt, 250, scale)
    assert_close(nchw(dl), g["ce_grad"], rtol=1e-4, atol=1e-7, what="ce grad")
    pw = d(g["ce_pw"].contiguous())
    out = H.cross_entropy_forward(logits, tgt, 250, pixel_weights=pw)
    assert_close(out[0] / M, g["ce_loss_pw"], rtol=1e-5, what="ce pw loss")
    dl = H.cross_entropy_backward(logits, tgt, 250, torch.full((1,), 1.0 / M).to(device), pixel_weights=pw)
    assert_close(nchw(dl), g["ce_grad_pw"], rtol=1e-4, atol=1e-7, what="ce pw grad")
    # logits as a padded-pitch slice
    wide = torch.zeros(logits.shape[:3] + (24,)).to(device)
    wide[..., :19] = logits
    out2 = H.cross_entropy_forward(wide[..., :19], tgt, 250)
    assert torch.equal(out2.cpu(), H.cross_entropy_forward(logits, tgt, 250).cpu())
    # all-ignored -> 0/0 = nan like the reference
    oi = H.cross_entropy_forward(logits, torch.full_like(tgt, 250), 250)
    assert float(oi[1]) == 0.0 and bool(torch.isnan(g["ce_loss_allignored"]))
    # mix (bit exact): NCHW and channels-last inputs, int64 / float masks, half-batch branch, labels
    for mk, xk, ok in (("mix_mask_f", "mix_img", "mix_img_f"), ("mix_mask_i", "mix_img", "mix_img_i"),
                       ("mix_mask_i", "mix_soft", "mix_soft_i"), ("mix_mask_half", "mix_img", "mix_img_half")):
        assert torch.equal(H.mix(d(g[mk]), d(g[xk])).cpu(), g[ok]), ok
        cl = d(g[xk]).contiguous(memory_format=torch.channels_last)
        o = H.mix(d(g[mk]), cl)
        assert o.stride() == cl.stride() and torch.equal(o.cpu(), g[ok]), ok + " channels_last"
    assert torch.equal(H.mix_labels(d(g["mix_mask_i"]), d(g["mix_lbl"])).cpu(), g["mix_target_i"])
    # masks (bit exact)
    m = H.depthcomp_mask(d(g["dc_depths"]), 0.03, 0.0)
    assert m.dtype == torch.int64 and torch.equal(m.cpu(), g["dc_mask_m003_ft0"])
    assert torch.equal(H.depthcomp_mask(d(g["dc_depths"]), 0.03, 0.25).cpu(), g["dc_mask_m003_ft025"])
    assert torch.equal(H.depth_threshold_mask(d(g["dm_depth"]), float(g["dm_thr1"][0])).cpu(), g["dm_mask1"])
    t2 = g["dm_thr2"]
    assert torch.equal(H.depth_threshold_mask(d(g["dm_depth"]), float(t2.min()), float(t2.max()), True).cpu(), g["dm_mask2"])
    assert torch.equal(H.class_mask(d(g["cm_pred"]), d(g["cm_classes"])).cpu(), g["cm_mask"])


# ---------------------------------------------------------------------------------------------
# loss kernels one by one against torch autograd / the golden vectors
# ---------------------------------------------------------------------------------------------
def run_loss_kernel_cases(device, golden):
    from oracle import photometric as P, geometry as G
    d = lambda t: t.to(device)
    g = golden("ssim_smooth")
    # reprojection error fwd / bwd (SSIM + L1)
    for no_ssim in (False, True):
        x = g["x"].clone().requires_grad_(True)
        err = P.reprojection_error(x, g["y"], no_ssim)
        gen = torch.Generator().manual_seed(1)
        ge = torch.randn(err.shape, generator=gen)
        err.backward(ge)
        B, _, Hh, W = g["x"].shape
        buf = torch.zeros(B, 2, Hh, W).to(device)
        H.reprojection_error(d(g["x"]), d(g["y"]), no_ssim, buf[:, 1])
        assert_close(buf[:, 1:2], err, rtol=1e-4, atol=1e-6, what="reproj err")
        assert float(buf[:, 0].abs().sum()) == 0
        gbuf = torch.zeros(B, 2, Hh, W)
        gbuf[:, 1] = ge[:, 0]
        gp = H.reprojection_error_backward(d(g["x"]), d(g["y"]), d(gbuf)[:, 1], no_ssim)
        assert_close(gp, x.grad, rtol=1e-3, atol=2e-5, what="reproj err bwd no_ssim=%s" % no_ssim)
    # clamp branch (near-identical images)
    B, _, Hh, W = g["x"].shape
    e2 = torch.zeros(B, 1, Hh, W).to(device)
    H.reprojection_error(d(g["x"]), d(g["y2"]), False, e2[:, 0])
    assert_close(e2, P.reprojection_error(g["x"], g["y2"]), rtol=1e-4, atol=1e-6, what="reproj err clamp")
    # smoothness
    disp = g["sm_disp"].clone().requires_grad_(True)
    mean_disp = disp.mean(2, True).mean(3, True)
    sm = P.edge_aware_smoothness(disp / (mean_disp + 1e-7), g["sm_img"])
    sm.backward()
    out, mean = H.smoothness_forward(d(g["sm_disp"]), d(g["sm_img"]))
    assert_close(out, sm.reshape(1), rtol=1e-4, what="smooth fwd")
    gd = torch.zeros_like(g["sm_disp"]).to(device)
    H.smoothness_backward(d(g["sm_disp"]), d(g["sm_img"]), mean, 1.0, gd)
    assert_close(gd, disp.grad, rtol=1e-3, atol=1e-6, what="smooth bwd")
    # warp fwd / bwd at two scales against the reference's own vectors + torch autograd of the oracle
    gl = golden("loss_default")
    inv_K, K = gl["in_inv_K_0"], gl["in_K_0"]
    for s in (0, 2):
        for f, tag in ((-1, "m1"), (1, "p1")):
            src = gl["in_color_%d_0" % f]
            color, grid, depth = H.warp_forward(d(gl["disp_%d" % s]), d(inv_K), d(K), d(gl["T_" + tag]), d(src), 0.1, 100,
                                                True, True)
            assert_close(depth, gl["depth_%d" % s], rtol=1e-5, what="depth")
            assert_close(grid, gl["sample_%s_%d" % (tag, s)], rtol=1e-4, atol=1e-5, what="grid")
            assert_close(color, gl["color_%s_%d" % (tag, s)], rtol=1e-3, atol=1e-4, what="warped color")
            # backward vs autograd through the oracle
            disp = gl["disp_%d" % s].clone().requires_grad_(True)
            T = gl["T_" + tag].clone().requires_grad_(True)
            Bq, _, Hh, W = src.shape
            up = F.interpolate(disp, [Hh, W], mode="bilinear", align_corners=False)
            up.retain_grad()
            dep = G.disp_to_depth(up, 0.1, 100)[1]
            col = G.warp(src, G.project(G.backproject(dep, inv_K), K, T, Hh, W))
            gen = torch.Generator().manual_seed(2)
            gc = torch.randn(col.shape, generator=gen)
            col.backward(gc)
            gup = torch.zeros(Bq, Hh, W).to(device)
            gT = torch.zeros(Bq, 4, 4).to(device)
            H.warp_backward(d(gc), d(gl["disp_%d" % s]), d(inv_K), d(K), d(gl["T_" + tag]), d(src), 0.1, 100, gup, gT)
            assert_close(gup, up.grad[:, 0], rtol=1e-3, atol=1e-5, what="warp bwd disp s=%d" % s)
            assert_close(gT, T.grad, rtol=1e-3, atol=1e-5, what="warp bwd T")
    # automask min fwd / bwd: two source frames (every shipped configuration), one, and monodepth2's three / a longer set
    gen = torch.Generator().manual_seed(4)
    for nfr in (2, 1, 3, 5):
        ident, reproj = torch.rand(2, nfr, 6, 9, generator=gen), torch.rand(2, nfr, 6, 9, generator=gen)
        if nfr == 3:
            reproj[0, 1, 2, 3] = reproj[0, 0, 2, 3] = 0.0        # a tie between candidates: the first index wins like torch.min's
        noise = torch.randn(2, nfr, 6, 9, generator=gen)
        for avg in (False, True):
            for use_ident in (True, False):
                i_, r_ = ident, reproj
                nz = noise[:, :1].contiguous() if avg else noise
                if avg:
                    i_c, r_c = ident.mean(1, keepdim=True), reproj.mean(1, keepdim=True)
                else:
                    i_c, r_c = ident, reproj
                comb = torch.cat([i_c + nz * 0.00001, r_c], 1) if use_ident else r_c
                if comb.shape[1] == 1:
                    mn, ix = comb[:, 0], torch.zeros_like(comb[:, 0], dtype=torch.long)
                else:
                    mn, ix = torch.min(comb, 1)
                what = "automask %d frames avg=%s ident=%s" % (nfr, avg, use_ident)
                out, sel, isel = H.automask_min(d(i_) if use_ident else None, d(nz) if use_ident else None, d(r_), avg)
                assert_close(out, mn.sum().reshape(1), rtol=1e-5, what=what + " sum")
                if avg and nfr > 2:     # a mean of three is a rounded quotient: candidates within an ulp may swap
                    assert float((sel.cpu().long() != ix).float().mean()) < 0.02, what
                else:
                    assert torch.equal(sel.cpu().long(), ix), what
                    if use_ident:
                        assert torch.equal(isel.cpu(), (ix > i_c.shape[1] - 1).float()), what
                gr = H.automask_min_backward(sel, use_ident, nfr, avg, 0.25)
                r2 = reproj.clone().requires_grad_(True)
                r2c = r2.mean(1, keepdim=True) if avg else r2
                comb2 = torch.cat([(i_c + nz * 0.00001), r2c], 1) if use_ident else r2c
                (comb2.min(1)[0].sum() * 0.25 if comb2.shape[1] > 1 else comb2.sum() * 0.25).backward()
                assert_close(gr, r2.grad, what=what + " bwd")


# ---------------------------------------------------------------------------------------------
# trainer-side callers: product vs the reference's vectors, bit-exact where integer / same
# fp32 operation order
# ---------------------------------------------------------------------------------------------
def run_trainer_cases(device, golden):
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from trainer_fixture import Tiny
    from improving_segmentation_with_selfsupervised_depth_amd import trainer as T
    g = golden("trainer")
    branches = json.loads(str(g["ema_branches_json"]))
    for tag, br in branches.items():
        for it in (0, 3, 5000):
            model, ema = Tiny(1).to(device), Tiny(2).to(device)
            out = T.update_ema_variables(ema, model, 0.99, it, **br)
            assert out is ema
            for n, p in ema.named_parameters():
                want = g["ema_%s_it%d_%s" % (tag, it, n)]
                got = p.data.cpu() if p.numel() < 2000 else p.data.cpu()[::97]
                assert torch.equal(got, want), ("ema", tag, it, n, float((got - want).abs().max()))
    # a second step on the same pair reuses the cached table and still matches two reference steps
    model, ema = Tiny(1).to(device), Tiny(2).to(device)
    T.update_ema_variables(ema, model, 0.99, 3)
    T.update_ema_variables(ema, model, 0.99, 4)
    mref, eref = Tiny(1), Tiny(2)
    for it in (3, 4):
        a = min(1 - 1 / (it + 1), 0.99)
        for e_, p_ in zip(eref.parameters(), mref.parameters()):
            e_.data[:] = a * e_.data + (1 - a) * p_.data
    for (n, p), q in zip(ema.named_parameters(), eref.parameters()):
        assert torch.equal(p.data.cpu(), q.data), ("ema two steps", n)

    soft, student = g["pl_soft"].to(device), g["pl_student"].detach().clone().to(device).requires_grad_(True)
    L_u, label = T.calc_pseudo_label_loss(soft, student, float(g["pl_consistency_weight"]))
    assert torch.equal(label.cpu(), g["pl_label"]), "pseudo label"
    assert_close(L_u, g["pl_loss"], rtol=1e-5, what="pseudo-label loss")
    L_u.backward()
    assert_close(student.grad, g["pl_grad"], rtol=1e-4, atol=1e-8, what="pseudo-label grad")
    lab2, count, maxp, pw = H.pseudo_label(soft, 0.968, 250, want_max=True)
    mx = g["pl_soft"].max(1)[0]
    assert torch.equal(maxp.cpu(), mx)
    assert int(count.cpu()[0]) == int((mx >= 0.968).sum())
    assert torch.equal(pw.cpu(), torch.full(mx.shape, int((mx >= 0.968).sum()) / mx.numel(), dtype=torch.float32))


def run_mix_use_gt_cases(device, golden):
    """mix_use_gt (train.py:667-672) on the reference's own vectors (tests/golden/usegt.npz: Trainer.train_step_segmentation_
    unlabeled run with a stand-in self): teacher softmax -> one-hot select -> depthcomp mask -> mix -> pseudo labels -> loss."""
    from improving_segmentation_with_selfsupervised_depth_amd import trainer as T
    from improving_segmentation_with_selfsupervised_depth_amd.loader import transformsgpu
    g = golden("usegt")
    logits = g["teacher_logits"].to(device)
    soft = T.teacher_softmax(logits)
    before = soft.clone()
    onehot, flags = g["onehot_lbl"].to(device), g["is_labeled"].to(device)
    H.onehot_select_(soft, onehot, flags)
    assert torch.equal(soft[0].cpu(), g["onehot_lbl"][0].float()), "labeled sample: the one-hot planes, bit-exact"
    assert torch.equal(soft[1], before[1]), "unlabeled sample: untouched, bit-exact"
    for dt in (torch.float32, torch.uint8):                       # other plane dtypes a pipeline may hand over
        s2 = before.clone()
        H.onehot_select_(s2, onehot.to(dt), [1, 0])               # flags as a host list
        assert torch.equal(s2, soft)
    none = before.clone()
    H.onehot_select_(none, onehot, torch.zeros(2, dtype=torch.bool))
    assert torch.equal(none, before)
    mask = T.generate_mix_mask("depthcomp", None, g["img"].to(device), g["pseudo_depth"].to(device), float(g["margin"]), float(g["ft"]))
    mixed, _ = transformsgpu.mix(mask=mask, data=g["img"].to(device))
    assert torch.equal(mixed.cpu(), g["mixed_img"]), "mixed image"
    soft_mixed, _ = transformsgpu.mix(mask=mask, data=soft)
    assert_close(soft_mixed, g["soft_mixed"], rtol=2e-6, atol=1e-9, what="mixed teacher distribution")
    # pixels that came from the labeled sample are exactly 0 / 1
    lab_px = (mask[0] == 1).cpu()
    assert torch.equal(soft_mixed[0].cpu()[:, lab_px], g["soft_mixed"][0][:, lab_px])
    student = g["mixed_img"].new_zeros(0)
    w, b = g["student_weight"].detach().clone().to(device).requires_grad_(True), g["student_bias"].detach().clone().to(device).requires_grad_(True)
    geom = H.ConvGeom(3, 19, 3, 1, 1, 1, False, 0, False)
    from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn
    x = Fn.to_nhwc(mixed, pad_to=4)
    wp = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 1))
    y = Fn.ConvFn.apply(x, None, wp, b, H.ConvGeom(4, 19, 3, 1, 1, 1, False, 0, False), "none")
    L_2, label = T.calc_pseudo_label_loss(soft_mixed, Fn.to_nchw(y), 1.0)
    assert torch.equal(label.cpu(), g["pseudo_label"]), "pseudo labels (one-hot rows of ignored pixels -> 250)"
    assert_close(L_2, g["L_2"], rtol=1e-4, what="L_2 with mix_use_gt")
    L_2.backward()
    assert_close(w.grad, g["grad_weight"], rtol=1e-3, atol=1e-6, what="student weight gradient")
    assert_close(b.grad, g["grad_bias"], rtol=1e-3, atol=1e-6, what="student bias gradient")
    # per-image foreground thresholds (train.py:592-599): the [B] threshold vector == per-image scalar calls
    d = g["pseudo_depth"].to(device)
    ft = torch.tensor([0.05, 0.4]).to(device)
    per = H.depthcomp_mask(d, 0.03, ft)
    for i in range(2):
        assert torch.equal(per[i], H.depthcomp_mask(d, 0.03, float(ft[i]))[i])


def run_dead_tap_rows_case(device):
    """ASPP-style dilated 3x3 with zero padding and a dilation beyond the map height (model_parts.py:5-32 at rate 18 on a 16-row
    map): tap rows 0 and 2 only ever see padding.  The kernels skip such rows per tile, which this test PROVES rather than
    assumes: their weights are set to +inf (0 * inf = nan would poison every output of a kernel that multiplies them) and the
    result must equal the convolution with those weights zeroed."""
    gen = torch.Generator().manual_seed(5)
    B, Hh, W, C, Cout, dil = 2, 16, 64, 128, 64, 18
    g = H.ConvGeom(C, Cout, 3, 1, dil, dil, False, 0, False)
    x = torch.randn(B, Hh, W, C, generator=gen)
    w = torch.randn(Cout, C, 3, 3, generator=gen) * 0.1
    w0 = w.clone()
    w0[:, :, 0, :] = 0
    w0[:, :, 2, :] = 0
    winf = w.clone()
    winf[:, :, 0, :] = float("inf")
    winf[:, :, 2, :] = float("inf")
    ref = F.conv2d(nchw(x), w0, None, 1, dil, dil)
    y = H.conv_forward(g, x.to(device), None, H.pack_weight(winf.to(device), False), None, "none")
    assert bool(torch.isfinite(y).all()), "a dead tap row was multiplied"
    assert_close(nchw(y), ref, what="dead tap rows fwd")
    dy = torch.randn(B, Hh, W, Cout, generator=gen)
    xr = nchw(x).clone().requires_grad_(True)
    F.conv2d(xr, w0, None, 1, dil, dil).backward(nchw(dy))
    dx, _ = H.conv_dgrad(g, dy.to(device), H.pack_weight(winf.to(device), True), winf.to(device), (Hh, W))
    assert bool(torch.isfinite(dx).all()), "a dead tap row was multiplied (data gradient)"
    assert_close(nchw(dx), xr.grad, what="dead tap rows dgrad")
    # weight gradient: the dead rows' gradient is exactly zero, the live row's the reference's
    wr = w.clone().requires_grad_(True)
    F.conv2d(nchw(x), wr, None, 1, dil, dil).backward(nchw(dy))
    dw = H.conv_wgrad(g, x.to(device), None, dy.to(device))
    assert float(dw[:, :, 0, :].abs().max()) == 0.0 and float(dw[:, :, 2, :].abs().max()) == 0.0
    assert_close(dw, wr.grad, rtol=1e-3, what="dead tap rows wgrad")


def run_stem_cases(device, cases=None):
    """Network stems (resnet_encoder.py:40-52, :90-93): (image - 0.45) / 0.225 -> Conv2d(3n, 64, 7, 2, 3, bias=False), forward
    (+ BatchNorm statistics partials) and weight gradient of the dedicated path against torch's fp64 conv2d and its autograd, the
    generic route of this package (to_nhwc + ConvFn) as a second witness."""
    import torch.nn.functional as F
    from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn
    from improving_segmentation_with_selfsupervised_depth_amd.models.layers import Conv2d
    if cases is None:   # B, C, H, W, Cout
        cases = [(2, 3, 16, 64, 64), (1, 6, 16, 64, 64), (1, 3, 15, 37, 64), (2, 6, 9, 21, 32), (1, 3, 2, 2, 64)]
    assert H.STEM
    taken0 = dict(H.STEM_TAKEN)
    for (B, C, Hh, W, Cout) in cases:
        gen = torch.Generator().manual_seed(Hh * 100 + W + C)
        img = torch.rand(B, C, Hh, W, generator=gen)
        conv = Conv2d(C, Cout, 7, 2, 3, bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(Cout, C, 7, 7, generator=gen) * 0.1)
        conv = conv.to(device).train()
        Ho, Wo = (Hh - 1) // 2 + 1, (W - 1) // 2 + 1
        dy = torch.randn(B, Ho, Wo, Cout, generator=gen)
        what = "stem B%d %dx%d c%d->%d" % (B, Hh, W, C, Cout)
        wq = conv.weight.detach().cpu().double().requires_grad_(True)
        yt = F.conv2d((img.double() - 0.45) / 0.225, wq, None, 2, 3)
        yt.backward(dy.double().permute(0, 3, 1, 2))
        y = conv.forward_image(img.to(device), 0.45, 0.225)
        assert y is not None, what + ": not taken"
        assert_close(y, yt.permute(0, 2, 3, 1), rtol=1e-4, atol=2e-5, what=what + " forward")
        part = getattr(y, "_bn_partials", None)
        if part is not None and part[0] is not None:
            st = part[0].sum(0).cpu()
            flat = yt.permute(0, 2, 3, 1).reshape(-1, Cout)
            assert_close(st[0], flat.sum(0), rtol=1e-5, atol=1e-4, what=what + " statistics: sum")
            assert_close(st[1], (flat * flat).sum(0), rtol=1e-5, atol=1e-4, what=what + " statistics: sum of squares")
        y.backward(dy.to(device))
        assert_close(conv.weight.grad, wq.grad, rtol=1e-4, atol=2e-5 * max(1.0, float(wq.grad.abs().max())), what=what + " dW")
        g1 = conv.weight.grad.clone()
        conv.weight.grad = None
        y2 = conv.forward_image(img.to(device), 0.45, 0.225)
        y2.backward(dy.to(device))
        assert torch.equal(y, y2) and torch.equal(conv.weight.grad, g1), what + " deterministic"
        # the generic route on the same inputs
        conv.weight.grad = None
        y9 = conv(Fn.to_nhwc(img.to(device), 0.45, 0.225, pad_to=4))
        assert_close(y, y9, rtol=1e-4, atol=2e-5, what=what + " forward vs generic route")
        # an image that needs a gradient is not this path's business
        assert conv.forward_image(img.to(device).requires_grad_(True), 0.45, 0.225) is None
    took = {k: H.STEM_TAKEN[k] - taken0[k] for k in taken0}
    assert took["fwd"] == 2 * len(cases) and took["wgrad"] == 2 * len(cases), took
    # other geometries are declined
    assert Conv2d(3, 64, 3, 2, 1, bias=False).forward_image(torch.rand(1, 3, 8, 8), 0.45, 0.225) is None
    assert Conv2d(3, 64, 7, 2, 3, bias=True).forward_image(torch.rand(1, 3, 8, 8), 0.45, 0.225) is None


def run_upfold_cases(device, cases=None):
    """Upsample-folded route of Conv3x3 on [upsample(x0) | x1] (depth_decoder.py:93-101): forward (+ bias + ELU), both data-
    gradients (+ fused activation backward) and the weight gradient against torch's fp64 interpolate -> ReflectionPad2d -> conv2d
    and its autograd; the plain 9-tap route of this package on the same inputs as a second witness."""
    import torch.nn.functional as F
    if cases is None:   # B, H, W, C0, C1, Cout, act, bias
        cases = [(2, 8, 64, 32, 0, 32, "elu", True), (1, 8, 64, 32, 32, 64, "elu", True), (1, 4, 256, 64, 32, 160, "none", False),
                 (1, 6, 64, 32, 32, 32, "relu", True)]
    assert H.UPFOLD
    taken0 = dict(H.UPFOLD_TAKEN)
    min_macs, H.UPFOLD_MIN_SAVED_MACS = H.UPFOLD_MIN_SAVED_MACS, 0.0     # the size gate would send these small cases to the plain route
    try:
        _run_upfold_cases(device, cases)
    finally:
        H.UPFOLD_MIN_SAVED_MACS = min_macs
    n = len(cases)
    took = {k: H.UPFOLD_TAKEN[k] - taken0[k] for k in taken0}
    assert took["fwd"] == n and took["wgrad"] == 2 * n and took["dgrad"] >= 2 * n, ("a case fell back to the 9-tap route", took)
    g = H.ConvGeom(32, 32, 3, 1, 1, 1, True, 0, True)
    assert H.upfold_ok(g) and not H.upfold_ok(g, 2 * 64 * 64) and H.upfold_ok(g, 16 * 512 * 1024)   # the size gate


def _run_upfold_cases(device, cases):
    import torch.nn.functional as F
    for (B, Hh, W, C0, C1, Cout, act, bias) in cases:
        gen = torch.Generator().manual_seed(Hh * 1000 + W + C0)
        g = H.ConvGeom(C0, Cout, 3, 1, 1, 1, True, C1, True)
        assert H.upfold_ok(g)
        x0 = torch.randn(B, Hh // 2, W // 2, C0, generator=gen)
        x1 = torch.randn(B, Hh, W, C1, generator=gen) if C1 else None
        wt = torch.randn(Cout, C0 + C1, 3, 3, generator=gen) * 0.1
        bs = torch.randn(Cout, generator=gen) if bias else None
        dy = torch.randn(B, Hh, W, Cout, generator=gen)
        what = "upfold B%d %dx%d %d+%d->%d %s" % (B, Hh, W, C0, C1, Cout, act)
        # fp64 truth
        a0 = x0.double().permute(0, 3, 1, 2).requires_grad_(True)
        a1 = x1.double().permute(0, 3, 1, 2).requires_grad_(True) if C1 else None
        wq = wt.double().requires_grad_(True)
        up = F.interpolate(a0, scale_factor=2, mode="nearest")
        xin = torch.cat([up, a1], 1) if C1 else up
        z = F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), wq, None if bs is None else bs.double())
        yt = {"elu": F.elu, "relu": torch.relu, "none": lambda v: v}[act](z)
        z.backward(dy.double().permute(0, 3, 1, 2))          # dy is the PRE-activation gradient, as ConvFn.backward receives it
        d = lambda t: None if t is None else t.to(device)
        X0, X1, Wt, Bs, Dy = d(x0), d(x1), d(wt), d(bs), d(dy)
        wp, wd = H.pack_weight_both(Wt)
        wf, wdf = H.upfold_pack(Wt, C0)
        y = H.conv_forward(g, X0, X1, wp, Bs, act, wfold=wf)
        assert_close(y, yt.permute(0, 2, 3, 1), rtol=1e-4, atol=1e-5, what=what + " forward")
        y9 = H.conv_forward(g, X0, X1, wp, Bs, act)
        assert_close(y, y9, rtol=1e-4, atol=1e-5, what=what + " forward vs 9-tap route")
        dx0, dx1 = H.conv_dgrad(g, Dy, wd, Wt, (Hh, W), fold=(wf, wdf))
        assert_close(dx0, a0.grad.permute(0, 2, 3, 1), rtol=1e-4, atol=1e-5, what=what + " d/dx0")
        if C1:
            assert_close(dx1, a1.grad.permute(0, 2, 3, 1), rtol=1e-4, atol=1e-5, what=what + " d/dx1")
        # fused activation backward of the tensor x0 came from
        ysaved = torch.nn.functional.elu(torch.randn(B, Hh // 2, W // 2, C0, generator=gen)).to(device)
        f0, f1 = H.conv_dgrad(g, Dy, wd, Wt, (Hh, W), fold=(wf, wdf), actgrad=(ysaved, "elu"))
        der = torch.where(ysaved > 0, torch.ones_like(ysaved), ysaved + 1)
        assert_close(f0, dx0 * der, rtol=1e-5, atol=1e-6, what=what + " d/dx0 with activation backward")
        if C1:
            assert torch.equal(f1, dx1)
        only1 = H.conv_dgrad(g, Dy, wd, Wt, (Hh, W), fold=(wf, wdf), need0=False) if C1 else None
        if C1:
            assert only1[0] is None and torch.equal(only1[1], dx1)
        dw = H.conv_wgrad(g, X0, X1, Dy)
        assert_close(dw, wq.grad, rtol=1e-4, atol=1e-5, what=what + " dW")
        dw2 = H.conv_wgrad(g, X0, X1, Dy)
        assert torch.equal(dw, dw2), what + " dW deterministic"


def run_upfold_random(device, n=24, seed=7):
    """random geometries of the upsample-folded route against the package's own 9-tap route (itself pinned on torch elsewhere):
    odd tile counts, sub-grid rows that are not a multiple of the 128-row tile, partial last tiles, channel counts that mix
    tile shapes (96, 160, 192), tiny maps.  Shapes the folded weight gradient does not take (rows not a multiple of 32 pixels
    wide) must fall back silently and still be right."""
    rng = np.random.RandomState(seed)
    min_macs, H.UPFOLD_MIN_SAVED_MACS = H.UPFOLD_MIN_SAVED_MACS, 0.0
    t0 = dict(H.UPFOLD_TAKEN)
    try:
        for it in range(n):
            B = int(rng.randint(1, 4))
            h2, w2 = int(rng.choice([2, 3, 5, 8, 13])), int(rng.choice([4, 7, 16, 32, 33, 64, 96, 128]))
            C0, C1 = int(rng.choice([32, 64, 96])), int(rng.choice([0, 0, 32, 64]))
            Cout = int(rng.choice([32, 64, 96, 160]))
            act = str(rng.choice(["none", "elu"]))
            Hh, W = 2 * h2, 2 * w2
            g = H.ConvGeom(C0, Cout, 3, 1, 1, 1, True, C1, True)
            gen = torch.Generator().manual_seed(1000 + it)
            x0 = torch.randn(B, h2, w2, C0, generator=gen).to(device)
            x1 = torch.randn(B, Hh, W, C1, generator=gen).to(device) if C1 else None
            wt = (torch.randn(Cout, C0 + C1, 3, 3, generator=gen) * 0.1).to(device)
            bs = torch.randn(Cout, generator=gen).to(device) if act != "none" else None
            dy = torch.randn(B, Hh, W, Cout, generator=gen).to(device)
            what = "upfold random #%d B%d %dx%d %d+%d->%d %s" % (it, B, Hh, W, C0, C1, Cout, act)
            wp, wd = H.pack_weight_both(wt)
            fold = H.upfold_pack(wt, C0)
            assert_close(H.conv_forward(g, x0, x1, wp, bs, act, wfold=fold[0]), H.conv_forward(g, x0, x1, wp, bs, act),
                         rtol=1e-4, atol=1e-5, what=what + " forward")
            ys = torch.nn.functional.elu(torch.randn(B, h2, w2, C0, generator=gen)).to(device)
            a0, a1 = H.conv_dgrad(g, dy, wd, wt, (Hh, W), fold=fold, actgrad=(ys, "elu"))
            b0, b1 = H.conv_dgrad(g, dy, wd, wt, (Hh, W), actgrad=(ys, "elu"))
            assert_close(a0, b0, rtol=1e-4, atol=1e-5, what=what + " d/dx0")
            if C1:
                assert_close(a1, b1, rtol=1e-4, atol=1e-5, what=what + " d/dx1")
            dw_f = H.conv_wgrad(g, x0, x1, dy)
            saved, H.UPFOLD = H.UPFOLD, False
            try:
                dw_9 = H.conv_wgrad(g, x0, x1, dy)
            finally:
                H.UPFOLD = saved
            assert_close(dw_f, dw_9, rtol=1e-4, atol=1e-5, what=what + " dW")
    finally:
        H.UPFOLD_MIN_SAVED_MACS = min_macs
    took = {k: H.UPFOLD_TAKEN[k] - t0[k] for k in t0}
    assert took["fwd"] >= n // 2 and took["dgrad"] >= n // 2 and took["wgrad"] >= n // 6, took


def run_depthmix_teacher_cases(device):
    """teacher softmax (train.py:666) and online-depth normalisation (train.py:690-697) kernels vs the torch ops"""
    from improving_segmentation_with_selfsupervised_depth_amd import trainer as T
    gen = torch.Generator().manual_seed(5)
    for (B, C, Hh, W) in ((2, 19, 24, 40), (1, 19, 7, 37), (2, 5, 16, 16)):
        lg = (torch.randn(B, Hh, W, C, generator=gen) * 4).to(device)
        want = torch.softmax(lg.cpu().permute(0, 3, 1, 2), dim=1)
        got = T.teacher_softmax(lg.permute(0, 3, 1, 2))
        assert got.is_contiguous() and tuple(got.shape) == (B, C, Hh, W)
        assert_close(got, want, rtol=2e-6, atol=1e-9, what="teacher softmax")
        wide = (torch.randn(B, Hh, W, C + 13, generator=gen) * 4).to(device)       # a channel slice of a wider buffer
        got = H.softmax_to_nchw(wide[..., 3:3 + C])
        assert_close(got, torch.softmax(wide.cpu()[..., 3:3 + C].permute(0, 3, 1, 2), dim=1), rtol=2e-6, atol=1e-9, what="softmax slice")
        got = T.teacher_softmax(lg.permute(0, 3, 1, 2).contiguous())                 # dense NCHW logits take the layout kernel
        assert_close(got, want, rtol=2e-6, atol=1e-9, what="teacher softmax (NCHW in)")
    for shape in ((2, 1, 33, 65), (3, 1, 8, 8), (1, 1, 300, 700)):
        d = torch.rand(shape, generator=gen).to(device)
        want = d.cpu().clone()
        for j in range(shape[0]):
            lo, hi = want[j].min(), want[j].max()
            want[j] = (torch.clamp(want[j], lo, hi) - lo) / (hi - lo)
        got = T.normalize_online_depth(d)
        assert torch.equal(got.cpu(), want), "min-max normalised disparity must be bit-exact"


def run_valtail_kernel_cases(device, golden):
    """generate_depth_test_pred and the 8-bit depth export vs the reference's vectors"""
    from improving_segmentation_with_selfsupervised_depth_amd.loss import get_monodepth_loss
    g = golden("valtail")
    B, _, Hh, W = g["rnd_disp_0"].shape
    cfg = {"training": {"batch_size": B, "monodepth_loss": dict(
        num_scales=4, frame_ids=[0, -1, 1], height=Hh, width=W, min_depth=0.1, max_depth=100, test_min_depth=1e-3,
        test_max_depth=80, disparity_smoothness=1e-3, no_ssim=False, avg_reprojection=False, disable_automasking=False)}}
    lo = get_monodepth_loss(cfg, is_train=False)
    out = {("disp", s): g["rnd_disp_%d" % s].to(device) for s in range(4)}
    lo.generate_depth_test_pred(out)
    for s in range(4):
        assert_close(out[("depth", 0, s)], g["rnd_depth_%d" % s], rtol=1e-5, atol=0, what="test depth %d" % s)
    u8 = H.minmax_normalize(g["disp_0"].to(device), as_uint8=True)
    assert torch.equal(u8[:, 0].cpu(), g["export_u8"]), "8-bit depth estimate must be bit-exact"


def run_conv_actgrad_cases(device):
    """data-gradient with the activation backward of its input fused into the epilogue == plain data-gradient times
    act'(saved output), on every route: staged epilogue (LDS-DMA / reflection-adjoint loops), fused 2x2 upsample sum,
    two-source split, generic scalar epilogue, stencil kernels (strip and per-pixel), shapes that fall back to a second pass"""
    gen = torch.Generator().manual_seed(11)
    cases = [  # C0, C1, up0, Cout, k, dil, pad, reflect, (H, W) of the virtual input
        (32, 0, False, 64, 3, 1, 1, True, (10, 12)), (32, 0, True, 32, 3, 1, 1, True, (12, 16)),
        (32, 32, True, 64, 3, 1, 1, True, (8, 16)), (64, 0, False, 64, 1, 1, 0, False, (9, 7)),
        (12, 0, False, 20, 3, 1, 1, False, (7, 9)), (8, 8, True, 16, 3, 1, 1, True, (8, 8)),
        (16, 0, False, 1, 3, 1, 1, True, (6, 10)), (64, 0, False, 1, 3, 1, 1, True, (8, 32)), (64, 0, False, 1, 3, 1, 1, False, (8, 32)),
    ]
    for (C0, C1, up0, Cout, k, dil, pad, reflect, (Hh, W)) in cases:
        g = H.ConvGeom(C0, Cout, k, 1, dil, pad, reflect, C1, up0)
        wt = (torch.randn(Cout, C0 + C1, k, k, generator=gen) * 0.1).to(device)
        wd = H.pack_weight(wt, True)
        dy = torch.randn(2, Hh, W, Cout, generator=gen).to(device)
        h0, w0 = (Hh // 2, W // 2) if up0 else (Hh, W)
        for kind in ("elu", "relu", "sigmoid"):
            y = torch.randn(2, h0, w0, C0, generator=gen)
            y = {"elu": torch.nn.functional.elu(y), "relu": torch.relu(y), "sigmoid": torch.sigmoid(y)}[kind].to(device)
            d0, d1 = H.conv_dgrad(g, dy, wd, wt, (Hh, W))
            f0, f1 = H.conv_dgrad(g, dy, wd, wt, (Hh, W), actgrad=(y, kind))
            der = {"elu": torch.where(y > 0, torch.ones_like(y), y + 1), "relu": (y > 0).float(), "sigmoid": y * (1 - y)}[kind]
            what = "actgrad %s C0=%d C1=%d up=%s Cout=%d k=%d refl=%s" % (kind, C0, C1, up0, Cout, k, reflect)
            assert_close(f0, d0 * der, rtol=1e-6, atol=1e-6, what=what)
            if C1:
                assert torch.equal(f1, d1), what + " (second source untouched)"
        if not up0 and not C1:
            # accumulate epilogue: y_acc += act'(y) * dgrad
            base = torch.randn(2, Hh, W, C0, generator=gen).to(device)
            acc = base.clone()
            r0, _ = H.conv_dgrad(g, dy, wd, wt, (Hh, W), accumulate_into=acc, actgrad=(y, kind))
            if r0 is not None:
                assert_close(r0, base + d0 * der, rtol=1e-6, atol=1e-6, what=what + " accumulate")


def run_fused_photometric_vs_stage(device):
    """fused per-scale photometric kernels == the per-stage kernel chain (which the golden vectors pin), at sizes that
    are not multiples of the 32x8 tile, with a lower-resolution disparity, for every flag combination"""
    from oracle import geometry as G
    d = lambda t: None if t is None else t.to(device)
    for (B, Hh, W, hs, ws) in ((1, 13, 37, 7, 19), (2, 9, 70, 9, 70), (1, 17, 33, 5, 9)):
        for (no_ssim, avg, automask) in ((False, False, True), (True, False, True), (False, True, True), (False, False, False)):
            gen = torch.Generator().manual_seed(Hh * 100 + W)
            tgt = d(torch.rand(B, 3, Hh, W, generator=gen))
            srcs = [d(torch.rand(B, 3, Hh, W, generator=gen)) for _ in range(2)]
            disp = d(0.05 + 0.9 * torch.rand(B, 1, hs, ws, generator=gen))
            K = torch.tensor([[1.1 * W, 0, 0.5 * W, 0], [0, 1.1 * W, 0.5 * Hh, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]]).repeat(B, 1, 1)
            iK, K = d(torch.linalg.pinv(K)), d(K)
            Ts = [d(G.pose_matrix(0.02 * torch.randn(B, 1, 3, generator=gen), 0.05 * torch.randn(B, 1, 3, generator=gen),
                                  invert=(j == 0))) for j in range(2)]
            noise = d(torch.randn(B, 1 if avg else 2, Hh, W, generator=gen)) if automask else None
            cols = [H.warp_forward(disp, iK, K, Ts[j], srcs[j], 0.1, 100.0)[0] for j in range(2)]
            ident = torch.empty(B, 2, Hh, W, device=device) if automask else None
            reproj = torch.empty(B, 2, Hh, W, device=device)
            for j in range(2):
                if automask:
                    H.reprojection_error(srcs[j], tgt, no_ssim, ident[:, j])
                H.reprojection_error(cols[j], tgt, no_ssim, reproj[:, j])
            ssum0, sel0, isel0 = H.automask_min(ident, noise, reproj, avg)
            scale = 1.0 / (B * Hh * W)
            greproj = H.automask_min_backward(sel0, automask, 2, avg, scale)
            gup0, gT0 = torch.zeros(B, Hh, W, device=device), []
            for j in range(2):
                gpred = H.reprojection_error_backward(cols[j], tgt, greproj[:, j], no_ssim)
                gT = torch.zeros(B, 4, 4, device=device)
                H.warp_backward(gpred, disp, iK, K, Ts[j], srcs[j], 0.1, 100.0, gup0, gT)
                gT0.append(gT)
            what = "fused photometric %s no_ssim=%s avg=%s automask=%s" % ((B, Hh, W), no_ssim, avg, automask)
            ident1 = H.photometric_identity(srcs[0], srcs[1], tgt, no_ssim) if automask else None
            ssum1, sel1, isel1 = H.photometric_forward(cols[0], cols[1], tgt, ident1, noise, no_ssim, avg)
            assert torch.equal(sel1, sel0), what
            assert not automask or (torch.equal(isel1, isel0) and torch.equal(ident1, ident)), what
            assert_close(ssum1, ssum0, rtol=1e-6, what=what + " sum")
            gT1 = [torch.zeros(B, 4, 4, device=device) for _ in range(2)]
            gup1 = H.photometric_backward(cols[0], cols[1], tgt, sel1, automask, disp, iK, K, Ts[0], Ts[1], srcs[0], srcs[1], 0.1,
                                          100.0, no_ssim, avg, scale, None, gT1[0], gT1[1])
            assert_close(gup1, gup0, rtol=1e-4, atol=1e-5 * float(gup0.abs().max()), what=what + " d disp")
            for j in range(2):
                assert_close(gT1[j], gT0[j], rtol=1e-4, atol=1e-5 * float(gT0[j].abs().max()), what=what + " dT%d" % j)


def run_augment_cases(device):
    """strongTransform's colour jitter / blur kernels vs the (parity-unpinned) torch restatement of kornia 0.4.0 in
    oracle/augment.py, plus properties that hold whatever kornia's exact rounding is"""
    import math
    from oracle import augment as A
    from improving_segmentation_with_selfsupervised_depth_amd.loader import transformsgpu as TG
    gen = torch.Generator().manual_seed(31)
    B, Hh, W = 3, 40, 72
    x = torch.rand(B, 3, Hh, W, generator=gen)
    x[0, :, :4] = 0.0                     # black pixels (s = 0 / 0), grey pixels (min == max), saturated pixels
    x[1, :, :4] = 0.37
    x[2, 0, 5:9] = 1.0
    xd = x.to(device)
    # ---- jitter: every order of the four adjustments, factors at and inside the range limits
    import itertools
    for n, order in enumerate(itertools.permutations(range(4))):
        if n % 5 and n not in (1, 23):
            continue
        params, _ = TG.sample_color_jitter_params(B, 0.25, generator=gen)
        got, _ = TG.color_jitter(0.9, data=xd, params=params, order=list(order))
        want = A.color_jitter(x, params, list(order))
        assert_close(got, want, rtol=1e-5, atol=2e-6, what="colour jitter order %s" % (order,))
        assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    # factor 1 / hue 0: the identity up to the HSV round trip
    ident = torch.tensor([[1.0, 1.0, 1.0, 0.0]]).repeat(B, 1)
    got, _ = TG.color_jitter(0.9, data=xd, params=ident, order=[0, 1, 2, 3])
    assert_close(got, x, rtol=1e-5, atol=2e-6, what="jitter with neutral factors")
    # below the 0.2 switch and for non-RGB data the call is the identity (transformsgpu.py:13-14)
    assert TG.color_jitter(0.2, data=xd)[0] is xd
    soft = torch.rand(B, 19, 8, 8).to(device)
    assert TG.color_jitter(0.9, data=soft)[0] is soft
    # a hue shift of a full turn-fraction changes the hue only: value (max channel) is preserved
    hp = torch.tensor([[1.0, 1.0, 1.0, 0.2]]).repeat(B, 1)
    got, _ = TG.color_jitter(0.9, data=xd, params=hp, order=[3, 0, 1, 2])
    assert_close(got.max(1)[0], x.max(1)[0], rtol=1e-5, atol=2e-6, what="hue shift keeps the value channel")
    # sampled parameters lie in kornia's ranges
    pr, od = TG.sample_color_jitter_params(512, 0.25, generator=gen)
    assert float(pr[:, :3].min()) >= 0.75 and float(pr[:, :3].max()) <= 1.25 and float(pr[:, 3].abs().max()) <= 0.25
    assert sorted(od) == [0, 1, 2, 3]
    # ---- blur
    for (hh, ww, sigma) in ((40, 72, 0.15), (40, 72, 1.15), (64, 128, 0.7), (33, 50, 1.0)):
        xi = torch.rand(2, 3, hh, ww, generator=gen)
        ky, kx = TG.blur_kernel_size(hh), TG.blur_kernel_size(ww)
        assert ky % 2 == 1 and kx % 2 == 1
        got, _ = TG.gaussian_blur(0.9, data=xi.to(device), sigma=sigma)
        want = A.gaussian_blur(xi, (ky, kx), sigma)
        assert_close(got, want, rtol=1e-5, atol=2e-6, what="gaussian blur %dx%d sigma %.2f" % (hh, ww, sigma))
        taps = TG.gaussian_taps(kx, sigma)
        assert abs(float(taps.sum()) - 1.0) < 1e-6 and taps.numel() % 2 == 1
        const = torch.full((1, 3, hh, ww), 0.625).to(device)
        got, _ = TG.gaussian_blur(0.9, data=const, sigma=sigma)
        assert_close(got, const.cpu(), rtol=1e-6, atol=1e-6, what="blur of a constant image")
    assert TG.gaussian_blur(0.5, data=xd)[0] is xd
    assert TG.blur_kernel_size(512) == 51 and TG.blur_kernel_size(1024) == 103 and TG.blur_kernel_size(2048) == 205


def run_metric_cases(device, golden):
    """runningScore mirror (device-resident confusion matrix) vs the reference's vectors: exact"""
    import numpy as np
    from improving_segmentation_with_selfsupervised_depth_amd.evaluation.metrics import runningScore
    g = golden("trainer")
    gt, pred, logits = g["cm_gt"].to(device), g["cm_pred"].to(device), g["cm_logits"].to(device)
    want = g["cm_matrix"].numpy()
    rs = runningScore(19)
    rs.update(gt, pred)
    rs.update(gt[:1], pred[:1])
    assert np.array_equal(rs.confusion_matrix, want), "confusion matrix from predictions"
    sc, cls_iu = rs.get_scores()
    np.testing.assert_allclose([sc["Overall Acc: \t"], sc["Mean Acc : \t"], sc["FreqW Acc : \t"], sc["Mean IoU : \t"]],
                               g["cm_scores"].numpy(), rtol=1e-12)
    np.testing.assert_allclose([cls_iu[i] for i in range(19)], g["cm_cls_iu"].numpy(), rtol=1e-12, equal_nan=True)
    # fused argmax, NCHW and channels-last logits; numpy inputs take the host path
    for lg in (logits, logits.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)):
        r2 = runningScore(19)
        r2.update_from_logits(gt, lg)
        r2.update(g["cm_gt"][:1].numpy(), g["cm_pred"][:1].numpy())
        assert np.array_equal(r2.confusion_matrix, want), "confusion matrix from logits"
    r2.reset()
    assert r2.confusion_matrix.sum() == 0


def run_residual_fusion_case(device):
    """Residual blocks with an identity skip: the first conv's data-gradient is accumulated onto the skip-path gradient in
    the kernel epilogue (functional.SplitFn).  Same input gradient as the plain autograd sum, and the fused path is taken."""
    from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn
    from improving_segmentation_with_selfsupervised_depth_amd.models.resnet_encoder import BasicBlock, Bottleneck
    gen = torch.Generator().manual_seed(5)
    for blk, cin in ((Bottleneck(32, 8), 32), (BasicBlock(16, 16), 16)):
        for p in blk.parameters():
            p.data = (torch.randn(p.shape, generator=gen) * 0.3).to(p.dtype)
        blk.to(device).train()
        x0 = torch.randn(2, 6, 10, cin, generator=gen).to(device)
        gy = torch.randn(2, 6, 10, cin, generator=gen).to(device)

        def grads(fused):
            leaf = x0.clone().requires_grad_(True)
            x = leaf * 1.0 if fused else leaf.detach().clone().requires_grad_(True)   # requires_grad either way
            if not fused:
                # plain path: call the block's pieces without the SplitFn (what forward does when a downsample exists)
                idt = x
                o = blk.bn1(blk.conv1(x), act="relu")
                if hasattr(blk, "conv3"):
                    o = blk.bn2(blk.conv2(o), act="relu")
                    y = blk.bn3(blk.conv3(o), residual=idt, act="relu")
                else:
                    y = blk.bn2(blk.conv2(o), residual=idt, act="relu")
                leaf = x
            else:
                y = blk(x)
            for q in blk.parameters():
                q.grad = None
            (y * gy).sum().backward()
            return leaf.grad.clone(), [q.grad.clone() for q in blk.parameters()]
        n0 = Fn.SplitFn.fused_count
        gx_f, gp_f = grads(True)
        assert Fn.SplitFn.fused_count == n0 + 1, "the in-kernel accumulation was not taken"
        gx_p, gp_p = grads(False)
        assert_close(gx_f, gx_p, rtol=1e-5, atol=1e-6, what="residual fusion dx")
        for a, b in zip(gp_f, gp_p):
            assert_close(a, b, rtol=1e-5, atol=1e-6, what="residual fusion param grad")


# ---------------------------------------------------------------------------------------------
# the stand-alone layers of models/monodepth_layers.py (SSIM, get_smooth_loss, BackprojectDepth, Project3D, upsample,
# rot_from_axisangle, get_translation_matrix) against the reference's own vectors (geom.npz, ssim_smooth.npz)
# ---------------------------------------------------------------------------------------------
def run_monodepth_layer_callables(device, golden):
    from improving_segmentation_with_selfsupervised_depth_amd.models import monodepth_layers as ML
    d = lambda t: t.to(device)
    g = golden("ssim_smooth")
    ssim = ML.SSIM()
    x = d(g["x"]).clone().requires_grad_(True)
    v = ssim(x, d(g["y"]))
    assert_close(v, g["ssim"], rtol=1e-4, atol=1e-6, what="SSIM map")
    (v * d(g["w"])).sum().backward()
    assert_close(x.grad, g["grad_x"], rtol=1e-3, atol=2e-5, what="SSIM map adjoint")
    assert_close(ssim(d(g["x"]), d(g["y2"])), g["ssim2"], rtol=1e-4, atol=1e-6, what="SSIM map (clamp branch)")
    # symmetry: the adjoint w.r.t. the second image of SSIM(y, x) equals the one w.r.t. the first of SSIM(x, y)
    y = d(g["x"]).clone().requires_grad_(True)
    (ssim(d(g["y"]), y) * d(g["w"])).sum().backward()
    assert_close(y.grad, g["grad_x"], rtol=1e-3, atol=2e-5, what="SSIM map adjoint (second argument)")
    disp = d(g["sm_disp"]).clone().requires_grad_(True)
    sm = ML.get_smooth_loss(disp, d(g["sm_img"]))
    assert_close(sm, g["smooth"], rtol=1e-5, what="get_smooth_loss")
    (3.0 * sm).backward()
    assert_close(disp.grad, 3.0 * g["grad_sm_disp"], rtol=1e-4, atol=1e-7, what="get_smooth_loss adjoint")
    q = golden("geom")
    B, _, Hh, W = q["depth"].shape
    cam = ML.BackprojectDepth(B, Hh, W)(d(q["depth"]), d(q["inv_K"]))
    assert_close(cam, q["cam_points"], rtol=1e-5, atol=1e-6, what="BackprojectDepth")
    grid = ML.Project3D(B, Hh, W)(d(q["cam_points"]), d(q["K"]), d(q["T"]))
    assert_close(grid, q["grid"], rtol=1e-4, atol=1e-5, what="Project3D")
    # the two layers are differentiable like the reference's torch code: depth -> points -> grid, gradients w.r.t. the
    # depth and the pose against the oracle's restatement under torch autograd (float64)
    from oracle import geometry as OG
    bp, pj = ML.BackprojectDepth(B, Hh, W), ML.Project3D(B, Hh, W)
    assert sorted(k for k, _ in bp.named_parameters()) == ["id_coords", "ones", "pix_coords"]
    assert tuple(bp.pix_coords.shape) == (B, 3, Hh * W) and not bp.pix_coords.requires_grad
    gen0 = torch.Generator().manual_seed(11)
    wgt0 = torch.randn(B, Hh, W, 2, generator=gen0)
    dep = d(q["depth"]).clone().requires_grad_(True)
    Tm = d(q["T"]).clone().requires_grad_(True)
    (pj(bp(dep, d(q["inv_K"])), d(q["K"]), Tm) * d(wgt0)).sum().backward()
    dep_o = q["depth"].double().clone().requires_grad_(True)
    T_o = q["T"].double().clone().requires_grad_(True)
    (OG.project(OG.backproject(dep_o, q["inv_K"].double()), q["K"].double(), T_o, Hh, W) * wgt0.double()).sum().backward()
    assert_close(dep.grad, dep_o.grad.float(), rtol=2e-4, atol=1e-6 * float(dep_o.grad.abs().max()), what="d grid / d depth")
    assert_close(Tm.grad[:, :3], T_o.grad.float()[:, :3], rtol=2e-4, atol=1e-5 * float(T_o.grad.abs().max()), what="d grid / d T")
    try:
        pj(bp(dep, d(q["inv_K"])), d(q["K"]).clone().requires_grad_(True), Tm)
        raise AssertionError("a gradient for the intrinsics must be refused, not dropped")
    except NotImplementedError:
        pass
    R = ML.rot_from_axisangle(d(q["axisangle"]))
    T = ML.get_translation_matrix(d(q["translation"]))
    assert_close(torch.matmul(T.cpu(), R.cpu()), q["M_fwd"], rtol=1e-5, atol=1e-6, what="T @ R == transformation_from_parameters")
    assert torch.equal(R[:, 3].cpu(), torch.tensor([0.0, 0, 0, 1]).expand(B, 4)) and torch.equal(T[:, :3, :3].cpu(), torch.eye(3).expand(B, 3, 3))
    gen = torch.Generator().manual_seed(3)
    z = torch.randn(2, 5, 4, 6, generator=gen)
    zu = d(z).clone().requires_grad_(True)
    up = ML.upsample(zu)
    assert torch.equal(up.cpu(), torch.nn.functional.interpolate(z, scale_factor=2, mode="nearest"))
    wgt = torch.randn(up.shape, generator=gen)
    (up * d(wgt)).sum().backward()
    assert_close(zu.grad, wgt.reshape(2, 5, 4, 2, 6, 2).sum((3, 5)), rtol=1e-6, atol=1e-6, what="upsample adjoint")


def run_jitter_blur_properties(device):
    """Properties that hold for ANY faithful implementation of kornia 0.4.0's ColorJitter / GaussianBlur2d (the restated
    operators are parity-unpinned, DESIGN.md 4: these narrow what can be wrong without a kornia wheel): zero-strength identity,
    hue periodicity in whole turns, grey pixels untouched by hue / saturation, brightness / contrast commute with clamping on
    unsaturated pixels, blur taps normalised, separable passes == the 2-D outer-product kernel, blur commutes with flips."""
    import math
    from improving_segmentation_with_selfsupervised_depth_amd.loader import transformsgpu as TG
    gen = torch.Generator().manual_seed(41)
    B, Hh, W = 3, 40, 56
    x = torch.rand(B, 3, Hh, W, generator=gen)
    xd = x.to(device)
    # s = 0: every factor is neutral whatever the draw, every order is the identity
    p0, _ = TG.sample_color_jitter_params(B, 0.0, generator=gen)
    assert torch.equal(p0, torch.tensor([1.0, 1.0, 1.0, 0.0]).expand(B, 4))
    for order in ([0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1]):
        got, _ = TG.color_jitter(0.9, data=xd, params=p0, order=order)
        assert_close(got, x, rtol=1e-5, atol=2e-6, what="zero-strength jitter, order %s" % (order,))
    # hue: a whole turn (factor +-1 = 2 pi) is the identity; two half turns compose to the identity
    for f in (1.0, -1.0, 2.0):
        hp = torch.tensor([1.0, 1.0, 1.0, f]).expand(B, 4).contiguous()
        got, _ = TG.color_jitter(0.9, data=xd, params=hp, order=[3, 0, 1, 2])
        assert_close(got, x, rtol=1e-4, atol=1e-5, what="hue shift by %g turns" % f)
    half = torch.tensor([1.0, 1.0, 1.0, 0.5]).expand(B, 4).contiguous()
    once, _ = TG.color_jitter(0.9, data=xd, params=half, order=[3, 0, 1, 2])
    twice, _ = TG.color_jitter(0.9, data=once, params=half, order=[3, 0, 1, 2])
    assert_close(twice, x, rtol=1e-4, atol=2e-5, what="two half turns of hue")
    assert float((once.cpu() - x).abs().max()) > 0.05                 # ... and half a turn is not the identity
    # hue keeps max(r, g, b) (the V channel) and min (V (1 - S)) of every pixel
    q = torch.tensor([1.0, 1.0, 1.0, 0.21]).expand(B, 4).contiguous()
    hq, _ = TG.color_jitter(0.9, data=xd, params=q, order=[3, 0, 1, 2])
    assert_close(hq.max(1)[0], x.max(1)[0], rtol=1e-5, atol=2e-6, what="hue keeps the value channel")
    assert_close(hq.min(1)[0], x.min(1)[0], rtol=1e-4, atol=1e-5, what="hue keeps value * (1 - saturation)")
    # grey pixels (r = g = b): hue and saturation leave them alone
    grey = x[:, :1].repeat(1, 3, 1, 1)
    gp = torch.tensor([1.0, 1.0, 1.37, -0.33]).expand(B, 4).contiguous()
    gg, _ = TG.color_jitter(0.9, data=grey.to(device), params=gp, order=[2, 3, 0, 1])
    assert_close(gg, grey, rtol=1e-5, atol=2e-6, what="grey under hue + saturation")
    # saturation 0 -> grey at the value channel; brightness / contrast are exact affine maps before clamping
    sp = torch.tensor([1.0, 1.0, 0.0, 0.0]).expand(B, 4).contiguous()
    s0, _ = TG.color_jitter(0.9, data=xd, params=sp, order=[2, 0, 1, 3])
    assert_close(s0, x.max(1, keepdim=True)[0].repeat(1, 3, 1, 1), rtol=1e-5, atol=2e-6, what="saturation factor 0")
    bp = torch.tensor([1.2, 0.9, 1.0, 0.0]).expand(B, 4).contiguous()
    bc, _ = TG.color_jitter(0.9, data=xd, params=bp, order=[0, 1, 2, 3])
    assert_close(bc, torch.clamp(torch.clamp(x + 0.2, 0, 1) * 0.9, 0, 1), rtol=1e-5, atol=2e-6, what="brightness then contrast")
    # ---- blur: taps are a normalised, symmetric, single-peaked Gaussian cut to its fp32 support
    for k, sigma in ((5, 0.15), (5, 1.15), (51, 0.6), (103, 1.15), (205, 0.15)):
        g = TG.gaussian_taps(k, sigma)
        assert g.numel() % 2 == 1 and g.numel() <= k and abs(float(g.double().sum()) - 1.0) < 1e-6
        assert torch.equal(g, g.flip(0)) and int(g.argmax()) == g.numel() // 2 and float(g.min()) > 0
        full = torch.exp(-(torch.arange(k, dtype=torch.float32) - k // 2) ** 2 / float(2 * sigma ** 2))
        full = full / full.sum()
        r = g.numel() // 2
        assert torch.equal(g, full[k // 2 - r:k // 2 + r + 1])
        assert float(full[:k // 2 - r].abs().sum()) == 0.0 and float(full[k // 2 + r + 1:].abs().sum()) == 0.0   # only zeros are cut off
    # separable passes == ONE 2-D correlation with the outer-product kernel on the reflect-padded image (kornia's filter2D)
    sigma = 0.8
    ky, kx = TG.blur_kernel_size(Hh), TG.blur_kernel_size(W)
    got, _ = TG.gaussian_blur(0.9, data=xd, sigma=sigma)
    gy = torch.exp(-(torch.arange(ky, dtype=torch.float64) - ky // 2) ** 2 / (2 * sigma ** 2))
    gx = torch.exp(-(torch.arange(kx, dtype=torch.float64) - kx // 2) ** 2 / (2 * sigma ** 2))
    k2 = torch.outer(gy / gy.sum(), gx / gx.sum())
    xp = torch.nn.functional.pad(x.double(), (kx // 2, kx // 2, ky // 2, ky // 2), mode="reflect")
    want = torch.nn.functional.conv2d(xp, k2[None, None].repeat(3, 1, 1, 1), groups=3)
    assert_close(got, want.float(), rtol=1e-5, atol=2e-6, what="separable blur vs the 2-D kernel (float64)")
    # mean-preserving up to the border handling; commutes with flips (symmetric taps, reflection border)
    fl, _ = TG.gaussian_blur(0.9, data=xd.flip(2).flip(3).contiguous(), sigma=sigma)
    assert_close(fl.flip(2).flip(3), got.cpu(), rtol=1e-6, atol=1e-6, what="blur commutes with flips")
    assert float(got.min()) >= float(x.min()) - 1e-6 and float(got.max()) <= float(x.max()) + 1e-6   # a convex combination
    # the kernel size rule of the reference (transformsgpu.py:26-27): odd, within one of 0.1 n
    for n in (40, 56, 100, 512, 513, 1024, 2048):
        k = TG.blur_kernel_size(n)
        assert k % 2 == 1 and abs(k - 0.1 * n) <= 1.0 + 1e-9, (n, k)


# ---------------------------------------------------------------------------------------------
# Winograd F(2x2,3x3) route (csrc/winograd.hip + the grouped position GEMMs) against float64 and against the direct kernel
# ---------------------------------------------------------------------------------------------
def run_winograd_cases(device, shapes=((2, 16, 32, 64, 64), (1, 32, 16, 96, 128), (2, 8, 64, 128, 64), (1, 12, 20, 64, 64))):
    """forward (zero and mirrored padding, with the BatchNorm statistics partials), data-gradient and the autograd glue of the
    Winograd route: error against a float64 convolution within 3x the direct kernel's own (plus 1e-6 of the largest value),
    statistics partials summing to the column sums of the output, transformed packs cached per weight_pack_scope."""
    from improving_segmentation_with_selfsupervised_depth_amd.models import layers as L
    from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn
    old = (H.WINOGRAD_MIN_CH, H.WINOGRAD_MIN_MACS)
    old_fused = H.WINO_FUSED
    H.WINOGRAD_MIN_CH, H.WINOGRAD_MIN_MACS = 32, 0.0
    H.WINO_FUSED = False          # this case is about the grouped-GEMM route (run_winograd_fused_cases: the one-kernel route)
    try:
        gen = torch.Generator().manual_seed(23)
        for (B, Hh, W, C, Co) in shapes:
            x = torch.randn(B, C, Hh, W, generator=gen)
            w = torch.randn(Co, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5
            dy = torch.randn(B, Co, Hh, W, generator=gen)
            xd, wd_, dyd = nhwc(x).to(device).contiguous(), w.to(device), nhwc(dy).to(device).contiguous()
            uf, ud = H.winograd_pack(wd_)
            wp, wdp = H.pack_weight_both(wd_)
            for refl in (False, True):
                g = H.ConvGeom(C, Co, 3, 1, 1, 1, refl, 0, False)
                assert H.winograd_ok(g, B, Hh, W)
                xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1), mode="reflect" if refl else "constant")
                want = torch.nn.functional.conv2d(xp, w.double())
                n0 = H.WINOGRAD_TAKEN["fwd"]
                y, part = H.conv_forward(g, xd, None, wp, None, want_stats=True, wino=uf)
                assert H.WINOGRAD_TAKEN["fwd"] == n0 + 1, "the Winograd route declined %s" % ((B, Hh, W, C, Co),)
                direct = H.conv_forward(g, xd, None, wp, None)
                sc = float(want.abs().max())
                e_w, e_d = float((nchw(y).double().cpu() - want).abs().max()), float((nchw(direct).double().cpu() - want).abs().max())
                assert e_w <= 3 * e_d + 1e-6 * sc, ("forward", refl, (B, Hh, W, C, Co), e_w, e_d, sc)
                rows = part.shape[0]
                assert part.shape == (rows, 2, Co)
                yy = y.double().reshape(-1, Co)
                assert_close(part[:, 0].sum(0), yy.sum(0), rtol=1e-9, atol=1e-9 * float(yy.abs().sum(0).max()), what="Winograd statistics: sums")
                assert_close(part[:, 1].sum(0), (yy * yy).sum(0), rtol=1e-9, atol=1e-9, what="Winograd statistics: sums of squares")
            # data-gradient (zero padding): conv of dy with the flipped, transposed kernel
            g = H.ConvGeom(C, Co, 3, 1, 1, 1, False, 0, False)
            want = torch.nn.functional.conv_transpose2d(dy.double(), w.double(), padding=1)
            n0 = H.WINOGRAD_TAKEN["dgrad"]
            dx, _ = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W), wino=ud)
            assert H.WINOGRAD_TAKEN["dgrad"] == n0 + (1 if C % 64 == 0 else 0)     # 96 gradient channels: the direct route
            dxd, _ = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W))
            sc = float(want.abs().max())
            e_w, e_d = float((nchw(dx).double().cpu() - want).abs().max()), float((nchw(dxd).double().cpu() - want).abs().max())
            assert e_w <= 3 * e_d + 1e-6 * sc, ("dgrad", (B, Hh, W, C, Co), e_w, e_d, sc)
            # weight gradient: sixteen position GEMMs over the tiles + G^T dU G (zero and mirrored padding)
            for refl in (False, True):
                g = H.ConvGeom(C, Co, 3, 1, 1, 1, refl, 0, False)
                xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1), mode="reflect" if refl else "constant")
                want = torch.nn.grad.conv2d_weight(xp, w.shape, dy.double())
                n0 = H.WINOGRAD_TAKEN["wgrad"]
                dw = H.conv_wgrad(g, xd, None, dyd)
                assert H.WINOGRAD_TAKEN["wgrad"] == n0 + 1
                H.WINOGRAD = False
                dwd = H.conv_wgrad(g, xd, None, dyd)
                H.WINOGRAD = True
                sc = float(want.abs().max())
                e_w, e_d = float((dw.double().cpu() - want).abs().max()), float((dwd.double().cpu() - want).abs().max())
                assert e_w <= 3 * e_d + 2e-6 * sc, ("wgrad", refl, (B, Hh, W, C, Co), e_w, e_d, sc)
        # dilated window (layer4 of the dilated ResNet: dilation 2 = padding 2, four sub-lattices), forward and data-gradient
        B, Hh, W, C, Co = 2, 16, 32, 64, 64
        x = torch.randn(B, C, Hh, W, generator=gen)
        w = torch.randn(Co, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5
        dy = torch.randn(B, Co, Hh, W, generator=gen)
        xd, wd_, dyd = nhwc(x).to(device).contiguous(), w.to(device), nhwc(dy).to(device).contiguous()
        uf, ud = H.winograd_pack(wd_)
        wp, wdp = H.pack_weight_both(wd_)
        g = H.ConvGeom(C, Co, 3, 1, 2, 2, False, 0, False)
        n0 = dict(H.WINOGRAD_TAKEN)
        y, part = H.conv_forward(g, xd, None, wp, None, want_stats=True, wino=uf)
        dx, _ = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W), wino=ud)
        assert (H.WINOGRAD_TAKEN["fwd"], H.WINOGRAD_TAKEN["dgrad"]) == (n0["fwd"] + 1, n0["dgrad"] + 1)
        want = torch.nn.functional.conv2d(x.double(), w.double(), padding=2, dilation=2)
        assert_close(nchw(y), want.float(), rtol=1e-4, atol=2e-6 * float(want.abs().max()), what="Winograd, dilation 2: forward")
        assert_close(part[:, 0].sum(0), y.double().reshape(-1, Co).sum(0), rtol=1e-9, atol=1e-9, what="Winograd, dilation 2: statistics")
        want = torch.nn.functional.conv_transpose2d(dy.double(), w.double(), padding=2, dilation=2)
        assert_close(nchw(dx), want.float(), rtol=1e-4, atol=2e-6 * float(want.abs().max()), what="Winograd, dilation 2: data-gradient")
        n0 = H.WINOGRAD_TAKEN["wgrad"]
        dw = H.conv_wgrad(g, xd, None, dyd)
        assert H.WINOGRAD_TAKEN["wgrad"] == n0 + 1
        want = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=2, dilation=2)
        assert_close(dw, want.float(), rtol=1e-4, atol=3e-6 * float(want.abs().max()), what="Winograd, dilation 2: weight gradient")
        # the decoder's Conv3x3 on [x | skip] at one resolution: two sources, mirrored padding, bias + ELU in the output transform
        C0, C1 = 32, 96
        x0, x1 = torch.randn(B, C0, Hh, W, generator=gen), torch.randn(B, C1, Hh, W, generator=gen)
        w = torch.randn(Co, C0 + C1, 3, 3, generator=gen) * (2.0 / (9 * (C0 + C1))) ** 0.5
        bias = torch.randn(Co, generator=gen) * 0.1
        g = H.ConvGeom(C0, Co, 3, 1, 1, 1, True, C1, False)
        uf, _ = H.winograd_pack(w.to(device))
        wp = H.pack_weight(w.to(device))
        n0 = H.WINOGRAD_TAKEN["fwd"]
        y = H.conv_forward(g, nhwc(x0).to(device).contiguous(), nhwc(x1).to(device).contiguous(), wp, bias.to(device), act="elu", wino=uf)
        assert H.WINOGRAD_TAKEN["fwd"] == n0 + 1
        want = torch.nn.functional.elu(torch.nn.functional.conv2d(
            torch.nn.functional.pad(torch.cat([x0, x1], 1).double(), (1, 1, 1, 1), mode="reflect"), w.double(), bias.double()))
        assert_close(nchw(y), want.float(), rtol=1e-4, atol=2e-6 * float(want.abs().max()), what="Winograd: two sources + mirrored padding + bias + ELU")
        dyc = torch.randn(B, Co, Hh, W, generator=gen)
        n0 = H.WINOGRAD_TAKEN["wgrad"]
        dw = H.conv_wgrad(g, nhwc(x0).to(device).contiguous(), nhwc(x1).to(device).contiguous(), nhwc(dyc).to(device).contiguous())
        assert H.WINOGRAD_TAKEN["wgrad"] == n0 + 1
        want = torch.nn.grad.conv2d_weight(torch.nn.functional.pad(torch.cat([x0, x1], 1).double(), (1, 1, 1, 1), mode="reflect"),
                                           w.shape, dyc.double())
        assert_close(dw, want.float(), rtol=1e-4, atol=3e-6 * float(want.abs().max()), what="Winograd: two sources, weight gradient")
        # a weight_pack_scope(model) transforms every eligible weight in one launch; the packs equal the one-by-one ones
        net = torch.nn.Sequential(L.Conv2d(64, 64, 3, padding=1, bias=False), L.Conv2d(64, 128, 3, padding=2, dilation=2, bias=True),
                                  L.Conv2d(64, 64, 1), L.Conv2d(64, 64, 3, stride=2, padding=1)).to(device)
        with L.weight_pack_scope(net):
            assert net[0]._wino_cache.get("packs") is not None and net[1]._wino_cache.get("packs") is not None
            assert not net[2]._wino_cache.get("packs") and not net[3]._wino_cache.get("packs")
            for m in (net[0], net[1]):
                uf1, ud1 = H.winograd_pack(m.weight)
                assert torch.equal(m._wino_cache["packs"][0], uf1) and torch.equal(m._wino_cache["packs"][1], ud1)
        # autograd glue: Conv2d -> BatchNorm2d through the Winograd route == the same modules on the direct route
        B, Hh, W, C, Co = 2, 16, 32, 64, 64
        torch.manual_seed(5)
        conv, bn = L.Conv2d(C, Co, 3, padding=1, bias=False).to(device), L.BatchNorm2d(Co).to(device)
        conv.train(); bn.train()
        x = torch.randn(B, Hh, W, C, generator=gen).to(device)
        res = []
        for on in (True, False):
            H.WINOGRAD = on
            xi = x.clone().requires_grad_(True)
            conv.zero_grad(); bn.zero_grad()
            n0 = dict(H.WINOGRAD_TAKEN)
            with L.weight_pack_scope():
                y = bn(conv(xi), act="relu")
                y2 = bn(conv(xi), act="relu")          # second forward in the scope: the transformed packs are reused
                (y * y).sum().backward()
            took = (H.WINOGRAD_TAKEN["fwd"] - n0["fwd"], H.WINOGRAD_TAKEN["dgrad"] - n0["dgrad"])
            assert took == ((2, 1) if on else (0, 0)), took
            assert torch.equal(y, y2)
            res.append((y.detach(), xi.grad.detach(), conv.weight.grad.detach().clone(), bn.weight.grad.detach().clone()))
        H.WINOGRAD = True
        for a, b, what in zip(res[0], res[1], ("output", "input gradient", "weight gradient", "BatchNorm weight gradient")):
            assert_close(a, b, rtol=1e-3, atol=1e-4 * float(b.abs().max()), what="Winograd vs direct through Conv2d + BatchNorm2d: " + what)
    finally:
        H.WINOGRAD_MIN_CH, H.WINOGRAD_MIN_MACS = old
        H.WINOGRAD = True
        H.WINO_FUSED = old_fused


def run_winograd_fused_cases(device, shapes=((1, 8, 16, 64, 64), (2, 12, 20, 64, 128), (1, 16, 32, 128, 64), (1, 6, 4, 128, 128))):
    keep = H.WINO_FUSED_REFLECT_DGRAD_MIN_PIX
    H.WINO_FUSED_REFLECT_DGRAD_MIN_PIX = 0          # the router's size gate of the mirrored data-gradient: these are small maps
    try:
        _run_winograd_fused_cases(device, shapes)
    finally:
        H.WINO_FUSED_REFLECT_DGRAD_MIN_PIX = keep


def _run_winograd_fused_cases(device, shapes):
    """the one-kernel Winograd route (csrc/winograd_fused.hip) against a float64 convolution: error within 3x the direct kernel's
    own, whole and partial 4x8-tile blocks, one and two channel fills, forward with bias + ELU, with the BatchNorm statistics
    partials, and the data-gradient through the flipped pack"""
    gen = torch.Generator().manual_seed(29)
    for (B, Hh, W, C, Co) in shapes:
        x = torch.randn(B, C, Hh, W, generator=gen)
        w = torch.randn(Co, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5
        bias = torch.randn(Co, generator=gen)
        dy = torch.randn(B, Co, Hh, W, generator=gen)
        xd, wd_, dyd, bd = nhwc(x).to(device).contiguous(), w.to(device), nhwc(dy).to(device).contiguous(), bias.to(device)
        wp, wdp = H.pack_weight_both(wd_)
        g = H.ConvGeom(C, Co, 3, 1, 1, 1, False, 0, False)
        uf, ud = H.winograd_fused_pack(wd_, False), H.winograd_fused_pack(wd_, True)
        assert uf.shape == (16, C, Co) and ud.shape == (16, Co, C)
        want = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
        sc = float(want.abs().max())
        what = "fused Winograd %s" % ((B, Hh, W, C, Co),)
        y, part = H.winograd_fused("conv_fwd", xd, uf, want_stats=True)
        direct = H.conv_forward(g, xd, None, wp, None)
        e_w, e_d = float((nchw(y).double().cpu() - want).abs().max()), float((nchw(direct).double().cpu() - want).abs().max())
        assert e_w <= 3 * e_d + 1e-6 * sc, (what, "forward", e_w, e_d, sc)
        yy = y.double().reshape(-1, Co)
        assert part.shape[1:] == (2, Co)
        assert_close(part[:, 0].sum(0), yy.sum(0), rtol=1e-9, atol=1e-9 * float(yy.abs().sum(0).max()), what=what + " statistics: sums")
        assert_close(part[:, 1].sum(0), (yy * yy).sum(0), rtol=1e-9, atol=1e-9, what=what + " statistics: sums of squares")
        y2, none = H.winograd_fused("conv_fwd", xd, uf, bias=bd, act="elu")
        assert none is None
        want2 = torch.nn.functional.elu(want + bias.double().view(1, -1, 1, 1))
        assert_close(nchw(y2).cpu(), want2.float(), rtol=1e-4, atol=3 * e_d + 1e-5 * sc, what=what + " bias + ELU")
        wantg = torch.nn.functional.conv_transpose2d(dy.double(), w.double(), padding=1)
        dx, _ = H.winograd_fused("conv_dgrad", dyd, ud)
        dxd, _ = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W))
        scg = float(wantg.abs().max())
        e_w, e_d = float((nchw(dx).double().cpu() - wantg).abs().max()), float((nchw(dxd).double().cpu() - wantg).abs().max())
        assert e_w <= 3 * e_d + 1e-6 * scg, (what, "data-gradient", e_w, e_d, scg)
        base = torch.randn(B, Hh, W, C, generator=gen).to(device)
        acc = base.clone()
        r = H.winograd_fused("conv_dgrad", dyd, ud, accumulate_into=acc)
        assert r is not None and r[0] is acc
        assert_close(acc, base + dx, rtol=1e-6, atol=1e-6 * scg, what=what + " accumulate")
    # mirrored padding (the decoder's single-source Conv3x3, forward)
    B, Hh, W, C, Co = 2, 12, 24, 64, 64
    x = torch.randn(B, C, Hh, W, generator=gen)
    w = torch.randn(Co, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5
    uf = H.winograd_fused_pack(w.to(device), False)
    y, _ = H.winograd_fused("conv_fwd", nhwc(x).to(device).contiguous(), uf, reflect=True)
    want = torch.nn.functional.conv2d(torch.nn.functional.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double())
    assert_close(nchw(y).cpu(), want.float(), rtol=1e-4, atol=3e-6 * float(want.abs().max()), what="fused Winograd, mirrored padding")
    # round 5: data-gradient of the mirrored convolution on the one-kernel route (zero-padded launch + the border launches of
    # segsde_reflect_adjoint_borders), with the activation derivative in its epilogue and accumulating onto a collector tensor --
    # against float64 autograd of ReflectionPad2d(1) -> conv, error within 3x the direct (reflection-adjoint) kernel's
    old_macs = H.WINOGRAD_MIN_MACS
    H.WINOGRAD_MIN_MACS = 0.0
    try:
        for (B, Hh, W, C, Co) in ((2, 12, 24, 64, 64), (1, 8, 16, 128, 64), (1, 4, 8, 64, 128), (1, 10, 36, 64, 64)):
            what = "fused Winograd, mirrored data-gradient %s" % ((B, Hh, W, C, Co),)
            w = torch.randn(Co, C, 3, 3, generator=gen) * (2.0 / (9 * C)) ** 0.5
            dy = torch.randn(B, Co, Hh, W, generator=gen)
            a = torch.nn.functional.elu(torch.randn(B, C, Hh, W, generator=gen))        # the saved output of the producing ELU
            xin = torch.zeros(B, C, Hh, W, dtype=torch.float64, requires_grad=True)
            torch.nn.functional.conv2d(torch.nn.functional.pad(xin, (1, 1, 1, 1), mode="reflect"), w.double()).backward(dy.double())
            wantg = xin.grad
            deriv = torch.where(a > 0, torch.ones_like(a), a + 1).double()
            scg = float(wantg.abs().max())
            wd_, dyd, ad = w.to(device), nhwc(dy).to(device).contiguous(), nhwc(a).to(device).contiguous()
            wfp, wdp = H.pack_weight_both(wd_)
            ud = H.winograd_fused_pack(wd_, True)
            g = H.ConvGeom(C, Co, 3, 1, 1, 1, True, 0, False)
            n0 = dict(H.WINO_FUSED_TAKEN)
            dx, _ = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W), wino=ud)          # border terms: implicit-GEMM border launches (no forward pack given)
            assert H.WINO_FUSED_TAKEN["dgrad_refl"] == n0["dgrad_refl"] + 1, what + ": the one-kernel route declined"
            dxd, _ = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W))
            e_w, e_d = float((nchw(dx).double().cpu() - wantg).abs().max()), float((nchw(dxd).double().cpu() - wantg).abs().max())
            assert e_w <= 3 * e_d + 1e-6 * scg, (what, e_w, e_d, scg)
            dx2, _ = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W), wino=ud, wfpack=wfp)   # border terms: the two-launch border kernel
            e_2 = float((nchw(dx2).double().cpu() - wantg).abs().max())
            assert e_2 <= 3 * e_d + 1e-6 * scg, (what, "border kernel", e_2, e_d, scg)
            dz_old, _ = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W), wino=ud, actgrad=(ad, "elu"))
            assert_close(nchw(dz_old).cpu(), (wantg * deriv).float(), rtol=1e-4, atol=3 * e_d + 1e-6 * scg, what=what + " x ELU' (border launches)")
            wdp_keep = wdp
            dz, _ = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W), wino=ud, actgrad=(ad, "elu"), wfpack=wfp)
            assert H.ACTGRAD_FUSED[0] and H.WINO_FUSED_TAKEN["dgrad_actgrad"] == n0["dgrad_actgrad"] + 2
            assert_close(nchw(dz).cpu(), (wantg * deriv).float(), rtol=1e-4, atol=3 * e_d + 1e-6 * scg, what=what + " x ELU'")
            base = torch.randn(B, Hh, W, C, generator=gen).to(device)
            acc = base.clone()
            dz2, _ = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W), wino=ud, actgrad=(ad, "elu"), accumulate_into=acc, wfpack=wfp)
            assert dz2 is acc and H.ACTGRAD_FUSED[0]
            assert_close(acc, base + dz, rtol=1e-6, atol=1e-6 * scg, what=what + " x ELU', accumulated")
            # zero padding + derivative (layer1 / layer2 never need it, but the epilogue is the same code)
            g0 = H.ConvGeom(C, Co, 3, 1, 1, 1, False, 0, False)
            want0 = torch.nn.functional.conv_transpose2d(dy.double(), w.double(), padding=1) * deriv
            dz0, _ = H.conv_dgrad(g0, dyd, wdp, wd_, (Hh, W), wino=ud, actgrad=(ad, "elu"))
            assert_close(nchw(dz0).cpu(), want0.float(), rtol=1e-4, atol=3 * e_d + 1e-6 * scg, what=what + " zero padding x ELU'")
        # the decoder's Conv3x3 on [upsample(x0) | x1] with mirrored padding, bias, ELU: upsampling and concat in the patch loader
        old_fold = H.WINO_FUSED2_MIN_FOLD
        H.WINO_FUSED2_MIN_FOLD = 0.0
        try:
            for (B, Hh, W, C0, C1, Co) in ((1, 8, 16, 64, 64, 64), (2, 12, 24, 128, 64, 128), (1, 8, 16, 64, 0, 64), (1, 20, 12, 64, 128, 64)):
                what = "fused Winograd on [up(x0) | x1] %s" % ((B, Hh, W, C0, C1, Co),)
                x0 = torch.randn(B, C0, Hh // 2, W // 2, generator=gen)
                x1 = torch.randn(B, C1, Hh, W, generator=gen) if C1 else None
                w = torch.randn(Co, C0 + C1, 3, 3, generator=gen) * (2.0 / (9 * (C0 + C1))) ** 0.5
                bias = torch.randn(Co, generator=gen)
                xin = torch.nn.functional.interpolate(x0.double(), scale_factor=2, mode="nearest")
                if C1:
                    xin = torch.cat([xin, x1.double()], 1)
                want = torch.nn.functional.elu(torch.nn.functional.conv2d(torch.nn.functional.pad(xin, (1, 1, 1, 1), mode="reflect"), w.double())
                                               + bias.double().view(1, -1, 1, 1))
                sc = float(want.abs().max())
                x0d, x1d = nhwc(x0).to(device).contiguous(), (nhwc(x1).to(device).contiguous() if C1 else None)
                wd_ = w.to(device)
                wp = H.pack_weight(wd_)
                uf = H.winograd_fused_pack(wd_, False)
                g = H.ConvGeom(C0, Co, 3, 1, 1, 1, True, C1, True)
                assert H.winograd_fused_ok(g, B, Hh, W)
                n0 = H.WINO_FUSED_TAKEN["fwd2"]
                y = H.conv_forward(g, x0d, x1d, wp, bias.to(device), act="elu", wino=uf)
                assert H.WINO_FUSED_TAKEN["fwd2"] == n0 + 1, what + ": declined"
                direct = H.conv_forward(g, x0d, x1d, wp, bias.to(device), act="elu")
                e_w, e_d = float((nchw(y).double().cpu() - want).abs().max()), float((nchw(direct).double().cpu() - want).abs().max())
                assert e_w <= 3 * e_d + 1e-6 * sc, (what, e_w, e_d, sc)
            # the same layers' data-gradient: the upsampled source's low-resolution gradient on the folded route, the skip source's
            # on the one-kernel Winograd route (a column slice of the flipped pack + the border kernel on the forward pack's slice),
            # also accumulating onto a gradient another consumer of the skip feature left
            old_fm = H.UPFOLD_MIN_SAVED_MACS
            H.UPFOLD_MIN_SAVED_MACS = 0.0
            try:
                for (B, Hh, W, C0, C1, Co) in ((1, 8, 16, 64, 64, 64), (2, 12, 24, 64, 128, 64), (1, 20, 12, 128, 64, 128)):
                    what = "data-gradient of [up(x0) | x1] %s" % ((B, Hh, W, C0, C1, Co),)
                    w = torch.randn(Co, C0 + C1, 3, 3, generator=gen) * (2.0 / (9 * (C0 + C1))) ** 0.5
                    dy = torch.randn(B, Co, Hh, W, generator=gen)
                    x0r = torch.zeros(B, C0, Hh // 2, W // 2, dtype=torch.float64, requires_grad=True)
                    x1r = torch.zeros(B, C1, Hh, W, dtype=torch.float64, requires_grad=True)
                    xin = torch.cat([torch.nn.functional.interpolate(x0r, scale_factor=2, mode="nearest"), x1r], 1)
                    torch.nn.functional.conv2d(torch.nn.functional.pad(xin, (1, 1, 1, 1), mode="reflect"), w.double()).backward(dy.double())
                    wd_, dyd = w.to(device), nhwc(dy).to(device).contiguous()
                    wfp, wdp = H.pack_weight_both(wd_)
                    ud = H.winograd_fused_pack(wd_, True)
                    g = H.ConvGeom(C0, Co, 3, 1, 1, 1, True, C1, True)
                    assert H.upfold_ok(g, B * Hh * W) and H.winograd_fused_dgrad2_ok(g, B, Hh, W), what
                    fold = H.upfold_pack(wd_, C0)
                    d0, d1 = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W))                      # the plain route: the yardstick for the error
                    n0 = dict(H.WINO_FUSED_TAKEN)
                    f0, f1 = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W), fold=fold, wino=ud, wfpack=wfp)
                    assert H.WINO_FUSED_TAKEN["dgrad2"] == n0["dgrad2"] + 1, what + ": the skip-source launch declined"
                    for got, dir_, want, tag in ((f0, d0, x0r.grad, "upsampled source"), (f1, d1, x1r.grad, "skip source")):
                        sc = float(want.abs().max())
                        e_w, e_d = float((nchw(got).double().cpu() - want).abs().max()), float((nchw(dir_).double().cpu() - want).abs().max())
                        assert e_w <= 3 * e_d + 2e-6 * sc, (what, tag, e_w, e_d, sc)
                    base = torch.randn(B, Hh, W, C1, generator=gen).to(device)
                    acc = base.clone()
                    _, a1 = H.conv_dgrad(g, dyd, wdp, wd_, (Hh, W), fold=fold, wino=ud, wfpack=wfp, accumulate_skip_into=acc, need0=False)
                    assert a1 is acc and H.SKIP_ACCUMULATED[0]
                    assert_close(acc, base + f1, rtol=1e-6, atol=1e-6 * float(f1.abs().max()), what=what + ", accumulated onto the collector")
            finally:
                H.UPFOLD_MIN_SAVED_MACS = old_fm
        finally:
            H.WINO_FUSED2_MIN_FOLD = old_fold
    finally:
        H.WINOGRAD_MIN_MACS = old_macs
    B, Hh, W, C, Co = 2, 12, 24, 64, 64
    # autograd glue: Conv2d -> BatchNorm2d and a mirrored Conv2d + bias + ELU through the one-kernel route == the direct route;
    # a weight_pack_scope(model) transforms the eligible weights in one launch, in the route's own layout
    from improving_segmentation_with_selfsupervised_depth_amd.models import layers as L
    old = (H.WINOGRAD_MIN_MACS, H.WINO_FUSED)
    H.WINOGRAD_MIN_MACS = 0.0
    try:
        net = torch.nn.Sequential(L.Conv2d(64, 128, 3, padding=1, bias=False), L.Conv2d(512, 512, 3, padding=1, bias=False),
                                  L.Conv2d(64, 64, 3, padding=2, dilation=2)).to(device)
        with L.weight_pack_scope(net):
            assert net[0]._wino_cache.get("kn") is True and net[1]._wino_cache.get("kn") is False
            uf1, ud1 = H.winograd_pack(net[0].weight, kn=True)
            assert torch.equal(net[0]._wino_cache["packs"][0], uf1) and torch.equal(net[0]._wino_cache["packs"][1], ud1)
            assert not net[2]._wino_cache.get("packs")
        torch.manual_seed(7)
        conv, bn = L.Conv2d(C, Co, 3, padding=1, bias=False).to(device), L.BatchNorm2d(Co).to(device)
        conv2 = L.Conv2d(Co, Co, 3, padding=1, bias=True, reflect=True).to(device)
        for m in (conv, bn, conv2):
            m.train()
        xin = torch.randn(B, Hh, W, C, generator=gen).to(device)
        res = []
        for on in (True, False):
            H.WINO_FUSED = on
            xi = xin.clone().requires_grad_(True)
            for m in (conv, bn, conv2):
                m.zero_grad()
            n0 = dict(H.WINO_FUSED_TAKEN)
            with L.weight_pack_scope():
                yv = conv2(bn(conv(xi), act="relu"), act="elu")
                (yv * yv).sum().backward()
            took = (H.WINO_FUSED_TAKEN["fwd"] - n0["fwd"], H.WINO_FUSED_TAKEN["dgrad"] - n0["dgrad"])
            assert took == ((2, 2) if on else (0, 0)), took      # conv + mirrored conv2 forward, both data-gradients (round 5: the mirrored one too)
            res.append((yv.detach(), xi.grad.detach(), conv.weight.grad.detach().clone(), conv2.weight.grad.detach().clone(),
                        conv2.bias.grad.detach().clone(), bn.weight.grad.detach().clone()))
        for a, b, what in zip(res[0], res[1], ("output", "input gradient", "weight gradient", "mirrored conv weight gradient",
                                               "bias gradient", "BatchNorm weight gradient")):
            assert_close(a, b, rtol=1e-3, atol=1e-4 * float(b.abs().max()), what="fused Winograd vs direct through Conv2d + BatchNorm2d: " + what)
    finally:
        H.WINOGRAD_MIN_MACS, H.WINO_FUSED = old


# ---------------------------------------------------------------------------------------------
# round 5: weight gradient on the one-kernel Winograd scheme (csrc/winograd_wgrad.hip) against float64 autograd and against the
# direct weight-gradient kernel: one / two sources, nearest-upsampled first source, zero / mirrored padding, whole and partial
# tile blocks, every staging variant (SEGSDE_WGRAD_FUSED_VAR is read once per process: the variants are separate test runs)
# ---------------------------------------------------------------------------------------------
def run_winograd_fused_wgrad_cases(device, shapes=None):
    gen = torch.Generator().manual_seed(41)
    if shapes is None:
        #          B  H   W   C0   C1  Co  up     reflect
        shapes = ((1, 8, 16, 64, 0, 64, False, False), (2, 12, 20, 64, 0, 128, False, True), (1, 16, 32, 128, 0, 64, False, False),
                  (1, 6, 36, 32, 0, 64, False, True), (2, 8, 16, 64, 32, 64, True, True), (1, 12, 24, 32, 96, 128, True, True),
                  (1, 8, 16, 64, 0, 64, True, True), (1, 4, 4, 32, 32, 64, False, False),
                  # maps with INTERIOR tile blocks (the loads' vector-instruction-free path), one source and [upsample | skip]
                  (1, 40, 64, 32, 0, 64, False, True), (2, 24, 48, 32, 32, 64, True, True), (1, 26, 52, 32, 0, 64, False, False))
    old = (H.WINOGRAD_MIN_MACS, H.WINO_FUSED_WGRAD_MIN_FOLD)
    H.WINOGRAD_MIN_MACS, H.WINO_FUSED_WGRAD_MIN_FOLD = 0.0, 0.0
    try:
        for (B, Hh, W, C0, C1, Co, up, refl) in shapes:
            what = "fused Winograd weight gradient %s" % ((B, Hh, W, C0, C1, Co, up, refl),)
            x0 = torch.randn(B, C0, Hh // 2 if up else Hh, W // 2 if up else W, generator=gen)
            x1 = torch.randn(B, C1, Hh, W, generator=gen) if C1 else None
            dy = torch.randn(B, Co, Hh, W, generator=gen)
            xin = torch.nn.functional.interpolate(x0.double(), scale_factor=2, mode="nearest") if up else x0.double()
            if C1:
                xin = torch.cat([xin, x1.double()], 1)
            w = torch.zeros(Co, C0 + C1, 3, 3, dtype=torch.float64, requires_grad=True)
            xp = torch.nn.functional.pad(xin, (1, 1, 1, 1), mode="reflect" if refl else "constant")
            torch.nn.functional.conv2d(xp, w).backward(dy.double())
            want = w.grad
            sc = float(want.abs().max())
            g = H.ConvGeom(C0, Co, 3, 1, 1, 1, refl, C1, up)
            x0d, x1d, dyd = nhwc(x0).to(device).contiguous(), (nhwc(x1).to(device).contiguous() if C1 else None), nhwc(dy).to(device).contiguous()
            if not (g.up0 or g.C1) or g.up0:
                assert H.winograd_fused_wgrad_ok(g, B, Hh, W), what
                n0 = H.WINO_FUSED_TAKEN["wgrad"]
                dw = H.conv_wgrad(g, x0d, x1d, dyd)
                assert H.WINO_FUSED_TAKEN["wgrad"] == n0 + 1, what + ": declined"
            else:
                # two sources without upsampling: the router keeps them on the grouped route; the kernel itself takes them
                dw = _wgrad_fused_direct(g, x0d, x1d, dyd)
            H.WINO_FUSED_WGRAD = False
            try:
                dwd = H.conv_wgrad(g, x0d, x1d, dyd)
            finally:
                H.WINO_FUSED_WGRAD = True
            e_w, e_d = float((dw.double().cpu() - want).abs().max()), float((dwd.double().cpu() - want).abs().max())
            assert e_w <= 3 * e_d + 1e-6 * sc, (what, e_w, e_d, sc)
    finally:
        H.WINOGRAD_MIN_MACS, H.WINO_FUSED_WGRAD_MIN_FOLD = old


def _wgrad_fused_direct(g, x0, x1, dy):
    import ctypes
    from improving_segmentation_with_selfsupervised_depth_amd import _lib
    B, H0, W0, _ = x0.shape
    Hh, W = (2 * H0, 2 * W0) if g.up0 else (H0, W0)
    Co = dy.shape[3]
    d = _lib.ConvDesc(B=B, H=Hh, W=W, C0=g.C0, C1=g.C1, ld0=H.nhwc_ld(x0), ld1=H.nhwc_ld(x1) if x1 is not None else 0, up0=int(g.up0),
                      Ho=Hh, Wo=W, Cout=Co, ldy=Co, ldy2=0, nsplit=0, KH=3, KW=3, stride=1, dil=1, pad=1,
                      pad_mode=H.PAD_REFLECT if g.reflect else H.PAD_ZERO, in_div=1, act=0, sum2x2=0)
    L = _lib.lib()
    nbytes = L.segsde_conv2d_wgrad_winograd_fused_workspace(ctypes.byref(d))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    dw = torch.empty((Co, g.Cin, 3, 3), dtype=torch.float32, device=dy.device)
    H.check(L.segsde_conv2d_wgrad_winograd_fused(ctypes.byref(d), H._p(x0), H._p(x1), H._p(dy), H.nhwc_ld(dy), H._p(dw), H._p(ws), nbytes,
                                                 H._stream(dy)), "wgrad_winograd_fused")
    return dw


# ---------------------------------------------------------------------------------------------
# round 5: nn.Dropout on the stacked segmentation features (JointSegDepthDecoder(layer_dropout > 0),
# models/joint_segmentation_depth_decoder.py:50): the counter-based mask kernel and the module path
# ---------------------------------------------------------------------------------------------
def run_dropout_case(device):
    from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(2, 24, 40, 48, generator=gen) + 3.0).to(device)          # no zeros in the input: a zero output is a dropped element
    p, seed = 0.3, 12345
    y = H.dropout(x, p, seed)
    kept = y != 0
    frac = float(kept.float().mean())
    assert abs(frac - (1 - p)) < 0.02, frac
    assert_close(y[kept], x[kept] / (1 - p), rtol=1e-6, atol=0, what="dropout: kept elements are scaled by 1 / (1 - p)")
    assert torch.equal(H.dropout(x, p, seed), y), "same seed, same mask"
    assert not torch.equal(H.dropout(x, p, seed + 1) != 0, kept), "another seed, another mask"
    assert torch.equal(H.dropout(x, 0.0, seed), x)
    xs = x[..., 8:40]                                                          # a channel slice: the kernel takes its pixel pitch
    ys = H.dropout(xs, p, 77)
    ks = ys != 0
    assert_close(ys[ks], xs[ks] / (1 - p), rtol=1e-6, atol=0, what="dropout on a channel slice")
    xi = x.clone().requires_grad_(True)
    out = Fn.DropoutFn.apply(xi, p, seed)
    w = torch.randn(x.shape, generator=gen).to(device)
    (out * w).sum().backward()
    assert_close(xi.grad, torch.where(kept, w / (1 - p), torch.zeros_like(w)), rtol=1e-6, atol=0, what="dropout adjoint: the same mask")
    # the module: layer_dropout > 0 drops in train mode only, and the state_dict layout is the reference's (a parameter-free module at
    # head.0 either way)
    from improving_segmentation_with_selfsupervised_depth_amd.models.joint_segmentation_depth_decoder import JointSegDepthDecoder
    torch.manual_seed(3)
    args = dict(num_ch_enc=[16, 16, 32, 32, 32], num_ch_dec=[16, 16, 16, 16, 16], num_classes=5, layers=[9], head_inter=False,
                layer_out_channels=16, head_inter_channels=16,
                depth_args=dict(scales=range(4), max_scale_size=[64, 128], num_ch_dec=[16, 16, 16, 16, 16], intermediate_aspp=False))
    net = JointSegDepthDecoder(layer_dropout=0.5, **args).to(device)
    ref = JointSegDepthDecoder(layer_dropout=0, **args).to(device)
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    ref.load_state_dict(net.state_dict())
    feats = [torch.randn(2, c, 32 >> i, 64 >> i, generator=gen).to(device) for i, c in enumerate([16, 16, 32, 32, 32])]
    net.eval(), ref.eval()
    with torch.no_grad():
        assert_close(net(feats), ref(feats), rtol=0, atol=0, what="layer_dropout is the identity in eval mode")
    net.train(), ref.train()
    a, b = net(feats), net(feats)
    assert not torch.equal(a, b), "train mode: a fresh mask per forward"
    a.sum().backward()
    assert all(torch.isfinite(q.grad).all() for q in net.parameters() if q.grad is not None)


# ---------------------------------------------------------------------------------------------
# BatchNorm statistics from convolution-epilogue partials: the three launch plans of segsde_bn_stats_from_partials (few rows: one
# small kernel; a few hundred: the one-launch wide kernel of round 5; thousands: reduce + finalize) against float64
# ---------------------------------------------------------------------------------------------
def run_bn_partials_case(device):
    gen = torch.Generator().manual_seed(17)
    for rows, C in ((40, 64), (300, 70), (256, 256), (1024, 64), (1500, 32), (65, 5)):
        M = rows * 128
        part = torch.rand(rows, 2, C, generator=gen, dtype=torch.float64)
        part[:, 0] = (part[:, 0] - 0.5) * 128.0            # tile sums of x
        part[:, 1] = part[:, 1] * 128.0 + 40.0             # tile sums of x^2 (keeps the variance positive)
        s, q = part[:, 0].sum(0), part[:, 1].sum(0)
        mu = s / M
        var = (q / M - mu * mu).clamp_min(0)
        rm, rv = torch.zeros(C).to(device), torch.ones(C).to(device)
        nbt = torch.zeros((), dtype=torch.int64).to(device)
        mean, invstd = H.bn_stats_from_partials(part.to(device), M, rm, rv, 0.1, 1e-5, num_batches_tracked=nbt)
        what = "bn_stats_from_partials rows=%d C=%d" % (rows, C)
        assert_close(mean, mu.float(), rtol=1e-6, atol=1e-7, what=what + " mean")
        assert_close(invstd, (1.0 / torch.sqrt(var + 1e-5)).float(), rtol=1e-6, atol=0, what=what + " invstd")
        assert_close(rm, (0.1 * mu).float(), rtol=1e-6, atol=1e-8, what=what + " running_mean")
        assert_close(rv, (0.9 + 0.1 * var * M / (M - 1)).float(), rtol=1e-6, atol=0, what=what + " running_var")
        assert int(nbt) == 1


def run_f16_operand_convolutions(device):
    """segsde_conv_desc.compute = 1 (the `amp: True` arithmetic: operands rounded to fp16 in the kernel, v_mfma_f32_32x32x16_f16,
    fp32 accumulation) on the three directions of the implicit GEMM: within fp16 round-off of a float64 convolution (tolerance
    1.5e-3 of the result's maximum; measured 3e-4), the fp32 mode of the same launches within 1e-5.  Shapes that take the LDS-DMA
    loop (channels a multiple of 32): 1x1, 3x3 zero- and mirror-padded, dilated, two-source upsampled (the folded route)."""
    import torch.nn.functional as F
    from improving_segmentation_with_selfsupervised_depth_amd import hipops as H
    torch.manual_seed(5)
    B = 2
    cases = [(16, 32, 64, 128, 1, 1, 0, False), (16, 32, 64, 64, 3, 1, 1, False), (16, 32, 64, 64, 3, 1, 1, True),
             (24, 32, 128, 64, 3, 6, 6, False)]
    for Hh, W, C, Co, k, dil, pad, refl in cases:
        x = torch.relu(torch.randn(B, Hh, W, C)).to(device)
        dy = torch.randn(B, Hh, W, Co).to(device)
        w = (torch.randn(Co, C, k, k) * (2.0 / (k * k * C)) ** 0.5).to(device)
        xr = x.cpu().permute(0, 3, 1, 2).double().requires_grad_(True)
        wr = w.cpu().double().requires_grad_(True)
        xin = F.pad(xr, (1, 1, 1, 1), mode="reflect") if refl else xr
        yr = F.conv2d(xin, wr, padding=0 if refl else pad, dilation=dil)
        yr.backward(dy.cpu().permute(0, 3, 1, 2).double())
        want = (yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad)
        wp, wdp = H.pack_weight_both(w)
        for mode, tol in ((0, 1e-5), (1, 1.5e-3)):
            H.COMPUTE_F16[0] = bool(mode)
            try:
                g = H.ConvGeom(C, Co, k, 1, dil, pad, refl, 0, False)
            finally:
                H.COMPUTE_F16[0] = False
            assert g.compute == mode
            got = (H.conv_forward(g, x, None, wp, None), H.conv_dgrad(g, dy, wdp, w, (Hh, W))[0], H.conv_wgrad(g, x, None, dy))
            for name, a, b in zip(("forward", "data-gradient", "weight gradient"), got, want):
                err = float((a.cpu().double() - b).abs().max() / b.abs().max())
                assert err <= tol, ("compute=%d %s %s: %.2e > %.1e" % (mode, (Hh, W, C, Co, k, dil, refl), name, err, tol))
            if mode == 1:
                e32 = float((H.conv_forward(H.ConvGeom(C, Co, k, 1, dil, pad, refl, 0, False), x, None, wp, None).cpu().double()
                             - want[0]).abs().max() / want[0].abs().max())
                e16 = float((got[0].cpu().double() - want[0]).abs().max() / want[0].abs().max())
                assert e16 > 10 * e32, "the half-precision mode did not take the half-precision kernel (%g vs %g)" % (e16, e32)


def run_torch_ops(device, golden):
    """torch.ops.segsde.* (improving_segmentation_with_selfsupervised_depth_amd/torch_ops.py): every registered operator runs on the
    HIP kernels and differentiates like the package's modules -- an encoder-style chain against ATen in float64, the loss stages
    against the reference vectors, the mask / mix operators against their defining formulas (bit-exact)"""
    import improving_segmentation_with_selfsupervised_depth_amd as pkg
    ops = torch.ops.segsde
    d = lambda t: t.to(device)
    gen = torch.Generator().manual_seed(21)
    assert len(pkg.torch_ops.names()) >= 18 and all(hasattr(ops, n.split("::")[1]) for n in pkg.torch_ops.names())
    # conv -> BatchNorm + ReLU -> max-pool -> reflected 3x3 on [upsampled | skip] -> resize -> global pool, gradients to everything
    B, Hh, W = 2, 16, 24
    x = torch.randn(B, 8, Hh, W, generator=gen)
    w1, b1 = 0.2 * torch.randn(16, 8, 3, 3, generator=gen), 0.1 * torch.randn(16, generator=gen)
    gam, bet = 1.0 + 0.1 * torch.randn(16, generator=gen), 0.1 * torch.randn(16, generator=gen)
    skip = torch.randn(B, 8, Hh, W, generator=gen)
    w2 = 0.2 * torch.randn(8, 24, 3, 3, generator=gen)
    wgt = torch.randn(B, 8, 10, 14, generator=gen)
    leaves = [t.clone().to(device).requires_grad_(True) for t in (x, w1, b1, gam, bet, skip, w2)]
    X, W1, B1, G, Bt, S, W2 = leaves
    rm, rv = torch.zeros(16, device=device), torch.ones(16, device=device)
    h = ops.conv2d(ops.to_nhwc(X), W1, B1, None, 1, 2, 2)
    h = ops.batch_norm_act(h, G, Bt, rm, rv, None, True, 0.1, 1e-5, "relu")
    h = ops.max_pool_3x3_s2(h)
    h = ops.conv2d(h, W2, None, ops.to_nhwc(S), 1, 1, 1, True, True)
    r = ops.resize_bilinear(h, [10, 14], False)
    out = (ops.to_nchw(r) * d(wgt)).sum() + 3.0 * ops.global_avg_pool(h).sum()
    out.backward()
    ref = [t.double().clone().requires_grad_(True) for t in (x, w1, b1, gam, bet, skip, w2)]
    Xr, W1r, B1r, Gr, Btr, Sr, W2r = ref
    rmr, rvr = torch.zeros(16, dtype=torch.float64), torch.ones(16, dtype=torch.float64)
    hr = F.conv2d(Xr, W1r, B1r, 1, 2, 2)
    hr = F.relu(F.batch_norm(hr, rmr, rvr, Gr, Btr, True, 0.1, 1e-5))
    hr = F.max_pool2d(hr, 3, 2, 1)
    hr = F.conv2d(F.pad(torch.cat([F.interpolate(hr, scale_factor=2, mode="nearest"), Sr], 1), (1, 1, 1, 1), mode="reflect"), W2r)
    rr = F.interpolate(hr, size=(10, 14), mode="bilinear", align_corners=False)
    outr = (rr * wgt.double()).sum() + 3.0 * hr.mean((2, 3)).sum()
    outr.backward()
    assert_close(out, outr.float(), rtol=1e-4, what="torch.ops chain value")
    assert_close(rm, rmr.float(), rtol=1e-4, atol=1e-6, what="torch.ops running mean")
    for a, b, name in zip(leaves, ref, ("x", "w1", "b1", "gamma", "beta", "skip", "w2")):
        # (the bias in front of a training-mode BatchNorm has gradient zero: what arrives is rounding noise -> an absolute floor)
        assert_close(a.grad, b.grad.float(), rtol=1e-3, atol=1e-4 * max(float(b.grad.abs().max()), 0.1), what="torch.ops chain d/d" + name)
    # pose matrix, geometry, warp against the reference vectors
    q = golden("geom")
    aa, tr = d(q["axisangle"]).clone().requires_grad_(True), d(q["translation"]).clone().requires_grad_(True)
    M = ops.pose_matrix(aa, tr, True)
    assert_close(M, q["M_inv"], rtol=1e-5, atol=1e-6, what="torch.ops pose_matrix")
    (M * d(q["M_weight"])).sum().backward()
    assert_close(aa.grad, q["grad_aa_inv"], rtol=1e-3, atol=1e-6, what="torch.ops pose_matrix adjoint")
    Bq, _, Hq, Wq = q["depth"].shape
    cam = ops.backproject_depth(d(q["depth"]), d(q["inv_K"]))
    assert_close(cam, q["cam_points"], rtol=1e-5, atol=1e-6, what="torch.ops backproject_depth")
    assert_close(ops.project3d(cam, d(q["K"]), d(q["T"]), Hq, Wq), q["grid"], rtol=1e-4, atol=1e-5, what="torch.ops project3d")
    gl = golden("loss_default")
    col, grid, depth = ops.warp(d(gl["disp_2"]), d(gl["in_inv_K_0"]), d(gl["in_K_0"]), d(gl["T_m1"]), d(gl["in_color_-1_0"]), 0.1, 100.0)
    assert_close(col, gl["color_m1_2"], rtol=1e-3, atol=1e-4, what="torch.ops warp colour")
    assert_close(grid, gl["sample_m1_2"], rtol=1e-4, atol=1e-5, what="torch.ops warp grid")
    assert_close(depth, gl["depth_2"], rtol=1e-4, atol=1e-5, what="torch.ops warp depth")
    # SSIM / smoothness / reprojection error / auto-mask
    g = golden("ssim_smooth")
    xs = d(g["x"]).clone().requires_grad_(True)
    v = ops.ssim(xs, d(g["y"]))
    assert_close(v, g["ssim"], rtol=1e-4, atol=1e-6, what="torch.ops ssim")
    (v * d(g["w"])).sum().backward()
    assert_close(xs.grad, g["grad_x"], rtol=1e-3, atol=2e-5, what="torch.ops ssim adjoint")
    disp = d(g["sm_disp"]).clone().requires_grad_(True)
    sm = ops.smooth_loss(disp, d(g["sm_img"]))
    sm.backward()
    assert_close(sm, g["smooth"], rtol=1e-5, what="torch.ops smooth_loss")
    assert_close(disp.grad, g["grad_sm_disp"], rtol=1e-4, atol=1e-7, what="torch.ops smooth_loss adjoint")
    from oracle import photometric as OP
    tgt, pred = gl["in_color_0_0"], gl["color_m1_0"]
    err = ops.reprojection_error(d(pred), d(tgt), False)
    assert_close(err, OP.reprojection_error(pred, tgt, False), rtol=1e-4, atol=1e-6, what="torch.ops reprojection_error")
    ident, reproj = torch.rand(2, 3, 6, 9, generator=gen), torch.rand(2, 3, 6, 9, generator=gen)
    noise = torch.randn(2, 3, 6, 9, generator=gen)
    ssum, sel, isel = ops.automask_min(d(ident), d(noise), d(reproj), False)
    mn, ix = torch.min(torch.cat([ident + noise * 0.00001, reproj], 1), 1)
    assert torch.equal(sel.cpu().long(), ix) and torch.equal(isel.cpu(), (ix > 2).float())
    assert_close(ssum, mn.sum().reshape(1), rtol=1e-5, what="torch.ops automask_min")
    ssum2, sel2, isel2 = ops.automask_min(None, None, d(reproj), False)
    assert isel2 is None and torch.equal(sel2.cpu().long(), reproj.min(1)[1])
    # segmentation loss, mix, depthcomp mask
    logits = torch.randn(2, 19, 12, 20, generator=gen)
    target = torch.randint(0, 19, (2, 12, 20), generator=gen)
    target[0, :3] = 250
    lg = d(logits).clone().requires_grad_(True)
    ce = ops.cross_entropy2d(lg, d(target))
    ce.backward()
    lr = logits.double().clone().requires_grad_(True)
    cer = F.cross_entropy(lr, target, ignore_index=250)
    cer.backward()
    assert_close(ce, cer.float(), rtol=1e-5, what="torch.ops cross_entropy2d")
    assert_close(lg.grad, lr.grad.float(), rtol=1e-3, atol=1e-8, what="torch.ops cross_entropy2d adjoint")
    mask = (torch.rand(2, 12, 20, generator=gen) > 0.5).long()
    data = torch.randn(2, 3, 12, 20, generator=gen)
    mixed = ops.mix(d(mask), d(data))
    mf = mask.float().unsqueeze(1)
    assert torch.equal(mixed.cpu(), mf * data + (1 - mf) * torch.roll(data, -1, 0))
    labels = ops.mix(d(mask), d(target))
    assert torch.equal(labels.cpu(), mask * target + (1 - mask) * torch.roll(target, -1, 0))
    depths = torch.rand(2, 12, 20, generator=gen)
    dm = ops.depthcomp_mask(d(depths), 0.05, 0.3)
    want = ((depths >= torch.roll(depths, -1, 0) - 0.05) & (depths >= 0.3)).long()
    assert torch.equal(dm.cpu(), want)
