"""Sampled-window reference for full-size convolution checks (TEST INFRASTRUCTURE).

At the benchmark's own sizes (batch 16, 512x1024, single activations above 2^31 bytes) a full CPU reference is out of
reach, but any single output of a convolution / data-gradient / weight-gradient depends on a small window.  These
helpers recompute, in float64 and straight from the definition

    y[b,ho,wo,n] = bias[n] + sum_{kh,kw,c} X[b, src(ho,kh), src(wo,kw), c] * w[n,c,kh,kw]

(X = [nearest-x2-upsampled x0 | x1], src = reflection or zero padding, stride, dilation), the values at sampled
positions only.  tests/test_spotcheck_ref.py pins them against torch autograd at small sizes on CPU.
Tensors are NHWC; torch indexing on whatever device the tensors live on is used for the gathers only."""
import numpy as np
import torch


def src_index(o, t, n_in, stride, dil, pad, reflect):
    """input coordinate read by output coordinate o at tap t, or -1 (zero padding)"""
    i = o * stride - pad + t * dil
    if reflect:
        if i < 0:
            i = -i
        elif i >= n_in:
            i = 2 * n_in - 2 - i
        return i
    return i if 0 <= i < n_in else -1


def _virtual(x0, x1, up0):
    H = x0.shape[1] * (2 if up0 else 1)
    W = x0.shape[2] * (2 if up0 else 1)
    return H, W


def _gather(x0, x1, up0, b, h, w):
    """channel vector of the virtual input [up(x0) | x1] at (b,h,w) as float64 on the CPU"""
    v0 = x0[b, h // 2 if up0 else h, w // 2 if up0 else w]
    if x1 is not None:
        v0 = torch.cat([v0, x1[b, h, w]])
    return v0.double().cpu()


def conv_samples(x0, x1, up0, w_oihw, bias, stride, dil, pad, reflect, samples):
    """forward values [len(samples), Cout] (float64) at output positions samples = [(b, ho, wo), ...]"""
    H, W = _virtual(x0, x1, up0)
    k = w_oihw.shape[2]
    wd = w_oihw.double().cpu()
    out = []
    for (b, ho, wo) in samples:
        acc = torch.zeros(w_oihw.shape[0], dtype=torch.float64)
        if bias is not None:
            acc += bias.double().cpu()
        for kh in range(k):
            hi = src_index(ho, kh, H, stride, dil, pad, reflect)
            if hi < 0:
                continue
            for kw in range(k):
                wi = src_index(wo, kw, W, stride, dil, pad, reflect)
                if wi < 0:
                    continue
                acc += wd[:, :, kh, kw] @ _gather(x0, x1, up0, b, hi, wi)
        out.append(acc)
    return torch.stack(out)


def _pairs(u, n_in, n_out, k, stride, dil, pad, reflect):
    """all (o, t) whose source coordinate is u -- brute force over a window that certainly contains them"""
    res = []
    lo = max(0, (u + pad - (k - 1) * dil) // stride - 2)
    hi = min(n_out - 1, (u + pad) // stride + 2)
    cand = set(range(lo, hi + 1))
    if reflect:          # mirrored pre-images live next to the borders
        cand |= set(range(0, min(n_out, k + 1))) | set(range(max(0, n_out - k - 1), n_out))
    for o in sorted(cand):
        for t in range(k):
            if src_index(o, t, n_in, stride, dil, pad, reflect) == u:
                res.append((o, t))
    return res


def dgrad_samples(dy, w_oihw, in_hw, c_lo, c_hi, up, stride, dil, pad, reflect, samples):
    """data-gradient values [len(samples), c_hi - c_lo] (float64) w.r.t. input channels c_lo:c_hi at STORED positions
    samples = [(b, h, w), ...]; up=True: the source is stored at half resolution (its gradient sums the 2x2 block)."""
    H, W = in_hw
    _, Ho, Wo, _ = dy.shape
    k = w_oihw.shape[2]
    wd = w_oihw.double().cpu()[:, c_lo:c_hi]            # [Cout, c, k, k]
    out = []
    for (b, h, w) in samples:
        acc = torch.zeros(c_hi - c_lo, dtype=torch.float64)
        cells = [(2 * h + a, 2 * w + c) for a in (0, 1) for c in (0, 1)] if up else [(h, w)]
        for (uh, uw) in cells:
            ph = _pairs(uh, H, Ho, k, stride, dil, pad, reflect)
            pw = _pairs(uw, W, Wo, k, stride, dil, pad, reflect)
            for (oh, th) in ph:
                for (ow, tw) in pw:
                    acc += dy[b, oh, ow].double().cpu() @ wd[:, :, th, tw]
        out.append(acc)
    return torch.stack(out)


def wgrad_samples(x0, x1, up0, dy, k, stride, dil, pad, reflect, taps):
    """weight-gradient values (float64 list) at taps = [(n, c, kh, kw), ...]: full reductions over every output pixel,
    done with float64 torch ops on the tensors' device"""
    H, W = _virtual(x0, x1, up0)
    B, Ho, Wo, _ = dy.shape
    C0 = x0.shape[3]
    dev = dy.device
    out = []
    for (n, c, kh, kw) in taps:
        hi = torch.tensor([src_index(o, kh, H, stride, dil, pad, reflect) for o in range(Ho)], device=dev)
        wi = torch.tensor([src_index(o, kw, W, stride, dil, pad, reflect) for o in range(Wo)], device=dev)
        vh, vw = hi >= 0, wi >= 0
        hi, wi = hi.clamp(min=0), wi.clamp(min=0)
        if c < C0:
            plane = x0[..., c]
            if up0:
                hi, wi = hi // 2, wi // 2
        else:
            plane = x1[..., c - C0]
        g = plane[:, hi][:, :, wi].double()                       # [B, Ho, Wo]
        g = g * (vh.double()[None, :, None] * vw.double()[None, None, :])
        out.append(float((g * dy[..., n].double()).sum()))
    return out


def pick_positions(B, H, W, n, seed, extra=()):
    """n sample positions (b, h, w): corners / borders of the first and the LAST image (largest byte offsets) + random"""
    rng = np.random.RandomState(seed)
    pos = [(0, 0, 0), (0, 0, W - 1), (0, H - 1, 0), (B - 1, H - 1, W - 1), (B - 1, H - 1, 0), (B - 1, 0, W - 1),
           (B - 1, H - 2, W - 2), (B - 1, 1, 1), (B - 1, H // 2, W - 1), (B - 1, H - 1, W // 2), (B // 2, 0, W // 2)]
    pos += list(extra)
    while len(pos) < n:
        pos.append((int(rng.randint(B)), int(rng.randint(H)), int(rng.randint(W))))
    return pos[:max(n, len(pos))]
