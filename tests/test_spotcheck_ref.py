"""CPU: the sampled-window reference used by the full-size GPU checks (tests/spotcheck.py) agrees with torch autograd of
F.conv2d on small shapes -- reflection / zero padding, nearest x2 upsample + concat, stride, dilation."""
import pytest
import torch
import torch.nn.functional as F

import spotcheck as SC

CASES = [
    # name, C0, C1, up0, Cout, k, stride, dil, pad, reflect, (B, H, W)  [H, W = virtual input size]
    ("refl_up", 5, 0, True, 6, 3, 1, 1, 1, True, (2, 8, 12)),
    ("refl_up_cat", 4, 3, True, 5, 3, 1, 1, 1, True, (2, 8, 12)),
    ("dil", 6, 0, False, 4, 3, 1, 3, 3, False, (2, 9, 11)),
    ("s2", 4, 0, False, 5, 3, 2, 1, 1, False, (2, 10, 12)),
    ("1x1", 7, 0, False, 3, 1, 1, 1, 0, False, (2, 5, 6)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_sampled_reference_matches_autograd(case):
    name, C0, C1, up0, Cout, k, stride, dil, pad, reflect, (B, H, W) = case
    gen = torch.Generator().manual_seed(1)
    x0 = torch.randn(B, H // 2 if up0 else H, W // 2 if up0 else W, C0, generator=gen, dtype=torch.float64, requires_grad=True)
    x1 = torch.randn(B, H, W, C1, generator=gen, dtype=torch.float64, requires_grad=True) if C1 else None
    w = torch.randn(Cout, C0 + C1, k, k, generator=gen, dtype=torch.float64, requires_grad=True)
    bias = torch.randn(Cout, generator=gen, dtype=torch.float64)
    xin = x0.permute(0, 3, 1, 2)
    if up0:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    if x1 is not None:
        xin = torch.cat([xin, x1.permute(0, 3, 1, 2)], 1)
    if reflect:
        y = F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w, bias, stride, 0, dil)
    else:
        y = F.conv2d(xin, w, bias, stride, pad, dil)
    y_nhwc = y.permute(0, 2, 3, 1)
    dy = torch.randn(y_nhwc.shape, generator=gen, dtype=torch.float64)
    (y_nhwc * dy).sum().backward()
    Ho, Wo = y_nhwc.shape[1:3]
    pos = [(b, h, ww) for b in range(B) for h in range(Ho) for ww in range(Wo)][::3]
    got = SC.conv_samples(x0.detach(), None if x1 is None else x1.detach(), up0, w.detach(), bias, stride, dil, pad, reflect, pos)
    want = torch.stack([y_nhwc[p].detach() for p in pos])
    assert torch.allclose(got, want, rtol=1e-10, atol=1e-10)
    h0, w0 = x0.shape[1:3]
    pos0 = [(b, h, ww) for b in range(B) for h in range(h0) for ww in range(w0)][::2]
    got = SC.dgrad_samples(dy, w.detach(), (H, W), 0, C0, up0, stride, dil, pad, reflect, pos0)
    want = torch.stack([x0.grad[p] for p in pos0])
    assert torch.allclose(got, want, rtol=1e-10, atol=1e-10)
    if x1 is not None:
        pos1 = [(b, h, ww) for b in range(B) for h in range(H) for ww in range(W)][::5]
        got = SC.dgrad_samples(dy, w.detach(), (H, W), C0, C0 + C1, False, stride, dil, pad, reflect, pos1)
        want = torch.stack([x1.grad[p] for p in pos1])
        assert torch.allclose(got, want, rtol=1e-10, atol=1e-10)
    taps = [(n, c, kh, kw) for n in range(Cout) for c in range(C0 + C1) for kh in range(k) for kw in range(k)][::4]
    got = SC.wgrad_samples(x0.detach(), None if x1 is None else x1.detach(), up0, dy, k, stride, dil, pad, reflect, taps)
    want = [float(w.grad[t]) for t in taps]
    assert torch.allclose(torch.tensor(got), torch.tensor(want), rtol=1e-10, atol=1e-10)
