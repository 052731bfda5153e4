"""CPU restatement of the kornia 0.4.0 operators behind the reference's colour jitter and blur
(/root/reference/loader/transformsgpu.py:10-30: kornia.augmentation.ColorJitter(s, s, s, s) and
kornia.filters.GaussianBlur2d(kernel, (sigma, sigma))).

TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED**: kornia (requirements.txt:16, kornia==0.4.0) is a third-party dependency that
is neither under /root/reference nor installed in this image, and the reference holds no vectors for it.  What follows
restates kornia 0.4.0's published algorithm in plain torch, written independently of csrc/augment.hip (tensor ops, the
2-D blur as ONE depthwise conv2d with the outer-product kernel on a reflect-padded image, exactly how kornia's filter2D
does it) so that the two implementations check each other:

  ColorJitter: per sample factors; the four adjustments are applied in a random order (one permutation per batch)
    brightness  x <- clamp(x + (bf - 1), 0, 1)                     (kornia.color.adjust_brightness is additive)
    contrast    x <- clamp(x * cf, 0, 1)                           (kornia.color.adjust_contrast is multiplicative)
    saturation  HSV, s <- clamp(s * sf, 0, 1), back to RGB
    hue         HSV, h <- fmod(h + 2 pi hf, 2 pi), back to RGB     (hue in radians)
  rgb_to_hsv: v = max, s = (max - min) / max (0 where max = 0), h = 2 pi ((sector + offset) / 6 mod 1)
  GaussianBlur2d: kernel = outer(g_y, g_x), g(x) = exp(-(x - k // 2)^2 / (2 sigma^2)) normalised to sum 1, reflect border.
"""
import math

import torch
import torch.nn.functional as F


def rgb_to_hsv(img):
    r, g, b = img[..., 0, :, :], img[..., 1, :, :], img[..., 2, :, :]
    maxc, minc = img.max(-3)[0], img.min(-3)[0]
    v = maxc
    deltac = maxc - minc
    s = deltac / v
    s = torch.where(torch.isnan(s), torch.zeros_like(s), s)
    dc = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
    rc, gc, bc = (maxc - r) / dc, (maxc - g) / dc, (maxc - b) / dc
    h = 4.0 + gc - rc
    h = torch.where(g == maxc, 2.0 + rc - bc, h)
    h = torch.where(r == maxc, bc - gc, h)
    h = torch.where(minc == maxc, torch.zeros_like(h), h)
    h = (h / 6.0) % 1.0
    return torch.stack([2 * math.pi * h, s, v], dim=-3)


def hsv_to_rgb(img):
    h, s, v = img[..., 0, :, :] / (2 * math.pi), img[..., 1, :, :], img[..., 2, :, :]
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    hi = hi.long()
    idx = torch.stack([hi, hi + 6, hi + 12], dim=-3)
    out = torch.stack((v, q, p, p, t, v, t, v, v, q, p, p, p, p, t, v, v, q), dim=-3)
    return torch.gather(out, -3, idx)


def color_jitter(x, params, order):
    """x [B,3,H,W]; params [B,4] = (brightness, contrast, saturation, hue) factors; order: permutation of 0..3"""
    bf, cf, sf, hf = (params[:, i].reshape(-1, 1, 1, 1).to(x.dtype) for i in range(4))
    for op in order:
        if op == 0:
            x = torch.clamp(x + (bf - 1), 0, 1)
        elif op == 1:
            x = torch.clamp(x * cf, 0, 1)
        elif op == 2:
            hsv = rgb_to_hsv(x)
            x = hsv_to_rgb(torch.stack([hsv[:, 0], torch.clamp(hsv[:, 1] * sf[:, 0], 0, 1), hsv[:, 2]], 1))
        else:
            hsv = rgb_to_hsv(x)
            x = hsv_to_rgb(torch.stack([torch.fmod(hsv[:, 0] + hf[:, 0] * 2 * math.pi, 2 * math.pi), hsv[:, 1], hsv[:, 2]], 1))
    return x


def gaussian_kernel1d(k, sigma, dtype=torch.float32):
    x = torch.arange(k, dtype=dtype) - k // 2
    g = torch.exp(-x ** 2 / float(2 * sigma ** 2))
    return g / g.sum()


def gaussian_blur(x, kernel_size, sigma):
    """kornia.filters.GaussianBlur2d(kernel_size=(ky, kx), sigma=(sigma, sigma)) -- filter2D, border_type='reflect'"""
    ky, kx = kernel_size
    k2 = torch.outer(gaussian_kernel1d(ky, sigma, x.dtype), gaussian_kernel1d(kx, sigma, x.dtype))
    C = x.shape[1]
    xp = F.pad(x, (kx // 2, kx // 2, ky // 2, ky // 2), mode="reflect")
    return F.conv2d(xp, k2[None, None].repeat(C, 1, 1, 1), groups=C)
