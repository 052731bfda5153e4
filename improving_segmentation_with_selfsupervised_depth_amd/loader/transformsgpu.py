"""Mirror of loader/transformsgpu.py: ``mix`` (DepthMix / ClassMix composite) on a single HIP kernel, bit-exact with
the reference's per-sample ``m*x_i + (1-m)*x_{(i+1)%B}`` loop (transformsgpu.py:33-47).  ``color_jitter`` /
``gaussian_blur`` are kornia 0.4.0 wrappers in the reference (third-party, absent here: SURVEY.md 8f item 2); they run on
HIP kernels that restate kornia 0.4.0's published algorithm (csrc/augment.hip) -- parity unpinned, see DESIGN.md."""
import numpy as np
import torch

from .. import hipops as H


def _mask(mask):
    return mask if mask.dtype in (torch.int64, torch.float32) else mask.float()


def mix(mask, data=None, target=None):
    if data is not None:
        if mask.shape[0] == data.shape[0] or mask.shape[0] == data.shape[0] / 2:
            data = H.mix(_mask(mask), data.float())
    if target is not None:
        if target.dtype == torch.int64 and mask.dtype == torch.int64:
            target = H.mix_labels(mask, target)
        else:
            target = H.mix(_mask(mask), target.float().unsqueeze(1)).squeeze(1)
    return data, target


def sample_color_jitter_params(batch_size, s=0.25, generator=None):
    """kornia 0.4.0 random_color_jitter_generator for ColorJitter(brightness=s, contrast=s, saturation=s, hue=s): per sample
    brightness / contrast / saturation factors ~ U[1 - s, 1 + s] (bounded to [0, 2]), hue factor ~ U[-s, s] (bounded to
    [-0.5, 0.5]), one random order of the four adjustments for the whole batch.  -> (params [B,4] on the CPU, order list)"""
    lo, hi = max(0.0, 1.0 - s), min(2.0, 1.0 + s)
    u = torch.rand(batch_size, 4, generator=generator)
    params = torch.empty(batch_size, 4)
    params[:, :3] = lo + (hi - lo) * u[:, :3]
    hs = min(0.5, s)
    params[:, 3] = -hs + 2 * hs * u[:, 3]
    order = torch.randperm(4, generator=generator).tolist()
    return params, order


def gaussian_taps(kernel_size, sigma):
    """kornia 0.4.0 get_gaussian_kernel1d: exp(-(x - k // 2)^2 / (2 sigma^2)), x = 0..k-1, normalised to sum 1 (fp32), cut
    down to its non-zero support (taps that underflow to exactly 0 cannot contribute)"""
    x = torch.arange(kernel_size, dtype=torch.float32) - kernel_size // 2
    g = torch.exp(-x ** 2 / float(2 * sigma ** 2))
    g = g / g.sum()
    nz = torch.nonzero(g).flatten()
    r = int(max(kernel_size // 2 - int(nz[0]), int(nz[-1]) - kernel_size // 2))
    return g[kernel_size // 2 - r:kernel_size // 2 + r + 1].contiguous()


def blur_kernel_size(n):
    """reference transformsgpu.py:26-27: an odd size of about 0.1 * n"""
    return int(np.floor(np.ceil(0.1 * n) - 0.5 + np.ceil(0.1 * n) % 2))


def color_jitter(jitter, data=None, target=None, s=0.25, params=None, order=None):
    """reference transformsgpu.py:10-17 (kornia ColorJitter(s, s, s, s) when jitter > 0.2).  ``params`` / ``order``: optional
    pre-sampled parameters (tests); otherwise sampled here with torch's CPU generator."""
    if data is not None and data.shape[1] == 3 and jitter > 0.2:
        if params is None:
            params, order = sample_color_jitter_params(data.shape[0], s)
        data = H.color_jitter(data.float(), params.to(data.device), order)
    return data, target


def gaussian_blur(blur, data=None, target=None, sigma=None):
    """reference transformsgpu.py:20-30 (kornia GaussianBlur2d with kernel ~ 0.1 * (H, W), sigma ~ U[0.15, 1.15] when
    blur > 0.5).  ``sigma``: optional fixed value (tests)."""
    if data is not None and data.shape[1] == 3 and blur > 0.5:
        if sigma is None:
            sigma = np.random.uniform(0.15, 1.15)
        ky, kx = blur_kernel_size(data.shape[2]), blur_kernel_size(data.shape[3])
        if ky // 2 >= data.shape[2] or kx // 2 >= data.shape[3]:
            raise ValueError("image too small for reflection padding of a %dx%d blur kernel" % (ky, kx))
        data = H.gaussian_blur(data.float(), gaussian_taps(ky, sigma).to(data.device), gaussian_taps(kx, sigma).to(data.device))
    return data, target
