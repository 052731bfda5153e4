"""Mirror of loader/transformsgpu.py: ``mix`` (DepthMix / ClassMix composite) on a single HIP kernel, bit-exact with
the reference's per-sample ``m*x_i + (1-m)*x_{(i+1)%B}`` loop (transformsgpu.py:33-47).  ``color_jitter`` /
``gaussian_blur`` are kornia 0.4.0 wrappers in the reference (third-party): not built yet."""
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


def color_jitter(jitter, data=None, target=None, s=0.25):
    if data is not None and data.shape[1] == 3 and jitter > 0.2:
        raise NotImplementedError("kornia ColorJitter (transformsgpu.py:10-17) is listed as 'next' in SURVEY.md 8f")
    return data, target


def gaussian_blur(blur, data=None, target=None):
    if data is not None and data.shape[1] == 3 and blur > 0.5:
        raise NotImplementedError("kornia GaussianBlur2d (transformsgpu.py:20-30) is listed as 'next' in SURVEY.md 8f")
    return data, target
