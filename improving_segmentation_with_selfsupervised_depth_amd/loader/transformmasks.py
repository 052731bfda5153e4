"""Mirror of loader/transformmasks.py mask builders + the depthcomp mask of Trainer.generate_mix_mask
(train.py:585-604), each one HIP kernel, bit-exact (comparisons only)."""
import numpy as np
import torch

from .. import hipops as H


def generate_cutout_mask(img_size, seed=None):
    """transformmasks.py:8-24 (CutMix-style box mask; unused by the reference's trainer, kept for scripts that import it):
    host-side numpy like the reference -- a [H, W] float64 array of ones with one zeroed box of half the image area, drawn
    from numpy's global generator after ``np.random.seed(seed)`` (same draws, same order: bit-identical masks)."""
    np.random.seed(seed)
    area = img_size[0] * img_size[1] / 2
    w = np.random.randint(img_size[1] / 2, img_size[1] + 1)
    h = np.round(area / w)
    x0 = np.random.randint(0, img_size[1] - w + 1)
    y0 = np.random.randint(0, img_size[0] - h + 1)
    mask = np.ones(img_size)
    mask[y0:int(y0 + h), x0:int(x0 + w)] = 0
    return mask.astype(float)


def generate_class_mask(pred, classes):
    """transformmasks.py:27-30 -> int64, same shape as pred"""
    return H.class_mask(pred, classes)


def generate_depth_mask(depth, threshold):
    """transformmasks.py:33-41"""
    if threshold.shape[0] == 1:
        return H.depth_threshold_mask(depth.float(), float(threshold[0]))
    if threshold.shape[0] == 2:
        return H.depth_threshold_mask(depth.float(), float(torch.min(threshold)), float(torch.max(threshold)), True)
    raise NotImplementedError


def generate_depthcomp_mask(depths, margin, foreground_threshold):
    """train.py:585-604 with partner (i+1) % B (identical to the reference at its asserted batch size 2).
    depths: [B,1,H,W] min-max-normalised disparities -> int64 [B,H,W]."""
    return H.depthcomp_mask(depths.float(), margin, foreground_threshold)
