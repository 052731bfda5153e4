"""Mirror of loader/transformmasks.py mask builders + the depthcomp mask of Trainer.generate_mix_mask
(train.py:585-604), each one HIP kernel, bit-exact (comparisons only)."""
import torch

from .. import hipops as H


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
