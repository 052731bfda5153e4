"""Mirror of loss/loss.py: cross_entropy2d on the fused HIP log-softmax + NLL kernels (reads NHWC logits in place:
the reference's NCHW -> (NHW, C) transpose copy, loss.py:25, disappears)."""
import torch

from .. import functional as Fn
from .. import hipops as H

IGNORE_INDEX = 250


class _CrossEntropyFn(Fn.Function):
    @staticmethod
    def forward(ctx, logits, target, class_weight, pixel_weights, mean_over_all):
        logits = Fn._c(logits)
        target = target.contiguous()
        out = H.cross_entropy_forward(logits, target, IGNORE_INDEX, class_weight, pixel_weights)
        M = target.numel()
        den = torch.full((), float(M), device=logits.device) if mean_over_all else out[1]
        ctx.save_for_backward(logits, target, class_weight, pixel_weights, den)
        return out[0] / den

    @staticmethod
    def backward(ctx, g):
        logits, target, cw, pw, den = ctx.saved_tensors
        scale = (g / den).reshape(1).contiguous()
        return H.cross_entropy_backward(logits, target, IGNORE_INDEX, scale, cw, pw), None, None, None, None


@Fn.fp32_region
def cross_entropy2d(input, target, class_weight=None, pixel_weights=None):
    """reference loss.py:17-37.  input: [N,C,H,W] logits (NCHW-logical), target: int64 [N,Ht,Wt]."""
    n, c, h, w = input.size()
    nt, ht, wt = target.size()
    x = Fn.to_nhwc(input)
    if h != ht and w != wt:
        x = Fn.resize_bilinear(x, (ht, wt), align_corners=True)
    mean_over_all = False
    if pixel_weights is not None:
        mean_over_all = True          # reduction="none" followed by torch.mean over every pixel (loss.py:28-36)
        # reference loss.py:29-31 checks the weights for NaN on the host (a device sync per call).  Weights made by
        # hipops.pseudo_label (count / total, a finite number by construction) carry a marker and skip that check.
        if not getattr(pixel_weights, "_segsde_finite", False) and torch.any(torch.isnan(pixel_weights)):
            print("WARN cross_entropy2d pixel_weights contains NaN. Skip weighting.")
            pixel_weights = None
        else:
            pixel_weights = pixel_weights.detach().reshape(-1).float().contiguous()
    return _CrossEntropyFn.apply(x, target.reshape(-1), class_weight, pixel_weights, mean_over_all)


def berhu(input, target, mask, apply_log=False):
    """reference loss.py:5-15 (pseudo-depth distillation; off in every benchmark config).  Plain torch ops."""
    threshold = 0.2
    if apply_log:
        input, target = torch.log(1 + input), torch.log(1 + target)
    absdiff = torch.abs(target - input) * mask
    C = threshold * torch.max(absdiff).item()
    return torch.mean(torch.where(absdiff <= C, absdiff, (absdiff * absdiff + C * C) / (2 * C)))
