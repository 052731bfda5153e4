"""Oracle: segmentation loss + DepthMix/ClassMix masks and composite
(TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import torch
import torch.nn.functional as F


def cross_entropy2d(logits, target, class_weight=None, pixel_weights=None):
    """/root/reference/loss/loss.py:17-37 (ignore_index=250)."""
    n, c, h, w = logits.shape
    _, ht, wt = target.shape
    if h != ht and w != wt:
        logits = F.interpolate(logits, size=(ht, wt), mode="bilinear", align_corners=True)
    flat = logits.permute(0, 2, 3, 1).reshape(-1, c)
    tgt = target.reshape(-1)
    loss = F.cross_entropy(flat, tgt, weight=class_weight,
                           reduction="mean" if pixel_weights is None else "none", ignore_index=250)
    if pixel_weights is not None:
        if not torch.any(torch.isnan(pixel_weights)):
            loss = pixel_weights.reshape(-1).detach() * loss
        loss = loss.mean()
    return loss


def mix(mask, data=None, target=None):
    """/root/reference/loader/transformsgpu.py:33-47.
    out_i = m_i*x_i + (1-m_i)*x_{(i+1)%B}; half-batch branch when mask has B/2 rows."""
    def full(m, x):
        B = x.shape[0]
        return torch.stack([m[i] * x[i] + (1 - m[i]) * x[(i + 1) % B] for i in range(B)], 0)
    if data is not None:
        if mask.shape[0] == data.shape[0]:
            data = full(mask, data)
        elif mask.shape[0] == data.shape[0] / 2:
            half = data.shape[0] // 2
            a = torch.stack([mask[i] * data[2 * i] + (1 - mask[i]) * data[2 * i + 1] for i in range(half)], 0)
            b = torch.stack([(1 - mask[i]) * data[2 * i] + mask[i] * data[2 * i + 1] for i in range(half)], 0)
            data = torch.cat([a, b], 0)
    if target is not None:
        target = full(mask, target)
    return data, target


def generate_class_mask(pred, classes):
    """/root/reference/loader/transformmasks.py:27-30 -> int64 [H,W]."""
    return (pred.unsqueeze(0) == classes.reshape(-1, 1, 1)).sum(0)


def generate_depth_mask(depth, threshold):
    """/root/reference/loader/transformmasks.py:33-41."""
    if threshold.shape[0] == 1:
        return depth.ge(threshold).float()
    if threshold.shape[0] == 2:
        return depth.ge(threshold.min()).le(threshold.max()).float()
    raise NotImplementedError


def depthcomp_mask(depths, margin, foreground_threshold):
    """/root/reference/train.py:585-604 generalised to partner (i+1)%B
    (identical to the reference at its asserted B==2).  depths [B,1,H,W] ->
    int64 [B,H,W]."""
    B = depths.shape[0]
    out = []
    for i in range(B):
        own, other = depths[i], depths[(i + 1) % B]
        m = torch.ge(own, other - margin).long()
        m = m * torch.ge(own, foreground_threshold).long()
        out.append(m)
    return torch.cat(out, 0)


def normalize_disparity(disp):
    """/root/reference/train.py:692-697: per-sample min-max normalisation of disp0."""
    out = disp.clone()
    for j in range(out.shape[0]):
        lo, hi = out[j].min(), out[j].max()
        out[j] = (torch.clamp(out[j], lo, hi) - lo) / (hi - lo)
    return out


def pseudo_label(teacher_softmax, ignore_index=250, threshold=0.968):
    """/root/reference/train.py:644-648 -> (pseudo_label int64 [B,H,W], unlabeled_weight float)."""
    max_probs, lab = torch.max(teacher_softmax, dim=1)
    lab = lab.clone()
    lab[max_probs == 0] = ignore_index
    weight = (max_probs.ge(threshold).long() == 1).sum().item() / float(lab.numel())
    return lab, weight


def depth_estimate_u8(disp0):
    """/root/reference/loader/depth_estimator.py:83-91: per image clamp, min-max normalise, ToPILImage (float tensor ->
    mul(255).byte()).  disp0 [B,1,H,W] -> uint8 [B,H,W]."""
    out = []
    for depth in disp0:
        dmin, dmax = torch.min(depth), torch.max(depth)
        depth = (torch.clamp(depth, dmin, dmax) - dmin) / (dmax - dmin)
        out.append(depth.squeeze(0).mul(255).byte())
    return torch.stack(out)
