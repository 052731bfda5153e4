"""Oracle: monodepth geometry (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/models/monodepth_layers.py (line refs per function).
Written functionally; no nn.Module state (the reference bakes batch size and
H,W into module buffers, monodepth_layers.py:148-167).
"""
import torch
import torch.nn.functional as F


def disp_to_depth(disp, min_depth, max_depth):
    """monodepth_layers.py:18-27 -> (scaled_disp, depth)."""
    lo = 1.0 / max_depth
    hi = 1.0 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1.0 / scaled


def rotation_from_axisangle(vec):
    """monodepth_layers.py:66-105.  vec [B,1,3] -> [B,4,4] (Rodrigues)."""
    angle = vec.norm(p=2, dim=2, keepdim=True)          # [B,1,1]
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1.0 - ca
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))  # [B,1,1]
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    B = vec.shape[0]
    rows = [
        [x * xC + ca, xyC - zs, zxC + ys],
        [xyC + zs, y * yC + ca, yzC - xs],
        [zxC - ys, yzC + xs, z * zC + ca],
    ]
    R = torch.zeros(B, 4, 4, dtype=vec.dtype, device=vec.device)
    for i in range(3):
        for j in range(3):
            R[:, i, j] = rows[i][j].reshape(B)
    R[:, 3, 3] = 1
    return R


def translation_matrix(t):
    """monodepth_layers.py:49-63.  t [B,1,3] or [B,3] -> [B,4,4]."""
    t = t.reshape(-1, 3)
    T = torch.zeros(t.shape[0], 4, 4, dtype=t.dtype, device=t.device)
    T[:, 0, 0] = T[:, 1, 1] = T[:, 2, 2] = T[:, 3, 3] = 1
    T[:, :3, 3] = t
    return T


def pose_matrix(axisangle, translation, invert=False):
    """monodepth_layers.py:30-46 (transformation_from_parameters)."""
    R = rotation_from_axisangle(axisangle)
    t = translation
    if invert:
        R = R.transpose(1, 2)
        t = -t
    T = translation_matrix(t)
    return R @ T if invert else T @ R


def backproject(depth, inv_K):
    """monodepth_layers.py:145-174.  depth [B,1,H,W] -> homogeneous [B,4,H*W]."""
    B, _, H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, dtype=depth.dtype, device=depth.device),
                          torch.arange(W, dtype=depth.dtype, device=depth.device), indexing="ij")
    pix = torch.stack([u.reshape(-1), v.reshape(-1), torch.ones(H * W, dtype=depth.dtype, device=depth.device)], 0)
    pix = pix.unsqueeze(0).expand(B, -1, -1)
    cam = inv_K[:, :3, :3] @ pix
    cam = depth.reshape(B, 1, -1) * cam
    ones = torch.ones(B, 1, H * W, dtype=depth.dtype, device=depth.device)
    return torch.cat([cam, ones], 1)


def project(points, K, T, H, W, eps=1e-7):
    """monodepth_layers.py:177-199 -> sampling grid [B,H,W,2] in [-1,1]."""
    B = points.shape[0]
    P = (K @ T)[:, :3, :]
    cam = P @ points
    pix = cam[:, :2, :] / (cam[:, 2:3, :] + eps)
    pix = pix.reshape(B, 2, H, W).permute(0, 2, 3, 1)
    gx = pix[..., 0] / (W - 1)
    gy = pix[..., 1] / (H - 1)
    return (torch.stack([gx, gy], -1) - 0.5) * 2


def warp(src, grid):
    """loss/monodepth_loss.py:94-98: bilinear, border padding, align_corners=True."""
    return F.grid_sample(src, grid, mode="bilinear", padding_mode="border", align_corners=True)
