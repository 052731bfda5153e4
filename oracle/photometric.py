"""Oracle: photometric reprojection + SSIM + smoothness loss
(TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/loss/monodepth_loss.py and
/root/reference/models/monodepth_layers.py:208-254.
"""
import torch
import torch.nn.functional as F

from . import geometry as G

SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2


def ssim_dissimilarity(x, y):
    """monodepth_layers.py:224-254: clamp((1-SSIM)/2, 0, 1), 3x3 mean window, reflect pad 1."""
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    yp = F.pad(y, (1, 1, 1, 1), mode="reflect")
    box = lambda t: F.avg_pool2d(t, 3, 1)
    mu_x, mu_y = box(xp), box(yp)
    sig_x = box(xp * xp) - mu_x * mu_x
    sig_y = box(yp * yp) - mu_y * mu_y
    sig_xy = box(xp * yp) - mu_x * mu_y
    num = (2 * mu_x * mu_y + SSIM_C1) * (2 * sig_xy + SSIM_C2)
    den = (mu_x * mu_x + mu_y * mu_y + SSIM_C1) * (sig_x + sig_y + SSIM_C2)
    return torch.clamp((1 - num / den) / 2, 0, 1)


def edge_aware_smoothness(disp, img):
    """monodepth_layers.py:208-221."""
    dx = (disp[:, :, :, :-1] - disp[:, :, :, 1:]).abs()
    dy = (disp[:, :, :-1, :] - disp[:, :, 1:, :]).abs()
    ix = (img[:, :, :, :-1] - img[:, :, :, 1:]).abs().mean(1, keepdim=True)
    iy = (img[:, :, :-1, :] - img[:, :, 1:, :]).abs().mean(1, keepdim=True)
    return (dx * torch.exp(-ix)).mean() + (dy * torch.exp(-iy)).mean()


def reprojection_error(pred, target, no_ssim=False):
    """monodepth_loss.py:104-116 -> [B,1,H,W]."""
    l1 = (target - pred).abs().mean(1, keepdim=True)
    if no_ssim:
        return l1
    return 0.85 * ssim_dissimilarity(pred, target).mean(1, keepdim=True) + 0.15 * l1


class MonodepthLossOracle:
    """Same constructor kwargs / method protocol as the reference MonodepthLoss
    (monodepth_loss.py:16-52), restated functionally.

    ``tiebreak_noise``: optional dict scale -> tensor replacing the fresh
    ``torch.randn(...)`` of monodepth_loss.py:163-164 (already *not* multiplied
    by 1e-5), so golden vectors can pin the automask selection.
    """

    def __init__(self, num_scales, frame_ids, height, width, batch_size, min_depth, max_depth,
                 test_min_depth, test_max_depth, disparity_smoothness, no_ssim, avg_reprojection,
                 disable_automasking, crop_h=None, crop_w=None, is_train=True):
        self.num_scales = num_scales
        self.frame_ids = list(frame_ids)
        # monodepth_loss.py:22-23
        self.height = height if crop_h is None or not is_train else crop_h
        self.width = width if crop_w is None or not is_train else crop_w
        self.batch_size = batch_size
        self.min_depth, self.max_depth = min_depth, max_depth
        self.test_min_depth, self.test_max_depth = test_min_depth, test_max_depth
        self.disparity_smoothness = disparity_smoothness
        self.no_ssim = no_ssim
        self.avg_reprojection = avg_reprojection
        self.disable_automasking = disable_automasking

    def generate_depth_test_pred(self, outputs):
        """monodepth_loss.py:54-62."""
        for s in range(self.num_scales):
            d = F.interpolate(outputs[("disp", s)], [self.height, self.width], mode="bilinear", align_corners=False)
            outputs[("depth", 0, s)] = G.disp_to_depth(d, self.test_min_depth, self.test_max_depth)[1]

    def generate_images_pred(self, inputs, outputs):
        """monodepth_loss.py:64-102."""
        H, W = self.height, self.width
        assert tuple(outputs[("disp", 0)].shape[-2:]) == (H, W)
        for s in range(self.num_scales):
            d = F.interpolate(outputs[("disp", s)], [H, W], mode="bilinear", align_corners=False)
            depth = G.disp_to_depth(d, self.min_depth, self.max_depth)[1]
            outputs[("depth", 0, s)] = depth
            for f in self.frame_ids[1:]:
                T = inputs["stereo_T"] if f == "s" else outputs[("cam_T_cam", 0, f)]
                pts = G.backproject(depth, inputs[("inv_K", 0)])
                grid = G.project(pts, inputs[("K", 0)], T, H, W)
                outputs[("sample", f, s)] = grid
                outputs[("color", f, s)] = G.warp(inputs[("color", f, 0)], grid)
                if not self.disable_automasking:
                    outputs[("color_identity", f, s)] = inputs[("color", f, 0)]

    def compute_losses(self, inputs, outputs, tiebreak_noise=None):
        """monodepth_loss.py:118-192."""
        losses, total = {}, 0.0
        target = inputs[("color", 0, 0)]
        for s in range(self.num_scales):
            disp = outputs[("disp", s)]
            color = inputs[("color", 0, s)]
            reproj = torch.cat([reprojection_error(outputs[("color", f, s)], target, self.no_ssim)
                                for f in self.frame_ids[1:]], 1)
            if self.avg_reprojection:
                reproj = reproj.mean(1, keepdim=True)
            if not self.disable_automasking:
                ident = torch.cat([reprojection_error(inputs[("color", f, 0)], target, self.no_ssim)
                                   for f in self.frame_ids[1:]], 1)
                if self.avg_reprojection:
                    ident = ident.mean(1, keepdim=True)
                if tiebreak_noise is not None:
                    noise = tiebreak_noise[s].to(ident)
                else:
                    noise = torch.randn(ident.shape).to(ident)
                ident = ident + noise * 0.00001
                combined = torch.cat([ident, reproj], 1)
            else:
                combined = reproj
            if combined.shape[1] == 1:
                to_opt = combined
            else:
                to_opt, idx = torch.min(combined, dim=1)
            if not self.disable_automasking:
                outputs["identity_selection/{}".format(s)] = (idx > ident.shape[1] - 1).float()
            loss = to_opt.mean()
            mean_disp = disp.mean(2, True).mean(3, True)
            norm_disp = disp / (mean_disp + 1e-7)
            loss = loss + self.disparity_smoothness * edge_aware_smoothness(norm_disp, color) / (2 ** s)
            total = total + loss
            losses["loss/{}".format(s)] = loss
        losses["loss"] = total / self.num_scales
        return losses
