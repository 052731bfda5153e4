#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's
own Python code (imported from /root/reference) on seeded inputs.

Run in the build container only (``python tests/golden/make_golden.py``); the
reference never travels to the GPU box -- only the .npz/.json data written here
does.  No reference source is copied: this script imports it, feeds it tensors
and stores inputs + outputs (+ gradients).

Stubs needed to import the reference here:
  * torchvision (absent): tests/golden/_tv_standin.py -- our restatement;
  * kornia (absent): empty module (only color_jitter / gaussian_blur use it);
  * torch.utils.tensorboard (absent): empty SummaryWriter, to import train.py
    for Trainer.generate_mix_mask (train.py:572-642);
  * package ``loader``: its __init__ pulls PIL dataset loaders -> bypassed with a
    namespace module so that loader.transformsgpu / transformmasks import alone.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import _tv_standin  # noqa: E402

_tv_standin.install()
sys.path.insert(0, REF)
_loader = types.ModuleType("loader")
_loader.__path__ = [os.path.join(REF, "loader")]
_loader.build_loader = None
sys.modules["loader"] = _loader
_de = types.ModuleType("loader.depth_estimator")
_de.DepthEstimator = None
sys.modules["loader.depth_estimator"] = _de
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb

import models.monodepth_layers as ref_layers  # noqa: E402
from loss.monodepth_loss import MonodepthLoss as RefMonodepthLoss  # noqa: E402
from loss.loss import cross_entropy2d as ref_ce  # noqa: E402
from loader import transformsgpu as ref_tg, transformmasks as ref_tm  # noqa: E402
from models import get_model as ref_get_model  # noqa: E402
from models.depth_decoder import DepthDecoder as RefDepthDecoder  # noqa: E402
from models.joint_segmentation_depth_decoder import JointSegDepthDecoder as RefJSD, PAD as RefPAD  # noqa: E402
from models.model_parts import SelfAttention as RefSA, ASPP as RefASPP  # noqa: E402
from models.pose_decoder import PoseDecoder as RefPoseDecoder  # noqa: E402
from models.resnet_encoder import ResnetEncoder as RefResnetEncoder  # noqa: E402

from oracle import nets as onets  # noqa: E402

torch.set_num_threads(8)


def np_(t):
    return t.detach().cpu().numpy()


OUT_DIR = [HERE]        # --check redirects the generators into a scratch directory


def save(name, d):
    path = os.path.join(OUT_DIR[0], name + ".npz")
    np.savez_compressed(path, **{k: (np_(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()})
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def dropout_eval(module):
    """Parity runs: disable nn.Dropout/Dropout2d only (BN stays in train mode)."""
    for m in module.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()


# ---------------------------------------------------------------------------
def make_loss_inputs(B, H, W, gen):
    """Seeded loss inputs: smooth-ish images so SSIM/warp are non-degenerate."""
    def img():
        lo = torch.rand(B, 3, H // 4, W // 4, generator=gen)
        return (F.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False) * 0.8
                + 0.2 * torch.rand(B, 3, H, W, generator=gen)).contiguous()
    inputs = {}
    for f in (0, -1, 1):
        inputs[("color", f, 0)] = img()
    for s in range(1, 4):
        inputs[("color", 0, s)] = F.avg_pool2d(inputs[("color", 0, 0)], 2 ** s)
    K = torch.tensor([[1.1 * W, 0, 0.5 * W, 0], [0, 1.1 * W, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
                     dtype=torch.float32)
    K = K.unsqueeze(0).repeat(B, 1, 1)
    K[1, 0, 2] += 1.5  # per-sample intrinsics differ
    inputs[("K", 0)] = K
    inputs[("inv_K", 0)] = torch.from_numpy(np.stack([np.linalg.pinv(k) for k in K.numpy()])).float()
    disps = {s: (0.02 + 0.96 * torch.rand(B, 1, H // 2 ** s, W // 2 ** s, generator=gen)) for s in range(4)}
    aa = 0.02 * torch.randn(B, 2, 1, 3, generator=gen)
    tr = 0.04 * torch.randn(B, 2, 1, 3, generator=gen)
    return inputs, disps, aa, tr


def gen_loss():
    B, H, W = 2, 32, 64
    base = dict(num_scales=4, frame_ids=[0, -1, 1], height=H, width=W, batch_size=B, min_depth=0.1, max_depth=100,
                test_min_depth=1e-3, test_max_depth=80, disparity_smoothness=1e-3, no_ssim=False,
                avg_reprojection=False, disable_automasking=False)
    variants = {"default": {}, "no_ssim": {"no_ssim": True}, "avg_reprojection": {"avg_reprojection": True},
                "disable_automasking": {"disable_automasking": True}}
    for vi, (name, over) in enumerate(variants.items()):
        gen = torch.Generator().manual_seed(100 + vi)
        inputs, disps, aa, tr = make_loss_inputs(B, H, W, gen)
        cfg = dict(base, **over)
        loss_obj = RefMonodepthLoss(**cfg)
        out = {}
        dleaf = {s: d.clone().requires_grad_(True) for s, d in disps.items()}
        for s in range(4):
            out[("disp", s)] = dleaf[s]
        Tleaf = {}
        for i, f in enumerate((-1, 1)):
            T = ref_layers.transformation_from_parameters(aa[:, i], tr[:, i], invert=(f < 0))
            Tleaf[f] = T.detach().clone().requires_grad_(True)
            out[("cam_T_cam", 0, f)] = Tleaf[f]
        # capture the tie-break noise of monodepth_loss.py:163-164
        nch = 1 if cfg["avg_reprojection"] else 2
        noise = {s: torch.randn(B, nch, H, W, generator=gen) for s in range(4)}
        queue = [noise[s] for s in range(4)]
        real_randn = torch.randn

        def fake_randn(*a, **k):
            return queue.pop(0)
        loss_obj.generate_images_pred(inputs, out)
        torch.randn = fake_randn
        try:
            losses = loss_obj.compute_losses(inputs, out)
        finally:
            torch.randn = real_randn
        losses["loss"].backward()
        d = {"cfg_json": json.dumps(cfg), "axisangle": aa, "translation": tr}
        for k, v in inputs.items():
            d["in_%s_%s_%s" % (k[0], k[1], k[2]) if len(k) == 3 else "in_%s_%s" % k] = v
        for s in range(4):
            d["disp_%d" % s] = disps[s]
            d["grad_disp_%d" % s] = dleaf[s].grad
            d["loss_%d" % s] = losses["loss/%d" % s]
            d["depth_%d" % s] = out[("depth", 0, s)]
            if not cfg["disable_automasking"]:
                d["noise_%d" % s] = noise[s]
                d["identity_selection_%d" % s] = out["identity_selection/%d" % s]
            for f in (-1, 1):
                tag = "m1" if f < 0 else "p1"
                if s in (0, 2):
                    d["sample_%s_%d" % (tag, s)] = out[("sample", f, s)]
                    d["color_%s_%d" % (tag, s)] = out[("color", f, s)]
        for f in (-1, 1):
            tag = "m1" if f < 0 else "p1"
            d["T_%s" % tag] = Tleaf[f]
            d["grad_T_%s" % tag] = Tleaf[f].grad
        d["loss"] = losses["loss"]
        save("loss_" + name, d)


def gen_loss_stereo():
    """the stereo-only frame set (0, "s") of the reference's loss (monodepth_loss.py:82-85: the stereo frame is warped with the
    batch's fixed inputs["stereo_T"]; one source frame, so the auto-mask's noise is [B, 1, H, W] and no pose takes a gradient),
    with and without auto-masking / averaging"""
    B, H, W = 2, 32, 64
    base = dict(num_scales=4, frame_ids=[0, "s"], height=H, width=W, batch_size=B, min_depth=0.1, max_depth=100,
                test_min_depth=1e-3, test_max_depth=80, disparity_smoothness=1e-3, no_ssim=False,
                avg_reprojection=False, disable_automasking=False)
    variants = {"default": {}, "avg_reprojection": {"avg_reprojection": True}, "disable_automasking": {"disable_automasking": True}}
    d = {}
    for vi, (name, over) in enumerate(variants.items()):
        gen = torch.Generator().manual_seed(300 + vi)
        inputs, disps, _, _ = make_loss_inputs(B, H, W, gen)
        inputs[("color", "s", 0)] = inputs.pop(("color", 1, 0))
        inputs.pop(("color", -1, 0))
        T = torch.eye(4).unsqueeze(0).repeat(B, 1, 1)
        T[0, 0, 3], T[1, 0, 3] = 0.1, -0.1                  # monodepth2's baseline: +-0.1 by the side of the camera
        T[:, 1, 3] = 0.003 * torch.randn(B, generator=gen)
        inputs["stereo_T"] = T
        cfg = dict(base, **over)
        loss_obj = RefMonodepthLoss(**cfg)
        dleaf = {s_: disps[s_].clone().requires_grad_(True) for s_ in range(4)}
        out = {("disp", s_): dleaf[s_] for s_ in range(4)}
        noise = {s_: torch.randn(B, 1, H, W, generator=gen) for s_ in range(4)}
        queue = [noise[s_] for s_ in range(4)]
        real_randn = torch.randn
        loss_obj.generate_images_pred(inputs, out)
        torch.randn = lambda *a, **k: queue.pop(0)
        try:
            losses = loss_obj.compute_losses(inputs, out)
        finally:
            torch.randn = real_randn
        losses["loss"].backward()
        d[name + "_cfg_json"] = json.dumps(cfg)
        for k in (("color", 0, 0), ("color", "s", 0), ("color", 0, 1), ("color", 0, 2), ("color", 0, 3), ("K", 0), ("inv_K", 0)):
            d[name + "_in_" + "_".join(str(x) for x in k)] = inputs[k]
        d[name + "_stereo_T"] = T
        d[name + "_loss"] = losses["loss"]
        for s_ in range(4):
            d[name + "_disp_%d" % s_] = disps[s_]
            d[name + "_grad_disp_%d" % s_] = dleaf[s_].grad
            d[name + "_loss_%d" % s_] = losses["loss/%d" % s_]
            if not cfg["disable_automasking"]:
                d[name + "_noise_%d" % s_] = noise[s_]
                d[name + "_identity_selection_%d" % s_] = out["identity_selection/%d" % s_]
        d[name + "_color_s_0"] = out[("color", "s", 0)]
        d[name + "_sample_s_0"] = out[("sample", "s", 0)]
    save("loss_stereo", d)


def gen_loss_frames4():
    """monodepth2's four-frame set (0, -1, 1, "s") through the reference's loss (monodepth_loss.py:80-85 picks the pose per frame,
    :136-177 loops over however many source frames there are): three source frames, so the auto-mask's noise is [B, 3, H, W] (or
    [B, 1, H, W] when averaging) and the minimum runs over six (two) candidates."""
    B, H, W = 2, 32, 64
    frames = [-1, 1, "s"]
    base = dict(num_scales=4, frame_ids=[0] + frames, height=H, width=W, batch_size=B, min_depth=0.1, max_depth=100,
                test_min_depth=1e-3, test_max_depth=80, disparity_smoothness=1e-3, no_ssim=False,
                avg_reprojection=False, disable_automasking=False)
    variants = {"default": {}, "no_ssim": {"no_ssim": True}, "avg_reprojection": {"avg_reprojection": True},
                "disable_automasking": {"disable_automasking": True}}
    d = {}
    for vi, (name, over) in enumerate(variants.items()):
        gen = torch.Generator().manual_seed(400 + vi)
        inputs, disps, aa, tr = make_loss_inputs(B, H, W, gen)
        lo = torch.rand(B, 3, H // 4, W // 4, generator=gen)
        inputs[("color", "s", 0)] = (F.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False) * 0.8
                                     + 0.2 * torch.rand(B, 3, H, W, generator=gen)).contiguous()
        T = torch.eye(4).unsqueeze(0).repeat(B, 1, 1)
        T[0, 0, 3], T[1, 0, 3] = 0.1, -0.1
        T[:, 1, 3] = 0.003 * torch.randn(B, generator=gen)
        inputs["stereo_T"] = T
        cfg = dict(base, **over)
        loss_obj = RefMonodepthLoss(**cfg)
        dleaf = {s_: disps[s_].clone().requires_grad_(True) for s_ in range(4)}
        out = {("disp", s_): dleaf[s_] for s_ in range(4)}
        Tleaf = {}
        for i, f in enumerate((-1, 1)):
            Tf = ref_layers.transformation_from_parameters(aa[:, i], tr[:, i], invert=(f < 0))
            Tleaf[f] = Tf.detach().clone().requires_grad_(True)
            out[("cam_T_cam", 0, f)] = Tleaf[f]
        nch = 1 if cfg["avg_reprojection"] else 3
        noise = {s_: torch.randn(B, nch, H, W, generator=gen) for s_ in range(4)}
        queue = [noise[s_] for s_ in range(4)]
        real_randn = torch.randn
        loss_obj.generate_images_pred(inputs, out)
        torch.randn = lambda *a, **k: queue.pop(0)
        try:
            losses = loss_obj.compute_losses(inputs, out)
        finally:
            torch.randn = real_randn
        losses["loss"].backward()
        d[name + "_cfg_json"] = json.dumps(cfg)
        for k in [("color", f, 0) for f in [0] + frames] + [("color", 0, 1), ("color", 0, 2), ("color", 0, 3), ("K", 0), ("inv_K", 0)]:
            d[name + "_in_" + "_".join(str(x) for x in k)] = inputs[k]
        d[name + "_stereo_T"] = T
        d[name + "_loss"] = losses["loss"]
        for f in (-1, 1):
            tag = "m1" if f < 0 else "p1"
            d["%s_T_%s" % (name, tag)] = Tleaf[f]
            d["%s_grad_T_%s" % (name, tag)] = Tleaf[f].grad
        for s_ in range(4):
            d[name + "_disp_%d" % s_] = disps[s_]
            d[name + "_grad_disp_%d" % s_] = dleaf[s_].grad
            d[name + "_loss_%d" % s_] = losses["loss/%d" % s_]
            if not cfg["disable_automasking"]:
                d[name + "_noise_%d" % s_] = noise[s_]
                d[name + "_identity_selection_%d" % s_] = out["identity_selection/%d" % s_]
        d[name + "_color_s_0"] = out[("color", "s", 0)]
        d[name + "_color_m1_2"] = out[("color", -1, 2)]
    save("loss_frames4", d)


def gen_geom():
    gen = torch.Generator().manual_seed(7)
    B, H, W = 3, 6, 10
    d = {}
    disp = torch.rand(B, 1, H, W, generator=gen)
    sd_, dep = ref_layers.disp_to_depth(disp, 0.1, 100)
    d.update(disp=disp, scaled_disp=sd_, depth=dep)
    aa = (0.3 * torch.randn(B, 1, 3, generator=gen)).requires_grad_(True)
    tr = (0.5 * torch.randn(B, 1, 3, generator=gen)).requires_grad_(True)
    wgt = torch.randn(B, 4, 4, generator=gen)
    for inv in (False, True):
        M = ref_layers.transformation_from_parameters(aa, tr, invert=inv)
        (M * wgt).sum().backward()
        tag = "inv" if inv else "fwd"
        d["M_" + tag] = M
        d["grad_aa_" + tag] = aa.grad.clone()
        d["grad_tr_" + tag] = tr.grad.clone()
        aa.grad = None
        tr.grad = None
    d.update(axisangle=aa, translation=tr, M_weight=wgt)
    # zero-angle edge case (axis = v/(|v|+1e-7))
    aa0 = torch.zeros(1, 1, 3)
    d["M_zero"] = ref_layers.transformation_from_parameters(aa0, torch.ones(1, 1, 3))
    K = torch.tensor([[12.0, 0, 5.1, 0], [0, 11.0, 2.9, 0], [0, 0, 1, 0], [0, 0, 0, 1]]).unsqueeze(0).repeat(B, 1, 1)
    invK = torch.from_numpy(np.stack([np.linalg.pinv(k) for k in K.numpy()])).float()
    bp = ref_layers.BackprojectDepth(B, H, W)
    pr = ref_layers.Project3D(B, H, W)
    pts = bp(dep, invK)
    T = ref_layers.transformation_from_parameters(aa.detach(), tr.detach() * 0.1)
    grid = pr(pts, K, T)
    d.update(K=K, inv_K=invK, cam_points=pts, T=T, grid=grid)
    save("geom", d)


def gen_ssim_smooth():
    gen = torch.Generator().manual_seed(11)
    x = torch.rand(2, 3, 9, 13, generator=gen).requires_grad_(True)
    y = torch.rand(2, 3, 9, 13, generator=gen)
    ssim = ref_layers.SSIM()
    v = ssim(x, y)
    w = torch.rand(v.shape, generator=gen)
    (v * w).sum().backward()
    d = dict(x=x, y=y, ssim=v, w=w, grad_x=x.grad)
    # near-identical images exercise the clamp-at-0 branch
    y2 = (x.detach() + 1e-4 * torch.randn(x.shape, generator=gen)).clamp(0, 1)
    d["y2"] = y2
    d["ssim2"] = ssim(x.detach(), y2)
    disp = torch.rand(2, 1, 9, 13, generator=gen).requires_grad_(True)
    img = torch.rand(2, 3, 9, 13, generator=gen)
    sm = ref_layers.get_smooth_loss(disp, img)
    sm.backward()
    d.update(sm_disp=disp, sm_img=img, smooth=sm, grad_sm_disp=disp.grad)
    save("ssim_smooth", d)


def gen_segmix():
    gen = torch.Generator().manual_seed(13)
    d = {}
    # cross entropy: ignore pixels, pixel weights, size mismatch
    B, C, H, W = 2, 19, 6, 8
    logits = torch.randn(B, C, H, W, generator=gen).requires_grad_(True)
    tgt = torch.randint(0, C, (B, H, W), generator=gen)
    tgt[0, 0, :3] = 250
    tgt[1, 2, 4] = 250
    loss = ref_ce(logits, tgt)
    loss.backward()
    d.update(ce_logits=logits, ce_target=tgt, ce_loss=loss, ce_grad=logits.grad.clone())
    logits.grad = None
    pw = torch.rand(B, H, W, generator=gen)
    loss = ref_ce(logits, tgt, pixel_weights=pw)
    loss.backward()
    d.update(ce_pw=pw, ce_loss_pw=loss, ce_grad_pw=logits.grad.clone())
    logits.grad = None
    tgt_big = torch.randint(0, C, (B, 2 * H, 2 * W), generator=gen)
    tgt_big[0, :2] = 250
    loss = ref_ce(logits, tgt_big)
    loss.backward()
    d.update(ce_target_big=tgt_big, ce_loss_big=loss, ce_grad_big=logits.grad.clone())
    allign = torch.full((B, H, W), 250, dtype=torch.long)
    d["ce_loss_allignored"] = ref_ce(logits.detach(), allign)
    # mix: full-batch, half-batch, target branch; int64 and float masks
    B, H, W = 4, 5, 7
    img = torch.rand(B, 3, H, W, generator=gen)
    soft = torch.softmax(torch.randn(B, 19, H, W, generator=gen), 1)
    m_f = (torch.rand(B, H, W, generator=gen) > 0.5).float()
    m_i = (torch.rand(B, H, W, generator=gen) > 0.5).long()
    m_half = (torch.rand(B // 2, H, W, generator=gen) > 0.5).long()
    d.update(mix_img=img, mix_soft=soft, mix_mask_f=m_f, mix_mask_i=m_i, mix_mask_half=m_half)
    d["mix_img_f"], _ = ref_tg.mix(m_f, data=img)
    d["mix_img_i"], _ = ref_tg.mix(m_i, data=img)
    d["mix_soft_i"], _ = ref_tg.mix(m_i, data=soft)
    d["mix_img_half"], _ = ref_tg.mix(m_half, data=img)
    lbl = torch.randint(0, 19, (B, H, W), generator=gen)
    _, d["mix_target_i"] = ref_tg.mix(m_i, target=lbl)
    d["mix_lbl"] = lbl
    # masks
    pred = torch.randint(0, 19, (H, W), generator=gen)
    classes = torch.tensor([3, 7, 11])
    d.update(cm_pred=pred, cm_classes=classes, cm_mask=ref_tm.generate_class_mask(pred, classes))
    depth = torch.rand(1, H, W, generator=gen)
    thr1 = torch.tensor([0.37])
    thr2 = torch.tensor([0.6, 0.2])
    d.update(dm_depth=depth, dm_thr1=thr1, dm_thr2=thr2, dm_mask1=ref_tm.generate_depth_mask(depth, thr1),
             dm_mask2=ref_tm.generate_depth_mask(depth, thr2))
    # depthcomp mask from Trainer.generate_mix_mask (train.py:585-604), B must be 2
    import train as ref_train
    fake = types.SimpleNamespace(cfg={"training": {"batch_size": 2}}, mix_mask="depthcomp", depthcomp_margin=0.03,
                                 depthcomp_foreground_threshold=0.0, device=torch.device("cpu"))
    depths = torch.rand(2, 1, 16, 24, generator=gen)
    depths[0, 0, :4] = depths[1, 0, :4] - 0.03  # exact-margin ties
    depths[1, 0, 4:8] = depths[0, 0, 4:8]
    d["dc_depths"] = depths
    d["dc_mask_m003_ft0"] = ref_train.Trainer.generate_mix_mask(fake, "depthcomp", None, None, depths)
    fake.depthcomp_foreground_threshold = 0.25
    d["dc_mask_m003_ft025"] = ref_train.Trainer.generate_mix_mask(fake, "depthcomp", None, None, depths)
    # pseudo label (train.py:644-648)
    mp, pl = torch.max(soft, dim=1)
    d["pl_label"] = pl
    d["pl_weight"] = torch.sum(mp.ge(0.968).long() == 1).item() / np.prod(pl.shape)
    # generate_cutout_mask (transformmasks.py:8-24; numpy's global generator re-seeded per call)
    for tag, size, seed in (("a", (24, 40), 3), ("b", (17, 33), 11), ("c", (64, 128), 2020)):
        d["cutout_%s" % tag] = ref_tm.generate_cutout_mask(size, seed=seed)
        d["cutout_%s_args" % tag] = np.array([size[0], size[1], seed])
    save("segmix", d)


def _sd_to_npz(prefix, sd, d):
    for k, v in sd.items():
        d[prefix + k] = v


def _grads(module):
    return {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in module.named_parameters()}


def gen_blocks():
    gen = torch.Generator().manual_seed(17)
    d = {}

    def rand_init(m):
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn(p.shape, generator=gen) * 0.2)
            for n_, b in m.named_buffers():
                if n_.endswith("running_mean"):
                    b.copy_(0.1 * torch.randn(b.shape, generator=gen))
                if n_.endswith("running_var"):
                    b.copy_(1 + 0.1 * torch.rand(b.shape, generator=gen))

    def run(tag, m, xs, fwd=None, seeded=None):
        if seeded is None:
            rand_init(m)
        else:
            # big modules: weights are regenerated in the test from this seed, in named_parameters order
            g2 = torch.Generator().manual_seed(seeded)
            with torch.no_grad():
                for p in m.parameters():
                    p.copy_(torch.randn(p.shape, generator=g2) * 0.05)
            d[tag + "_seed"] = seeded
        dropout_eval(m)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        xs = [x.clone().requires_grad_(True) for x in xs]
        y = m(*xs) if fwd is None else fwd(m, xs)
        ys = y if isinstance(y, (tuple, list)) else [y]
        tot = 0
        for i, yy in enumerate(ys):
            w = torch.randn(yy.shape, generator=gen)
            d["%s_w%d" % (tag, i)] = w
            d["%s_y%d" % (tag, i)] = yy
            tot = tot + (yy * w).sum()
        tot.backward()
        if seeded is None:
            _sd_to_npz(tag + "_sd_", sd0, d)
            if any(k.endswith("running_mean") for k in sd0):
                _sd_to_npz(tag + "_sdafter_", m.state_dict(), d)
        for i, x in enumerate(xs):
            d["%s_x%d" % (tag, i)] = x
            d["%s_gx%d" % (tag, i)] = x.grad
        for k, g in _grads(m).items():
            if g.numel() <= 20000:
                d["%s_g_%s" % (tag, k)] = g
            else:
                d["%s_gnorm_%s" % (tag, k)] = g.double().norm()
                d["%s_gslice_%s" % (tag, k)] = g.reshape(-1)[:4096]

    x = torch.randn(2, 8, 7, 9, generator=gen)
    run("convblock", ref_layers.ConvBlock(8, 12), [x])
    run("convblock_bn", ref_layers.ConvBlock(8, 12, bn=True), [x])
    run("conv3x3", ref_layers.Conv3x3(8, 1), [x])
    run("selfatt", RefSA(8, 8), [x])
    run("aspp", RefASPP(16, [1, 2, 3], True, 8), [torch.randn(2, 16, 6, 8, generator=gen)])
    run("posedec", RefPoseDecoder([4, 4, 8, 8, 16], num_input_features=1, num_frames_to_predict_for=2),
        [torch.randn(2, 16, 3, 5, generator=gen)], fwd=lambda m, xs: m([[xs[0]]]), seeded=4242)
    save("blocks", d)


def gen_decoders():
    gen = torch.Generator().manual_seed(19)
    d = {}
    enc = [8, 8, 16, 16, 32]
    B = 2
    # dilated-style pyramid: f4 has the same size as f3 (no upsample at i=4)
    shapes_dil = [(16, 24), (8, 12), (4, 6), (2, 3), (2, 3)]
    shapes_str = [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]

    def feats(shapes):
        return [torch.randn(B, c, h, w, generator=gen) for c, (h, w) in zip(enc, shapes)]

    def rand_init(m):
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn(p.shape, generator=gen) * 0.15)

    def record(tag, m, fs, out, keys):
        tot = 0
        for k in keys:
            name = "%s_out_%s" % (tag, "_".join(str(x) for x in k) if isinstance(k, tuple) else k)
            w = torch.randn(out[k].shape, generator=gen)
            d[name] = out[k]
            d[name + "_w"] = w
            tot = tot + (out[k] * w).sum()
        tot.backward()
        for i, f in enumerate(fs):
            d["%s_f%d" % (tag, i)] = f
            d["%s_gf%d" % (tag, i)] = f.grad if f.grad is not None else torch.zeros_like(f)
        for k, g in _grads(m).items():
            d["%s_g_%s" % (tag, k)] = g

    # 1) DepthDecoder with ASPP, dilated pyramid
    args1 = dict(intermediate_aspp=True, aspp_rates=[1, 2, 3], num_ch_dec=[8, 8, 8, 16, 16], max_scale_size=[32, 48])
    m = RefDepthDecoder(enc, range(4), **args1)
    rand_init(m)
    dropout_eval(m)
    _sd_to_npz("dd1_sd_", {k: v.clone() for k, v in m.state_dict().items()}, d)
    fs = [f.requires_grad_(True) for f in feats(shapes_dil)]
    out = m(fs)
    record("dd1", m, fs, out, [("disp", 0), ("disp", 1), ("disp", 2), ("disp", 3), ("upconv", 0), ("upconv", 3)])
    d["dd1_args_json"] = json.dumps(args1)
    # 2) DepthDecoder without ASPP, batch_norm, strided pyramid, split execution (exec_layer / x)
    args2 = dict(num_ch_dec=[8, 8, 8, 16, 16], batch_norm=True, max_scale_size=[64, 96])
    m = RefDepthDecoder(enc, range(4), **args2)
    rand_init(m)
    _sd_to_npz("dd2_sd_", {k: v.clone() for k, v in m.state_dict().items()}, d)
    fs = [f.requires_grad_(True) for f in feats(shapes_str)]
    o1 = m(fs, exec_layer=[4, 3, 2])
    o1 = dict(o1)
    o2 = m(fs, x=o1[("upconv", 2)] * 1.5, exec_layer=[1, 0])
    out = dict(o1)
    out.update(o2)
    record("dd2", m, fs, out, [("disp", 0), ("disp", 2), ("upconv", 2)])
    _sd_to_npz("dd2_sdafter_", m.state_dict(), d)
    d["dd2_args_json"] = json.dumps(args2)
    # 3) JointSegDepthDecoder, exp-210 style args (layers=[9], head_inter=False) and head_inter=True
    for tag, sa in (("jsd1", dict(layers=[9], head_inter=False, output_stride=1, layer_out_channels=8,
                                  head_inter_channels=8)),
                    ("jsd2", dict(layers=[9, 7], head_inter=True, output_stride=2, layer_out_channels=8,
                                  head_inter_channels=8))):
        m = RefJSD(enc, args1["num_ch_dec"], 5, weights="none", depth_args=dict(args1), **sa)
        rand_init(m)
        dropout_eval(m)
        _sd_to_npz(tag + "_sd_", {k: v.clone() for k, v in m.state_dict().items()}, d)
        fs = [f.requires_grad_(True) for f in feats(shapes_dil)]
        out = {"semantics": m(fs)}
        record(tag, m, fs, out, ["semantics"])
        _sd_to_npz(tag + "_sdafter_", m.state_dict(), d)
        d[tag + "_args_json"] = json.dumps(sa)
    # 4) PAD (exp-212 style: distillation_layer 7, final 9, side output)
    for tag, sa in (("pad1", dict(final_layer=9, output_stride=1, distillation_layer=7, side_output=True)),
                    ("pad2", dict(final_layer=9, output_stride=2, distillation_layer=6, side_output=True))):
        m = RefPAD(enc, args1["num_ch_dec"], 5, weights="none", depth_args=dict(args1), **sa)
        rand_init(m)
        dropout_eval(m)
        _sd_to_npz(tag + "_sd_", {k: v.clone() for k, v in m.state_dict().items()}, d)
        fs = [f.requires_grad_(True) for f in feats(shapes_dil)]
        out = m(fs)
        record(tag, m, fs, out, ["semantics", "intermediate_semantics", ("disp", 0), ("disp", 3)])
        d[tag + "_args_json"] = json.dumps(sa)
    save("decoders", d)


def sd_hash(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(np_(v)).tobytes())
    return h.hexdigest()


def model_cfgs():
    mono = dict(frame_ids=[0, -1, 1], num_scales=4, height=64, width=128)
    common = dict(arch="joint_segmentation_depth", pose_model_input="pairs", provide_uncropped_for_pose=False,
                  backbone_pretraining="none", depth_pretraining="none", pose_pretraining="none",
                  freeze_backbone=False, freeze_depth=False, freeze_pose=False, freeze_segmentation=False,
                  disable_monodepth=False, disable_pose=False, enable_imnet_encoder=False, **mono)
    dec = dict(intermediate_aspp=True, aspp_rates=[6, 12, 18], num_ch_dec=[64, 128, 128, 256, 256],
               max_scale_size=[64, 128])
    jsd_args = dict(weights="none", layers=[9], head_inter_channels=64, layer_out_channels=64, head_dropout=0.1,
                    layer_dropout=0, head_inter=False, output_stride=1)
    pad_args = dict(weights="none", output_stride=1, distillation_layer=7, side_output=True, final_layer=9)
    return {
        "r18_mono": dict(common, backbone_name="resnet18", replace_stride_with_dilation=None,
                         segmentation_name=None, segmentation_args=None, depth_args=dict(dec)),
        "r18_jsd": dict(common, backbone_name="resnet18", replace_stride_with_dilation=None,
                        segmentation_name="joint_seg_depth_dec", segmentation_args=dict(jsd_args),
                        depth_args=dict(dec)),
        "r50_mono": dict(common, backbone_name="resnet50", replace_stride_with_dilation=[False, False, True],
                         segmentation_name=None, segmentation_args=None, depth_args=dict(dec)),
        "r101_jsd": dict(common, backbone_name="resnet101", replace_stride_with_dilation=[False, False, True],
                         segmentation_name="joint_seg_depth_dec", segmentation_args=dict(jsd_args),
                         depth_args=dict(dec)),
        "r101_pad": dict(common, backbone_name="resnet101", replace_stride_with_dilation=[False, False, True],
                         segmentation_name="mtl_pad", segmentation_args=dict(pad_args), depth_args=dict(dec)),
    }


def gen_nets():
    """Full reference models: state_dict key/shape contract for every benchmark
    config + end-to-end outputs/loss/grad-norms for ResNet-18 models whose
    weights are regenerated from a seed by oracle.nets.build_state_dict."""
    cfgs = model_cfgs()
    contract = {}
    for name, cfg in cfgs.items():
        m = ref_get_model(cfg, 19)
        contract[name] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
        contract[name + "__trainable"] = [k for k, p in m.named_parameters() if p.requires_grad]
        print(name, len(contract[name]), "state_dict entries,",
              sum(p.numel() for p in m.parameters()) / 1e6, "M params")
        del m
    with open(os.path.join(OUT_DIR[0], "state_dict_contract.json"), "w") as f:
        json.dump({"cfgs": cfgs, "contract": contract}, f)
    d = {}
    for name in ("r18_mono", "r18_jsd"):
        cfg = cfgs[name]
        torch.manual_seed(0)
        m = ref_get_model(cfg, 19)
        sd = onets.build_state_dict(cfg, 19, seed=1234, randomize_bn=True)
        m.load_state_dict(sd, strict=True)  # also checks the key contract both ways
        dropout_eval(m)
        m.train()
        dropout_eval(m)
        gen = torch.Generator().manual_seed(5)
        B, H, W = 2, 64, 128
        inputs, _, _, _ = make_loss_inputs(B, H, W, gen)
        for f in (0, -1, 1):
            inputs[("color_aug", f, 0)] = inputs[("color", f, 0)]
        lbl = torch.randint(0, 19, (B, H, W), generator=gen)
        lbl[:, :3] = 250
        out = m(inputs)
        loss_obj = RefMonodepthLoss(num_scales=4, frame_ids=[0, -1, 1], height=H, width=W, batch_size=B,
                                    min_depth=0.1, max_depth=100, test_min_depth=1e-3, test_max_depth=80,
                                    disparity_smoothness=1e-3, no_ssim=False, avg_reprojection=False,
                                    disable_automasking=False)
        noise = [torch.randn(B, 2, H, W, generator=gen) for _ in range(4)]
        q = list(noise)
        real = torch.randn
        loss_obj.generate_images_pred(inputs, out)
        torch.randn = lambda *a, **k: q.pop(0)
        try:
            losses = loss_obj.compute_losses(inputs, out)
        finally:
            torch.randn = real
        total = losses["loss"]
        if "semantics" in out:
            seg = ref_ce(out["semantics"], lbl)
            total = total + seg
            d[name + "_seg_loss"] = seg
            d[name + "_semantics"] = out["semantics"]
        total.backward()
        d[name + "_sd_hash"] = sd_hash(sd)
        d[name + "_mono_loss"] = losses["loss"]
        for s in range(4):
            d[name + "_disp_%d" % s] = out[("disp", s)]
            d[name + "_noise_%d" % s] = noise[s]
        for f, tag in ((-1, "m1"), (1, "p1")):
            d[name + "_T_" + tag] = out[("cam_T_cam", 0, f)]
        d[name + "_lbl"] = lbl
        for k, v in inputs.items():
            if k[0] in ("color", "K", "inv_K"):
                d[name + "_in_" + "_".join(str(x) for x in k)] = v
        names, norms = [], []
        for k, p in m.named_parameters():
            names.append(k)
            norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
        d[name + "_grad_names"] = np.array(names)
        d[name + "_grad_norms"] = np.array(norms, dtype=np.float64)
        d[name + "_grad_conv1"] = m.models["encoder"].encoder.conv1.weight.grad
        d[name + "_bn1_running_mean_after"] = m.models["encoder"].encoder.bn1.running_mean
    save("nets", d)


def gen_encoder():
    """ResnetEncoder wiring (normalisation, feature taps, dilation flags) on tiny inputs."""
    d = {}
    gen = torch.Generator().manual_seed(23)
    for tag, nl, rswd, nimg in (("r18", 18, None, 1), ("r50dil", 50, [False, False, True], 1), ("r18x2", 18, None, 2)):
        kw = {} if nimg > 1 else {"replace_stride_with_dilation": rswd}
        m = RefResnetEncoder(nl, False, num_input_images=nimg, **kw)
        cfg = dict(backbone_name="resnet%d" % nl, replace_stride_with_dilation=rswd)
        sd = {}
        onets._resnet_sd(sd, "encoder.", nl, nimg, rswd, torch.Generator().manual_seed(77), True)
        m.encoder.fc = torch.nn.Identity()
        m.encoder.avgpool = torch.nn.Identity()
        m.load_state_dict(sd, strict=True)
        m.train()
        x = torch.rand(2, 3 * nimg, 64, 96, generator=gen)
        fs = m(x)
        d[tag + "_x"] = x
        d[tag + "_sd_hash"] = sd_hash(sd)
        for i, f in enumerate(fs):
            d["%s_f%d" % (tag, i)] = f if f.numel() < 40000 else f[:, :8]
            d["%s_f%d_shape" % (tag, i)] = np.array(f.shape)
    save("encoder", d)


def gen_trainer():
    """Trainer.update_ema_variables (train.py:346-358) and Trainer.calc_pseudo_label_loss (train.py:644-651), called
    unbound with a stand-in ``self`` that carries only the attributes the methods read."""
    import train as ref_train
    d = {}
    gen = torch.Generator().manual_seed(31)

    from trainer_fixture import Tiny      # parameters from an integer formula: reproducible without any RNG stream

    names = [n for n, _ in Tiny(0).named_parameters()]
    d["ema_param_names"] = np.array(names)
    branches = {"all": dict(save_monodepth_ema=False, segmentation_name="joint_seg_depth_dec", freeze_backbone=False),
                "pad": dict(save_monodepth_ema=False, segmentation_name="mtl_pad", freeze_backbone=False),
                "mono": dict(save_monodepth_ema=True, segmentation_name="mtl_pad", freeze_backbone=False),
                "mono_frozen": dict(save_monodepth_ema=True, segmentation_name=None, freeze_backbone=True)}
    d["ema_branches_json"] = json.dumps(branches)
    for tag, br in branches.items():
        fake = types.SimpleNamespace(cfg={"training": {"save_monodepth_ema": br["save_monodepth_ema"]},
                                          "model": {"segmentation_name": br["segmentation_name"],
                                                    "freeze_backbone": br["freeze_backbone"]}})
        fake.extract_monodepth_ema_params = types.MethodType(ref_train.Trainer.extract_monodepth_ema_params, fake)
        fake.extract_pad_ema_params = types.MethodType(ref_train.Trainer.extract_pad_ema_params, fake)
        for it in (0, 3, 5000):
            model, ema = Tiny(1), Tiny(2)
            ref_train.Trainer.update_ema_variables(fake, ema, model, 0.99, it)
            for n, p in ema.named_parameters():
                d["ema_%s_it%d_%s" % (tag, it, n)] = p.data if p.numel() < 2000 else p.data[::97]
    # pseudo labels
    B, C, H, W = 2, 19, 12, 16
    logits_t = 4.0 * torch.randn(B, C, H, W, generator=gen)
    logits_t[:, 3, :4] += 12.0                                   # confident region (softmax >= 0.968)
    soft = torch.softmax(logits_t, dim=1)
    soft[0, :, 5, 3:9] = 0.0                                     # pixels the mix left empty: max == 0 -> ignore_index
    soft[1, 2, 7, 7] = soft[1, 5, 7, 7] = soft[1].max()          # an exact tie: the first maximum wins
    student = torch.randn(B, C, H, W, generator=gen).requires_grad_(True)
    fake = types.SimpleNamespace(unlabeled_loader=types.SimpleNamespace(ignore_index=250), consistency_weight=1.5,
                                 device=torch.device("cpu"))
    L_u, label = ref_train.Trainer.calc_pseudo_label_loss(fake, soft.clone(), student)
    L_u.backward()
    d.update(pl_soft=soft, pl_student=student.detach(), pl_loss=L_u.detach(), pl_label=label, pl_grad=student.grad,
             pl_consistency_weight=np.float32(1.5))
    # validation metric: evaluation/metrics.py runningScore on seeded labels / logits (train.py:848-851)
    from evaluation.metrics import runningScore as RefScore
    n = 19
    logits = torch.randn(3, n, 10, 14, generator=gen)
    logits[:, 18] -= 100.0                                         # a class that is never predicted (nan IoU path)
    gt = torch.randint(0, n - 1, (3, 10, 14), generator=gen)
    gt[0, :2] = 250                                                # ignore label
    gt[1, 3, 3] = -1
    gt[(torch.rand(3, 10, 14, generator=gen) < 0.6)] = 0           # a dominant class
    pred = logits.max(1)[1]
    rs = RefScore(n)
    rs.update(gt.numpy(), pred.numpy())
    rs.update(gt[:1].numpy(), pred[:1].numpy())                    # a second batch accumulates
    sc, cls_iu = rs.get_scores()
    d.update(cm_logits=logits, cm_gt=gt, cm_pred=pred, cm_matrix=rs.confusion_matrix,
             cm_scores=np.array([sc["Overall Acc: \t"], sc["Mean Acc : \t"], sc["FreqW Acc : \t"], sc["Mean IoU : \t"]]),
             cm_cls_iu=np.array([cls_iu[i] for i in range(n)]))
    save("trainer", d)


def gen_usegt():
    """The reference's OWN ``Trainer.train_step_segmentation_unlabeled`` (train.py:653-724) with ``mix_use_gt`` on, called
    unbound with a stand-in ``self``: the teacher is a table of fixed logits, the student one 3x3 convolution on the mixed
    image (so that the pseudo-label loss has a gradient to record), depths come from ``pseudo_depth``
    (depthmix_online_depth off), mask "depthcomp".  Sample 0 is flagged labeled, sample 1 is not: the label planes of sample 1
    are garbage that must never be read.  Stored: inputs, the mixed image the student saw, the mixed teacher distribution
    and pseudo labels ``calc_pseudo_label_loss`` received / produced, L_2 and the student's gradients."""
    import train as ref_train
    gen = torch.Generator().manual_seed(77)
    B, C, H, W = 2, 19, 12, 20
    teacher_logits = 3.0 * torch.randn(B, C, H, W, generator=gen)
    img = torch.rand(B, 3, H, W, generator=gen)
    depth = torch.rand(B, 1, H, W, generator=gen)
    lbl = torch.randint(0, C, (B, H, W), generator=gen)
    lbl[0, :2, :5] = 250                                         # ignored pixels of the labeled sample: all-zero planes
    dense = lbl.clone()
    dense[dense == 250] = C
    onehot = F.one_hot(dense, C + 2)[..., :C].permute(0, 3, 1, 2).contiguous()     # the loader's recipe (:237-242), int64
    onehot[1] = torch.randint(0, 2, (C, H, W), generator=gen)    # unlabeled sample: never read
    is_labeled = torch.tensor([True, False])
    student = torch.nn.Conv2d(3, C, 3, padding=1)
    with torch.no_grad():
        student.weight.copy_(0.5 * torch.randn(student.weight.shape, generator=gen))
        student.bias.copy_(0.1 * torch.randn(C, generator=gen))
    seen = {}

    class Teacher:
        use_pose_net = True

        def __call__(self, inputs):
            return {"semantics": teacher_logits.clone()}

    def student_model(inputs):
        seen["mixed_img"] = inputs[("color_aug", 0, 0)].clone()
        return {"semantics": student(inputs[("color_aug", 0, 0)])}

    fake = types.SimpleNamespace(
        ema_model=Teacher(), model=student_model, mix_use_gt=True, depthmix_online_depth=False, mix_mask="depthcomp",
        depthcomp_margin=0.03, depthcomp_foreground_threshold=0.1, unlabeled_color_jitter=False, unlabeled_blur=False,
        unlabeled_backward_first_pseudo_label=False, consistency_weight=1.0, device=torch.device("cpu"),
        unlabeled_loader=types.SimpleNamespace(ignore_index=250), scaler=types.SimpleNamespace(scale=lambda x: x),
        cfg={"training": {"batch_size": B, "monodepth_lambda": 1.0, "print_interval": 1000, "log_path": "/tmp"}})
    fake.generate_mix_mask = types.MethodType(ref_train.Trainer.generate_mix_mask, fake)

    def calc(teacher_softmax, student_logits):
        seen["soft_mixed"] = teacher_softmax.clone()
        L, lab = ref_train.Trainer.calc_pseudo_label_loss(fake, teacher_softmax, student_logits)
        seen["pseudo_label"] = lab.clone()
        return L, lab
    fake.calc_pseudo_label_loss = calc
    inputs = {("color_aug", 0, 0): img.clone(), "pseudo_depth": depth, "onehot_lbl": onehot, "is_labeled": is_labeled,
              "filename": ["a", "b"]}
    total, mono = ref_train.Trainer.train_step_segmentation_unlabeled(fake, inputs, 0)
    assert mono == 0 and fake.ema_model.use_pose_net is False
    d = dict(teacher_logits=teacher_logits, img=img, pseudo_depth=depth, onehot_lbl=onehot, is_labeled=is_labeled,
             student_weight=student.weight.detach(), student_bias=student.bias.detach(), mixed_img=seen["mixed_img"],
             soft_mixed=seen["soft_mixed"], pseudo_label=seen["pseudo_label"], L_2=total.detach(),
             grad_weight=student.weight.grad, grad_bias=student.bias.grad, margin=np.float32(0.03), ft=np.float32(0.1))
    save("usegt", d)


def gen_poseall():
    """pose_model_input = "all" (models/joint_segmentation_depth.py:52-68): the three frames through ONE 9-channel pose
    network, both poses predicted together -- reference outputs for the oracle / product to match."""
    cfg = dict(model_cfgs()["r18_mono"], pose_model_input="all")
    torch.manual_seed(0)
    m = ref_get_model(cfg, 19)
    sd = onets.build_state_dict(cfg, 19, seed=55, randomize_bn=True)
    m.load_state_dict(sd, strict=True)
    m.train()
    gen = torch.Generator().manual_seed(8)
    B, H, W = 2, 64, 128
    inputs, _, _, _ = make_loss_inputs(B, H, W, gen)
    for f in (0, -1, 1):
        inputs[("color_aug", f, 0)] = inputs[("color", f, 0)]
    out = m(inputs)
    loss = sum((out[("cam_T_cam", 0, f)] ** 2).sum() for f in (-1, 1)) + out[("axisangle", 0, 1)].sum()
    loss.backward()
    d = {"sd_hash": sd_hash(sd), "cfg_json": json.dumps(cfg)}
    for f in (0, -1, 1):
        d["in_color_%d" % f] = inputs[("color", f, 0)]
    for f, tag in ((-1, "m1"), (1, "p1")):
        d["T_" + tag] = out[("cam_T_cam", 0, f)]
    d["axisangle"], d["translation"] = out[("axisangle", 0, 1)], out[("translation", 0, 1)]
    d["grad_pose_conv1"] = m.models["pose_encoder"].encoder.conv1.weight.grad
    d["grad_pose_last"] = m.models["pose"].net[3].weight.grad
    d["pose_state_keys"] = np.array([k for k in m.state_dict() if k.startswith("models.pose")][:4])
    save("poseall", d)


def gen_valtail():
    """Validation tail: MonodepthLoss.generate_depth_test_pred (loss/monodepth_loss.py:54-62),
    JointSegmentationMonodepth.predict_test_disp in eval mode (models/joint_segmentation_depth.py:72-75, called by
    loader/depth_estimator.py:80-81) and the min-max-normalised 8-bit disparity DepthEstimator.prepare_depth_estimates
    stores (loader/depth_estimator.py:83-91: clamp, normalise, ToPILImage = mul(255).byte())."""
    cfg = model_cfgs()["r18_mono"]
    torch.manual_seed(0)
    m = ref_get_model(cfg, 19)
    sd = onets.build_state_dict(cfg, 19, seed=77, randomize_bn=True)
    m.load_state_dict(sd, strict=True)
    m.eval()
    gen = torch.Generator().manual_seed(6)
    B, H, W = 2, 64, 128
    inputs, _, _, _ = make_loss_inputs(B, H, W, gen)
    loss_obj = RefMonodepthLoss(num_scales=4, frame_ids=[0, -1, 1], height=H, width=W, batch_size=B, min_depth=0.1,
                                max_depth=100, test_min_depth=1e-3, test_max_depth=80, disparity_smoothness=1e-3,
                                no_ssim=False, avg_reprojection=False, disable_automasking=False, is_train=False)
    with torch.no_grad():
        out = m.predict_test_disp(inputs)
        loss_obj.generate_depth_test_pred(out)
    d = {"sd_hash": sd_hash(sd), "in_color_0_0": inputs[("color", 0, 0)]}
    for s_ in range(4):
        d["disp_%d" % s_] = out[("disp", s_)]
        d["depth_%d" % s_] = out[("depth", 0, s_)]
    imgs = []
    for depth in out[("disp", 0)].cpu():
        dmin, dmax = torch.min(depth), torch.max(depth)
        depth = torch.clamp(depth, dmin, dmax)
        depth = (depth - dmin) / (dmax - dmin)
        imgs.append(depth.squeeze(0).mul(255).byte())          # torchvision ToPILImage on a float tensor: mul(255).byte()
    d["export_u8"] = torch.stack(imgs)
    # generate_depth_test_pred alone on seeded disparities (no network in front)
    disps = {s_: (0.02 + 0.96 * torch.rand(B, 1, H // 2 ** s_, W // 2 ** s_, generator=gen)) for s_ in range(4)}
    o2 = {("disp", s_): disps[s_] for s_ in range(4)}
    loss_obj.generate_depth_test_pred(o2)
    for s_ in range(4):
        d["rnd_disp_%d" % s_] = disps[s_]
        d["rnd_depth_%d" % s_] = o2[("depth", 0, s_)]
    save("valtail", d)


ALL = ["loss", "loss_stereo", "loss_frames4", "geom", "ssim_smooth", "segmix", "blocks", "decoders", "encoder", "nets", "trainer", "usegt", "valtail", "poseall"]


def check(which):
    """Regenerate ``which`` into a scratch directory and compare every array with the committed file of the same name, bit for
    bit (dtype, shape, bytes).  Returns the number of mismatching / missing arrays; prints one line per file."""
    import tempfile
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        OUT_DIR[0] = tmp
        try:
            for w in which:
                globals()["gen_" + w]()
        finally:
            OUT_DIR[0] = HERE
        for fn in sorted(os.listdir(tmp)):
            ref_path = os.path.join(HERE, fn)
            if fn.endswith(".json"):
                same = os.path.exists(ref_path) and json.load(open(ref_path)) == json.load(open(os.path.join(tmp, fn)))
                print("CHECK %-32s %s" % (fn, "identical" if same else "DIFFERS"))
                bad += 0 if same else 1
                continue
            new = np.load(os.path.join(tmp, fn), allow_pickle=False)
            if not os.path.exists(ref_path):
                print("CHECK %-32s no committed file" % fn)
                bad += len(new.files)
                continue
            old = np.load(ref_path, allow_pickle=False)
            diff = [k for k in sorted(set(new.files) | set(old.files))
                    if k not in new.files or k not in old.files or new[k].dtype != old[k].dtype
                    or new[k].shape != old[k].shape or new[k].tobytes() != old[k].tobytes()]
            print("CHECK %-32s %4d arrays, %d differ%s" % (fn, len(new.files), len(diff), (": " + ", ".join(diff[:6])) if diff else ""))
            bad += len(diff)
    return bad


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--check":
        n = check(args[1:] or ALL)
        print("make_golden --check:", "OK" if n == 0 else "%d arrays differ from the committed fixtures" % n)
        sys.exit(1 if n else 0)
    for w in args or ALL:
        globals()["gen_" + w]()
