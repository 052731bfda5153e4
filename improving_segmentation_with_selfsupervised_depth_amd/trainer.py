"""Trainer-side callers of the hot path, same names and argument meaning as the methods of
the reference's ``Trainer`` (train.py), as free functions taking the config values the methods read from ``self``:

* ``extract_ema_params`` (train.py:124-135), ``update_ema_variables`` (train.py:346-358): the EMA teacher update as ONE
  multi-tensor launch over all parameter tensors instead of a Python loop of ~880 x 3 elementwise kernels;
* ``calc_pseudo_label_loss`` (train.py:644-651): max / argmax / empty-pixel masking / confidence count in one pass over
  the mixed teacher softmax, the confidence weight kept on the device (the reference's ``.item()`` round trip is gone),
  then the package's cross_entropy2d.
"""
import numpy as np
import torch

from . import functional as Fn
from . import hipops as H
from .loss.loss import cross_entropy2d

__all__ = ["extract_ema_params", "EmaUpdater", "update_ema_variables", "calc_pseudo_label_loss", "teacher_softmax",
           "normalize_online_depth", "generate_mix_mask", "train_step_segmentation_unlabeled", "create_ema_model", "extract_param_dict",
           "get_params", "get_train_params", "train_step"]


def extract_ema_params(model, ema_model, model_names):
    """train.py:124-135"""
    relevant_params, relevant_ema_params = [], []
    for k, v in model.models.items():
        if k in model_names:
            relevant_params.extend(v.parameters())
    for k, v in ema_model.models.items():
        if k in model_names:
            relevant_ema_params.extend(v.parameters())
    return relevant_params, relevant_ema_params


def _select(model, ema_model, save_monodepth_ema, segmentation_name, freeze_backbone):
    """parameter lists exactly as Trainer.update_ema_variables picks them (train.py:347-352, 317-326)"""
    if save_monodepth_ema:
        names = ["depth"] + ([] if freeze_backbone else ["encoder"])
        return extract_ema_params(model, ema_model, names)
    if segmentation_name == "mtl_pad":
        return extract_ema_params(model, ema_model, ["depth", "encoder", "mtl_decoder"])
    return list(model.parameters()), list(ema_model.parameters())


class EmaUpdater:
    """Caches the device-side chunk table for a (model, ema_model) pair; ``step`` is one kernel launch."""

    def __init__(self, model_params, ema_params):
        model_params, ema_params = list(model_params), list(ema_params)
        assert len(model_params) == len(ema_params), f"len(mp)={len(model_params)}; len(mcp)={len(ema_params)}"
        self._key = tuple((e.data_ptr(), p.data_ptr(), p.numel()) for e, p in zip(ema_params, model_params))
        self._like = ema_params[0] if ema_params else None
        self._ema = list(ema_params)
        self._table = H.multi_tensor_table([e.data for e in ema_params], [p.data for p in model_params])

    def matches(self, model_params, ema_params):
        return self._key == tuple((e.data_ptr(), p.data_ptr(), p.numel()) for e, p in zip(ema_params, model_params))

    def step(self, alpha_teacher, iteration):
        # "Use the true average until the exponential average is more correct" (train.py:353-354); the two scalars are
        # rounded to fp32 the way torch rounds a Python double that multiplies a float32 tensor
        alpha = min(1 - 1 / (iteration + 1), alpha_teacher)
        if self._table is not None:
            H.multi_tensor_lerp(self._table, np.float32(alpha), np.float32(1 - alpha), self._like)
            # the kernel writes through raw pointers: tell autograd's version counters, which the convolutions' weight-pack
            # caches (models/layers.py) and any saved-tensor check rely on
            with torch.no_grad():
                for e in self._ema:
                    torch.autograd.graph.increment_version(e)


_UPDATERS = {}


def update_ema_variables(ema_model, model, alpha_teacher, iteration, save_monodepth_ema=False, segmentation_name=None,
                         freeze_backbone=False):
    """Trainer.update_ema_variables (train.py:346-358); the three keyword arguments are the cfg entries the method reads
    (cfg["training"]["save_monodepth_ema"], cfg["model"]["segmentation_name"], cfg["model"]["freeze_backbone"])."""
    mp, ep = _select(model, ema_model, save_monodepth_ema, segmentation_name, freeze_backbone)
    mp, ep = list(mp), list(ep)
    key = (id(model), id(ema_model))
    up = _UPDATERS.get(key)
    if up is None or not up.matches(mp, ep):
        up = _UPDATERS[key] = EmaUpdater(mp, ep)
    up.step(alpha_teacher, iteration)
    return ema_model


def calc_pseudo_label_loss(teacher_softmax, student_logits, consistency_weight, ignore_index=250, threshold=0.968):
    """Trainer.calc_pseudo_label_loss (train.py:644-651) -> (L_u, pseudo_label).  ``teacher_softmax``: [B,C,H,W]."""
    pseudo_label, _, _, pixel_weight = H.pseudo_label(teacher_softmax.detach(), threshold, ignore_index)
    L_u = consistency_weight * cross_entropy2d(input=student_logits, target=pseudo_label, pixel_weights=pixel_weight)
    return L_u, pseudo_label


def teacher_softmax(logits):
    """train.py:666 ``torch.softmax(logits_u_w.detach(), dim=1)``: [B,C,H,W] logits (NCHW-logical; the segmentation head
    hands them over channels-last) -> dense NCHW probabilities, one HIP pass."""
    from . import functional as Fn
    return H.softmax_to_nchw(Fn.to_nhwc(logits.detach()))


def normalize_online_depth(disp):
    """train.py:690-697: the student's ("disp", 0), per sample min-max normalised to [0, 1] (detached)."""
    return H.minmax_normalize(disp.detach())


def generate_mix_mask(mode, argmax_u_w, unlabeled_imgs, depths, depthcomp_margin=0.03, depthcomp_foreground_threshold=0.0):
    """Trainer.generate_mix_mask (train.py:572-642): "class" / "depthcomp" / "depth" / None on the HIP mask kernels
    ("depthhist" is CPU numpy histogram code in the reference and is not part of the path)."""
    from .loader import transformmasks
    B = unlabeled_imgs.shape[0]
    dev = unlabeled_imgs.device
    if mode == "class":
        masks = []
        for i in range(B):
            classes = torch.unique(argmax_u_w[i])
            classes = classes[classes != 250]
            n = classes.shape[0]
            pick = torch.as_tensor(np.random.choice(n, int((n - n % 2) / 2), replace=False), dtype=torch.int64, device=dev)
            masks.append(transformmasks.generate_class_mask(argmax_u_w[i], classes[pick]).unsqueeze(0))
        return torch.cat(masks)
    if mode == "depthcomp":
        ft = depthcomp_foreground_threshold
        if isinstance(ft, (tuple, list)):
            # train.py:592-599: an independent threshold per image, drawn on the device RNG; it stays on the device
            ft_l, ft_u = ft
            assert ft_u > ft_l
            ft = torch.rand(B, device=dev) * (ft_u - ft_l) + ft_l
        return transformmasks.generate_depthcomp_mask(depths, depthcomp_margin, ft)
    if mode == "depth":
        masks = []
        for i in range(B):
            thr = torch.rand(1) * (0.4 - 0.1) + 0.1
            masks.append(transformmasks.generate_depth_mask(depths[i], thr))
        return torch.cat(masks)
    if mode is None:
        return torch.ones((B,) + tuple(unlabeled_imgs.shape[2:]), device=dev)
    raise NotImplementedError(f"Unknown mix_mask {mode}")


def train_step_segmentation_unlabeled(model, ema_model, monodepth_loss_calculator, unlabeled_inputs, mix_mask="depthcomp",
                                      depthmix_online_depth=True, monodepth_lambda=1.0, consistency_weight=1.0,
                                      backward_first_pseudo_label=False, depthcomp_margin=0.03,
                                      depthcomp_foreground_threshold=0.0, color_jitter=False, blur=False, reducer=None,
                                      last_backward=True, mix_use_gt=False, scaler=None):
    """Trainer.train_step_segmentation_unlabeled (train.py:653-724) with the ``self.*`` values as arguments: teacher
    forward -> softmax; student forward on the unmixed frames -> monodepth loss backward and the online depth; mix mask;
    DepthMix of images and teacher softmax; student forward on the mixed frames; pseudo-label loss backward.
    Returns (L_2 + L_1, mono_loss) like the reference.  ``reducer`` (ddp.GradAllReducer) keeps the gradient all-reduce
    out of every backward() except the last one of the step (``last_backward``: the L_2 backward is that one).
    ``mix_use_gt`` (train.py:667-672, on in the exp-212 block, experiments.py:343-357): samples of the unlabeled batch with
    ``unlabeled_inputs["is_labeled"][i]`` set take ``unlabeled_inputs["onehot_lbl"][i]`` instead of the teacher's softmax."""
    import contextlib
    import random
    from .loader import transformsgpu
    from .models.layers import weight_pack_scope

    @contextlib.contextmanager
    def nosync():
        """a backward() that is not the step's last: no gradient all-reduce, and its gradients are folded into the ones
        already accumulated with one multi-tensor add instead of one small add kernel per parameter (the reference's
        three backward passes per step, train.py:486-514 + 676-702, otherwise cost ~500 launches each)"""
        params = [p for p in model.parameters() if p.grad is not None]
        held = [p.grad for p in params]
        for p in params:
            p.grad = None
        try:
            with (reducer.no_sync() if reducer is not None else contextlib.nullcontext()):
                yield
        finally:
            new, old = [], []
            for p, g in zip(params, held):
                if p.grad is None:
                    p.grad = g
                else:
                    new.append(p.grad)
                    old.append(g)
            if new:
                torch._foreach_add_(new, old)

    def _scaled(loss):           # train.py:687, 696, 724: every loss of the unlabeled step goes through the GradScaler too
        return scaler.scale(loss) if scaler is not None else loss

    def strong_transform(parameters, data=None, target=None):
        data, target = transformsgpu.mix(mask=parameters["Mix"], data=data, target=target)
        data, target = transformsgpu.color_jitter(jitter=parameters["ColorJitter"], data=data, target=target)
        data, target = transformsgpu.gaussian_blur(blur=parameters["GaussianBlur"], data=data, target=None)
        return data, target

    unlabeled_imgs = unlabeled_inputs[("color_aug", 0, 0)]
    # first step: teacher -> pseudo-label distribution (train.py:663-672)
    ema_model.use_pose_net = False
    with torch.no_grad():
        logits_u_w = ema_model(unlabeled_inputs)["semantics"]
    softmax_u_w = teacher_softmax(logits_u_w)
    if mix_use_gt:
        if "is_labeled" not in unlabeled_inputs or "onehot_lbl" not in unlabeled_inputs:
            raise KeyError('mix_use_gt needs unlabeled_inputs["is_labeled"] and ["onehot_lbl"] (loader load_onehot, train.py:222)')
        H.onehot_select_(softmax_u_w, unlabeled_inputs["onehot_lbl"], unlabeled_inputs["is_labeled"])
    argmax_u_w = None
    if isinstance(mix_mask, str) and mix_mask == "class":
        argmax_u_w = H.pseudo_label(softmax_u_w, 2.0, -1, want_weight=False)[0]
    # second step: student on the unaugmented frames -> online depth + monodepth loss (train.py:676-702)
    mono_loss, L_1 = 0, 0
    with weight_pack_scope():          # both student passes run on the same weights: their convolutions pack once
        if depthmix_online_depth:
            outputs_1 = model(unlabeled_inputs)
            if monodepth_lambda > 0:
                monodepth_loss_calculator.generate_images_pred(unlabeled_inputs, outputs_1)
                mono_losses = monodepth_loss_calculator.compute_losses(unlabeled_inputs, outputs_1)
                mono_loss = monodepth_lambda * mono_losses["loss"]
                with nosync():
                    _scaled(mono_loss).backward(retain_graph=backward_first_pseudo_label)
                depths = normalize_online_depth(outputs_1[("disp", 0)])
            else:
                depths = unlabeled_inputs["pseudo_depth"]
            if backward_first_pseudo_label:
                L_1, _ = calc_pseudo_label_loss(softmax_u_w, outputs_1["semantics"], consistency_weight)
                with nosync():
                    _scaled(L_1).backward()
            del outputs_1
        elif "pseudo_depth" in unlabeled_inputs:
            depths = unlabeled_inputs["pseudo_depth"]
        else:
            depths = [None] * unlabeled_imgs.shape[0]
        # third step: mix (train.py:704-724)
        if torch.is_tensor(mix_mask):
            MixMask = mix_mask                    # a precomputed mask (tests; pre-generated masks of a data pipeline)
        else:
            MixMask = generate_mix_mask(mix_mask, argmax_u_w, unlabeled_imgs, depths, depthcomp_margin,
                                        depthcomp_foreground_threshold)
        strong_parameters = {"Mix": MixMask, "ColorJitter": random.uniform(0, 1) if color_jitter else 0,
                             "GaussianBlur": random.uniform(0, 1) if blur else 0}
        inputs_u_s, _ = strong_transform(strong_parameters, data=unlabeled_imgs)
        mixed_inputs = dict(unlabeled_inputs)
        mixed_inputs[("color_aug", 0, 0)] = inputs_u_s
        outputs = model(mixed_inputs)
    softmax_u_w_mixed, _ = strong_transform(strong_parameters, data=softmax_u_w)
    L_2, pseudo_label = calc_pseudo_label_loss(softmax_u_w_mixed, outputs["semantics"], consistency_weight)
    if last_backward:
        if reducer is not None:
            reducer.complete_unreachable([L_2])
        _scaled(L_2).backward()
    else:
        with nosync():
            _scaled(L_2).backward()
    train_step_segmentation_unlabeled.last = {"MixMask": MixMask, "depths": depths, "pseudo_label": pseudo_label,
                                              "softmax_u_w": softmax_u_w, "inputs_u_s": inputs_u_s}   # debug images :726-744
    return L_2 + L_1, mono_loss


# ----------------------------------------------------------------------------------------------
# the labeled step around it (train.py:39-101, 442-549)
# ----------------------------------------------------------------------------------------------
def extract_param_dict(model):
    """train.py:42-53: sub-model name -> parameters; a PAD decoder stands for both "segmentation" and "depth" """
    from .models.joint_segmentation_depth_decoder import PAD
    out, pad = {}, False
    for name, sub in model.models.items():
        if isinstance(sub, PAD):
            out["segmentation"], out["depth"], pad = sub.segmentation_params(), sub.depth_params(), True
        elif not (pad and name in ("depth", "segmentation")):
            out[name] = sub.parameters()
    return out


def create_ema_model(model, cfg, n_classes):
    """``Trainer.create_ema_model`` (train.py:328-344): a second model without pose networks whose (selected) parameters
    start as detached copies of the student's"""
    from copy import deepcopy
    from .models import get_model
    ema_cfg = deepcopy(cfg["model"])
    ema_cfg["disable_pose"] = True
    ema_model = get_model(ema_cfg, n_classes)
    mp, mcp = _select(model, ema_model, cfg["training"]["save_monodepth_ema"], cfg["model"]["segmentation_name"],
                      cfg["model"]["freeze_backbone"])
    mp, mcp = list(mp), list(mcp)
    assert len(mp) == len(mcp), f"len(mp)={len(mp)}; len(mcp)={len(mcp)}"
    with torch.no_grad():
        for src, dst in zip(mp, mcp):
            dst.detach_()
            dst.data = src.detach().to(dst.device).clone()
    return ema_model


def get_params(model, submodules):
    """train.py:56-64"""
    table = extract_param_dict(model)
    for sm in submodules:
        assert sm in table, f"{sm} not in {table.keys()}"
    return [p for name, ps in table.items() if name in submodules for p in ps]


def get_train_params(model, cfg):
    """train.py:67-101: optimizer parameter groups from ``backbone_lr`` / ``pose_lr`` / ``depth_lr`` / ``segmentation_lr``"""
    opt, rest, groups = cfg["training"]["optimizer"], extract_param_dict(model), []
    if "backbone_lr" in opt:
        groups.append({"params": rest.pop("encoder"), "lr": opt["backbone_lr"]})
    if "pose_lr" in opt and "pose_encoder" in model.models:
        groups.append({"params": [*rest.pop("pose_encoder"), *rest.pop("pose")], "lr": opt["pose_lr"]})
    for key, name in (("depth_lr", "depth"), ("segmentation_lr", "segmentation")):
        if key in opt:
            groups.append({"params": rest.pop(name), "lr": opt[key]})
    if not groups:
        return model.parameters()
    groups.append({"params": [p for ps in rest.values() for p in ps]})
    return groups


def train_step(model, optimizer, inputs, step, cfg, loss_fn, monodepth_loss_calculator, ema_model=None, scheduler=None,
               unlabeled_inputs=None, reducer=None, mIoU=0, scaler=None):
    """``Trainer.train_step`` (train.py:442-549) as a free function over the objects the method reads from ``self``
    (``cfg`` is the same nested dict; ``unlabeled_inputs`` the batch the method draws from its unlabeled loader when
    ``cfg["training"]["unlabeled_segmentation"]`` is set).  Same order of forward / ``backward()`` calls, gradient clipping,
    optimizer / scheduler step and EMA update; returns the same dict of detached losses.  ``amp: True`` (the dec6 configs,
    train.py:300,468-528) keeps the reference's protocol -- forward and segmentation loss under ``autocast``, every loss through
    ``scaler.scale(...)``, ``unscale_`` before the clipping, ``scaler.step`` / ``update`` -- while the kernels compute in fp32
    (functional.fp32_region): at least the reference's precision, no reduced-precision speed-up.  ``scaler``: a
    ``torch.amp.GradScaler`` kept by the caller across steps (one is created and kept on the optimizer object otherwise).
    ``reducer`` (ddp.GradAllReducer): every backward but the step's last runs under ``no_sync()`` and ``finish()`` is called
    before the clipping."""
    import contextlib
    from .loss.loss import berhu
    tr = cfg["training"]
    amp = bool(tr.get("amp", False))
    if scaler is None:
        scaler = getattr(optimizer, "_segsde_scaler", None)
        if scaler is None or scaler.is_enabled() != amp:
            scaler = torch.amp.GradScaler("cuda", enabled=amp)          # train.py:300
            optimizer._segsde_scaler = scaler
    autocast = lambda: torch.autocast(device_type="cuda", enabled=amp)   # noqa: E731
    unl = tr.get("unlabeled_segmentation", None)
    dev = next(model.parameters()).device
    model.train()
    if ema_model is not None:
        ema_model.train()
    for k, v in inputs.items():
        if torch.is_tensor(v):
            inputs[k] = v.to(dev, non_blocking=True)
    if unl is not None:
        for k in unlabeled_inputs.keys():
            if torch.is_tensor(unlabeled_inputs[k]):
                unlabeled_inputs[k] = unlabeled_inputs[k].to(dev, non_blocking=True)
    optimizer.zero_grad()
    zero = torch.tensor(0)
    segmentation_loss = segmentation_total = mono_loss = feat_dist_loss = mono_total = pseudo_depth_loss = zero
    if cfg["model"].get("freeze_backbone_bn", False):
        for m in model.models["encoder"].modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.train(False)
    do_mono, do_seg = tr["monodepth_lambda"] > 0, tr["segmentation_lambda"] > 0
    do_pd = tr.get("pseudo_depth_lambda", 0) > 0

    def hold(last):           # all but the last backward() of the step keep the gradient all-reduce back
        return reducer.no_sync() if (reducer is not None and not last) else contextlib.nullcontext()

    # One encoder backward per forward, whatever the number of backward() calls on its losses (functional.defer_trunk).  The
    # contract is that the forward's LAST backward() releases its graph: true whenever the segmentation loss is on (train.py:510 is
    # a plain backward(); the unlabeled step's forwards end on plain calls as well, train.py:687,696 + the mixed pass) and false
    # for a monodepth-only step, which ends on retain_graph=True (train.py:486) -- there the gate stays off.
    # ``cfg["training"]["defer_trunk_backward"] = False`` switches it off for any configuration.
    if hasattr(model, "defer_trunk_backward"):
        model.defer_trunk_backward = bool(tr.get("defer_trunk_backward", True)) and do_seg and (do_mono or do_pd or unl is not None)
    with autocast():
        outputs = model(inputs)
    if do_mono:
        if amp:
            for k, v in list(outputs.items()):       # train.py:472-474
                if ("depth" in k or "cam_T_cam" in k) and torch.is_tensor(v) and v.dtype != torch.float32:
                    outputs[k] = v.to(torch.float32)
        monodepth_loss_calculator.generate_images_pred(inputs, outputs)
        mono_loss = tr["monodepth_lambda"] * monodepth_loss_calculator.compute_losses(inputs, outputs)["loss"]
        if tr["feat_dist_lambda"] > 0:
            feat_dist_loss = tr["feat_dist_lambda"] * torch.dist(outputs["encoder_features"], outputs["imnet_features"], p=2)
        mono_total = mono_loss + feat_dist_loss
        with hold(not (do_pd or do_seg)):
            scaler.scale(mono_total).backward(retain_graph=True)
    if do_pd:
        with torch.no_grad():      # the bottom tenth of the frame (the ego vehicle) does not count, train.py:491-494
            keep = torch.ones(outputs["disp", 0].shape, device=dev)
            keep[:, :, int(outputs["disp", 0].shape[2] * 0.9):, :] = 0
        pseudo_depth_loss = berhu(outputs["disp", 0], inputs["pseudo_depth"], keep) * tr["pseudo_depth_lambda"]
        with hold(not do_seg):
            scaler.scale(pseudo_depth_loss).backward(retain_graph=True)
    if do_seg:
        with autocast():
            segmentation_loss = loss_fn(input=outputs["semantics"], target=inputs["lbl"])
            if "intermediate_semantics" in outputs:
                segmentation_loss = (segmentation_loss + loss_fn(input=outputs["intermediate_semantics"], target=inputs["lbl"])) / 2
            segmentation_loss = segmentation_loss * tr["segmentation_lambda"]
            segmentation_total = segmentation_loss
        if reducer is not None and unl is None and (do_mono or do_pd):
            reducer.complete_unreachable([segmentation_total])     # the depth / pose gradients are final: reduce them under this backward
        with hold(unl is None):
            scaler.scale(segmentation_total).backward()
        if unl is not None:
            u_loss, u_mono = train_step_segmentation_unlabeled(
                model, ema_model, monodepth_loss_calculator, unlabeled_inputs, mix_mask=unl.get("mix_mask", None),
                depthmix_online_depth=unl.get("depthmix_online_depth", False), monodepth_lambda=tr["monodepth_lambda"],
                consistency_weight=unl["consistency_weight"], backward_first_pseudo_label=unl["backward_first_pseudo_label"],
                depthcomp_margin=unl["depthcomp_margin"], depthcomp_foreground_threshold=unl["depthcomp_foreground_threshold"],
                color_jitter=unl.get("color_jitter"), blur=unl.get("blur"), reducer=reducer,
                mix_use_gt=unl.get("mix_use_gt", False), scaler=scaler)
            # train.py:510-514: ``segmentation_total_loss = segmentation_loss`` binds a second NAME to the same tensor and the
            # unlabeled loss is then added IN PLACE -- the reference's returned 'segmentation_loss' includes the unlabeled term
            # whenever the unlabeled step runs (and equals 'segmentation_total_loss'); kept, so that logged curves compare
            segmentation_total = segmentation_total + u_loss
            segmentation_loss = segmentation_total
            mono_total = mono_total + u_mono
    Fn.flush_deferred_trunks()      # (nothing is parked when the sequence above ended on a releasing call)
    if reducer is not None:
        reducer.finish()
    if tr.get("clip_grad_norm") is not None:
        scaler.unscale_(optimizer)           # train.py:517-518 (a no-op scaler without amp)
        clipped = get_params(model, ["encoder", "segmentation"]) if tr.get("disable_depth_grad_clip", False) \
            else model.parameters()
        torch.nn.utils.clip_grad_norm_(clipped, tr["clip_grad_norm"])
    scaler.step(optimizer)                   # train.py:527-528
    scaler.update()
    if scheduler is not None:
        if isinstance(scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
            scheduler.step(metrics=mIoU)
        else:
            scheduler.step()
    if ema_model is not None:
        update_ema_variables(ema_model, model, 0.99, step, save_monodepth_ema=tr["save_monodepth_ema"],
                             segmentation_name=cfg["model"]["segmentation_name"],
                             freeze_backbone=cfg["model"]["freeze_backbone"])
    total = segmentation_total + mono_total + pseudo_depth_loss
    return {"segmentation_loss": segmentation_loss.detach(), "mono_loss": mono_loss.detach(),
            "pseudo_depth_loss": pseudo_depth_loss.detach(), "feat_dist_loss": feat_dist_loss.detach(),
            "segmentation_total_loss": segmentation_total.detach(), "mono_total_loss": mono_total.detach(),
            "total_loss": total.detach()}
