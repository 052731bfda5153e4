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

from . import hipops as H
from .loss.loss import cross_entropy2d

__all__ = ["extract_ema_params", "EmaUpdater", "update_ema_variables", "calc_pseudo_label_loss"]


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
        self._table = H.multi_tensor_table([e.data for e in ema_params], [p.data for p in model_params])

    def matches(self, model_params, ema_params):
        return self._key == tuple((e.data_ptr(), p.data_ptr(), p.numel()) for e, p in zip(ema_params, model_params))

    def step(self, alpha_teacher, iteration):
        # "Use the true average until the exponential average is more correct" (train.py:353-354); the two scalars are
        # rounded to fp32 the way torch rounds a Python double that multiplies a float32 tensor
        alpha = min(1 - 1 / (iteration + 1), alpha_teacher)
        if self._table is not None:
            H.multi_tensor_lerp(self._table, np.float32(alpha), np.float32(1 - alpha), self._like)


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
