"""CPU restatement of the trainer-side callers of the path.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product
never does).  Pinned: tests/golden/trainer.npz holds inputs/outputs of the reference's own
``Trainer.update_ema_variables`` / ``Trainer.calc_pseudo_label_loss`` (run by tests/golden/make_golden.py, which
imports /root/reference/train.py), and tests/test_oracle_golden.py checks these functions against them.
"""
import numpy as np
import torch

from .segmix import cross_entropy2d


def select_ema_params(model, ema_model, save_monodepth_ema, segmentation_name, freeze_backbone):
    """parameter lists as /root/reference/train.py:347-352 picks them (extract_ema_params :124-135, :317-326)"""
    def extract(names):
        mp = [p for k, v in model.models.items() if k in names for p in v.parameters()]
        ep = [p for k, v in ema_model.models.items() if k in names for p in v.parameters()]
        return mp, ep
    if save_monodepth_ema:
        return extract(["depth"] + ([] if freeze_backbone else ["encoder"]))
    if segmentation_name == "mtl_pad":
        return extract(["depth", "encoder", "mtl_decoder"])
    return list(model.parameters()), list(ema_model.parameters())


def update_ema_variables(ema_params, model_params, alpha_teacher, iteration):
    """/root/reference/train.py:353-357 -- in place on ``ema_params`` (lists of tensors already selected the way
    train.py:347-352 selects them)."""
    alpha_teacher = min(1 - 1 / (iteration + 1), alpha_teacher)
    for ema_param, param in zip(ema_params, model_params):
        ema_param.data[:] = alpha_teacher * ema_param.data + (1 - alpha_teacher) * param.data
    return ema_params


def calc_pseudo_label_loss(teacher_softmax, student_logits, consistency_weight, ignore_index=250):
    """/root/reference/train.py:644-651 -> (L_u, pseudo_label)"""
    max_probs, pseudo_label = torch.max(teacher_softmax, dim=1)
    pseudo_label = pseudo_label.clone()
    pseudo_label[max_probs == 0] = ignore_index
    unlabeled_weight = torch.sum(max_probs.ge(0.968).long() == 1).item() / np.prod(pseudo_label.shape)
    pixel_weight = unlabeled_weight * torch.ones(max_probs.shape)
    L_u = consistency_weight * cross_entropy2d(student_logits, pseudo_label, pixel_weights=pixel_weight)
    return L_u, pseudo_label
