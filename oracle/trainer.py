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


def use_gt(softmax_u_w, unlabeled_inputs):
    """/root/reference/train.py:667-672 (``mix_use_gt``): labeled samples of the unlabeled batch take their one-hot label
    planes (loader/sequence_segmentation_loader.py:237-246: int64, all zero on ignored pixels) instead of the teacher's
    softmax; the assignment casts to the softmax's dtype."""
    out = softmax_u_w.clone()
    with torch.no_grad():
        for i in range(out.shape[0]):
            if bool(unlabeled_inputs["is_labeled"][i]):
                out[i] = unlabeled_inputs["onehot_lbl"][i]
    return out


def train_step_segmentation_unlabeled(sd_student, sd_teacher, model_cfg, loss_oracle, unlabeled_inputs, margin=0.03,
                                      foreground_threshold=0.0, consistency_weight=1.0, monodepth_lambda=1.0,
                                      tiebreak_noise=None, mask_override=None, dropout=False, mix_use_gt=False):
    """/root/reference/train.py:653-724 with exp-212 flags (mix_mask "depthcomp", depthmix_online_depth True,
    backward_first_pseudo_label False, jitter / blur off): teacher forward -> softmax (:664-666); student forward on the
    unmixed frames -> monodepth loss backward (:679-689) and min-max normalised online disparity (:690-697); depthcomp
    mask (:585-604); mix of image and teacher softmax (:717-722); student forward on the mixed frames; pseudo-label loss
    backward (:723-724).  ``sd_student`` leaves accumulate .grad.  Returns a dict of the intermediate tensors.
    ``mix_use_gt`` (:667-672, on in the exp-212 block experiments.py:343-357): the teacher softmax of every sample with
    ``unlabeled_inputs["is_labeled"][i]`` is replaced by ``unlabeled_inputs["onehot_lbl"][i]`` before anything reads it."""
    from . import nets as N, segmix as S
    with torch.no_grad():
        out_t = N.model_forward({k: v.detach() for k, v in sd_teacher.items()}, model_cfg, dict(unlabeled_inputs), train=True,
                                dropout=dropout, use_pose_net=False)
    softmax_u_w = torch.softmax(out_t["semantics"].detach(), dim=1)
    softmax_u_w = use_gt(softmax_u_w, unlabeled_inputs) if mix_use_gt else softmax_u_w
    out_1 = N.model_forward(sd_student, model_cfg, dict(unlabeled_inputs), train=True, dropout=dropout)
    loss_oracle.generate_images_pred(unlabeled_inputs, out_1)
    mono = monodepth_lambda * loss_oracle.compute_losses(unlabeled_inputs, out_1, tiebreak_noise=tiebreak_noise)["loss"]
    mono.backward()
    depths = S.normalize_disparity(out_1[("disp", 0)].detach())
    mask = S.depthcomp_mask(depths, margin, foreground_threshold) if mask_override is None else mask_override
    img_mixed, _ = S.mix(mask, data=unlabeled_inputs[("color_aug", 0, 0)])
    inp2 = dict(unlabeled_inputs)
    inp2[("color_aug", 0, 0)] = img_mixed
    out_2 = N.model_forward(sd_student, model_cfg, inp2, train=True, dropout=dropout)
    soft_mixed, _ = S.mix(mask, data=softmax_u_w)
    L_2, label = calc_pseudo_label_loss(soft_mixed, out_2["semantics"], consistency_weight)
    L_2.backward()
    return {"softmax_u_w": softmax_u_w, "depths": depths, "mask": mask, "img_mixed": img_mixed, "soft_mixed": soft_mixed,
            "L_2": L_2.detach(), "mono_loss": mono.detach(), "pseudo_label": label, "disp0": out_1[("disp", 0)].detach()}
