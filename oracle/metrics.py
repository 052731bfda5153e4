"""CPU restatement of the reference's segmentation validation metric.

TEST INFRASTRUCTURE ONLY.  Pinned: tests/golden/trainer.npz (keys ``cm_*``) holds the confusion matrix and scores of the
reference's own ``evaluation.metrics.runningScore`` on seeded labels; tests/test_oracle_golden.py checks this file
against them."""
import numpy as np


def fast_hist(label_true, label_pred, n_class):
    """/root/reference/evaluation/metrics.py:12-17"""
    mask = (label_true >= 0) & (label_true < n_class)
    return np.bincount(n_class * label_true[mask].astype(int) + label_pred[mask], minlength=n_class ** 2).reshape(n_class, n_class)


def confusion_matrix(label_trues, label_preds, n_class):
    """/root/reference/evaluation/metrics.py:19-25 (numpy int arrays [B,H,W])"""
    cm = np.zeros((n_class, n_class))
    for lt, lp in zip(label_trues, label_preds):
        cm += fast_hist(lt.flatten(), lp.flatten(), n_class)
    return cm


def scores(hist):
    """/root/reference/evaluation/metrics.py:34-44 -> (overall acc, mean acc, freq-weighted acc, mean IoU, per-class IoU)"""
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        mean_iu = np.nanmean(iu)
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
    return acc, acc_cls, fwavacc, mean_iu, iu
