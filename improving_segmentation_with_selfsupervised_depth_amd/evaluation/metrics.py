"""Mirror of evaluation/metrics.py ``runningScore`` with the confusion matrix accumulated on the
device: the reference copies predictions and labels to the host every validation batch and runs ``np.bincount`` per
image (metrics.py:12-25, train.py:848-851); here ``update`` is one kernel (optionally fused with the class argmax,
``update_from_logits``) into a device-resident int64 histogram, copied to the host only when the scores are read."""
import numpy as np
import torch

from .. import _lib
from .. import hipops as H


class runningScore(object):
    def __init__(self, n_classes):
        self.n_classes = n_classes
        self._host = np.zeros((n_classes, n_classes))      # contributions that arrived as numpy arrays
        self._dev = None

    def _hist(self, device):
        if self._dev is None or self._dev.device != device:
            assert self._dev is None, "runningScore was updated from two devices"
            self._dev = torch.zeros(self.n_classes * self.n_classes, dtype=torch.int64, device=device)
        return self._dev

    @property
    def confusion_matrix(self):
        """float64 [n, n] like the reference's attribute (metrics.py:10)"""
        m = self._host.copy()
        if self._dev is not None:
            m += self._dev.cpu().numpy().reshape(self.n_classes, self.n_classes).astype(np.float64)
        return m

    def _fast_hist(self, label_true, label_pred, n_class):
        """metrics.py:12-17 (host fallback for numpy inputs)"""
        mask = (label_true >= 0) & (label_true < n_class)
        return np.bincount(n_class * label_true[mask].astype(int) + label_pred[mask], minlength=n_class ** 2).reshape(
            n_class, n_class)

    def update(self, label_trues, label_preds):
        """metrics.py:19-25; device tensors stay on the device"""
        if torch.is_tensor(label_trues) and torch.is_tensor(label_preds) and (label_trues.is_cuda or _lib.HOST_POINTERS_OK):
            H.confusion_update(self._hist(label_trues.device), label_trues.detach(), pred=label_preds.detach())
            return
        if torch.is_tensor(label_trues):
            label_trues = label_trues.detach().cpu().numpy()
        if torch.is_tensor(label_preds):
            label_preds = label_preds.detach().cpu().numpy()
        for lt, lp in zip(label_trues, label_preds):
            self._host += self._fast_hist(lt.flatten(), lp.flatten(), self.n_classes)

    def update_from_logits(self, label_trues, semantics):
        """train.py:848-851 in one kernel: ``pred = semantics.data.max(1)[1]`` fused with the histogram"""
        H.confusion_update(self._hist(semantics.device), label_trues.detach(), logits=semantics.detach())

    def get_scores(self):
        """metrics.py:27-56"""
        hist = self.confusion_matrix
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.diag(hist) / hist.sum(axis=1)
        acc_cls = np.nanmean(acc_cls)
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        mean_iu = np.nanmean(iu)
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
        cls_iu = dict(zip(range(self.n_classes), iu))
        return ({"Overall Acc: \t": acc, "Mean Acc : \t": acc_cls, "FreqW Acc : \t": fwavacc, "Mean IoU : \t": mean_iu},
                cls_iu)

    def reset(self):
        self._host = np.zeros((self.n_classes, self.n_classes))
        if self._dev is not None:
            self._dev.zero_()
