"""Segmentation metrics of the evaluation path (drop-in for the reference's tools/seg_opr/metric.py: hist_info,
compute_score), with the per-pixel part on the device.

`hist_info(n_cl, pred, gt)` keeps the reference's signature and return value (hist, labeled, correct) - pred is the uint8
class map the engine produced on the GPU, gt a uint8 / int32 / int64 label tensor - and runs fs_hist_info (integer atomics,
bit-exact with np.bincount) instead of copying a (1024, 2048) map per image to the host.  `HistAccumulator` keeps the counts of
a whole validation run on the device and reads them back once (compute_metric, train/eval.py:55-68)."""
import ctypes

import numpy as np
import torch

from . import kernels as K
from ._lib import call

np.seterr(divide='ignore', invalid='ignore')


def _gt_bytes(gt):
    try:
        return {torch.uint8: 1, torch.int32: 4, torch.int64: 8}[gt.dtype]
    except KeyError:
        raise TypeError("labels must be uint8, int32 or int64, got %s" % gt.dtype)


class HistAccumulator:
    def __init__(self, n_cl, device="cuda"):
        self.n_cl = n_cl
        self.hist = torch.zeros(n_cl * n_cl, dtype=torch.int64, device=device)
        self.counts = torch.zeros(2, dtype=torch.int64, device=device)          # labeled, correct

    def add(self, pred, gt):
        assert pred.shape == gt.shape, (pred.shape, gt.shape)
        assert pred.is_cuda and pred.dtype == torch.uint8 and pred.is_contiguous() and gt.is_cuda and gt.is_contiguous()
        call("fs_hist_info", K._stream(), K._p(pred), K._p(gt), _gt_bytes(gt), pred.numel(), self.n_cl, K._p(self.hist), K._p(self.counts))

    def result(self):
        """(hist (n_cl, n_cl) int64 numpy, labeled, correct) - one device-to-host read for the whole run."""
        h = self.hist.cpu().numpy().reshape(self.n_cl, self.n_cl)
        c = self.counts.cpu().numpy()
        return h, int(c[0]), int(c[1])


def hist_info(n_cl, pred, gt):
    """tools/seg_opr/metric.py:7-17."""
    acc = HistAccumulator(n_cl, pred.device)
    acc.add(pred.contiguous(), gt.contiguous())
    return acc.result()


def compute_score(hist, correct, labeled):
    """tools/seg_opr/metric.py:20-29 (19 x 19 host arithmetic)."""
    hist = np.asarray(hist, dtype=np.float64)
    iu = np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))
    mean_IU = np.nanmean(iu)
    mean_IU_no_back = np.nanmean(iu[1:])
    mean_pixel_acc = correct / labeled
    return iu, mean_IU, mean_IU_no_back, mean_pixel_acc
