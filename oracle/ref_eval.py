"""TEST INFRASTRUCTURE — CPU oracle of the evaluation path: tools/seg_opr/metric.py (hist_info, compute_score) and the
class-map reduction of tools/engine/evaluator.py:205-225,297-318, restated in numpy.  Pinned by tests/golden/eval.npz, which
oracle/make_golden.py produced by calling the reference's own metric.py."""
import numpy as np


def hist_info(n_cl, pred, gt):
    """metric.py:7-17."""
    assert pred.shape == gt.shape
    k = (gt >= 0) & (gt < n_cl)
    labeled = np.sum(k)
    correct = np.sum(pred[k] == gt[k])
    return np.bincount(n_cl * gt[k].astype(int) + pred[k].astype(int), minlength=n_cl ** 2).reshape(n_cl, n_cl), labeled, correct


def compute_score(hist, correct, labeled):
    """metric.py:20-29."""
    with np.errstate(divide='ignore', invalid='ignore'):
        iu = np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))
        return iu, np.nanmean(iu), np.nanmean(iu[1:]), correct / labeled


def class_map(logits_chw):
    """evaluator.py:311 score = exp(score); :215-223 permute, argmax over the class axis."""
    return np.exp(logits_chw.astype(np.float64)).transpose(1, 2, 0).argmax(2)
