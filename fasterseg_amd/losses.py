"""Step glue that stays on PyTorch ops (not part of the conv hot path, used as-is so the backward
signal of the benchmarked steps matches the reference): OHEM cross-entropy and the KL distillation term."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ProbOhemCrossEntropy2d(nn.Module):
    """Online-hard-example-mining CE (reference tools/seg_opr/loss_opr.py:43-93): pixels whose predicted probability of
    the true class is above max(thresh, the min_kept-th smallest such probability) are ignored."""

    def __init__(self, ignore_label, reduction='mean', thresh=0.6, min_kept=256, down_ratio=1, use_weight=False):
        super().__init__()
        self.ignore_label = ignore_label
        self.thresh = float(thresh)
        self.min_kept = int(min_kept)
        self.down_ratio = down_ratio
        if use_weight:
            raise NotImplementedError("class-weighted OHEM is not used by the reference training scripts")
        self.criterion = nn.CrossEntropyLoss(reduction=reduction, ignore_index=ignore_label)

    def forward(self, pred, target):
        b, c, h, w = pred.size()
        flat = target.reshape(-1)
        valid = flat.ne(self.ignore_label)
        flat = flat * valid.long()
        num_valid = int(valid.sum())
        if self.min_kept <= num_valid and num_valid > 0:
            with torch.no_grad():
                prob = F.softmax(pred, dim=1).transpose(0, 1).reshape(c, -1)
                true_prob = prob.gather(0, flat.unsqueeze(0)).squeeze(0).masked_fill(~valid, 1)
                threshold = self.thresh
                if self.min_kept > 0:
                    # k-th smallest probability = sorted[min_kept-1] (the reference argsorts, loss_opr.py:82-83); a device
                    # radix sort is ~30x faster than torch.kthvalue at 6.3 M pixels on ROCm
                    kth = torch.sort(true_prob).values[min(true_prob.numel(), self.min_kept) - 1]
                    if float(kth) > self.thresh:
                        threshold = kth
                kept = true_prob.le(threshold)
                flat = flat * kept.long()
                valid = valid & kept
        flat = flat.masked_fill(~valid, self.ignore_label)
        return self.criterion(pred, flat.view(b, h, w))


def distill_kl(student_logits, teacher_logits):
    """nn.KLDivLoss()(softmax(student).log(), softmax(teacher)) with the default 'mean' reduction (train/train.py:64,260)."""
    return F.kl_div(F.log_softmax(student_logits, dim=1), F.softmax(teacher_logits, dim=1), reduction='mean')
