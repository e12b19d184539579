"""TEST INFRASTRUCTURE — CPU restatement of the step glue's criteria (torch-CPU ops), used by tests/, bench.py's parity gates and
cpu_baseline legs only.  Pinned to tests/golden/loss.npz (outputs of the unmodified reference, oracle/make_golden.py gen_loss)
by tests/test_oracle_golden.py.

  ohem_ce            ProbOhemCrossEntropy2d.forward             tools/seg_opr/loss_opr.py:63-93
  student_step_loss  OHEM(p8) + 0.2 OHEM(p16) + 0.2 OHEM(p32) + KLDiv(log softmax(student p8), softmax(teacher))   train/train.py:246-262
"""
import torch
import torch.nn.functional as F


def ohem_ce(pred, target, ignore_label=255, thresh=0.7, min_kept=0):
    """Online hard example mining on the softmax probability of the true class (loss_opr.py:63-93): pixels whose true-class
    probability is at most max(thresh, the min_kept-th smallest such probability) are kept, the others ignored; mean CE."""
    c = pred.shape[1]
    flat = target.reshape(-1)
    valid = flat != ignore_label
    n_valid = int(valid.sum())
    keep = valid
    if min_kept <= n_valid and n_valid > 0:                                   # :72-86 (else: every valid pixel contributes)
        prob = F.softmax(pred, dim=1).transpose(0, 1).reshape(c, -1)
        true_prob = prob.gather(0, (flat * valid).unsqueeze(0)).squeeze(0)
        true_prob = torch.where(valid, true_prob, torch.ones_like(true_prob))  # masked_fill_(~valid, 1)
        if min_kept > 0:
            kth = true_prob.sort().values[min(true_prob.numel(), min_kept) - 1]
            threshold = kth if float(kth) > thresh else true_prob.new_tensor(thresh)
            keep = valid & (true_prob <= threshold)
    tgt = torch.where(keep, flat, torch.full_like(flat, ignore_label)).reshape(target.shape)
    return F.cross_entropy(pred, tgt, ignore_index=ignore_label)


def student_step_loss(p8, p16, p32, teacher_logits, target, min_kept, lamb=0.2, thresh=0.7, ignore_label=255):
    """train/train.py:246-262 on full-resolution logits."""
    loss = ohem_ce(p8, target, ignore_label, thresh, min_kept)
    loss = loss + lamb * ohem_ce(p16, target, ignore_label, thresh, min_kept)
    loss = loss + lamb * ohem_ce(p32, target, ignore_label, thresh, min_kept)
    kl = F.kl_div(F.softmax(p8, dim=1).log(), F.softmax(teacher_logits, dim=1), reduction="mean")    # nn.KLDivLoss() default
    return loss + kl
