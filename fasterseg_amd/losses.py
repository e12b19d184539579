"""Losses of the train steps: OHEM cross-entropy and the KL distillation term.

On CPU tensors both are the reference's own PyTorch op chain.  On the GPU the OHEM criterion runs on two HIP kernels
(fs_ohem_ce_fwd / fs_ohem_ce_bwd): at 12 x 19 x 512 x 1024 the logits are 478 MB per head and the ATen chain (softmax,
transposed copy, log_softmax, nll and their backwards) moves that tensor about ten times per head and synchronises the
host twice; the fused form reads it once forward and once backward and keeps the threshold logic on the device."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _OhemCE(torch.autograd.Function):
    """loss = mean over kept pixels of -log softmax(pred)[target];  kept = valid & (p_target <= max(thresh, k-th smallest
    p_target)) when at least min_kept pixels are valid, else kept = valid (loss_opr.py:70-93)."""

    @staticmethod
    def forward(ctx, pred, target, thresh, min_kept, ignore):
        from . import kernels as K
        B, C, H, W = pred.shape
        HW, P = H * W, B * H * W
        logits = pred.detach().contiguous()
        tgt = target.reshape(-1).contiguous()
        buf = torch.empty((3, P), dtype=torch.float32, device=pred.device)
        true_prob, nll, lse = buf[0], buf[1], buf[2]
        K.call("fs_ohem_ce_fwd", K._stream(), logits.data_ptr(), tgt.data_ptr(), B, C, HW, int(ignore), true_prob.data_ptr(),
               nll.data_ptr(), lse.data_ptr())
        valid = tgt.ne(ignore)
        num_valid = valid.sum()
        kept = valid
        if min_kept > 0:        # the reference builds kept_mask only inside `if self.min_kept > 0` (loss_opr.py:80-86)
            threshold = torch.full((), float(thresh), dtype=torch.float32, device=pred.device)
            kth = torch.sort(true_prob).values[min(P, min_kept) - 1]
            threshold = torch.maximum(threshold, kth)
            # OHEM only applies when enough valid pixels exist (loss_opr.py:74-76); decided on the device, no host sync
            apply = (num_valid >= min_kept) & (num_valid > 0)
            kept = valid & (true_prob.le(threshold) | ~apply)
        count = kept.sum()
        loss = (nll * kept).sum() / count
        ctx.save_for_backward(logits, tgt, lse, kept.to(torch.uint8), count)
        ctx.shape = (B, C, H, W)
        return loss

    @staticmethod
    def backward(ctx, g):
        from . import kernels as K
        logits, tgt, lse, kept, count = ctx.saved_tensors
        B, C, H, W = ctx.shape
        scale = (g.float() / count).reshape(1).contiguous()
        d = torch.empty_like(logits)
        K.call("fs_ohem_ce_bwd", K._stream(), logits.data_ptr(), tgt.data_ptr(), lse.data_ptr(), kept.data_ptr(), scale.data_ptr(),
               B, C, H * W, d.data_ptr())
        return d, None, None, None, None


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
        if pred.is_cuda and pred.dtype == torch.float32 and self.criterion.reduction == "mean" and target.dtype == torch.long:
            return _OhemCE.apply(pred, target, self.thresh, self.min_kept, self.ignore_label)
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
                if self.min_kept > 0:       # no mask at all when min_kept == 0 (loss_opr.py:80-86)
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


def _logits_desc(x, size):
    """fs_logits_desc of a low-resolution NHWC logits view (N, C, h, w) that is to be read at resolution `size`."""
    from . import kernels as K
    from ._lib import LogitsDesc
    N, C, h, w = x.shape
    cs = K.channel_stride(x)
    assert cs is not None and cs % 4 == 0 and cs >= ((C + 3) // 4) * 4, "logits must be an NHWC view with a channel stride padded to 4"
    return LogitsDesc(N, h, w, C, cs, int(size[0]), int(size[1]), K.dtype_code(x.dtype))


def _up_workspace(d, device):
    """Scratch of the two-launch backward (per-cell corner sums), fs_loss_up_workspace_bytes."""
    import ctypes
    from . import _lib
    n = int(_lib.lib().fs_loss_up_workspace_bytes(ctypes.byref(d)))
    return torch.empty(n // 4, dtype=torch.float32, device=device)


class _OhemCEUp(torch.autograd.Function):
    """_OhemCE on the bilinear up-sample of low-resolution logits, without the up-sampled tensor (fs_ohem_ce_up_fwd/_bwd):
    ProbOhemCrossEntropy2d(F.interpolate(pred_lo, size, 'bilinear', align_corners=True), target)
    (train/model_seg.py:357-362 + tools/seg_opr/loss_opr.py:63-93)."""

    @staticmethod
    def forward(ctx, pred_lo, target, thresh, min_kept, ignore):
        import ctypes
        from . import kernels as K
        B, H, W = target.shape
        d = _logits_desc(pred_lo, (H, W))
        P = B * H * W
        tgt = target.reshape(-1).contiguous()
        buf = torch.empty((3, P), dtype=torch.float32, device=pred_lo.device)
        true_prob, nll, lse = buf[0], buf[1], buf[2]
        x = pred_lo.detach()
        K.call("fs_ohem_ce_up_fwd", K._stream(), ctypes.byref(d), K._p(x), K._p(tgt), int(ignore), K._p(true_prob), K._p(nll), K._p(lse))
        valid = tgt.ne(ignore)
        num_valid = valid.sum()
        kept = valid
        if min_kept > 0:
            threshold = torch.full((), float(thresh), dtype=torch.float32, device=x.device)
            kth = torch.sort(true_prob).values[min(P, min_kept) - 1]
            threshold = torch.maximum(threshold, kth)
            apply = (num_valid >= min_kept) & (num_valid > 0)
            kept = valid & (true_prob.le(threshold) | ~apply)
        count = kept.sum()
        loss = (nll * kept).sum() / count
        ctx.save_for_backward(x, tgt, lse, kept.to(torch.uint8), count)
        ctx.desc = d
        return loss

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from . import kernels as K
        x, tgt, lse, kept, count = ctx.saved_tensors
        d = ctx.desc
        scale = (g.float() / count).reshape(1).contiguous()
        dx = K.empty_nhwc(d.N, d.C, d.h, d.w, x.dtype, x.device, cs=d.cs)
        ws = _up_workspace(d, x.device)
        K.call("fs_ohem_ce_up_bwd", K._stream(), ctypes.byref(d), K._p(x), K._p(tgt), K._p(lse), K._p(kept), K._p(scale), K._p(dx),
               K._p(ws), ws.numel() * 4)
        return dx, None, None, None, None


class _DistillKLUp(torch.autograd.Function):
    """distill_kl on the up-samples of two low-resolution logit maps (student / teacher may differ in resolution and dtype)."""

    @staticmethod
    def forward(ctx, student_lo, teacher_lo, size):
        import ctypes
        from . import kernels as K
        ds, dt = _logits_desc(student_lo, size), _logits_desc(teacher_lo, size)
        P = ds.N * ds.H * ds.W
        s, t = student_lo.detach(), teacher_lo.detach()
        buf = torch.empty((3, P), dtype=torch.float32, device=s.device)
        K.call("fs_kl_distill_up_fwd", K._stream(), ctypes.byref(ds), K._p(s), ctypes.byref(dt), K._p(t), K._p(buf[0]), K._p(buf[1]), K._p(buf[2]))
        ctx.save_for_backward(s, t, buf)
        ctx.descs = (ds, dt)
        return buf[0].sum() / (P * ds.C)

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from . import kernels as K
        s, t, buf = ctx.saved_tensors
        ds, dt = ctx.descs
        scale = (g.float() / (ds.N * ds.H * ds.W * ds.C)).reshape(1).contiguous()
        dx = K.empty_nhwc(ds.N, ds.C, ds.h, ds.w, s.dtype, s.device, cs=ds.cs)
        ws = _up_workspace(ds, s.device)
        K.call("fs_kl_distill_up_bwd", K._stream(), ctypes.byref(ds), K._p(s), ctypes.byref(dt), K._p(t), K._p(buf[1]), K._p(buf[2]),
               K._p(scale), K._p(dx), K._p(ws), ws.numel() * 4)
        return dx, None, None


def ohem_ce_lowres(criterion, pred_lo, target):
    """`criterion(F.interpolate(pred_lo, target.shape[-2:], mode='bilinear', align_corners=True), target)` for a
    ProbOhemCrossEntropy2d `criterion`, computed from the low-resolution NHWC logits (CUDA only)."""
    assert pred_lo.is_cuda and target.dtype == torch.long and criterion.criterion.reduction == "mean"
    return _OhemCEUp.apply(pred_lo, target, criterion.thresh, criterion.min_kept, criterion.ignore_label)


def distill_kl_lowres(student_lo, teacher_lo, size):
    """distill_kl(up(student_lo), up(teacher_lo)) with both bilinear up-samples to `size` evaluated inside the kernels."""
    assert student_lo.is_cuda and not teacher_lo.requires_grad
    return _DistillKLUp.apply(student_lo, teacher_lo, (int(size[0]), int(size[1])))


class _DistillKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher):
        from . import kernels as K
        B, C, H, W = student.shape
        P = B * H * W
        s, t = student.detach().contiguous(), teacher.detach().contiguous()
        buf = torch.empty((3, P), dtype=torch.float32, device=s.device)
        K.call("fs_kl_distill_fwd", K._stream(), s.data_ptr(), t.data_ptr(), B, C, H * W, buf[0].data_ptr(), buf[1].data_ptr(),
               buf[2].data_ptr())
        ctx.save_for_backward(s, t, buf)
        ctx.dims = (B, C, H * W)
        return buf[0].sum() / (P * C)

    @staticmethod
    def backward(ctx, g):
        from . import kernels as K
        s, t, buf = ctx.saved_tensors
        B, C, HW = ctx.dims
        scale = (g.float() / (B * C * HW)).reshape(1).contiguous()
        d = torch.empty_like(s)
        K.call("fs_kl_distill_bwd", K._stream(), s.data_ptr(), t.data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), scale.data_ptr(),
               B, C, HW, d.data_ptr())
        return d, None


def distill_kl(student_logits, teacher_logits):
    """nn.KLDivLoss()(softmax(student).log(), softmax(teacher)) with the default 'mean' reduction (train/train.py:64,260).
    CUDA fp32 logits take the two-kernel fused form (the teacher receives no gradient, as in the reference's no_grad forward)."""
    if (student_logits.is_cuda and student_logits.dtype == torch.float32 and teacher_logits.dtype == torch.float32
            and student_logits.shape == teacher_logits.shape and not teacher_logits.requires_grad):
        return _DistillKL.apply(student_logits, teacher_logits)
    return F.kl_div(F.log_softmax(student_logits, dim=1), F.softmax(teacher_logits, dim=1), reduction='mean')
