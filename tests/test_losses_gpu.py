"""Fused OHEM cross-entropy (fs_ohem_ce_fwd / fs_ohem_ce_bwd) against the reference's PyTorch op chain (the CPU path of
fasterseg_amd.losses.ProbOhemCrossEntropy2d, itself pinned to the reference fixture in tests/test_losses.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [
    dict(shape=(2, 19, 32, 48), thresh=0.7, min_kept=2 * 32 * 48 // 16, ignore_frac=0.05),      # k-th value above thresh or not
    dict(shape=(3, 19, 17, 23), thresh=0.05, min_kept=400, ignore_frac=0.1),                    # k-th smallest decides
    dict(shape=(1, 19, 16, 16), thresh=0.7, min_kept=10 ** 6, ignore_frac=0.0),                 # fewer valid than min_kept: no OHEM
    dict(shape=(2, 7, 20, 20), thresh=0.9, min_kept=0, ignore_frac=0.5),                        # threshold only
], ids=["typical", "kth", "not_enough_valid", "thresh_only"])
def test_fused_ohem_matches_torch_chain(case):
    from fasterseg_amd.losses import ProbOhemCrossEntropy2d
    g = torch.Generator().manual_seed(5)
    B, C, H, W = case["shape"]
    pred = (torch.randn(B, C, H, W, generator=g) * 2.0).requires_grad_(True)
    target = torch.randint(0, C, (B, H, W), generator=g)
    target[torch.rand(B, H, W, generator=g) < case["ignore_frac"]] = 255
    crit = ProbOhemCrossEntropy2d(ignore_label=255, thresh=case["thresh"], min_kept=case["min_kept"])
    ref = crit(pred, target)
    ref.backward()
    pred_d = pred.detach().cuda().requires_grad_(True)
    got = crit(pred_d, target.cuda())
    assert type(got.grad_fn).__name__ == "_OhemCEBackward"
    got.backward()
    assert abs(float(got) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref))), (float(got), float(ref))
    err = float((pred_d.grad.cpu() - pred.grad).abs().max())
    assert err <= 1e-6 + 1e-4 * float(pred.grad.abs().max()), err
    # scaled upstream gradient and a non-contiguous prediction
    pred_t = pred.detach().cuda().permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2).requires_grad_(True)
    (crit(pred_t, target.cuda()) * 0.2).backward()
    assert float((pred_t.grad.cpu() - 0.2 * pred.grad).abs().max()) <= 1e-6 + 1e-4 * float(pred.grad.abs().max())


def test_fused_kl_distillation_matches_torch_chain():
    from fasterseg_amd.losses import distill_kl
    g = torch.Generator().manual_seed(6)
    s = (torch.randn(2, 19, 24, 40, generator=g) * 2.0).requires_grad_(True)
    t = torch.randn(2, 19, 24, 40, generator=g) * 3.0
    ref = distill_kl(s, t)
    ref.backward()
    sd = s.detach().cuda().requires_grad_(True)
    got = distill_kl(sd, t.cuda())
    assert type(got.grad_fn).__name__ == "_DistillKLBackward"
    (got * 3.0).backward()
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)) + 1e-8
    assert float((sd.grad.cpu() - 3.0 * s.grad).abs().max()) <= 1e-9 + 1e-4 * float(s.grad.abs().max()) * 3.0
