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


def test_fused_ohem_matches_reference_fixture():
    """The HIP criterion against the values the reference's own tools/seg_opr/loss_opr.py produced (tests/golden/loss.npz,
    oracle/make_golden.py) - no product code on the reference side of the comparison."""
    import numpy as np
    from fasterseg_amd.losses import ProbOhemCrossEntropy2d
    from tests._util import load_npz
    store = load_npz("loss.npz")
    for i in range(4):
        pred = torch.tensor(store["ohem%d/pred" % i]).cuda().requires_grad_(True)
        target = torch.tensor(store["ohem%d/target" % i]).cuda()
        thresh, min_kept = store["ohem%d/cfg" % i]
        loss = ProbOhemCrossEntropy2d(255, thresh=float(thresh), min_kept=int(min_kept))(pred, target)
        assert type(loss.grad_fn).__name__ == "_OhemCEBackward"
        loss.backward()
        assert abs(float(loss.detach()) - float(store["ohem%d/loss" % i][0])) < 1e-5, i
        np.testing.assert_allclose(pred.grad.cpu().numpy(), store["ohem%d/grad" % i], atol=2e-6)


def _lowres_logits(shape, dtype, seed, cs=32):
    """(N, C, h, w) NHWC view with channel stride cs like a Head's classifier output, plus its fp32 value on the CPU."""
    from fasterseg_amd import kernels as K
    N, C, h, w = shape
    g = torch.Generator().manual_seed(seed)
    val = (torch.randn(N, C, h, w, generator=g) * 2.0).to(dtype).float()
    buf = K.empty_nhwc(N, C, h, w, dtype, "cuda", cs=cs, zero=True)
    buf.copy_(val.cuda().to(dtype))
    return buf, val


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", [
    dict(lo=(2, 19, 8, 12), up=8, thresh=0.7, min_kept=2 * 64 * 96 // 16, ignore_frac=0.05),
    dict(lo=(1, 19, 4, 6), up=16, thresh=0.7, min_kept=64 * 96 // 16, ignore_frac=0.1),
    dict(lo=(2, 19, 2, 3), up=32, thresh=0.2, min_kept=300, ignore_frac=0.0),
    dict(lo=(1, 19, 5, 7), up=8, thresh=0.7, min_kept=10 ** 7, ignore_frac=0.02),             # not enough valid pixels: no OHEM
], ids=["x8", "x16", "x32", "no_ohem"])
def test_ohem_from_lowres_logits_matches_upsample_then_reference_chain(case, dtype):
    """losses.ohem_ce_lowres(pred_lo) == ProbOhemCrossEntropy2d(F.interpolate(pred_lo, align_corners=True)) - the reference's
    train/model_seg.py:357-362 + loss_opr.py:63-93 on the CPU - in value and in the gradient w.r.t. the LOW-resolution logits."""
    import torch.nn.functional as F
    from fasterseg_amd.losses import ProbOhemCrossEntropy2d, ohem_ce_lowres
    N, C, h, w = case["lo"]
    H, W = h * case["up"], w * case["up"]
    buf, val = _lowres_logits(case["lo"], dtype, 21)
    g = torch.Generator().manual_seed(22)
    target = torch.randint(0, C, (N, H, W), generator=g)
    target[torch.rand(N, H, W, generator=g) < case["ignore_frac"]] = 255
    crit = ProbOhemCrossEntropy2d(255, thresh=case["thresh"], min_kept=case["min_kept"])
    ref_in = val.clone().requires_grad_(True)
    ref = crit(F.interpolate(ref_in, size=(H, W), mode="bilinear", align_corners=True), target)
    ref.backward()
    x = buf.detach().requires_grad_(True)
    got = ohem_ce_lowres(crit, x, target.cuda())
    (got * 0.5).backward()
    assert abs(float(got) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref))), (float(got), float(ref))
    gtol = (1e-6 + 2e-4 * float(ref_in.grad.abs().max())) if dtype == torch.float32 else 1e-2 * float(ref_in.grad.abs().max())
    err = float((x.grad.float().cpu() - 0.5 * ref_in.grad).abs().max())
    assert err <= gtol, (err, gtol)


@pytest.mark.parametrize("dtypes", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16)],
                         ids=["fp32", "bf16", "s_fp32_t_bf16"])
def test_kl_from_lowres_logits_matches_upsample_then_kldivloss(dtypes):
    import torch.nn.functional as F
    from fasterseg_amd.losses import distill_kl, distill_kl_lowres
    H, W = 64, 96
    s_buf, s_val = _lowres_logits((2, 19, 8, 12), dtypes[0], 31)
    t_buf, t_val = _lowres_logits((2, 19, 8, 12), dtypes[1], 32)
    ref_in = s_val.clone().requires_grad_(True)
    up = lambda t: F.interpolate(t, size=(H, W), mode="bilinear", align_corners=True)
    ref = distill_kl(up(ref_in), up(t_val))                                # nn.KLDivLoss chain on the CPU (train/train.py:260)
    ref.backward()
    x = s_buf.detach().requires_grad_(True)
    got = distill_kl_lowres(x, t_buf, (H, W))
    (got * 2.0).backward()
    assert abs(float(got) - float(ref)) <= 2e-5 * abs(float(ref)) + 1e-8
    gtol = (1e-9 + 2e-4 * float(ref_in.grad.abs().max())) if dtypes[0] == torch.float32 else 1e-2 * float(ref_in.grad.abs().max())
    assert float((x.grad.float().cpu() - 2.0 * ref_in.grad).abs().max()) <= 2.0 * gtol


def test_student_step_fused_loss_equals_materialised_loss():
    """One student distillation step with the loss heads fused into the up-sample vs the same step through the up-sampled
    (B, 19, H, W) tensors: same loss, same parameters afterwards."""
    from fasterseg_amd import train_step
    losses, states = [], []
    for fused in (True, False):
        torch.manual_seed(0)
        st = train_step.StudentDistillStep(2, 128, 256, fused_loss=fused)
        imgs, target = train_step.synthetic_batch(2, 128, 256, 0, "cuda")
        losses.append(float(st.step(imgs, target)))
        states.append({k: v.detach().float().cpu().clone() for k, v in st.student.state_dict().items()})
    assert abs(losses[0] - losses[1]) <= 1e-4 * abs(losses[1]), losses
    worst = max(float((states[0][k] - states[1][k]).abs().max()) for k in states[0])
    assert worst <= 1e-4, worst
