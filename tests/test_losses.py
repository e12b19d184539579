"""Step glue parity (CPU): fasterseg_amd.losses.ProbOhemCrossEntropy2d vs the reference's tools/seg_opr/loss_opr.py
(fixture tests/golden/loss.npz from oracle/make_golden.py)."""
import numpy as np
import torch

from fasterseg_amd.losses import ProbOhemCrossEntropy2d, distill_kl
from tests._util import load_npz


def test_ohem_matches_reference():
    store = load_npz("loss.npz")
    for i in range(4):
        pred = torch.tensor(store["ohem%d/pred" % i]).requires_grad_(True)
        target = torch.tensor(store["ohem%d/target" % i])
        thresh, min_kept = store["ohem%d/cfg" % i]
        loss = ProbOhemCrossEntropy2d(255, thresh=float(thresh), min_kept=int(min_kept))(pred, target)
        loss.backward()
        assert abs(float(loss.detach()) - float(store["ohem%d/loss" % i][0])) < 1e-5, i
        np.testing.assert_allclose(pred.grad.numpy(), store["ohem%d/grad" % i], atol=1e-6)


def test_distill_kl_is_torch_kldivloss():
    g = torch.Generator().manual_seed(1)
    s, t = torch.randn(2, 19, 6, 8, generator=g), torch.randn(2, 19, 6, 8, generator=g)
    want = torch.nn.KLDivLoss()(torch.softmax(s, 1).log(), torch.softmax(t, 1))
    assert torch.allclose(distill_kl(s, t), want, atol=1e-6)


def test_ohem_with_min_kept_zero_is_plain_cross_entropy():
    """The reference only builds its keep-mask inside `if self.min_kept > 0` (tools/seg_opr/loss_opr.py:80-86): with
    min_kept == 0 every valid pixel contributes, whatever `thresh` is."""
    import torch
    from fasterseg_amd.losses import ProbOhemCrossEntropy2d
    g = torch.Generator().manual_seed(3)
    pred = torch.randn(2, 19, 8, 12, generator=g) * 3
    target = torch.randint(0, 19, (2, 8, 12), generator=g)
    target[0, :2] = 255
    got = ProbOhemCrossEntropy2d(ignore_label=255, thresh=0.3, min_kept=0)(pred, target)
    want = torch.nn.functional.cross_entropy(pred, target, ignore_index=255)
    assert abs(float(got) - float(want)) < 1e-6
