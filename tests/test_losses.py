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
