"""Independent pin of the torch-CPU primitives the oracle drives: oracle/prim.c (plain C, double accumulation) vs
F.conv2d / F.interpolate(align_corners=True) / F.batch_norm on the argument patterns the hot path uses.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import prim


@pytest.mark.parametrize("case", [(2, 8, 9, 11, 12, 3, 1, 1), (1, 16, 10, 14, 8, 3, 2, 1), (2, 12, 6, 8, 19, 1, 1, 0),
                                  (1, 3, 16, 20, 8, 3, 2, 1), (1, 8, 8, 8, 4, 1, 2, 0)])
def test_conv2d(case):
    N, Cin, H, W, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(1)
    x, w, b = torch.randn(N, Cin, H, W, generator=g), torch.randn(Cout, Cin, k, k, generator=g), torch.randn(Cout, generator=g)
    np.testing.assert_allclose(prim.conv2d(x.numpy(), w.numpy(), b.numpy(), stride, pad), F.conv2d(x, w, b, stride, pad).numpy(),
                               rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape,size", [((2, 4, 9, 12), (18, 24)), ((1, 3, 16, 32), (8, 16)), ((1, 2, 7, 14), (3, 7)),
                                        ((1, 2, 3, 7), (7, 14)), ((1, 2, 4, 8), (32, 64)), ((1, 2, 1, 6), (4, 12)),
                                        ((1, 2, 6, 6), (1, 1))])
def test_bilinear_align_corners(shape, size):
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(2))
    want = F.interpolate(x, size=size, mode="bilinear", align_corners=True).numpy()
    np.testing.assert_allclose(prim.bilinear(x.numpy(), size), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("training", [False, True])
def test_batchnorm(training):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 6, 5, 7, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(6, generator=g) + 0.5, torch.randn(6, generator=g)
    rm, rv = torch.randn(6, generator=g) * 0.1, torch.rand(6, generator=g) + 0.5
    rm_np, rv_np = rm.numpy().copy(), rv.numpy().copy()
    want = F.batch_norm(x, rm, rv, gamma, beta, training, 0.1, 1e-5)
    got = prim.batchnorm(x.numpy(), gamma.numpy(), beta.numpy(), rm_np, rv_np, training, 0.1, 1e-5)
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rm_np, rm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv_np, rv.numpy(), rtol=1e-5, atol=1e-6)
