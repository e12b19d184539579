"""fs_conv_bn_act_train_fwd / _bwd through functional.conv_bn_act on both sides of the statistics switch (units.hip:stats_in_epilogue):
maps up to ~13 k pixels keep the BatchNorm statistics in the convolution's epilogue (float atomics per 32 output rows), larger ones
take the separate reduction pass over z.  Reference: F.conv2d -> F.batch_norm(training=True) -> relu and their autograd on the CPU in
fp32 (what operations.py:ConvNorm / seg_oprs.py:ConvBnRelu compute).  fp32 (with the ReLU) 1e-4-level; bf16 storage 3e-2 of max|ref|,
compared WITHOUT the ReLU: with it ~0.3 % of the outputs sit within a bf16 rounding of zero and flip their mask against the fp32
reference, which is a property of the storage type (measured: sparse O(1) errors in dz, 10-20 % of max|dx|), not of these kernels -
the rectified path is pinned in fp32 here and in bf16 by tests/test_bn_group_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, Cin, Cout, H, W, k, stride
    (1, 32, 32, 64, 96, 3, 1),        # 6144 px: statistics in the conv epilogue
    (2, 32, 32, 96, 128, 3, 1),       # 24576 px: separate pass
    (2, 64, 32, 128, 128, 3, 2),      # stride 2, 8192 px out: epilogue
    (4, 32, 64, 128, 128, 3, 1),      # 65536 px: separate pass, bf16 routed to the register-staged implicit GEMM
    (2, 64, 48, 96, 128, 1, 1),       # 1x1, 24576 px
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=["%dx%d-%d@%dx%d-k%d-s%d" % c for c in CASES])
def test_conv_bn_relu_unit_matches_torch(case, dtype):
    from fasterseg_amd import functional as FN
    N, cin, cout, H, W, k, stride = case
    g = torch.Generator().manual_seed(3)
    q = lambda t: t.to(dtype).float()
    x = q(torch.randn(N, cin, H, W, generator=g))
    w = q(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    rm0, rv0 = torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5
    pad = k // 2
    relu = dtype == torch.float32

    xr, wr, gr, br = (t.clone().requires_grad_(True) for t in (x, w, gamma, beta))
    rm, rv = rm0.clone(), rv0.clone()
    want = F.batch_norm(F.conv2d(xr, wr, None, stride, pad), rm, rv, gr, br, True, 0.1, 1e-5)
    if relu:
        want = F.relu(want)
    dy = q(torch.randn(want.shape, generator=g))
    want.backward(dy)

    xd, wd, gd, bd = (t.cuda().requires_grad_(True) for t in (x, w, gamma, beta))
    rmd, rvd = rm0.cuda(), rv0.cuda()
    FN.set_compute_dtype(dtype)
    try:
        got_nchw = FN.conv_bn_act(xd, wd, gd, bd, rmd, rvd, stride, pad, relu, True)      # NHWC storage, logical NCHW shape
        assert tuple(got_nchw.shape) == tuple(want.shape)
        got_nchw.float().backward(dy.cuda())
    finally:
        FN.set_compute_dtype(torch.float32)
    tol = 2e-4 if dtype == torch.float32 else 3e-2

    def close(a, b, what, scale=None):
        s = float(b.abs().max()) if scale is None else scale
        err = float((a.detach().float().cpu() - b.detach()).abs().max())
        assert err <= tol * max(s, 1e-6), (what, err, s)
    close(got_nchw, want, "y")
    close(rmd, rm, "running_mean", scale=1.0)
    close(rvd, rv, "running_var", scale=1.0)
    close(xd.grad, xr.grad, "dx")
    close(wd.grad, wr.grad, "dw")
    close(gd.grad, gr.grad, "dgamma")
    close(bd.grad, br.grad, "dbeta")
