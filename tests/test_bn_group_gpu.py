"""fs_bn_group_fwd / fs_bn_group_bwd (bn_col.hip): train-mode BatchNorm (+ReLU) of a small map in one launch, optionally over
several independently normalised groups of the batch, optionally summing the producing convolution's split-K slabs.

Reference: torch.nn.functional.batch_norm(training=True) + relu and their autograd on the CPU in fp32, applied per group in
order (= what the reference does when it evaluates one module on two inputs one after the other, model_search.py:322-329):
outputs, saved statistics, running statistics after the sequential updates, input gradient, dgamma / dbeta summed over the
groups.  fp32: 1e-5-level agreement; bf16 storage: 2e-2 of max|ref|.  Determinism: two runs are bit-identical."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, C, H, W, groups, relu
    (3, 96, 32, 64, 1, 1),        # supernet 1/8 scale at C3: 6144 pixels (beyond the register cache)
    (3, 192, 16, 32, 1, 1),
    (3, 384, 4, 8, 1, 0),         # 96 pixels: most lanes idle
    (6, 64, 16, 32, 2, 1),        # two groups of 3 images
    (4, 32, 7, 14, 2, 0),
    (4, 48, 14, 28, 4, 1),
    (3, 192, 8, 16, 1, 1),        # 384 px: register-resident kernel, both unroll slots live
    (6, 96, 8, 16, 2, 1),         # ... two groups of 384 px
    (2, 64, 8, 16, 1, 0),         # exactly 256 px
    (3, 40, 5, 7, 1, 1),          # 105 px: fewer pixels than lanes
    (2, 24, 16, 32, 2, 1),        # two groups of 512 px: the largest register-resident case
]


def _run(case, dtype, splits=0, seed=0, unit=False):
    from fasterseg_amd import kernels as K
    from fasterseg_amd._lib import call
    N, C, H, W, G, relu = case
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3).to(dtype).float()
    dy = torch.randn(N, C, H, W, generator=g).to(dtype).float()
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    pixels = N * H * W
    dev = "cuda"
    z = K.to_nhwc(x.to(dev), dtype)
    partials = None
    if splits:                                # random slabs that sum to x (in fp32); z starts as garbage and must be written
        parts = torch.randn(splits - 1, pixels, C, generator=g)
        xl = x.permute(0, 2, 3, 1).reshape(pixels, C)
        partials = torch.cat([parts, (xl - parts.sum(0))[None]], 0).contiguous().to(dev)
        z = K.empty_nhwc(N, C, H, W, dtype, dev)
        z.fill_(123.0)
    y = K.empty_nhwc(N, C, H, W, dtype, dev)
    saved = torch.empty(G * 4 * C, dtype=torch.float32, device=dev)
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    gd, bd = gamma.to(dev), beta.to(dev)
    if unit:          # fs_bn_act_train_fwd: chooses the one-launch kernel or the grouped grid-wide passes by the map size
        stats = torch.zeros(G * 2 * C, dtype=torch.float32, device=dev)
        call("fs_bn_act_train_fwd", K._stream(), pixels, C, G, K._p(z), K.channel_stride(z), K._p(gd), K._p(bd), 1e-5, 0.1,
             K._p(rm_d), K._p(rv_d), K._p(nbt), K._p(stats), K._p(saved), K._p(y), K.channel_stride(y), K.dtype_code(dtype), relu,
             *K.stream_workspace(dev))
    else:
        call("fs_bn_group_fwd", K._stream(), pixels, C, G, K._p(z), K.channel_stride(z), K._p(partials), splits, K._p(gd), K._p(bd), 1e-5, 0.1,
             K._p(rm_d), K._p(rv_d), K._p(nbt), K._p(saved), K._p(y), K.channel_stride(y), K.dtype_code(dtype), relu)
    dyg = K.to_nhwc(dy.to(dev), dtype)
    dz = K.empty_nhwc(N, C, H, W, dtype, dev)
    red = torch.empty(2 * C, dtype=torch.float32, device=dev)
    dgacc, dbacc = torch.full((C,), 2.0, device=dev), torch.full((C,), -1.0, device=dev)
    if unit:
        red = torch.zeros((G + 1 if G > 1 else 1) * 2 * C, dtype=torch.float32, device=dev)
        call("fs_bn_act_train_bwd", K._stream(), pixels, C, G, K._p(z), K.channel_stride(z), K._p(dyg), K.channel_stride(dyg), K._p(y),
             K.channel_stride(y), K._p(saved), K._p(gd), K._p(red), K.dtype_code(dtype), relu, K._p(dz), K.channel_stride(dz), K._p(dgacc),
             K._p(dbacc), *K.stream_workspace(dev))
    else:
        call("fs_bn_group_bwd", K._stream(), pixels, C, G, K._p(z), K.channel_stride(z), K._p(dyg), K.channel_stride(dyg), K._p(y),
             K.channel_stride(y), K._p(saved), K._p(gd), K.dtype_code(dtype), relu, K._p(dz), K.channel_stride(dz), K._p(red), K._p(dgacc),
             K._p(dbacc))
    torch.cuda.synchronize()
    red = red[:2 * C]
    got = dict(y=y.float().cpu(), z=z.float().cpu(), dz=dz.float().cpu(), red=red.cpu(), saved=saved.cpu().view(G, 4, C), rm=rm_d.cpu(),
               rv=rv_d.cpu(), nbt=int(nbt), dgacc=dgacc.cpu(), dbacc=dbacc.cpu())
    # ---- reference, group after group
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    outs, means, invstds = [], [], []
    ng = N // G
    for k in range(G):
        xs = xr[k * ng:(k + 1) * ng]
        o = F.batch_norm(xs, rm_r, rv_r, gr, br, True, 0.1, 1e-5)
        outs.append(torch.relu(o) if relu else o)
        m = xs.detach().mean((0, 2, 3))
        means.append(m)
        invstds.append(1.0 / torch.sqrt(xs.detach().var((0, 2, 3), unbiased=False) + 1e-5))
    yr = torch.cat(outs, 0)
    yr.backward(dy)
    want = dict(y=yr.detach(), dz=xr.grad, dgamma=gr.grad, dbeta=br.grad, rm=rm_r, rv=rv_r, mean=torch.stack(means), invstd=torch.stack(invstds))
    return got, want, x


def _close(a, b, dtype, what, scale=None):
    err = float((a - b).abs().max())
    ref = float(b.abs().max()) if scale is None else scale
    tol = (2e-5 + 2e-5 * ref) if dtype == torch.float32 else 2e-2 * max(ref, 1e-3)
    assert err <= tol, "%s: max err %.3e (max|ref| %.3e)" % (what, err, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=["%dx%dx%dx%d-g%d-r%d" % c for c in CASES])
def test_bn_group_matches_torch_per_group(case, dtype):
    got, want, x = _run(case, dtype)
    N, C, H, W, G, relu = case
    _close(got["y"], want["y"], dtype, "y")
    _close(got["dz"], want["dz"], dtype, "dz")
    _close(got["red"][C:], want["dgamma"], dtype, "dgamma")
    _close(got["red"][:C], want["dbeta"], dtype, "dbeta")
    _close(got["dgacc"] - 2.0, want["dgamma"], dtype, "dgamma accumulated")
    _close(got["dbacc"] + 1.0, want["dbeta"], dtype, "dbeta accumulated")
    _close(got["saved"][:, 0], want["mean"], torch.float32, "saved mean")
    _close(got["saved"][:, 1], want["invstd"], torch.float32, "saved invstd", scale=float(want["invstd"].abs().max()))
    _close(got["rm"], want["rm"], torch.float32, "running_mean after %d sequential updates" % G)
    _close(got["rv"], want["rv"], torch.float32, "running_var")
    assert got["nbt"] == G


UNIT_CASES = [
    (6, 64, 16, 32, 2, 1),        # 1536 px per group: the grid-wide grouped passes (blockIdx.y = group)
    (4, 96, 32, 64, 2, 1),        # 4096 px per group
    (6, 40, 24, 40, 3, 0),
    (2, 128, 32, 64, 1, 1),       # one group, grid-wide
    (4, 64, 8, 16, 2, 1),         # 256 px per group: routed to the one-launch kernel
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", UNIT_CASES, ids=["%dx%dx%dx%d-g%d-r%d" % c for c in UNIT_CASES])
def test_bn_unit_matches_torch_per_group(case, dtype):
    """fs_bn_act_train_fwd / _bwd across the size threshold (atomics in the grid-wide passes: fp32 bar 1e-4-level)."""
    got, want, x = _run(case, dtype, unit=True)
    N, C, H, W, G, relu = case
    t32 = torch.float32
    _close(got["y"], want["y"], dtype, "y")
    _close(got["dz"], want["dz"], dtype, "dz")
    _close(got["red"][C:], want["dgamma"], dtype, "dgamma", scale=5 * float(want["dgamma"].abs().max()))
    _close(got["red"][:C], want["dbeta"], dtype, "dbeta", scale=5 * float(want["dbeta"].abs().max()))
    _close(got["dgacc"] - 2.0, want["dgamma"], dtype, "dgamma accumulated", scale=5 * float(want["dgamma"].abs().max()))
    _close(got["dbacc"] + 1.0, want["dbeta"], dtype, "dbeta accumulated", scale=5 * float(want["dbeta"].abs().max()))
    _close(got["saved"][:, 0], want["mean"], t32, "saved mean", scale=5.0)
    _close(got["saved"][:, 1], want["invstd"], t32, "saved invstd", scale=5 * float(want["invstd"].abs().max()))
    _close(got["rm"], want["rm"], t32, "running_mean after %d sequential updates" % G, scale=5.0)
    _close(got["rv"], want["rv"], t32, "running_var", scale=5.0)
    assert got["nbt"] == G


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_bn_group_sums_split_k_slabs(dtype):
    case = (4, 64, 8, 16, 2, 1)
    got, want, x = _run(case, dtype, splits=3)
    _close(got["z"], x.to(dtype).float(), dtype, "z written from the slabs")
    _close(got["y"], want["y"], dtype, "y")
    _close(got["dz"], want["dz"], dtype, "dz")


def test_bn_group_is_bit_reproducible():
    a, _, _ = _run((6, 96, 16, 32, 2, 1), torch.float32, seed=4)
    b, _, _ = _run((6, 96, 16, 32, 2, 1), torch.float32, seed=4)
    for k in ("y", "dz", "red", "saved", "rm", "rv"):
        assert torch.equal(a[k], b[k]), k
