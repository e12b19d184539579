"""Kernel-level parity on a real MI355X: every C-ABI entry point against a plain PyTorch fp32 CPU reference of the
same op (F.conv2d / F.batch_norm / F.interpolate(align_corners=True) and their autograd), fp32 and bf16.

Tolerances: fp32 kernels use exact-fp32 arithmetic - the fp32 MFMA (v_mfma_f32_32x32x2_f32) or, round 6, eight bf16 MFMAs on three-way
split operands (fs_set_fp32_split; partial products down to 2^-24 relative) - so only summation order differs from the CPU:
|err| <= 1e-4 + 1e-4*|ref| (north_star asks 1e-3 on logits).  bf16 kernels are checked against the fp32 reference
evaluated on bf16-rounded operands with |err| <= 2e-2*max|ref| (storage rounding of the output, 2^-8 relative)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def K():
    from fasterseg_amd import kernels
    return kernels


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dtype):
    """round to the storage dtype and back (identity for fp32)"""
    return t.to(dtype).to(torch.float32)


def check(got, want, dtype, what=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs()
    if dtype == torch.float32:
        tol = 1e-4 + 1e-4 * want.abs()
    else:
        tol = 2e-2 * want.abs().max().clamp_min(1e-3) + 0 * want
    bad = err > tol
    assert not bad.any(), "%s: max err %.3e (max|ref| %.3e), %d bad of %d" % (
        what, float(err.max()), float(want.abs().max()), int(bad.sum()), bad.numel())


CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad
    (1, 32, 16, 24, 32, 3, 1, 1),
    (2, 64, 9, 13, 96, 3, 2, 1),
    (1, 128, 8, 8, 19, 1, 1, 0),
    (2, 48, 12, 10, 80, 3, 1, 1),
    (1, 256, 4, 8, 256, 3, 1, 1),
    (2, 96, 8, 12, 48, 1, 2, 0),
    (2, 96, 8, 12, 48, 1, 2, -1),     # FactorizedReduce second branch: x[:, :, 1:, 1:] (operations.py:523)
    (1, 32, 64, 128, 64, 3, 1, 1),
    (1, 16, 40, 36, 144, 3, 1, 1),
    (3, 384, 4, 8, 384, 3, 1, 1),
    (1, 8, 5, 7, 8, 3, 1, 1),
    (1, 32, 192, 256, 128, 3, 1, 1),   # M = 49152: the large-tile configurations under the heuristic
    (2, 16, 128, 160, 32, 3, 1, 1),
]


def ref_conv(x, w, stride, pad):
    if pad < 0:
        return F.conv2d(x[:, :, -pad:, -pad:], w, None, stride, 0)
    return F.conv2d(x, w, None, stride, pad)


@pytest.fixture
def force_cfg(request):
    """Every tile configuration of fs_conv2d_fwd must give the same answer; -1 is the production heuristic."""
    from fasterseg_amd import _lib
    _lib.lib().fs_debug_force_conv_cfg(request.param)
    yield request.param
    _lib.lib().fs_debug_force_conv_cfg(-1)


# -1: production heuristic (conv_igemm2.hip where it qualifies); -2: conv_igemm.hip's heuristic alone; 0..7: its configurations;
# 100..106: conv_igemm2.hip's configurations; 1000 * s + 100 + c: with s K slices (fp32 slabs + reduce launch)
@pytest.mark.parametrize("force_cfg", [-1, -2, 0, 1, 2, 3, 4, 5, 6, 7, 100, 101, 102, 103, 104, 105, 106, 3100, 2104], indirect=True,
                         ids=lambda c: "cfg%d" % c)
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv2d_fwd(case, dtype, force_cfg):
    k = K()
    N, Cin, H, W, Cout, ks, stride, pad = case
    x = q(rnd(N, Cin, H, W, seed=1), dtype)
    w = q(rnd(Cout, Cin, ks, ks, seed=2, scale=(2.0 / (Cin * ks * ks)) ** 0.5), dtype)
    scale = rnd(Cout, seed=3).abs() + 0.5
    shift = rnd(Cout, seed=4)
    ref_raw = ref_conv(x, w, stride, pad)
    ref = F.relu(ref_raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    xd = k.to_nhwc(x.cuda(), dtype)
    wp = k.pack_weight(w.cuda(), dtype)
    stats = torch.zeros(2 * Cout, device="cuda")
    ohw = tuple(ref_raw.shape[2:])
    y = k.conv2d(xd, wp, Cout, ks, ks, stride, pad, scale.cuda(), shift.cuda(), relu=True, stats=stats, out_hw=ohw)
    check(y, ref, dtype, "conv+affine+relu")
    # raw conv + BN statistics epilogue
    y2 = k.conv2d(xd, wp, Cout, ks, ks, stride, pad, out_hw=ohw)
    check(y2, ref_raw, dtype, "raw conv")
    cnt = ref_raw.numel() / Cout
    s1 = ref_raw.sum((0, 2, 3))
    s2 = (ref_raw * ref_raw).sum((0, 2, 3))
    assert torch.allclose(stats[:Cout].cpu(), s1, atol=2e-3 * cnt ** 0.5 + 1e-3, rtol=2e-3), "sum"
    assert torch.allclose(stats[Cout:].cpu(), s2, atol=1e-3, rtol=3e-3), "sumsq"


HALO_CASES = [(1, 32, 16, 32, 32), (2, 64, 9, 13, 96), (1, 48, 24, 40, 80), (1, 128, 8, 16, 128), (1, 16, 33, 50, 19),
              (2, 96, 17, 16, 64), (1, 192, 16, 16, 192), (1, 8, 5, 7, 8), (1, 64, 64, 128, 64)]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", HALO_CASES, ids=[str(c) for c in HALO_CASES])
def test_conv3x3_halo(case, dtype):
    """LDS-halo 3x3 kernel == F.conv2d(k3,s1,p1) incl. ragged tiles, channel tails, BN-stat epilogue and slices."""
    k = K()
    N, Cin, H, W, Cout = case
    x = q(rnd(N, Cin, H, W, seed=1), dtype)
    w = q(rnd(Cout, Cin, 3, 3, seed=2, scale=(2.0 / (Cin * 9)) ** 0.5), dtype)
    scale, shift = rnd(Cout, seed=3).abs() + 0.5, rnd(Cout, seed=4)
    raw = F.conv2d(x, w, None, 1, 1)
    ref = F.relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    xd = k.to_nhwc(x.cuda(), dtype)
    wf = k.pack_weight_frag(w.cuda(), dtype)
    stats = torch.zeros(2 * Cout, device="cuda")
    y = k.conv3x3_halo(xd, wf, Cout, scale.cuda(), shift.cuda(), relu=True, stats=stats)
    check(y, ref, dtype, "halo conv+affine+relu")
    cnt = raw.numel() / Cout
    assert torch.allclose(stats[:Cout].cpu(), raw.sum((0, 2, 3)), atol=2e-3 * cnt ** 0.5 + 1e-3, rtol=2e-3)
    assert torch.allclose(stats[Cout:].cpu(), (raw * raw).sum((0, 2, 3)), atol=1e-3, rtol=3e-3)
    if Cout % 8 == 0:
        wide = k.empty_nhwc(N, Cout + 32, H, W, dtype, "cuda", zero=True)
        k.conv3x3_halo(xd, wf, Cout, out=wide[:, 32:])
        check(wide[:, 32:], raw, dtype, "halo into slice")
        assert float(wide[:, :32].abs().max()) == 0.0


HALO_S2_CASES = [
    # N, Cin, H, W, Cout
    (1, 32, 64, 96, 64),          # stem.1 conv1 of the student (32 -> 64, stride 2), reduced map
    (2, 64, 40, 72, 64),
    (1, 32, 33, 47, 32),          # odd input size: ragged output tiles, last tap on the zero padding
    (1, 24, 18, 34, 48),          # channel tails
    (1, 96, 16, 32, 160),         # 128-wide tile + tail
    (1, 8, 6, 10, 8),             # smaller than one tile
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("tile", [0, 32, 64, 128])
@pytest.mark.parametrize("case", HALO_S2_CASES, ids=["%dx%dx%dx%d-%d" % c for c in HALO_S2_CASES])
def test_conv3x3_halo_stride2(case, dtype, tile):
    """The stride-2 form of the LDS-halo kernel (even / odd input columns de-interleaved in LDS) == F.conv2d(k3, s2, p1) for
    every output-channel tile, incl. odd sizes and the BN-statistics epilogue (operations.py:298-306 at stride 2)."""
    k = K()
    N, Cin, H, W, Cout = case
    x = q(rnd(N, Cin, H, W, seed=11), dtype)
    w = q(rnd(Cout, Cin, 3, 3, seed=12, scale=(2.0 / (Cin * 9)) ** 0.5), dtype)
    scale, shift = rnd(Cout, seed=13).abs() + 0.5, rnd(Cout, seed=14)
    raw = F.conv2d(x, w, None, 2, 1)
    ref = F.relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    xd = k.to_nhwc(x.cuda(), dtype)
    wf = k.pack_weight_frag(w.cuda(), dtype)
    stats = torch.zeros(2 * Cout, device="cuda")
    y = k.conv3x3_halo(xd, wf, Cout, scale.cuda(), shift.cuda(), relu=True, stats=stats, stride=2, tile=tile)
    assert tuple(y.shape) == tuple(ref.shape)
    check(y, ref, dtype, "stride-2 halo conv+affine+relu (tile %d)" % tile)
    cnt = raw.numel() / Cout
    assert torch.allclose(stats[:Cout].cpu(), raw.sum((0, 2, 3)), atol=2e-3 * cnt ** 0.5 + 1e-3, rtol=2e-3)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("tile", [32, 64, 128])
def test_conv3x3_halo_forced_tiles_stride1(dtype, tile):
    k = K()
    x = q(rnd(1, 64, 24, 40, seed=21), dtype)
    w = q(rnd(96, 64, 3, 3, seed=22, scale=0.06), dtype)
    ref = F.conv2d(x, w, None, 1, 1)
    y = k.conv3x3_halo(k.to_nhwc(x.cuda(), dtype), k.pack_weight_frag(w.cuda(), dtype), 96, tile=tile)
    check(y, ref, dtype, "halo tile %d" % tile)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_conv2d_into_channel_slice(dtype):
    """torch.cat fused away: the conv writes into a channel slice of a wider buffer and reads from one."""
    k = K()
    x = q(rnd(1, 64, 10, 12, seed=5), dtype)
    w = q(rnd(32, 32, 3, 3, seed=6, scale=0.1), dtype)
    wide = k.to_nhwc(x.cuda(), dtype)
    out = k.empty_nhwc(1, 96, 10, 12, dtype, "cuda", zero=True)
    k.conv2d(wide[:, 32:64], k.pack_weight(w.cuda(), dtype), 32, 3, 3, 1, 1, out=out[:, 64:96])
    ref = F.conv2d(x[:, 32:64], w, None, 1, 1)
    check(out[:, 64:96], ref, dtype, "slice out")
    assert float(out[:, :64].abs().max()) == 0.0


DGRAD_CASES = [(1, 32, 12, 16, 64, 3, 1, 1), (2, 32, 12, 16, 64, 3, 2, 1), (1, 64, 8, 12, 32, 1, 2, 0),
               (1, 64, 8, 12, 32, 1, 2, -1), (2, 48, 7, 14, 48, 3, 1, 1), (1, 64, 6, 10, 19, 1, 1, 0)]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", DGRAD_CASES, ids=[str(c) for c in DGRAD_CASES])
def test_conv2d_dgrad_and_wgrad(case, dtype):
    k = K()
    N, Cin, H, W, Cout, ks, stride, pad = case
    x = q(rnd(N, Cin, H, W, seed=1), dtype).requires_grad_(True)
    w = q(rnd(Cout, Cin, ks, ks, seed=2, scale=0.2), dtype).requires_grad_(True)
    y = ref_conv(x, w, stride, pad)
    dy = q(rnd(*y.shape, seed=3), dtype)
    y.backward(dy)
    Cp = Cout if Cout % 8 == 0 else 32          # classifier: channel-padded gradient buffer
    dyd = k.empty_nhwc(N, Cp, y.shape[2], y.shape[3], dtype, "cuda", zero=True)
    dyd[:, :Cout].copy_(dy.cuda().to(dtype))
    xd = k.to_nhwc(x.detach().cuda(), dtype)
    # data gradient = conv of dy with the flipped/transposed filter (zero-insertion for stride 2)
    wf = k.pack_weight(w.detach().cuda(), dtype, flip=True, rows=None)
    if Cp != Cout:
        wf_p = torch.zeros((Cin, ks, ks, Cp), dtype=dtype, device="cuda")
        wf_p[..., :Cout] = wf
        wf = wf_p
    dx = k.conv2d(dyd, wf, Cin, ks, ks, 1, ks - 1 - pad, transposed=(stride == 2), out_hw=(H, W))
    check(dx, x.grad, dtype, "dgrad")
    dw = k.conv2d_wgrad(xd, dyd, ks, ks, stride, pad)
    gw = torch.zeros(Cp, Cin, ks, ks, device="cuda")
    k.unpack_weight_grad(dw, gw, Cp, Cin)
    ref = w.grad
    got = gw[:Cout].cpu()
    tol = 2e-4 * ref.abs().max() + 1e-4 if dtype == torch.float32 else 2e-2 * ref.abs().max()
    assert float((got - ref).abs().max()) <= float(tol), "wgrad max err %.3e vs max|ref| %.3e" % (
        float((got - ref).abs().max()), float(ref.abs().max()))
    if Cp != Cout:
        assert float(gw[Cout:].abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [-1, 100, 101, 103, 105, 106], ids=lambda c: "cfg%d" % c)
@pytest.mark.parametrize("case", [(2, 96, 16, 32, 96, 3, 1, 1), (1, 48, 24, 40, 80, 3, 2, 1), (3, 384, 4, 8, 384, 3, 1, 1), (2, 64, 8, 12, 32, 1, 1, 0)],
                         ids=str)
def test_fp32_on_bf16_matrix_cores_matches_the_fp32_mfma(case, cfg):
    """Round 6 (fs_set_fp32_split): the fp32 convolution and weight gradient contracted with 8 bf16 MFMAs on three-way split operands
    against the same kernels on the fp32 MFMA, and both against an fp64 reference: the split form must be as close to fp64 as the native
    one (same order of magnitude of error: only l.l terms of 2^-32 are dropped), for every tile configuration that has a split form."""
    from fasterseg_amd import _lib
    k = K()
    lib = _lib.lib()
    N, Cin, H, W, Cout, ks, stride, pad = case
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, ks, ks, seed=2, scale=(2.0 / (Cin * ks * ks)) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    dy = rnd(*ref.shape, seed=3)
    wd = w.double().requires_grad_(True)
    F.conv2d(x.double(), wd, None, stride, pad).backward(dy.double())
    xd = k.to_nhwc(x.cuda(), torch.float32)
    dyd = k.to_nhwc(dy.cuda(), torch.float32)
    wp = k.pack_weight(w.cuda(), torch.float32)
    out = {}
    try:
        lib.fs_debug_force_conv_cfg(cfg)
        for split in (0, 1):
            lib.fs_set_fp32_split(split)
            assert lib.fs_get_fp32_split() == split
            y = k.conv2d(xd, wp, Cout, ks, ks, stride, pad, out_hw=tuple(ref.shape[2:]))
            dw = k.conv2d_wgrad(xd, dyd, ks, ks, stride, pad)
            gw = torch.zeros(Cout, Cin, ks, ks, device="cuda")
            k.unpack_weight_grad(dw, gw, Cout, Cin)
            out[split] = (k.to_nchw(y).double().cpu(), gw.double().cpu())
    finally:
        lib.fs_debug_force_conv_cfg(-1)
        lib.fs_set_fp32_split(1)
    for which, want in ((0, ref), (1, wd.grad)):
        scale = float(want.abs().max())
        e_native = float((out[0][which] - want).abs().max()) / scale
        e_split = float((out[1][which] - want).abs().max()) / scale
        assert e_native < 5e-6 and e_split < 5e-6, (which, e_native, e_split)
        assert e_split <= 4 * e_native + 2e-7, (which, e_native, e_split)          # the split form is as exact as the fp32 MFMA


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_factorized_reduce_pair_entry_points_and_time_op(dtype):
    """ABI 211: FactorizedReduce's two 1x1 stride-2 convolutions (search/operations.py:521-526: conv_1(x), conv_2(x[:, :, 1:, 1:])
    into the two channel halves of one map) and their two weight gradients as ONE grouped launch each, against F.conv2d + autograd; fs_time_op
    returns a positive per-launch time for a convolution geometry."""
    import ctypes
    from fasterseg_amd._lib import call
    k = K()
    N, Cin, H, W, half = 2, 64, 12, 16, 48
    x = q(rnd(N, Cin, H, W, seed=41), dtype).requires_grad_(True)
    w1 = q(rnd(half, Cin, 1, 1, seed=42, scale=0.2), dtype).requires_grad_(True)
    w2 = q(rnd(half, Cin, 1, 1, seed=43, scale=0.2), dtype).requires_grad_(True)
    ref = torch.cat([F.conv2d(x, w1, None, 2, 0), F.conv2d(x[:, :, 1:, 1:], w2, None, 2, 0)], dim=1)
    dy = q(rnd(*ref.shape, seed=44), dtype)
    ref.backward(dy)
    xd = k.to_nhwc(x.detach().cuda(), dtype)
    Ho, Wo = ref.shape[2], ref.shape[3]
    out = k.empty_nhwc(N, 2 * half, Ho, Wo, dtype, "cuda", zero=True)
    d1 = k.conv_desc(xd.shape, k.channel_stride(xd), half, 1, 1, 2, 0, 2 * half, dtype, 0, (Ho, Wo))
    d2 = k.conv_desc(xd.shape, k.channel_stride(xd), half, 1, 1, 2, -1, 2 * half, dtype, 0, (Ho, Wo))
    p1, p2 = k.pack_weight(w1.detach().cuda(), dtype), k.pack_weight(w2.detach().cuda(), dtype)
    es = out.element_size()
    st1 = torch.zeros(2 * half, device="cuda")
    st2 = torch.zeros(2 * half, device="cuda")
    call("fs_factorized_reduce_fwd", k._stream(), ctypes.byref(d1), k._p(xd), k._p(p1), out.data_ptr(), k._p(st1),
         ctypes.byref(d2), k._p(xd), k._p(p2), out.data_ptr() + half * es, k._p(st2))
    check(out, ref.detach(), dtype, "factorized reduce pair")
    got = k.to_nchw(out).float()
    for st, sl in ((st1, slice(0, half)), (st2, slice(half, 2 * half))):
        want = got[:, sl].sum(dim=(0, 2, 3))
        assert float((st[:half] - want).abs().max()) <= 2e-2 * float(want.abs().max()) + 1e-3
    # weight gradients of the pair: dy channel halves against x and the shifted x
    dyd = k.to_nhwc(dy.cuda(), dtype)
    g1 = k.conv_desc(xd.shape, k.channel_stride(xd), half, 1, 1, 2, 0, 2 * half, dtype, 0, (Ho, Wo))
    g2 = k.conv_desc(xd.shape, k.channel_stride(xd), half, 1, 1, 2, -1, 2 * half, dtype, 0, (Ho, Wo))
    dw1 = torch.zeros(half, 1, 1, Cin, device="cuda")
    dw2 = torch.zeros(half, 1, 1, Cin, device="cuda")
    call("fs_factorized_reduce_wgrad", k._stream(), ctypes.byref(g1), k._p(xd), dyd.data_ptr(), k._p(dw1),
         ctypes.byref(g2), k._p(xd), dyd.data_ptr() + half * es, k._p(dw2))
    for dw, w in ((dw1, w1), (dw2, w2)):
        want = w.grad.reshape(half, Cin)
        tol = (2e-4 if dtype == torch.float32 else 2e-2) * float(want.abs().max()) + 1e-4
        assert float((dw.reshape(half, Cin).cpu() - want).abs().max()) <= tol
    ms = ctypes.c_float(0.0)
    y = k.empty_nhwc(N, half, Ho, Wo, dtype, "cuda")
    d = k.conv_desc(xd.shape, k.channel_stride(xd), half, 1, 1, 2, 0, half, dtype, 0, (Ho, Wo))
    call("fs_time_op", k._stream(), ctypes.byref(d), k._p(xd), k._p(p1), k._p(y), 3, 20, ctypes.byref(ms))
    assert 1e-4 < ms.value < 5.0, ms.value                      # a 1x1 convolution on a 6 x 8 map: microseconds, not zero


S2_DGRAD_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad   (forward geometry; dx has the size of x)
    (2, 32, 12, 16, 64, 3, 2, 1),
    (1, 48, 9, 13, 32, 3, 2, 1),          # odd map: the four parity classes have different sizes
    (1, 16, 1, 7, 24, 3, 2, 1),           # one row: the odd-row classes are empty
    (2, 64, 8, 12, 32, 1, 2, 0),          # FactorizedReduce conv_1 (operations.py:521): only even pixels receive a gradient
    (2, 64, 8, 12, 32, 1, 2, -1),         # conv_2 on x[:, :, 1:, 1:] (:523): only odd pixels do
    (1, 64, 7, 9, 40, 1, 2, -1),
    (3, 96, 32, 64, 192, 3, 2, 1),        # a stride-2 MixedOp of the supernet at 1/8 scale, fused pair (2 x 96 output channels)
]


@pytest.mark.parametrize("force_cfg", [-1, -2, 100, 101, 102, 104, 105, 106], indirect=True, ids=lambda c: "cfg%d" % c)
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", S2_DGRAD_CASES, ids=[str(c) for c in S2_DGRAD_CASES])
def test_stride2_dgrad_by_parity_classes(case, dtype, force_cfg):
    """FS_CONV_TRANSPOSED through conv_igemm2.hip: the data gradient of a stride-2 convolution evaluated per output-parity class (only
    the taps that meet real pixels: 1 / 2 / 2 / 4 of 9) equals autograd's input gradient - and the zero-insertion form of
    conv_igemm.hip (cfg -2).  Every pixel of dx is written (classes without any tap get zeros)."""
    k = K()
    N, Cin, H, W, Cout, ks, stride, pad = case
    x = q(rnd(N, Cin, H, W, seed=11), dtype).requires_grad_(True)
    w = q(rnd(Cout, Cin, ks, ks, seed=12, scale=0.2), dtype)
    y = ref_conv(x, w, stride, pad)
    dy = q(rnd(*y.shape, seed=13), dtype)
    y.backward(dy)
    dyd = k.to_nhwc(dy.cuda(), dtype)
    wf = k.pack_weight(w.cuda(), dtype, flip=True)
    dx = k.empty_nhwc(N, Cin, H, W, dtype, "cuda")
    dx.fill_(float("nan"))
    k.conv2d(dyd, wf, Cin, ks, ks, 1, ks - 1 - pad, transposed=True, out_hw=(H, W), out=dx)
    assert bool(torch.isfinite(dx.float()).all()), "some pixel of dx was not written"
    check(dx, x.grad, dtype, "stride-2 dgrad")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(3, 96, 32, 64, 96, 3, 1, 1), (6, 384, 8, 16, 384, 3, 1, 1), (3, 192, 16, 32, 384, 3, 2, 1),
                                  (2, 64, 64, 128, 64, 3, 1, 1), (3, 96, 32, 64, 48, 1, 2, -1)], ids=str)
def test_wgrad_slab_reduction_is_deterministic_and_matches_atomics(case, dtype):
    """fs_conv2d_wgrad_ws in bit-reproducible mode (partial tiles in the workspace, summed in slab order by the last-arriving block) on
    the supernet's geometries: bit-identical from run to run, accumulates onto what the gradient tensor already holds, equals the fp32-atomics
    entry point up to summation order, and leaves the workspace's arrival counters zero."""
    import ctypes
    from fasterseg_amd import _lib
    k = K()
    N, Cin, H, W, Cout, ks, stride, pad = case
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    x = k.to_nhwc(rnd(N, Cin, H, W, seed=11).cuda(), dtype)
    dy = k.to_nhwc(rnd(N, Cout, Ho, Wo, seed=12).cuda(), dtype)
    d = k.conv_desc(x.shape, Cin, Cout, ks, ks, stride, pad, Cout, dtype, 0, (Ho, Wo))
    base = rnd(Cout, ks, ks, Cin, seed=13).cuda()
    ws = torch.empty(k.WORKSPACE_BYTES, dtype=torch.uint8, device="cuda")
    ws[:-k.WS_COUNTER_BYTES].fill_(0x7f)                      # garbage scratch, zero counters
    ws[-k.WS_COUNTER_BYTES:].zero_()
    st = k._stream()
    outs = []
    with k.deterministic():
        for _ in range(3):
            dw = base.clone()
            _lib.call("fs_conv2d_wgrad_ws", st, ctypes.byref(d), k._p(x), k._p(dy), k._p(dw), ks * ks * Cin, 1, Cin, k._p(ws), k.WORKSPACE_BYTES)
            outs.append(dw)
        torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "slab reduction is not bit-reproducible"
    assert int(ws[-k.WS_COUNTER_BYTES:].view(torch.int32).abs().sum()) == 0, "arrival counters were not left at zero"
    ref = base.clone()
    _lib.call("fs_conv2d_wgrad_strided", st, ctypes.byref(d), k._p(x), k._p(dy), k._p(ref), ks * ks * Cin, 1, Cin)
    torch.cuda.synchronize()
    scale = float((ref - base).abs().max())
    assert float((outs[0] - ref).abs().max()) <= 2e-5 * scale + 1e-5, (float((outs[0] - ref).abs().max()), scale)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("cout", [32, 48, 24])
@pytest.mark.parametrize("hw", [(34, 52), (33, 50), (26, 264), (17, 136)], ids=lambda v: "%dx%d" % v)
def test_stem_conv(cout, dtype, hw):
    """(34,52)/(26,264)/(17,136): W % 4 == 0 -> LDS-tiled kernel (partial tiles, several tiles per row, odd H);
    (33,50): gather kernel."""
    k = K()
    x = rnd(2, 3, hw[0], hw[1], seed=7)
    w = rnd(cout, 3, 3, 3, seed=8, scale=0.3)
    scale, shift = rnd(cout, seed=9).abs() + 0.5, rnd(cout, seed=10)
    ref = F.relu(F.conv2d(x, w, None, 2, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    wp = k.pack_weight(w.cuda(), torch.float32)
    y = k.conv_stem(x.cuda(), wp, cout, scale.cuda(), shift.cuda(), True, dtype)
    check(y, q(ref, dtype) if dtype != torch.float32 else ref, dtype, "stem")


@pytest.mark.parametrize("cout", [32, 48, 64, 8])
@pytest.mark.parametrize("hw", [(34, 52), (26, 264), (17, 136), (128, 256)])
def test_stem_mfma_form_matches_fp32_direct_form(cout, hw):
    """The matrix-core stem (split-bf16 operands: x_hi*w_hi + x_lo*w_hi + x_hi*w_lo) against the fp32 vector-ALU stem and
    against F.conv2d: the only difference allowed is the final bf16 rounding of the output (one ulp = 2^-8 relative)."""
    from fasterseg_amd import _lib
    k = K()
    x = rnd(2, 3, hw[0], hw[1], seed=17) * 2.0
    w = rnd(cout, 3, 3, 3, seed=18, scale=0.3)
    scale, shift = rnd(cout, seed=19).abs() + 0.5, rnd(cout, seed=20)
    ref = F.relu(F.conv2d(x, w, None, 2, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    wp = k.pack_weight(w.cuda(), torch.float32)
    lib = _lib.lib()
    try:
        lib.fs_debug_stem_mfma(1)
        y_mfma = k.conv_stem(x.cuda(), wp, cout, scale.cuda(), shift.cuda(), True, torch.bfloat16).float().cpu()
        lib.fs_debug_stem_mfma(0)
        y_valu = k.conv_stem(x.cuda(), wp, cout, scale.cuda(), shift.cuda(), True, torch.bfloat16).float().cpu()
    finally:
        lib.fs_debug_stem_mfma(1)
    # both are the bf16 rounding of (nearly) the same fp32 value: the split drops x_lo*w_lo (2^-16 of sum|x*w| ~ 1e-4 here),
    # so they differ by at most that plus one bf16 ulp of the result, and only where the rounding boundary is crossed
    tol = 2e-4 + ref.abs() * 2.0 ** -7
    assert ((y_mfma - y_valu).abs() <= tol).all(), float((y_mfma - y_valu).abs().max())
    assert float(((y_mfma - y_valu).abs() > 0).float().mean()) < 0.05
    assert ((y_mfma - ref).abs() <= tol).all()


RESIZE_CASES = [((2, 32, 9, 12), (18, 24)), ((1, 64, 16, 32), (8, 16)), ((2, 16, 7, 14), (3, 7)), ((1, 32, 3, 7), (7, 14)),
                ((1, 8, 4, 8), (32, 64)), ((1, 24, 5, 5), (5, 5)), ((1, 8, 1, 6), (4, 12)), ((1, 8, 6, 6), (1, 1))]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("case", RESIZE_CASES, ids=[str(c) for c in RESIZE_CASES])
def test_bilinear_fwd_bwd(case, relu, dtype):
    k = K()
    shape, size = case
    x = q(rnd(*shape, seed=11), dtype).requires_grad_(True)
    y = F.interpolate(x, size=size, mode="bilinear", align_corners=True)
    yr = F.relu(y) if relu else y
    dy = q(rnd(*yr.shape, seed=12), dtype)
    yr.backward(dy)
    xd = k.to_nhwc(x.detach().cuda(), dtype)
    out = k.bilinear(xd, size, relu=relu)
    check(out, yr, dtype, "bilinear fwd")
    if dtype == torch.float32:
        # relu mask must come from the same values; reuse the kernel's own output
        dx = k.bilinear_bwd(k.to_nhwc(dy.cuda(), dtype), out, shape, relu, dtype)
        check(dx, x.grad, dtype, "bilinear bwd")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("mode", [1, 2])
def test_bilinear_logits_nchw(dtype, mode):
    k = K()
    x = q(rnd(2, 19, 8, 12, seed=13), dtype).requires_grad_(True)
    y = F.interpolate(x, size=(64, 96), mode="bilinear", align_corners=True)
    dy = rnd(*y.shape, seed=14)
    y.backward(dy)
    buf = k.empty_nhwc(2, 32, 8, 12, dtype, "cuda", zero=True)
    buf[:, :19].copy_(x.detach().cuda().to(dtype))
    out = k.bilinear(buf, (64, 96), out_nchw=mode, channels=19)
    assert out.is_contiguous() and out.shape == (2, 19, 64, 96)
    assert out.dtype == (torch.float32 if mode == 1 else dtype)
    check(out, y, dtype if mode == 2 else (dtype if dtype != torch.float32 else torch.float32), "logits upsample")
    if mode == 1:
        dx = k.bilinear_bwd(dy.cuda(), None, (2, 19, 8, 12), False, dtype, out_nchw=1, dx_cs=32)
        check(dx, x.grad, dtype, "logits upsample bwd")
    # odd width -> scalar writer
    out2 = k.bilinear(buf, (20, 30), out_nchw=1, channels=19)
    check(out2, F.interpolate(x, size=(20, 30), mode="bilinear", align_corners=True), dtype, "scalar nchw")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 32, 9, 13), (3, 96, 4, 8), (1, 48, 16, 32), (2, 384, 3, 7), (1, 64, 128, 128)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_train_fwd_bwd(shape, relu, dtype):
    k = K()
    N, C, H, W = shape
    z = q(rnd(*shape, seed=15) * 1.5 + 0.3, dtype).requires_grad_(True)
    gamma = (rnd(C, seed=16).abs() + 0.5).requires_grad_(True)
    beta = rnd(C, seed=17).requires_grad_(True)
    rm, rv = rnd(C, seed=18) * 0.1, rnd(C, seed=19).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(z, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    yr = F.relu(y) if relu else y
    dy = q(rnd(*shape, seed=20), dtype)
    yr.backward(dy)
    zd = k.to_nhwc(z.detach().cuda(), dtype)
    stats = k.channel_stats(zd)
    rm_d, rv_d = rm.cuda(), rv.cuda()
    mean, invstd, scale, shift = k.bn_finalize(stats, N * H * W, gamma.detach().cuda(), beta.detach().cuda(), 1e-5, 0.1, rm_d, rv_d)
    out = k.affine_act(zd, scale, shift, relu)
    check(out, yr, dtype, "bn fwd")
    assert torch.allclose(rm_d.cpu(), rm_ref, atol=1e-4, rtol=1e-4)
    assert torch.allclose(rv_d.cpu(), rv_ref, atol=1e-4, rtol=1e-3)
    # the one-launch form (finalize folded into the normalise pass) gives the same output, saved statistics and buffers
    rm_f, rv_f, nbt = rm.cuda(), rv.cuda(), torch.tensor(3, dtype=torch.long, device="cuda")
    out_f, saved = k.bn_train_apply(zd, stats, gamma.detach().cuda(), beta.detach().cuda(), 1e-5, 0.1, rm_f, rv_f, nbt, relu)
    assert int(nbt) == 4
    tol = 1e-5 if dtype == torch.float32 else 1e-2          # same math, different association of the shift term
    assert torch.allclose(out_f.float(), out.float(), atol=tol, rtol=tol)
    assert torch.allclose(saved, torch.cat([mean, invstd, scale, shift]), atol=1e-6, rtol=1e-5)
    assert torch.allclose(rm_f, rm_d, atol=1e-6, rtol=1e-6) and torch.allclose(rv_f, rv_d, atol=1e-6, rtol=1e-6)
    if dtype == torch.float32:
        dz, dg, db = k.bn_backward(zd, k.to_nhwc(dy.cuda(), dtype), out, mean, invstd, gamma.detach().cuda(), relu)
        check(dz, z.grad, dtype, "bn dz")
        assert torch.allclose(dg.cpu(), gamma.grad, atol=2e-3, rtol=1e-3), float((dg.cpu() - gamma.grad).abs().max())
        assert torch.allclose(db.cpu(), beta.grad, atol=2e-3, rtol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_layout_copy_axpy_dot(dtype):
    k = K()
    x = q(rnd(2, 40, 7, 9, seed=21), dtype)
    y = q(rnd(2, 40, 7, 9, seed=22), dtype)
    xd, yd = k.to_nhwc(x.cuda(), dtype), k.to_nhwc(y.cuda(), dtype)
    assert torch.equal(k.to_nchw(xd).cpu(), x)
    assert torch.equal(xd.float().cpu(), x)                       # logical NCHW view agrees
    cat = k.cat_channels([xd, yd])
    assert torch.equal(cat.float().cpu(), torch.cat([x, y], 1))
    alpha = torch.tensor([0.37], device="cuda")
    acc = k.axpy(xd, alpha, k.empty_nhwc(2, 40, 7, 9, dtype, "cuda"), False)
    check(acc, 0.37 * x, dtype, "axpy")
    k.axpy(yd, alpha, acc, True)
    check(acc, q(0.37 * x, dtype) + 0.37 * y, dtype, "axpy acc")
    d = k.dot(xd, yd)
    assert abs(float(d) - float((x * y).sum())) < 1e-2 * float((x * y).abs().sum()) ** 0.5 + 1e-2


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("n", [1, 2, 5, 8])
def test_weighted_sum_fwd_bwd_dots(dtype, n):
    """fs_weighted_sum{,_bwd,_dots} against sum_k c_k x_k and its autograd (MixedOp mixing, model_search.py:76-78)."""
    k = K()
    shape = (2, 48, 5, 9)
    xs = [q(rnd(*shape, seed=40 + i), dtype).requires_grad_(True) for i in range(n)]
    coef = (rnd(n, seed=39) * 0.5).requires_grad_(True)
    ref = sum(c * x for c, x in zip(coef, xs))
    dy = q(rnd(*shape, seed=38), dtype)
    ref.backward(dy)
    xd = [k.to_nhwc(x.detach().cuda(), dtype) for x in xs]
    cd = coef.detach().cuda()
    out = k.weighted_sum(xd, cd)
    check(out, ref, dtype, "weighted_sum")
    dyd = k.to_nhwc(dy.cuda(), dtype)
    need = [i % 3 != 1 for i in range(n)]
    gxs = k.weighted_sum_bwd(dyd, cd, need)
    for i in range(n):
        if need[i]:
            check(gxs[i], xs[i].grad, dtype, "weighted_sum dx%d" % i)
        else:
            assert gxs[i] is None
    dots = k.weighted_sum_dots(dyd, xd).cpu()
    scale = float(dy.pow(2).sum().sqrt()) * max(float(x.detach().pow(2).sum().sqrt()) for x in xs)
    assert float((dots - coef.grad).abs().max()) < 1e-4 * scale + 1e-3


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("cfg", [-1, -2, 3, 5, 6, 7, 100, 101, 104, 2102])
def test_conv_two_segment_filter_bank_and_contraction(dtype, cfg):
    """fs_conv_desc.n_seg / n_jump (output channels >= n_seg read filter rows n_jump further: two filter banks as one GEMM) and
    k_seg / k_jump (input channels >= k_seg of every tap are k_jump elements further: the data gradient of the fused pair over both
    rotated packs) against the same convolutions on densely concatenated packs - the horizontal fusion of a MixedOp's
    'conv' + 'conv_2x'.conv1 (fusion.py), on the widths of a wide supernet cell."""
    import ctypes
    from fasterseg_amd import _lib
    k = K()
    O, I, cout, cin, N, H, W = 192, 160, 144, 128, 2, 12, 16
    wa, wb = q(rnd(O, I, 3, 3, seed=70) * 0.1, dtype).cuda(), q(rnd(O, I, 3, 3, seed=71) * 0.1, dtype).cuda()
    x = k.to_nhwc(q(rnd(N, cin, H, W, seed=72), dtype).cuda(), dtype)
    es = 4 if dtype == torch.float32 else 2
    # forward: two full-size [O][3][3][I] banks back to back (what optim.FlatSGD lays out for a pair)
    banks = torch.cat([k.pack_weight(wa, dtype).reshape(-1), k.pack_weight(wb, dtype).reshape(-1)])
    dense = torch.cat([k.pack_weight(wa, dtype, cout, cin).reshape(-1), k.pack_weight(wb, dtype, cout, cin).reshape(-1)])
    _lib.lib().fs_debug_force_conv_cfg(cfg)
    try:
        ref = k.conv2d(x, dense.view(2 * cout, 3, 3, cin), 2 * cout, 3, 3, 1, 1)
        got = k.empty_nhwc(N, 2 * cout, H, W, dtype, "cuda")
        d = k.conv_desc(x.shape, cin, 2 * cout, 3, 3, 1, 1, 2 * cout, dtype)
        d.w_os, d.w_ts, d.n_seg, d.n_jump = 9 * I, I, cout, O - cout
        ws, wsb = k.stream_workspace("cuda")
        _lib.call("fs_conv2d_fwd_ws", k._stream(), ctypes.byref(d), k._p(x), k._p(banks), None, None, k._p(got), None, ws, wsb)
        assert torch.equal(ref, got), float((ref.float() - got.float()).abs().max())
        # data gradient: contraction over the 2 * cout channels of dz, rotated packs [I][3][3][O] back to back
        dz = k.to_nhwc(q(rnd(N, 2 * cout, H, W, seed=73), dtype).cuda(), dtype)
        flips = torch.cat([k.pack_weight(wa, dtype, flip=True).reshape(-1), k.pack_weight(wb, dtype, flip=True).reshape(-1)])
        fa, fb = k.pack_weight(wa, dtype, cout, cin, flip=True), k.pack_weight(wb, dtype, cout, cin, flip=True)      # [cin][3][3][cout]
        ref = k.conv2d(dz, torch.cat([fa, fb], dim=3).contiguous(), cin, 3, 3, 1, 1)
        got = k.empty_nhwc(N, cin, H, W, dtype, "cuda")
        g = k.conv_desc(dz.shape, 2 * cout, cin, 3, 3, 1, 1, cin, dtype)
        g.w_os, g.w_ts, g.k_seg, g.k_jump = 9 * O, O, cout, I * 9 * O - cout
        _lib.call("fs_conv2d_fwd_ws", k._stream(), ctypes.byref(g), k._p(dz), k._p(flips), None, None, k._p(got), None, ws, wsb)
        assert torch.equal(ref, got), float((ref.float() - got.float()).abs().max())
    finally:
        _lib.lib().fs_debug_force_conv_cfg(-1)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("cfg", [-1, -2, 0, 3, 5, 6, 7, 100, 103, 105, 106])
def test_conv_reads_filter_block_of_wider_pack(dtype, cfg):
    """fs_conv_desc.w_os / w_ts: the [:cout][..][:cin] block of a full-size packed bank read in place equals the densely
    re-packed slice (USConv2d widths, slimmable_ops.py:42), forward and flipped (data-gradient) banks."""
    from fasterseg_amd import _lib
    k = K()
    O, I, cout, cin = 96, 64, 48, 40
    w = q(rnd(O, I, 3, 3, seed=60) * 0.2, dtype).cuda()
    x = k.to_nhwc(q(rnd(2, cin, 13, 17, seed=61), dtype).cuda(), dtype)
    full = k.pack_weight(w, dtype)                              # [O][3][3][I]
    dense = k.pack_weight(w, dtype, cout, cin)
    _lib.lib().fs_debug_force_conv_cfg(cfg)
    try:
        ref = k.conv2d(x, dense, cout, 3, 3, 1, 1)
        got = k.conv2d(x, full, cout, 3, 3, 1, 1, w_strides=(9 * I, I))
        assert torch.equal(ref, got)
        dz = k.to_nhwc(q(rnd(2, cout, 13, 17, seed=62), dtype).cuda(), dtype)
        full_f = k.pack_weight(w, dtype, flip=True)             # [I][3][3][O]
        dense_f = k.pack_weight(w, dtype, cout, cin, flip=True)
        ref = k.conv2d(dz, dense_f, cin, 3, 3, 1, 1)
        got = k.conv2d(dz, full_f, cin, 3, 3, 1, 1, w_strides=(9 * O, O))
        assert torch.equal(ref, got)
    finally:
        _lib.lib().fs_debug_force_conv_cfg(-1)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 384, 384, 4, 8), (2, 192, 96, 7, 14), (1, 256, 64, 16, 32), (3, 96, 96, 16, 32)])
@pytest.mark.parametrize("relu", [False, True])
def test_conv_split_k_matches_single_pass(shape, relu, dtype):
    """fs_conv2d_fwd_ws: long contractions of small maps are split over K across blocks (partial tiles in the workspace +
    reduce/epilogue launch).  Same result as the single-pass kernel (up to fp32 summation order), same BN statistics, and
    both agree with the fp32 CPU convolution."""
    k = K()
    N, cin, cout, H, W = shape
    x = q(rnd(N, cin, H, W, seed=70), dtype)
    w = q(rnd(cout, cin, 3, 3, seed=71) * (2.0 / (9 * cin)) ** 0.5, dtype)
    scale, shift = (rnd(cout, seed=72).abs() + 0.5), rnd(cout, seed=73)
    ref = F.conv2d(x, w, padding=1)
    want = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    want = F.relu(want) if relu else want
    xd = k.to_nhwc(x.cuda(), dtype)
    wp = k.pack_weight(w.cuda(), dtype)
    outs, stats = [], []
    for workspace in (False, True):
        st = torch.zeros(2 * cout, device="cuda")
        outs.append(k.conv2d(xd, wp, cout, 3, 3, 1, 1, scale.cuda(), shift.cuda(), relu, stats=st, workspace=workspace))
        stats.append(st.cpu())
    check(outs[1], want, dtype, "split-K conv vs CPU")
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((outs[0].float() - outs[1].float()).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    cnt = N * H * W
    assert torch.allclose(stats[1][:cout] / cnt, ref.mean((0, 2, 3)), atol=2e-3, rtol=1e-3)
    assert torch.allclose(stats[1][cout:] / cnt, ref.square().mean((0, 2, 3)), atol=2e-3, rtol=2e-3)
    assert torch.allclose(stats[0], stats[1], atol=1e-2 * cnt ** 0.5, rtol=2e-3)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("cfg", [-1, 3, 4, 5, 6])
@pytest.mark.parametrize("case", [((2, 32, 18, 40), (9, 20), False, 3, 1), ((1, 64, 8, 12), (16, 24), True, 3, 1),
                                  ((2, 40, 7, 9), (14, 18), True, 1, 0), ((1, 32, 33, 17), (16, 8), False, 3, 1)],
                         ids=["down", "up_relu", "up_1x1", "down_odd"])
def test_conv_virtual_resize(case, cfg, dtype):
    """fs_conv_desc.vr_*: conv(F.interpolate(x, size, bilinear, align_corners=True) [-> ReLU]) with the resampling folded
    into the gather == the two-launch path (the resampled map is rounded to the storage dtype in both)."""
    from fasterseg_amd import _lib
    k = K()
    (N, C, Hs, Ws), (Hv, Wv), vrelu, ksz, pad = case
    cout = 48
    x = q(rnd(N, C, Hs, Ws, seed=80), dtype)
    w = q(rnd(cout, C, ksz, ksz, seed=81) * (2.0 / (ksz * ksz * C)) ** 0.5, dtype)
    scale, shift = rnd(cout, seed=82).abs() + 0.5, rnd(cout, seed=83)
    mid = F.interpolate(x, size=(Hv, Wv), mode="bilinear", align_corners=True)
    mid = q(F.relu(mid) if vrelu else mid, dtype)
    want = F.relu(F.conv2d(mid, w, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    xd = k.to_nhwc(x.cuda(), dtype)
    wp = k.pack_weight(w.cuda(), dtype)
    _lib.lib().fs_debug_force_conv_cfg(cfg)
    try:
        got = k.conv2d(xd, wp, cout, ksz, ksz, 1, pad, scale.cuda(), shift.cuda(), True, vres=(Hv, Wv, vrelu))
        two = k.conv2d(k.bilinear(xd, (Hv, Wv), relu=vrelu), wp, cout, ksz, ksz, 1, pad, scale.cuda(), shift.cuda(), True)
    finally:
        _lib.lib().fs_debug_force_conv_cfg(-1)
    assert got.shape == (N, cout, Hv, Wv)
    check(got, want, dtype, "virtual-resize conv vs CPU")
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((got.float() - two.float()).abs().max()) <= tol * max(1.0, float(want.abs().max()))


def test_bn_finalize_counter_and_fused_param_grad_accumulation():
    """num_batches_tracked is bumped by fs_bn_finalize; fs_bn_bwd_apply adds dgamma/dbeta into the given buffers."""
    k = K()
    N, C, H, W = 2, 32, 6, 10
    z = rnd(N, C, H, W, seed=50)
    zd = k.to_nhwc(z.cuda(), torch.float32)
    gamma, beta = (rnd(C, seed=51).abs() + 0.5).cuda(), rnd(C, seed=52).cuda()
    nbt = torch.tensor(7, dtype=torch.long, device="cuda")
    stats = k.channel_stats(zd)
    mean, invstd, scale, shift = k.bn_finalize(stats, N * H * W, gamma, beta, 1e-5, 0.1, None, None, nbt)
    assert int(nbt) == 8
    out = k.affine_act(zd, scale, shift, True)
    dy = k.to_nhwc(rnd(N, C, H, W, seed=53).cuda(), torch.float32)
    dz0, dg0, db0 = k.bn_backward(zd, dy, out, mean, invstd, gamma, True)
    gacc, bacc = torch.full((C,), 2.0, device="cuda"), torch.full((C,), -1.0, device="cuda")
    dz1, dg1, db1 = k.bn_backward(zd, dy, out, mean, invstd, gamma, True, gacc, bacc)
    assert torch.equal(dz0, dz1)
    assert torch.allclose(gacc, dg0 + 2.0, atol=1e-5) and torch.allclose(bacc, db0 - 1.0, atol=1e-5)


def test_errors_are_raised_not_fatal():
    from fasterseg_amd._lib import FasterSegHipError
    k = K()
    x = k.to_nhwc(torch.randn(1, 8, 4, 4).cuda(), torch.float32)
    w = k.pack_weight(torch.randn(8, 8, 5, 5).cuda(), torch.float32)
    with pytest.raises(FasterSegHipError):
        k.conv2d(x, w, 8, 5, 5, 1, 2)            # 5x5 filters are not part of the hot path
