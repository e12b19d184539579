"""fs_zoom_cell_fwd (zoom_cell.hip: a whole zoomed-conv cell in one launch) on a real MI355X.

Reference = plain PyTorch fp32 on the CPU of what BasicResidual_downup_2x.forward computes in eval mode
(/root/reference search/operations.py:435-446): F.interpolate(1/2, bilinear, align_corners=True) -> conv3x3 -> BN(folded
scale/shift) -> ReLU -> conv3x3 -> BN -> [F.interpolate(x2)] -> ReLU, and the stride-1 BasicResidual2x (:352-359) with
both resamples off.  Tolerances: fp32 (exact-fp32 MFMA) 2e-4 + 2e-4*|ref|; bf16 vs the fp32 reference on bf16-rounded
operands 3e-2*max|ref| (two storage roundings: the sampled input and the mid map).  The fused launch is also compared
with the same cell issued as separate launches (resize, conv, conv, resize)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def K():
    from fasterseg_amd import kernels
    return kernels


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def reference(x, w1, s1, b1, w2, s2, b2, down, up):
    H, W = x.shape[2], x.shape[3]
    o = F.interpolate(x, size=(H // 2, W // 2), mode="bilinear", align_corners=True) if down else x
    o = torch.relu(F.conv2d(o, w1, padding=1) * s1[None, :, None, None] + b1[None, :, None, None])
    o = F.conv2d(o, w2, padding=1) * s2[None, :, None, None] + b2[None, :, None, None]
    if up:
        o = F.interpolate(o, size=(H, W), mode="bilinear", align_corners=True)
    return torch.relu(o)


CASES = [
    # N, Cin, C, H, W, down, up          C = Cmid = Cout
    (1, 32, 32, 128, 256, 1, 1),         # student cells 2-0 / 3-1 at C2
    (2, 64, 64, 64, 128, 1, 1),          # 64->64->64 (x3 at C2)
    (1, 128, 64, 64, 128, 1, 1),
    (1, 64, 128, 64, 128, 1, 0),         # stride-2 zoomed cell: no up-sample
    (1, 64, 192, 32, 64, 1, 1),          # 6 n-tiles
    (1, 128, 256, 32, 64, 1, 1),         # 8 n-tiles
    (1, 192, 128, 32, 48, 1, 1),         # 6 input chunks
    (2, 24, 48, 20, 52, 1, 1),           # ragged tiles, channel tails (Cin % 32, C % 32)
    (1, 40, 96, 12, 28, 1, 0),           # 3 n-tiles, ragged
    (1, 16, 160, 8, 24, 1, 1),           # 5 n-tiles -> 6-tile kernel on a zero-filled bank
    (1, 64, 32, 24, 40, 0, 0),           # plain conv_2x (stride 1)
    (2, 32, 64, 26, 30, 0, 0),
    (1, 8, 8, 4, 4, 1, 1),               # smaller than one tile
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[("%dx%d-%d-%dx%d-d%du%d" % c) for c in CASES])
def test_zoom_cell_matches_reference_and_split_launches(case, dtype):
    k = K()
    N, Cin, C, H, W, down, up = case
    if dtype == torch.float32 and C > 128:
        pytest.skip("fp32 keeps at most 128 mid channels in LDS (fs_zoom_cell_supported)")
    q = (lambda t: t.to(dtype).float())
    x = q(rnd(N, Cin, H, W, seed=1))
    w1 = rnd(C, Cin, 3, 3, seed=2, scale=(2.0 / (9 * Cin)) ** 0.5)
    w2 = rnd(C, C, 3, 3, seed=3, scale=(2.0 / (9 * C)) ** 0.5)
    s1, b1 = rnd(C, seed=4).abs() * 0.5 + 0.75, rnd(C, seed=5, scale=0.2)
    s2, b2 = rnd(C, seed=6).abs() * 0.5 + 0.75, rnd(C, seed=7, scale=0.2)
    want = reference(x, q(w1), s1, b1, q(w2), s2, b2, down, up)
    dev = "cuda"
    xg = k.to_nhwc(x.to(dev), dtype)
    d = k.zoom_desc(tuple(xg.shape), k.channel_stride(xg), C, C, down, up, k.round_up(C, k.vec_of(dtype)), dtype)
    assert k.zoom_cell_supported(d)
    w1f, w2f = k.pack_weight_frag(w1.to(dev), dtype), k.pack_weight_frag(w2.to(dev), dtype)
    sc = [t.to(dev).contiguous() for t in (s1, b1, s2, b2)]
    got = k.zoom_cell(xg, w1f, sc[0], sc[1], w2f, sc[2], sc[3], C, C, down=bool(down), up=bool(up))
    torch.cuda.synchronize()
    got_c = got.float().cpu()
    assert got_c.shape == want.shape
    err = (got_c - want).abs()
    if dtype == torch.float32:
        tol = 2e-4 + 2e-4 * want.abs()
    else:
        tol = 3e-2 * want.abs().max() + 0 * want
    assert not (err > tol).any(), "max err %.3e (max|ref| %.3e), %d bad of %d" % (
        float(err.max()), float(want.abs().max()), int((err > tol).sum()), err.numel())
    # the same cell as separate launches
    o = k.bilinear(xg, (H // 2, W // 2)) if down else xg
    o = k.conv2d(o, k.pack_weight(w1.to(dev), dtype), C, 3, 3, 1, 1, scale=sc[0], shift=sc[1], relu=True)
    o = k.conv2d(o, k.pack_weight(w2.to(dev), dtype), C, 3, 3, 1, 1, scale=sc[2], shift=sc[3], relu=not up)
    if up:
        o = k.bilinear(o, (H, W), relu=True)
    torch.cuda.synchronize()
    split = o.float().cpu()
    tol2 = (1e-4 + 1e-4 * want.abs().max()) if dtype == torch.float32 else 2e-2 * want.abs().max()
    assert float((got_c - split).abs().max()) <= float(tol2), "fused vs separate launches: %.3e" % float((got_c - split).abs().max())


def test_zoom_cell_reads_and_writes_channel_slices():
    """x is a channel slice of a wider buffer and y a slice of a concat buffer (torch.cat fused away, model_seg.py:307)."""
    k = K()
    dtype = torch.bfloat16
    N, Cin, C, H, W = 1, 32, 64, 32, 48
    q = (lambda t: t.to(dtype).float())
    x = q(rnd(N, Cin, H, W, seed=11))
    w1, w2 = rnd(C, Cin, 3, 3, seed=12, scale=0.08), rnd(C, C, 3, 3, seed=13, scale=0.06)
    ones, zeros = torch.ones(C), torch.zeros(C)
    want = reference(x, q(w1), ones, zeros, q(w2), ones, zeros, 1, 1)
    wide_in = k.empty_nhwc(N, 96, H, W, dtype, "cuda", zero=True)
    wide_in[:, 32:64].copy_(x.to("cuda").to(dtype))
    wide_out = k.empty_nhwc(N, 160, H, W, dtype, "cuda", zero=True)
    wide_out.fill_(7.0)
    k.zoom_cell(wide_in[:, 32:64], k.pack_weight_frag(w1.cuda(), dtype), None, None, k.pack_weight_frag(w2.cuda(), dtype), None, None,
                C, C, down=True, up=True, out=wide_out[:, 64:128])
    torch.cuda.synchronize()
    got = wide_out.float().cpu()
    assert float((got[:, 64:128] - want).abs().max()) <= 3e-2 * float(want.abs().max())
    assert (got[:, :64] == 7.0).all() and (got[:, 128:] == 7.0).all(), "wrote outside its channel slice"


def test_zoom_cell_rejects_unsupported_geometry_with_a_status():
    from fasterseg_amd import _lib
    k = K()
    x = k.empty_nhwc(1, 32, 15, 20, torch.bfloat16, "cuda", zero=True)          # odd height cannot be resampled 1/2 -> x2
    wf = k.pack_weight_frag(torch.zeros(32, 32, 3, 3, device="cuda"), torch.bfloat16)
    with pytest.raises(_lib.FasterSegHipError, match="unsupported geometry"):
        k.zoom_cell(x, wf, None, None, wf, None, None, 32, 32, down=True, up=True,
                    out=k.empty_nhwc(1, 32, 14, 20, torch.bfloat16, "cuda"))
