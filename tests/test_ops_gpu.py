"""Operator- and network-level parity on MI355X: the drop-in modules (fasterseg_amd.operations / seg_oprs / model_seg)
against (a) fixtures produced by the unmodified reference and (b) the CPU oracle on the same seeded inputs.
Tolerance (north_star): logits within 1e-3 in fp32; gradients 2e-3 relative to their scale."""
import pytest
import torch

from oracle import ref_ops
from oracle.seeded import resolve_aliases, seeded_input, seeded_state
from tests._util import arch_tensors, assert_close_golden, golden_get, load_json, load_npz, shapes_template

pytestmark = pytest.mark.gpu
WML = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]
OPS_INDEX = load_json("ops_index.json")


def build_module(rec):
    from fasterseg_amd import operations, seg_oprs
    if rec["type"] == "primitive":
        n, cin, h, w = rec["shape"]
        if rec.get("slimmable"):
            m = operations.OPS[rec["kind"]](rec["cin_max"], rec["cout"], rec["stride"], True, WML)
            m.set_ratio(tuple(rec["ratio"]))
        else:
            m = operations.OPS[rec["kind"]](cin, rec["cout"], rec["stride"], False, [1.])
        return m
    if rec["type"] == "convnorm":
        return operations.ConvNorm(rec["shape"][1], rec["cout"], rec["k"], rec["stride"], rec["pad"], slimmable=False)
    if rec["type"] == "head":
        return seg_oprs.Head(rec["shape"][1], 19, True)
    if rec["type"] == "ffm":
        return seg_oprs.FeatureFusion(64, 64, reduction=1, Fch=12, scale=8, branch=2)
    raise ValueError(rec["type"])


@pytest.mark.parametrize("rec", OPS_INDEX, ids=[r["name"] for r in OPS_INDEX])
def test_module_matches_reference_fixture(rec):
    store = load_npz("ops.npz")
    name = rec["name"]
    m = build_module(rec)
    sd = m.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == rec["state_shapes"], "state_dict keys/shapes differ from reference"
    m.load_state_dict(seeded_state(sd, rec["seed"]))
    m = m.cuda()
    m.train(rec["training"])
    x = seeded_input(tuple(rec["shape"]), rec["seed"]).cuda().requires_grad_(rec["training"])
    y = m(x)
    assert_close_golden(y, store, name + "/y", 2e-4, 2e-4, name)
    if not rec["training"]:
        return
    gy = seeded_input(tuple(y.shape), rec["seed"] + 17).cuda()
    (y * gy).sum().backward()
    gscale = float(abs(golden_get(store, name + "/gx")[0]).max())
    assert_close_golden(x.grad, store, name + "/gx", 2e-3 * gscale + 1e-5, 1e-3, name)
    params = dict(m.named_parameters())
    for key in store:
        if key.startswith(name + "/g/"):
            pname = key[len(name) + 3:].split("@")[0]
            p = params[pname]
            assert p.grad is not None, pname
            want, _ = golden_get(store, name + "/g/" + pname)
            assert_close_golden(p.grad, store, name + "/g/" + pname, 2e-3 * float(abs(want).max()) + 1e-5, 1e-3, name + ":" + pname)
    after = m.state_dict()
    for key in store:
        if key.startswith(name + "/s/"):
            assert_close_golden(after[key[len(name) + 3:]], store, key, 1e-5, 1e-4, name)


def build_net(idx, lasts, training):
    from fasterseg_amd import model_seg
    alphas, betas, ratios, _ = arch_tensors(idx)
    teacher = idx == 0
    net = model_seg.Network_Multi_Path_Infer(alphas, betas, ratios, num_classes=19, layers=16, Fch=12, width_mult_list=WML,
                                             stem_head_width=(1., 1.) if teacher else (8. / 12, 8. / 12), ignore_skip=teacher)
    net.train(training)
    net.build_structure(list(lasts))
    net.train(training)          # modules created by build_structure start in train mode (same in the reference)
    net.load_state_dict(seeded_state(net.state_dict(), 12345))
    return net.cuda()


@pytest.mark.parametrize("idx,shape", [(1, (1, 3, 128, 256)), (0, (1, 3, 64, 128))])
def test_derived_net_eval_logits(idx, shape):
    """BASELINE config C1: searched student arch_1 forward on 1x3x128x256 vs the reference's CPU forward."""
    store = load_npz("nets.npz")
    net = build_net(idx, [2, 1], False)
    x = seeded_input(shape, 5).cuda()
    with torch.no_grad():
        y = net(x)
    assert tuple(y.shape) == (shape[0], 19, shape[2], shape[3]) and y.is_contiguous() and y.dtype == torch.float32
    assert_close_golden(y[:, :, ::4, ::4], store, "arch%d_eval/logits_sub" % idx, 1e-3, 0.0, "logits")
    # and the full tensor against the oracle
    meta = load_json("arch_%d.json" % idx)["eval_21"]
    params = resolve_aliases(seeded_state(shapes_template(meta["state_shapes"]), 12345), meta)
    with torch.no_grad():
        want = ref_ops.derived_forward(params, meta, seeded_input(shape, 5), training=False)
    err = float((y.cpu() - want).abs().max())
    assert err <= 1e-3, "max |logit error| vs oracle %.3e" % err


def test_student_bf16_logits_close():
    """bf16 storage path of the same network (the C2 throughput configuration): argmax agreement and bounded error."""
    from fasterseg_amd import functional as FN
    meta = load_json("arch_1.json")["eval_21"]
    params = resolve_aliases(seeded_state(shapes_template(meta["state_shapes"]), 12345), meta)
    with torch.no_grad():
        want = ref_ops.derived_forward(params, meta, seeded_input((1, 3, 128, 256), 5), training=False)
    net = build_net(1, [2, 1], False)
    FN.set_compute_dtype(torch.bfloat16)
    try:
        with torch.no_grad():
            y = net(seeded_input((1, 3, 128, 256), 5).cuda())
    finally:
        FN.set_compute_dtype(torch.float32)
    rel = float((y.cpu() - want).abs().max() / want.abs().max())
    agree = float((y.cpu().argmax(1) == want.argmax(1)).float().mean())
    assert rel < 5e-2 and agree > 0.97, (rel, agree)


def _rel_l2(got, store, key):
    want, step = golden_get(store, key)
    got = got.detach().float().cpu().numpy().reshape(-1)[::step]
    want = want.reshape(-1)
    return float(((got - want) ** 2).sum() ** 0.5 / ((want ** 2).sum() ** 0.5 + 1e-30))


def test_student_train_step_matches_reference():
    """Student train-mode step (3 heads, batch-statistics BN down to 1x2 maps = 4 samples per channel).
    Logits are checked against the reference's own fp32 CPU run (1e-3).  Gradients are checked against the SAME reference
    modules run in fp64 (fixture arch1_train64): a single ReLU-mask flip caused by 1e-7 rounding moves individual
    gradient elements by O(1) — the reference's fp32 run itself differs from its fp64 run by e.g. 2.26 of 128 in
    ffm.conv_1x1.bn.bias[4] and 7e-3 relative L2 in d(loss)/d(input) — so element-wise exactness is pinned per operator
    above (2e-3, 61 cases) and whole-network gradients by relative L2 error against fp64: 3e-2 per tensor (five
    repeated runs of this step measured 6e-4 .. 1.2e-2: BN statistics are accumulated with atomics, so the set of
    flipped masks varies run to run)."""
    store = load_npz("nets.npz")
    net = build_net(1, [2, 1], True)
    x = seeded_input((2, 3, 128, 256), 6).cuda().requires_grad_(True)
    p8, p16, p32 = net(x)
    loss = (p8 * seeded_input(tuple(p8.shape), 7).cuda()).sum() + 0.2 * (p16 * seeded_input(tuple(p16.shape), 8).cuda()).sum() \
        + 0.2 * (p32 * seeded_input(tuple(p32.shape), 9).cuda()).sum()
    loss.backward()
    # the loss is a signed sum of ~4e5 terms of magnitude O(1): 0.5 absolute is ~1e-6 of sum|terms|
    assert abs(float(loss.detach()) - float(store["arch1_train64/loss"][0])) < 0.5
    for tag in ("arch1_train", "arch1_train64"):
        assert_close_golden(p8[:, :, ::4, ::4], store, tag + "/p8_sub", 1e-3, 1e-3)
        assert_close_golden(p16[:, :, ::4, ::4], store, tag + "/p16_sub", 1e-3, 1e-3)
        assert_close_golden(p32[:, :, ::4, ::4], store, tag + "/p32_sub", 1e-3, 1e-3)
    assert _rel_l2(x.grad, store, "arch1_train64/gx") < 3e-2
    params = dict(net.named_parameters())
    norms = load_json("arch1_train64_gradnorms.json")
    assert set(norms) == {k for k, p in params.items() if p.grad is not None}, "same set of parameters receives gradients"
    for k, want in norms.items():
        got = float(params[k].grad.norm())
        assert abs(got - want) <= 2e-2 * want + 1e-4, (k, got, want)
    for key in store:
        if key.startswith("arch1_train64/g/"):
            pname = key[len("arch1_train64/g/"):].split("@")[0]
            rel = _rel_l2(params[pname].grad, store, "arch1_train64/g/" + pname)
            assert rel < 3e-2, (key, rel)
