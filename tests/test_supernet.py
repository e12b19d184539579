"""Supernet (fasterseg_amd.model_search) vs fixtures produced by the reference's search/model_search.py.
CPU part: module tree / state_dict parity and the differentiable latency model (pure host logic over the LUT).
GPU part: eval forward, pretrain and search losses + gradients through every primitive at every width."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.seeded import seeded_input, seeded_state
from tests._util import assert_close_golden, golden_get, load_json, load_npz

WML = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]
CFG = dict(num_classes=19, layers=6, Fch=12, width_mult_list=WML, prun_modes=['max', 'arch_ratio'],
           stem_head_width=[(1, 1), (8. / 12, 8. / 12)])


def build():
    from fasterseg_amd import model_search
    net = model_search.Network_Multi_Path(criterion=torch.nn.CrossEntropyLoss(ignore_index=255), **CFG)
    sd = seeded_state(net.state_dict(), 777)
    for k in list(sd):
        if k.split("_")[0] in ("alpha", "beta", "ratio"):
            sd[k] = sd[k] * 5.0
    net.load_state_dict(sd)
    return net


def test_supernet_state_dict_matches_reference():
    meta = load_json("supernet_meta.json")
    net = build()
    sd = net.state_dict()
    assert list(sd.keys()) == list(meta["state_shapes"].keys())
    assert {k: list(v.shape) for k, v in sd.items()} == meta["state_shapes"]
    assert sum(p.numel() for p in net.parameters()) == meta["num_params"]


def test_full_size_supernet_param_count():
    from fasterseg_amd import model_search
    net = model_search.Network_Multi_Path(19, 16, None, 12, WML, ['max', 'arch_ratio'], [(1, 1), (8. / 12, 8. / 12)])
    assert sum(p.numel() for p in net.parameters()) == 252057316          # 252.06 M


def test_forward_latency_matches_reference():
    """The latency regulariser of the search (architect.py:66-72): LUT lookups weighted by differentiable alpha/beta/ratio
    scores, including the reference's BasicResidual2x key quirk for zoomed 2x cells."""
    from fasterseg_amd import operations
    store = load_npz("supernet.npz")
    net = build()
    saved = dict(operations.latency_lookup_table)
    operations.latency_lookup_table.clear()
    operations.latency_lookup_table.update(load_json("latency_lut_1080ti.json"))
    try:
        net.arch_idx = 1
        net.prun_mode = None
        for a, b, r in ((True, False, False), (False, True, False), (False, False, True), (True, True, True)):
            net.zero_grad()
            torch.manual_seed(9)
            lat = net.forward_latency((3, 1024, 2048), alpha=a, beta=b, ratio=r)
            lat.backward()
            key = "lat_%d%d%d" % (a, b, r)
            assert abs(float(lat.detach()) - float(store[key + "/value"][0])) < 1e-4, key
            for k in ("alpha_1_0", "alpha_1_2", "beta_1_1", "ratio_1_1"):
                gr = getattr(net, k).grad
                got = gr.numpy() if gr is not None else np.zeros(1, np.float32)
                np.testing.assert_allclose(got, store[key + "/g/" + k], atol=1e-5, rtol=1e-4)
    finally:
        operations.latency_lookup_table.clear()
        operations.latency_lookup_table.update(saved)


@pytest.mark.parametrize("arch_idx", [0, 1])
def test_linear_latency_equals_per_mixedop_evaluation(arch_idx):
    """forward_latency(beta=False) as one dot product over tabulated LUT rows (no host read-back of the Gumbel widths), and
    forward_latency(alpha=False, beta=True, ratio=False) as a tree product of per-assignment affine maps, against the
    per-MixedOp evaluation they replace: same value, same gradients w.r.t. alpha, beta and ratio, on perturbed architecture
    parameters, for the Gumbel ("arch_ratio", arch 1) and the fixed-width ("max", arch 0) modes."""
    from fasterseg_amd import model_search, operations
    saved = dict(operations.latency_lookup_table)
    operations.latency_lookup_table.clear()
    operations.latency_lookup_table.update(load_json("latency_lut_1080ti.json"))
    flag = model_search._LINEAR_LATENCY
    try:
        net = build()
        g = torch.Generator().manual_seed(4)
        for p in net._arch_parameters[arch_idx]:
            p.data.add_(torch.randn(p.shape, generator=g) * 0.5)
        net.arch_idx = arch_idx
        net.prun_mode = None
        for a, b, r in ((True, False, False), (False, False, True), (True, False, True), (False, False, False), (False, True, False)):
            got = []
            for linear in (False, True):
                model_search._LINEAR_LATENCY = linear
                net.zero_grad()
                torch.manual_seed(9)
                lat = net.forward_latency((3, 1024, 2048), alpha=a, beta=b, ratio=r)
                grads = {}
                if torch.is_tensor(lat) and lat.requires_grad:
                    lat.backward()
                    grads = {n: getattr(net, n).grad.clone() for kind in ("alphas", "betas", "ratios")
                             for n in net._arch_names[arch_idx][kind] if getattr(net, n).grad is not None}
                got.append((float(lat.detach()) if torch.is_tensor(lat) else float(lat), grads))
            (v0, g0), (v1, g1) = got
            assert abs(v0 - v1) <= 2e-6 * abs(v0), (a, b, r, v0, v1)
            assert set(g0) == set(g1), (a, b, r, sorted(g0), sorted(g1))
            for n in g0:
                assert torch.allclose(g0[n], g1[n], rtol=1e-4, atol=1e-6), (a, b, r, n, float((g0[n] - g1[n]).abs().max()))
    finally:
        model_search._LINEAR_LATENCY = flag
        operations.latency_lookup_table.clear()
        operations.latency_lookup_table.update(saved)


@pytest.mark.parametrize("arch_idx,mode", [(1, "arch_ratio"), (0, "max"), (1, "random")])
def test_batched_coefficients_equal_per_mixedop_products(arch_idx, mode):
    """`_coefficient_rows` (alpha row x in-score x out-score of every MixedOp of a pass in a few batched ops) against
    MixedOp._coefficients evaluated one MixedOp at a time (reference model_search.py:64-78): same values, same gradients w.r.t.
    the alpha and ratio parameters."""
    from fasterseg_amd import model_search
    net = build()
    g = torch.Generator().manual_seed(8)
    for p in net._arch_parameters[arch_idx]:
        p.data.add_(torch.randn(p.shape, generator=g) * 0.5)
    net.arch_idx = arch_idx
    x = torch.zeros(1)
    results = []
    for batched in (True, False):
        net.zero_grad()
        torch.manual_seed(3)
        np.random.seed(3)
        alphas, _ = net._arch_tensors()
        ratios = net.sample_prun_ratio(mode=mode)
        rows = net._coefficient_rows(alphas, ratios, mode) if batched else None
        vals, total = {}, 0
        w = torch.Generator().manual_seed(1)
        for i, cells in enumerate(net.cells):
            for j, cell in enumerate(cells):
                r = net._cell_ratio(i, j, ratios)
                for which, (op, out) in enumerate(((cell._op, r[1]), (getattr(cell, "downsample", None), r[2]))):
                    if op is None or out is None:
                        continue
                    a = rows[(i, j, which)] if batched else alphas[j][i - j]
                    c = op._coefficients(x, a, (r[0], out))
                    vals[(i, j, which)] = c.detach().clone()
                    total = total + (c * torch.randn(c.shape, generator=w)).sum()
        total.backward()
        grads = {n: getattr(net, n).grad.clone() for kind in ("alphas", "ratios") for n in net._arch_names[arch_idx][kind]
                 if getattr(net, n).grad is not None}
        results.append((vals, grads))
    (v0, g0), (v1, g1) = results
    assert set(v0) == set(v1) and len(v0) == sum(1 + int(c._down) for cells in net.cells for c in cells)
    for key in v0:
        assert torch.equal(v0[key], v1[key]), key
    assert set(g0) == set(g1)
    for n in g0:
        assert torch.allclose(g0[n], g1[n], rtol=1e-5, atol=1e-7), (n, float((g0[n] - g1[n]).abs().max()))


def test_batched_gumbel_sampling_equals_the_slot_by_slot_reference_draws():
    """sample_prun_ratio("arch_ratio") draws all 44 slots in one batch (round 5); the reference draws them one by one
    (model_search.py:19-44,243-247: gumbel_softmax(F.log_softmax(ratio[layer]), hard=True) per slot).  Same host-RNG consumption (the next
    draw after the call is the same), same one-hots, same soft scores, same gradients w.r.t. the ratio parameters."""
    from fasterseg_amd import model_search
    net = build()
    net.arch_idx = 1
    g = torch.Generator().manual_seed(5)
    for p in net._arch_parameters[1]:
        p.data.add_(torch.randn(p.shape, generator=g) * 0.7)
    names = net._arch_names[1]["ratios"]
    counts = (net._layers - 1, net._layers - 1, net._layers - 2)
    torch.manual_seed(11)
    got = net.sample_prun_ratio(mode="arch_ratio")
    after_got = torch.rand(3)
    assert got.stacked.shape == (sum(counts), len(net._width_mult_list)) and got.index.shape == (sum(counts),)
    wsum = torch.Generator().manual_seed(2)
    weights = [[torch.randn(len(net._width_mult_list), generator=wsum) for _ in range(counts[s])] for s in range(3)]
    sum((r * w).sum() for rs, ws in zip(got, weights) for r, w in zip(rs, ws)).backward()
    grads_got = [getattr(net, n).grad.clone() for n in names]
    net.zero_grad()
    torch.manual_seed(11)
    want = [[model_search.gumbel_softmax(F.log_softmax(getattr(net, names[s])[layer], dim=-1), hard=True) for layer in range(counts[s])]
            for s in range(3)]
    after_want = torch.rand(3)
    assert torch.equal(after_got, after_want)                    # the generator is left where the reference leaves it
    sum((r * w).sum() for rs, ws in zip(want, weights) for r, w in zip(rs, ws)).backward()
    n = 0
    for s in range(3):
        assert len(got[s]) == counts[s]
        for layer in range(counts[s]):
            a, b = got[s][layer], want[s][layer]
            assert torch.allclose(a, b, rtol=0, atol=1e-7), (s, layer)
            assert int(a.argmax()) == int(b.argmax()) == int(a._fs_index) == int(a._fs_index_t) == int(got.index[n])
            n += 1
    for a, n_ in zip(grads_got, names):
        assert torch.allclose(a, getattr(net, n_).grad, rtol=1e-5, atol=1e-7), n_


def _rel_l2(got, store, key):
    want, step = golden_get(store, key)
    got = got.detach().float().cpu().numpy().reshape(-1)[::step]
    return float(((got - want.reshape(-1)) ** 2).sum() ** 0.5 / ((want ** 2).sum() ** 0.5 + 1e-30))


@pytest.mark.gpu
def test_supernet_eval_forward():
    store = load_npz("supernet.npz")
    net = build().cuda().eval()
    x = seeded_input((2, 3, 128, 256), 31).cuda()
    for idx in (0, 1):
        net.arch_idx = idx
        net.prun_mode = "max"
        with torch.no_grad():
            preds = net(x)
        assert len(preds) == 5
        for i, p in enumerate(preds):
            assert p.shape == (2, 19, 128, 256)
            assert_close_golden(p[:, :, ::4, ::4], store, "eval_arch%d/pred%d_sub" % (idx, i), 2e-3, 1e-3, "arch%d pred%d" % (idx, i))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["pretrain", "search"])
def test_supernet_loss_and_gradients(mode):
    """`_loss` = 4 full supernet forwards (pretrain: max/min/random/random; search: arch0, arch1 (Gumbel widths), max, min)
    against the reference modules run in fp64 (batch-statistics BN on small maps: relative-L2 bars, see test_ops_gpu)."""
    store = load_npz("supernet.npz")
    net = build().cuda().train()
    x = seeded_input((2, 3, 128, 256), 31).cuda()
    target = torch.tensor(store["target"]).cuda()
    np.random.seed(5)
    torch.manual_seed(6)
    net.arch_idx = 0
    loss = net._loss(x, target, mode == "pretrain")
    loss.backward()
    want = float(store["%s_f64/loss" % mode][0])
    assert abs(float(loss.detach()) - want) < 2e-3 * abs(want), (float(loss.detach()), want)
    params = dict(net.named_parameters())
    norms = load_json("supernet_%s_f64_gradnorms.json" % mode)
    assert set(norms) == {k for k, p in params.items() if p.grad is not None}, "same parameters receive gradients"
    bad = [(k, float(params[k].grad.norm()), w) for k, w in norms.items()
           if abs(float(params[k].grad.norm()) - w) > 5e-2 * w + 1e-5]
    assert len(bad) <= len(norms) // 100, bad[:10]
    for key in store:
        if key.startswith("%s_f64/g/" % mode):
            pname = key[len("%s_f64/g/" % mode):].split("@")[0]
            rel = _rel_l2(params[pname].grad, store, "%s_f64/g/%s" % (mode, pname))
            assert rel < 5e-2, (key, rel)


def test_beta_table_cache_reads_once_per_value_and_catches_a_zero():
    """model_search._positive_table (round 5): the `betas > 0` table is read back once per VALUE of the beta parameters; when they change, the
    previous table is used while the new values are checked asynchronously, and a check that finds a 0 switches the blocking read back on.
    Driven here with host tensors (the device path differs only in the pinned flag + event behind which the check completes)."""
    import warnings
    from fasterseg_amd import model_search
    b1 = torch.softmax(torch.randn(3, 2), -1)
    b2 = torch.softmax(torch.randn(2, 2), -1)
    cache = {}
    t0 = model_search._positive_table([None, b1, b2], key=("v", 0), cache=cache)
    assert t0 == [None, [[True, True]] * 3, [[True, True]] * 2] and cache["table"][0] == ("v", 0)
    # same key: served from the cache, whatever the tensors now hold
    zero = torch.tensor([[1.0, 0.0]] * 3)
    assert model_search._positive_table([None, zero, b2], key=("v", 0), cache=cache) is t0
    # new key, still positive: the old table, one pending check that passes at the next call
    t1 = model_search._positive_table([None, b1 * 0.5, b2], key=("v", 1), cache=cache)
    assert t1 is t0 and len(cache["pending"]) == 1
    assert model_search._positive_table([None, b1 * 0.5, b2], key=("v", 1), cache=cache) is t0 and not cache["pending"] and not cache.get("blocking")
    # new key with a zero: optimistic answer now, caught at the next call, exact (blocking) answers from then on
    t2 = model_search._positive_table([None, zero, b2], key=("v", 2), cache=cache)
    assert t2 is t0 and len(cache["pending"]) == 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        t3 = model_search._positive_table([None, zero, b2], key=("v", 2), cache=cache)
    assert cache["blocking"] and any("beta reached 0" in str(x.message) for x in w)
    assert t3 == [None, [[True, False]] * 3, [[True, True]] * 2]
    assert model_search._positive_table([None, b1, b2], key=("v", 3), cache=cache) == t0          # blocking mode: exact every time
    # without a cache (host tensors in the model, or no key): the plain read
    assert model_search._positive_table([None, zero, b2]) == t3
