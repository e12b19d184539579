"""Host logic parity (CPU): arch decode + derived-network construction vs fixtures produced by the unmodified reference
(tests/golden/arch_*.json, decode_cases.json; generator oracle/make_golden.py)."""
import pytest
import torch

from fasterseg_amd import model_seg
from tests._util import arch_tensors, load_json

WML = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]


def build(idx, lasts, training):
    alphas, betas, ratios, _ = arch_tensors(idx)
    teacher = idx == 0
    net = model_seg.Network_Multi_Path_Infer(alphas, betas, ratios, num_classes=19, layers=16, Fch=12, width_mult_list=WML,
                                             stem_head_width=(1., 1.) if teacher else (8. / 12, 8. / 12), ignore_skip=teacher)
    net.train(training)
    net.build_structure(list(lasts))
    return net


@pytest.mark.parametrize("idx", [0, 1])
@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("lasts", [[2, 1], [2, 0], [1], [2]])
def test_shipped_arch_builds_like_reference(idx, training, lasts):
    meta = load_json("arch_%d.json" % idx)["%s_%s" % ("train" if training else "eval", "".join(map(str, lasts)))]
    net = build(idx, lasts, training)
    assert [[int(o) for o in ops] for ops in net.ops] == meta["ops"]
    assert net.paths == meta["paths"] and net.downs == meta["downs"]
    assert [[float(w) for w in ws] for ws in net.widths] == meta["widths"]
    assert net.branch_groups == meta["branch_groups"]
    assert (net.ch_16, net.ch_8_2, net.ch_8_1) == (meta["ch_16"], meta["ch_8_2"], meta["ch_8_1"])
    for key, cell in net.cells.items():
        want = meta["cells"][key]
        assert model_seg.PRIMITIVES.index({v: k for k, v in {
            'skip': 'FactorizedReduce', 'conv': 'BasicResidual1x', 'conv_downup': 'BasicResidual_downup_1x',
            'conv_2x': 'BasicResidual2x', 'conv_2x_downup': 'BasicResidual_downup_2x'}.items()}[type(cell._op._op).__name__]) == want["op"]
        assert (int(bool(cell._down)), cell._C_in, cell._C_out) == (want["down"], want["C_in"], want["C_out"])
    sd = net.state_dict()
    assert list(sd.keys()) == list(meta["state_shapes"].keys())          # same keys, same order
    assert {k: list(v.shape) for k, v in sd.items()} == meta["state_shapes"]
    assert sum(p.numel() for p in net.parameters()) == meta["num_params"]


def test_param_counts_match_paper_pins():
    # SURVEY.md §8c pins: 3,454,643 (student eval build), 22,188,931 (teacher)
    assert sum(p.numel() for p in build(1, [2, 1], False).parameters()) == 3454643
    assert sum(p.numel() for p in build(0, [2, 1], False).parameters()) == 22188931


CASES = load_json("decode_cases.json")


@pytest.mark.parametrize("i", range(len(CASES)))
def test_network_metas_property_cases(i):
    rec = CASES[i]
    t = lambda v: torch.tensor(v, dtype=torch.float32)
    alphas = [t(a) for a in rec["alphas"]]
    betas = [None, t(rec["betas"][1]), t(rec["betas"][2])]
    ratios = [t(r) for r in rec["ratios"]]
    nw = ratios[0].shape[1]
    wml = WML if nw == 5 else ([1.] if rec["ignore_skip"] else [4. / 12])
    got = []
    try:
        for last in (0, 1, 2):
            ops, path, downs, widths = model_seg.network_metas(alphas, betas, ratios, wml, rec["layers"], last,
                                                               ignore_skip=rec["ignore_skip"])
            got.append({"ops": [int(o) for o in ops], "path": path, "downs": downs, "widths": [float(w) for w in widths]})
    except AssertionError:
        assert rec["metas"] is None and got == rec["partial"]
        return
    assert got == rec["metas"]


def test_modules_are_picklable():
    """The reference evaluator pickles the network into spawned workers (tools/engine/evaluator.py:128-157)."""
    import pickle
    net = build(1, [2, 1], False)
    clone = pickle.loads(pickle.dumps(net))
    assert list(clone.state_dict().keys()) == list(net.state_dict().keys())


def test_registry_surface():
    from fasterseg_amd import genotypes, operations
    assert genotypes.PRIMITIVES == ['skip', 'conv', 'conv_downup', 'conv_2x', 'conv_2x_downup']
    assert list(operations.OPS.keys()) == genotypes.PRIMITIVES == list(operations.OPS_Class.keys())
    assert operations.OPS_name == [c.__name__ for c in operations.OPS_Class.values()]
    for name in genotypes.PRIMITIVES:
        for stride in (1, 2):
            for slim in (False, True):
                m = operations.OPS[name](48, 48 * stride, stride, slim, WML if slim else [1.])
                assert isinstance(m, operations.OPS_Class[name])
                if slim:
                    m.set_ratio((WML[1], WML[3]))
