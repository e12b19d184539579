"""Host-side graph rewrites of the inference engine (fasterseg_amd/engine.py `_fuse_resizes`) on a hand-built op list: no
GPU needed, the pass only looks at shapes and producer/consumer edges.  What it must do: share identical resamples of one
feature map, fold a resample into its consumer only when that consumer is a 1x1 implicit-GEMM conv and the only reader,
and never touch resamples that feed a concat, several readers or the final NCHW logits."""
import torch

from fasterseg_amd import engine
from fasterseg_amd.engine import _Tracer


class _W:          # stands in for a filter tensor: the tracer only reads .shape
    def __init__(self, *shape):
        self.shape = shape


def _engine(ops, out_sym, mode="1"):
    e = engine.InferenceEngine.__new__(engine.InferenceEngine)
    e.ops, e.out_sym, e.halo_min_pixels = ops, out_sym, 16384
    import os
    old = os.environ.get("FS_ENGINE_FUSE_RESIZE")
    os.environ["FS_ENGINE_FUSE_RESIZE"] = mode
    try:
        e._fuse_resizes()
    finally:
        if old is None:
            os.environ.pop("FS_ENGINE_FUSE_RESIZE")
        else:
            os.environ["FS_ENGINE_FUSE_RESIZE"] = old
    return e


def _net():
    t = _Tracer(torch.bfloat16)
    from fasterseg_amd.functional import SymTensor
    x = SymTensor((1, 32, 64, 128), torch.bfloat16)
    a = t.conv(x, _W(64, 32, 3, 3), None, None, 1, 1, True, False, 64, 32)            # op 0
    d1 = t.resize(a, (32, 64), False, 0)                                                # op 1: down, read by a 3x3 conv
    d2 = t.resize(a, (32, 64), False, 0)                                                # op 2: same resample again (another cell)
    c1 = t.conv(d1, _W(64, 64, 3, 3), None, None, 1, 1, True, False, 64, 64)            # op 3
    c2 = t.conv(d2, _W(64, 64, 3, 3), None, None, 1, 1, False, False, 64, 64)           # op 4
    u1 = t.resize(c1, (64, 128), True, 0)                                               # op 5: up + ReLU, read by a 1x1 conv only
    p = t.conv(u1, _W(48, 64, 1, 1), None, None, 1, 0, True, False, 48, 64)             # op 6
    u2 = t.resize(c2, (64, 128), True, 0)                                               # op 7: up, read by a concat
    cat = t.cat([p, u2])                                                                # op 8
    u3 = t.resize(cat, (128, 256), False, 0)                                            # op 9: two readers
    q1 = t.conv(u3, _W(32, 112, 1, 1), None, None, 1, 0, False, False, 32, 112)         # op 10
    q2 = t.conv(u3, _W(19, 112, 1, 1), None, None, 1, 0, False, False, 19, 112)         # op 11
    out = t.resize(q2, (512, 1024), False, 1)                                           # op 12: NCHW logits
    return t.ops, out, q1


def test_resize_rewrites_default_mode():
    ops, out, _ = _net()
    e = _engine(ops, out)
    assert e.shared_resizes == 1 and ops[2].get("dead") and ops[4]["x"] is ops[1]["out"]        # d2 -> d1
    assert not ops[1].get("dead") and ops[3].get("vres") is None                               # 3x3 consumers keep the launch
    assert e.fused_resizes == 1 and ops[5].get("dead")                                          # up+ReLU folded into the 1x1 conv
    assert ops[6]["vres"] == (64, 128, True) and ops[6]["x"] is ops[3]["out"]
    assert not ops[7].get("dead") and ops[8]["inputs"][1] is ops[7]["out"]                      # concat operand stays
    assert not ops[9].get("dead") and ops[10].get("vres") is None and ops[11].get("vres") is None   # two readers
    assert not ops[12].get("dead")                                                              # the logits up-sample


def test_resize_rewrites_off_and_aggressive():
    ops, out, _ = _net()
    e = _engine(ops, out, mode="0")
    assert e.fused_resizes == 0 and not any(op.get("vres") for op in ops)
    ops, out, _ = _net()
    e = _engine(ops, out, mode="2")                # any implicit-GEMM consumer ...
    assert not ops[1].get("dead")                  # ... but after sharing, the down-sample has two readers and stays
    assert e.fused_resizes == 1
    t = _Tracer(torch.bfloat16)
    from fasterseg_amd.functional import SymTensor
    x = SymTensor((1, 32, 64, 128), torch.bfloat16)
    a = t.conv(x, _W(64, 32, 3, 3), None, None, 1, 1, True, False, 64, 32)
    d = t.resize(a, (32, 64), False, 0)
    c = t.conv(d, _W(64, 64, 3, 3), None, None, 1, 1, True, False, 64, 64)
    big = t.resize(c, (128, 256), False, 0)        # 32768 pixels: its 3x3 consumer runs on the halo kernel -> never folded
    h = t.conv(big, _W(64, 64, 3, 3), None, None, 1, 1, True, False, 64, 64)
    out = t.resize(h, (512, 1024), False, 1)
    e = _engine(t.ops, out, mode="2")
    assert t.ops[1].get("dead") and t.ops[2]["vres"] == (32, 64, False) and t.ops[2]["x"] is t.ops[0]["out"]
    assert not t.ops[3].get("dead") and t.ops[4].get("vres") is None
    assert e.fused_resizes == 1


def test_list_scheduler_puts_independent_chains_on_separate_lanes(monkeypatch):
    """`_schedule` (HEFT with measured durations): a trunk that forks into a long and a short chain which join again.  The
    result must be a topological order with remapped dependencies, every cross-lane edge an event wait, the two chains on
    different lanes, and the estimated makespan the critical path (+ the two lane-crossing edges), not the serial sum."""
    monkeypatch.setenv("FS_ENGINE_SPLITK", "0")
    e = engine.InferenceEngine.__new__(engine.InferenceEngine)
    e.n_lanes, e.device = 3, "cpu"
    #        0 -> 1 -> 2 -> 3 -> 4 (long chain, 10 us each)     0 -> 5 -> 6 (short chain, 4 us each)     (4, 6) -> 7
    deps = {0: [], 1: [0], 2: [1], 3: [2], 4: [3], 5: [0], 6: [5], 7: [4, 6]}
    e.calls = [dict(fn="fs_zoom_cell_fwd", args=(), deps=list(deps[i]), label=str(i)) for i in range(8)]
    durs = [0.010] * 5 + [0.004] * 2 + [0.010]
    est = e._schedule(durs, 3)
    labels = [c["label"] for c in e.calls]
    pos = {l: k for k, l in enumerate(labels)}
    for k, c in enumerate(e.calls):
        assert all(d < k for d in c["deps"])                                   # topological
        assert sorted(labels[d] for d in c["deps"]) == sorted(str(d) for d in deps[int(c["label"])])     # same edges
        assert c["waits"] == [d for d in c["deps"] if e.calls[d]["lane"] != c["lane"]]
    long_lane = {e.calls[pos[str(i)]]["lane"] for i in (1, 2, 3, 4)}
    short_lane = {e.calls[pos[str(i)]]["lane"] for i in (5, 6)}
    assert len(long_lane) == 1 and len(short_lane) == 1 and long_lane != short_lane
    assert e.calls[pos["0"]]["signal"] and e.calls[pos["6"]]["signal"] or e.calls[pos["4"]]["signal"]
    assert abs(est - 0.060) < 0.004, est                                        # critical path 6 x 10 us (+ edges), serial sum is 68
