"""Latency harness contract: key grammar identical to the shipped 667-entry table, miss -> measure ->
insert -> persist behaviour, and on GPU the hipEvent timer itself."""
import json
import os

import numpy as np
import pytest
import torch

from tests._util import load_json


def test_generator_covers_exactly_the_shipped_keys():
    from fasterseg_amd.latency_lookup_table import entries
    keys = {k for k, _ in entries()}
    assert keys == set(load_json("latency_lut_1080ti.json"))


def test_forward_latency_lookup_and_miss_path(tmp_path, monkeypatch):
    from fasterseg_amd import operations
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(operations, "latency_lookup_table", {})
    monkeypatch.setattr(operations, "_journal_dirty", False)
    calls = []
    monkeypatch.setattr(operations, "compute_latency", lambda model, size: calls.append(size) or 0.123)
    op = operations.BasicResidual2x(32, 64, stride=2, slimmable=False)
    lat, size = op.forward_latency((32, 128, 256))
    assert (lat, size) == (0.123, (64, 64, 128)) and calls == [(1, 32, 128, 256)]
    key = "BasicResidual2x_H128_W256_Cin32_Cout64_stride2_dilation1"
    assert operations.latency_lookup_table == {key: 0.123}
    # a miss appends one journal line; the table file - the reference's on-disk format - is written once, by the flush (atexit)
    assert not os.path.exists("latency_lookup_table.npy")
    assert [json.loads(l) for l in open("latency_lookup_table.npy.journal")] == [[key, 0.123]]
    operations.flush_latency_table()
    assert np.load("latency_lookup_table.npy", allow_pickle=True).item() == {key: 0.123}
    assert not os.path.exists("latency_lookup_table.npy.journal")
    op.forward_latency((32, 128, 256))
    assert len(calls) == 1                                                                      # second call is a hit
    # the zoomed-2x quirk: priced under the BasicResidual2x key (reference operations.py:426-431)
    z = operations.BasicResidual_downup_2x(32, 64, stride=2, slimmable=False)
    assert z.forward_latency((32, 128, 256))[0] == 0.123 and len(calls) == 1
    s = operations.BasicResidual1x(96, 96, slimmable=True, width_mult_list=[4. / 12, 1.])
    s.set_ratio((4. / 12, 1.))
    with pytest.raises(AssertionError):
        s.forward_latency((96, 32, 64))          # c_in must equal int(C_in * ratio)


def test_latency_table_persistence_is_one_save_and_survives_a_killed_process(tmp_path, monkeypatch):
    """the reference rewrites the whole pickled dict on every miss (operations.py:116-122)."""
    from fasterseg_amd import operations
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(operations, "latency_lookup_table", {})
    monkeypatch.setattr(operations, "_journal_dirty", False)          # (restored at teardown: nothing is left for the atexit flush)
    saves = []
    real_save = np.save
    monkeypatch.setattr(np, "save", lambda *a, **k: (saves.append(a[0]), real_save(*a, **k))[1])
    for i in range(50):
        assert operations.lookup_latency("key_%d" % i, lambda i=i: 0.01 * i) == 0.01 * i
    assert saves == [] and sum(1 for _ in open("latency_lookup_table.npy.journal")) == 50
    operations.flush_latency_table()
    operations.flush_latency_table()                       # nothing new: no second write
    assert saves == ["latency_lookup_table.npy"]
    assert np.load("latency_lookup_table.npy", allow_pickle=True).item() == {"key_%d" % i: 0.01 * i for i in range(50)}
    # a process killed before its flush leaves the journal (possibly with a torn last line): the next import merges it
    operations.lookup_latency("late", lambda: 1.5)
    with open("latency_lookup_table.npy.journal", "a") as f:
        f.write('["torn", 0.')
    monkeypatch.setattr(operations, "latency_lookup_table", {})
    monkeypatch.setattr(operations, "_journal_dirty", False)
    operations._load_latency_table()
    assert operations.latency_lookup_table["late"] == 1.5 and len(operations.latency_lookup_table) == 51
    operations.flush_latency_table()
    assert np.load("latency_lookup_table.npy", allow_pickle=True).item()["late"] == 1.5
    assert not os.path.exists("latency_lookup_table.npy.journal")


@pytest.mark.gpu
def test_hip_timer_contract():
    from fasterseg_amd import operations
    from fasterseg_amd.latency import compute_latency_ms_hip
    layer = operations.BasicResidual2x(32, 32, slimmable=False)
    ms = compute_latency_ms_hip(layer, (1, 32, 128, 256), min_calib_ms=5, budget_ms=20)
    ms_eager = compute_latency_ms_hip(layer, (1, 32, 128, 256), graph=False, min_calib_ms=5, budget_ms=20)
    assert 0.002 < ms < 1.0 and 0.002 < ms_eager < 5.0
    assert ms <= ms_eager * 1.05, "device time of the captured forward cannot exceed the dispatch-inclusive eager time"
    assert layer.training, "the timer must restore the module's mode (it switches to eval() for the measurement)"


LUT_SAMPLE = ["BasicResidual1x_H128_W256_Cin96_Cout96_stride1_dilation1", "BasicResidual_downup_1x_H64_W128_Cin192_Cout192_stride1_dilation1",
              "BasicResidual2x_H32_W64_Cin384_Cout384_stride1_dilation1", "BasicResidual_downup_2x_H128_W256_Cin64_Cout64_stride1_dilation1",
              "BasicResidual2x_H64_W128_Cin128_Cout256_stride2_dilation1", "FactorizedReduce_H64_W128_Cin192_Cout384_stride2",
              "ConvNorm_H1024_W2048_Cin3_Cout48_kernel3_stride2", "ff_H128_W256_C96", "head_H128_W256_Cin96_Cout19"]


@pytest.mark.gpu
def test_shipped_lut_matches_the_current_kernels():
    """The shipped MI355X table (fasterseg_amd/fasterseg/latency_lookup_table_mi355x_bf16.json, what the C5 search step is regularised
    with) must describe THIS build: nine keys - every operator class, all three scales, both strides - are re-timed with the
    generator's own thunks and have to agree with the shipped values within +-30 % (hipEvent timing of ~10-100 us kernels on a
    shared box; a table from older kernels was off by 2x on the fused-cell shapes)."""
    from fasterseg_amd import functional as FN
    from fasterseg_amd import latency, latency_lookup_table
    import fasterseg_amd.operations as ops
    import fasterseg_amd.seg_oprs as sops
    shipped = latency_lookup_table.load_shipped("bf16")
    thunks = dict(latency_lookup_table.entries())
    assert set(LUT_SAMPLE) <= set(thunks) and set(thunks) == set(shipped)
    real = latency.compute_latency_ms_hip
    saved = (ops.compute_latency, sops.compute_latency)
    ops.compute_latency = sops.compute_latency = lambda model, size: real(model, size, min_calib_ms=20.0, budget_ms=60.0)
    FN.set_compute_dtype(torch.bfloat16)
    try:
        got = {k: float(thunks[k]()) for k in LUT_SAMPLE}
    finally:
        ops.compute_latency, sops.compute_latency = saved
        FN.set_compute_dtype(torch.float32)
    off = {k: (got[k], shipped[k]) for k in LUT_SAMPLE if not (0.7 * shipped[k] <= got[k] <= 1.3 * shipped[k])}
    assert not off, off


@pytest.mark.gpu
def test_hip_timer_refuses_uncapturable_forward():
    from fasterseg_amd.latency import compute_latency_ms_hip

    class Syncs(torch.nn.Module):
        def forward(self, x):
            return x + float(x.sum().item())          # a host read: not capturable
    with pytest.raises(RuntimeError, match="not hipGraph-capturable"):
        compute_latency_ms_hip(Syncs(), (1, 8, 4, 4), min_calib_ms=1, budget_ms=2)
    assert compute_latency_ms_hip(Syncs(), (1, 8, 4, 4), graph=False, min_calib_ms=1, budget_ms=2) > 0
