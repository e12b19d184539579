"""The inference engine — the code path bench.py times — against the CPU oracle AT the benchmarked shape.

`InferenceEngine` (static plan, fused zoomed cells, halo kernels, stream lanes, hipGraph) is built for the student (arch_1,
eval build) on seeded weights and compared with oracle.ref_ops.derived_forward, the fixture-pinned restatement of
/root/reference train/model_seg.py:337-366 (Network_Multi_Path_Infer.forward), on the same seeded input:
  fp32  full-tensor |logits - oracle| <= 1e-3 (north_star's bar)
  bf16  relative to max|logits| <= 5e-2 and arg-max agreement >= 97 %
at 1x3x1024x2048 (BASELINE configs[1]) and at 1x3x256x512, for every instantiation bench.py may select: the multi-lane
hipGraph, the single-lane graph, the direct launch list and the fs_exec_program_streams executor; with cells fused,
un-fused, and chosen by timing."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_cache = {}


def _student(shape):
    """(net on the GPU, seeded input, oracle logits) — the oracle forward at 1024x2048 takes ~1 s on the host."""
    if shape in _cache:
        return _cache[shape]
    from fasterseg_amd import archs
    from oracle import ref_ops
    from oracle.seeded import resolve_aliases, seeded_input, seeded_state
    with open(os.path.join(ROOT, "tests", "golden", "arch_1.json")) as f:
        meta = json.load(f)["eval_21"]
    net = archs.build_derived(1, training=False, lasts=[2, 1])
    state = seeded_state(net.state_dict(), 12345)
    net.load_state_dict(state)
    net = net.cuda().eval()
    x = seeded_input(shape, 3)
    with torch.no_grad():
        want = ref_ops.derived_forward(resolve_aliases({k: v.clone() for k, v in state.items()}, meta), meta, x, training=False)
    _cache[shape] = (net, x, want)
    return _cache[shape]


def _check(got, want, dtype, what):
    got = got.float().cpu()
    assert got.shape == want.shape
    err = float((got - want).abs().max())
    if dtype == torch.float32:
        assert err <= 1e-3, "%s: fp32 logits differ from the oracle by %.3e (> 1e-3)" % (what, err)
    else:
        rel = err / float(want.abs().max())
        agree = float((got.argmax(1) == want.argmax(1)).float().mean())
        assert rel <= 5e-2, "%s: bf16 logits rel. error %.3e (> 5e-2)" % (what, rel)
        assert agree >= 0.97, "%s: bf16 arg-max agreement %.4f (< 0.97)" % (what, agree)


SHAPES = [(1, 3, 1024, 2048), (1, 3, 256, 512)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", SHAPES, ids=["1024x2048", "256x512"])
@pytest.mark.parametrize("cells", ["1", "0", "auto"], ids=["fused", "split", "auto"])
def test_engine_logits_match_oracle_at_benchmark_shape(shape, dtype, cells):
    from fasterseg_amd import engine
    net, x, want = _student(shape)
    with torch.no_grad():
        eng = engine.InferenceEngine(net, shape, dtype=dtype, fuse_cells=cells)
        got = eng(x.cuda()).clone()
        torch.cuda.synchronize()
    _check(got, want, dtype, "selected instantiation (%s)" % (eng.capture_log,))
    fns = [c["fn"] for c in eng.calls]
    if shape == (1, 3, 1024, 2048):
        # the plan really contains the kernels the benchmark's time is made of
        assert "fs_conv3x3_s1_fwd" in fns, "no halo-kernel launch in the plan"
        assert any(c["fn"].startswith("fs_conv2d_fwd") and c["desc"].N * c["desc"].Ho * c["desc"].Wo >= 128 * 128 for c in eng.calls), \
            "no large-tile implicit-GEMM launch in the plan"
        if cells == "1" and dtype == torch.bfloat16:       # fp32 keeps the two widest cells (192, 256 channels) un-fused
            assert fns.count("fs_zoom_cell_fwd") >= 10 and len(fns) <= 40, (fns.count("fs_zoom_cell_fwd"), len(fns))
        if cells == "0":
            assert "fs_zoom_cell_fwd" not in fns
    # every other way the same plan can be issued
    with torch.no_grad():
        eng.input.copy_(x.cuda())
        eng.output.zero_()
        eng._launch_all()                                   # direct launches, one stream
        torch.cuda.synchronize()
        _check(eng.output.clone(), want, dtype, "direct launch list")
        for lanes in sorted({1, eng.n_lanes}):
            g = eng._capture_once(lanes)
            eng.output.zero_()
            g.replay()
            torch.cuda.synchronize()
            _check(eng.output.clone(), want, dtype, "%d-lane hipGraph" % lanes)
        eng.output.zero_()
        eng._run_program()                                  # fs_exec_program_streams
        torch.cuda.synchronize()
        _check(eng.output.clone(), want, dtype, "launch program")


def test_teacher_engine_matches_eager_operator_path():
    """The frozen teacher of the distillation step (train/train.py:246-250) runs through the engine as well."""
    from fasterseg_amd import archs, engine
    from oracle.seeded import seeded_input, seeded_state
    net = archs.build_derived(0, training=False)
    net.load_state_dict(seeded_state(net.state_dict(), 777))
    net = net.cuda().eval()
    shape = (2, 3, 256, 512)
    x = seeded_input(shape, 5).cuda()
    with torch.no_grad():
        want = net(x).float().cpu()                         # per-operator path (parity-tested against the fixtures)
        for cells in ("0", "1"):
            eng = engine.InferenceEngine(net, shape, dtype=torch.float32, fuse_cells=cells)
            got = eng(x).float().cpu()
            assert float((got - want).abs().max()) <= 1e-3, (cells, float((got - want).abs().max()))


def test_engine_plan_file_reproduces_the_tuned_plan(tmp_path, monkeypatch):
    """FS_ENGINE_PLAN: the build that tunes writes its per-layer / per-cell choices, a later build reads them back instead of
    timing and issues the same launches (and the same logits)."""
    from fasterseg_amd import engine
    shape = (1, 3, 256, 512)
    net, x, want = _student(shape)
    monkeypatch.setenv("FS_ENGINE_PLAN", str(tmp_path / "plan.json"))
    with torch.no_grad():
        a = engine.InferenceEngine(net, shape, dtype=torch.bfloat16)
        ga = a(x.cuda()).clone()
        assert a._plan_path.startswith(str(tmp_path / "plan.json") + ".logits.bf16.") and os.path.exists(a._plan_path)
        b = engine.InferenceEngine(net, shape, dtype=torch.bfloat16)
        gb = b(x.cuda()).clone()
        # a plan is positional: another input shape (or network) has another signature and tunes afresh instead of replaying it
        c = engine.InferenceEngine(net, (1, 3, 128, 256), dtype=torch.bfloat16)
        torch.cuda.synchronize()
    assert b._plan_in is not None and a._plan_in is None
    assert c._plan_in is None and c._plan_path != a._plan_path
    assert sorted(c["label"] + c["fn"] for c in a.calls) == sorted(c["label"] + c["fn"] for c in b.calls)
    assert [t[4] for t in a.autotuned] == [t[4] for t in b.autotuned]
    assert torch.equal(ga, gb)
    _check(gb, want, torch.bfloat16, "engine rebuilt from its plan file")
