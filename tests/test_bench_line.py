"""The line bench.py prints must stay small enough for the driver to parse (round 3 printed 24 KB and the driver
recorded `parsed: null`).  Built here from a canned full result - round 3's own 24 KB line - without a GPU."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canned():
    with open(os.path.join(ROOT, "profiles", "r03_bench_default.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000                       # the input really is the oversized line
    train = {k: v for k, v in full["workloads"].items() if k != "C2_student_infer"}
    c2 = {k: v for k, v in full.items() if k != "workloads"}
    return c2, train


def test_line_is_small_and_round_trips(tmp_path):
    c2, train = _canned()
    line, detail = bench.build_line(c2, train, 1, 20, "bf16", str(tmp_path / "bench_detail.json"))
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT, len(text)
    assert "\n" not in text
    back = json.loads(text)
    assert back == line
    assert back["parity"]["pass"] is True
    # the contract's keys and the two objects the judge reads
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert set(back["config"]) == {"workload", "parallelism"}
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-3
    # one small object per workload, numbers only
    assert set(back["workloads"]) == {"C2_student_infer", "C3_supernet_pretrain", "C4_student_train", "C5_supernet_search"}
    for name, w in back["workloads"].items():
        assert len(json.dumps(w)) < 700, (name, len(json.dumps(w)))
        if name != "C2_student_infer":                       # C2's objects are the top-level ones
            assert w["parity"]["pass"] is True
            assert "roofline" in w and "cpu_baseline" in w
            assert "ms_per_step_fp32" in w
    # nothing was lost: the big tables live in the detail object
    assert "kernels_in_step" in detail["C3_supernet_pretrain"]
    assert "conv_autotune" in detail["C2_student_infer"]["config"]


def test_line_without_optional_parts():
    c2, _ = _canned()
    for k in ("roofline", "cpu_baseline", "class_map", "frame_roofline"):
        c2.pop(k, None)
    line, _ = bench.build_line(c2, {}, 2, None, "bf16", None)
    assert line["n_gpus"] == 2 and "roofline" not in line and line["detail"] is None
    assert len(json.dumps(line)) < 1500


def test_step_traffic_is_null_when_the_pmc_table_predates_the_kernels():
    """`roofline.traffic` comes from committed PMC passes (profiles/r0N_<workload>_pmc.json); a table taken before a kernel was replaced
    must not be quoted for the new kernel."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fam = bench.STEP_FAMILY_KERNELS["conv_igemm (fwd + dgrad)"]
    tables = [os.path.join(root, "profiles", "%s_c3_pmc.json" % r) for r in ("r06", "r05")]          # the newest one is what bench.py quotes
    tables = [t for t in tables if os.path.exists(t)]
    if not tables:
        import pytest
        pytest.skip("needs a committed PMC table")
    with open(tables[0]) as f:
        table = json.load(f)
    per = lambda k: table[k]["hbm_read_bytes_per_launch"] + table[k]["hbm_write_bytes_per_launch"]
    now = {"conv_igemm2_kernel": {"launches": 1800}, "conv_igemm2_group_kernel": {"launches": 636}, "splitk_reduce_kernel": {"launches": 10}}
    got = bench.step_traffic("c3", fam, now)
    # the table's per-launch bytes of every kernel it knows, weighted by the launches of the step that asks (not by the PMC run's own mix)
    known = {k: v for k, v in now.items() if k in table}
    want = sum(v["launches"] * per(k) for k, v in known.items()) / sum(v["launches"] for v in known.values())
    assert abs(got - want) <= 1.0 and got > 1e6
    future = {"conv_igemm3_kernel": {"launches": 2000}, "conv_igemm2_kernel": {"launches": 100}}
    assert bench.step_traffic("c3", fam + ("conv_igemm3_kernel",), future) is None      # mostly kernels the table has never seen: null
    assert bench.step_traffic("c3", fam) > 1e6                                            # (no launch table given: the caller vouches)


def test_line_carries_the_fp32_leg_of_c2():
    """VERDICT r4 missing #4: north_star's ">= 163 fps with logits within 1e-3" in ONE driver record."""
    c2, train = _canned()
    c2["fp32"] = {"value": 1042.3, "unit": "frames/s", "ms_per_step": 0.9594, "steps": 2000, "max_abs_err": 6.2e-6, "argmax_agreement": 1.0,
                  "vs_baseline": 6.36, "launches": 51, "note": "x" * 300}
    line, _ = bench.build_line(c2, train, 1, 20, "bf16", None)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert set(line["fp32"]) == {"value", "unit", "ms_per_step", "max_abs_err", "vs_baseline"}
    assert line["fp32"]["max_abs_err"] <= 1e-3 and line["fp32"]["value"] >= 163


def test_an_oversized_line_degrades_instead_of_aborting():
    """after minutes of measurement a long field must cost fields, never the whole result line."""
    c2, train = _canned()
    line, _ = bench.build_line(c2, train, 1, 20, "bf16", None)
    for w in line["workloads"].values():
        w["padding"] = "p" * 1500                              # something the builder never trims on its own
    line["cpu_baseline"]["sample"] = "s" * 3000
    fitted = bench.fit_line(json.loads(json.dumps(line)))
    text = json.dumps(fitted)
    assert len(text) < bench.LINE_LIMIT
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert k in fitted, k
    # and a line that fits is left alone
    small, _ = bench.build_line(c2, train, 1, 20, "bf16", None)
    assert bench.fit_line(json.loads(json.dumps(small))) == small


def test_cpu_baseline_sweeps_the_thread_count():
    """the stated CPU baseline is the oracle's best over a thread sweep, with the count it was measured at."""
    import time
    import torch
    seen = []

    def fn():
        seen.append(torch.get_num_threads())
        time.sleep(0.01 if torch.get_num_threads() == min(2, before) else 0.03)
    before = torch.get_num_threads()
    out = bench._time_cpu(fn, 1, 0.5, "unit test", threads=(1, 2, 1))
    assert torch.get_num_threads() == before                   # restored
    assert out["kind"] == "port" and out["unit"] == "images/s" and out["value"] > 0
    assert out["cores"] == min(2, before)                      # the fastest candidate
    assert "threads swept" in out["sample"]
    assert set(seen) <= {1, min(2, before)}
