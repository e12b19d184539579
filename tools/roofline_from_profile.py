"""Recomputes the `roofline` fractions bench.py prints from the rocprofv3 tables committed under profiles/.

Train steps (C3 / C4 / C5):   python tools/roofline_from_profile.py step <bench.json> <C3|C4|C5> <profiles/rNN_cX_*_kernel_stats.csv>
    FLOPs per kernel family come from the launch census in the bench line (kernel_families: launches, TFLOPs x ms = FLOPs - pure
    geometry); durations come from the rocprofv3 per-step kernel table (tools/prof_step.sh: Name, CallsPerStep, AverageNs, MsPerStep).
    conv_igemm family = conv_igemm_kernel + splitk_reduce_kernel, conv_wgrad = wgrad_kernel, conv3x3_halo = conv3x3_halo_kernel.
C2 frame:                     python tools/roofline_from_profile.py frame <plan.json> <kernel_trace.csv> [bench.json]
    plan.json and the trace come from tools/profile_frame.py under rocprofv3 --kernel-trace (frames issued in plan order on one stream):
    the k-th kernel between two stem kernels is the k-th launch of the plan (+ its split-K reduction when one follows).
Prints, per family: launches, FLOPs, profiler time, TFLOP/s, fraction of the dense bf16 MFMA peak (2500 TFLOP/s) - and beside it the
fraction bench.py printed from its own HIP-event timing, with the ratio of the two.
"""
import csv, json, re, sys

PEAK = {"bf16": 2500.0, "fp32": 157.3}
FAMILY_KERNELS = {"conv_igemm (fwd + dgrad)": ("conv_igemm_kernel", "splitk_reduce_kernel"), "conv_wgrad": ("wgrad_kernel",),
                  "conv3x3_halo": ("conv3x3_halo_kernel",)}


def bench_line(path):
    return json.loads([l for l in open(path) if l.startswith('{"metric"')][-1])


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"[<(].*$", "", name)
    return name.replace("fs::", "")


def step(bench, which, stats_csv):
    d = bench_line(bench)
    key = [k for k in d["workloads"] if k.startswith(which)][0]
    w = d["workloads"][key]
    peak = PEAK[w.get("dtype", d["dtype"])]
    prof = {}
    with open(stats_csv) as f:
        for r in csv.DictReader(f):
            n = short(r["Name"])
            a = prof.setdefault(n, [0.0, 0.0])
            a[0] += float(r["CallsPerStep"]); a[1] += float(r["MsPerStep"])
    print("%s: step %.2f ms (bench); profiler kernel time %.2f ms over %.0f launches" % (key, w["ms_per_step"], sum(v[1] for v in prof.values()),
                                                                                     sum(v[0] for v in prof.values())))
    print("%-28s %9s %10s %10s %10s %9s | %9s %7s" % ("family", "launches", "GFLOP", "prof ms", "TFLOP/s", "frac", "bench", "ratio"))
    for fam, v in w["kernel_families"].items():
        flops = v["TFLOPs"] * 1e12 * v["ms_per_step"] * 1e-3
        ms = sum(prof.get(k, [0, 0])[1] for k in FAMILY_KERNELS[fam])
        calls = sum(prof.get(k, [0, 0])[0] for k in FAMILY_KERNELS[fam][:1])
        if ms <= 0:
            continue
        tf = flops / (ms * 1e-3) / 1e12
        print("%-28s %9.0f %10.1f %10.3f %10.2f %9.5f | %9.5f %7.3f" % (fam, calls, flops / 1e9, ms, tf, tf / peak, v["frac_of_mfma_peak"],
                                                                     tf / peak / v["frac_of_mfma_peak"]))
    r = w["roofline"]
    print("bench roofline: %s achieved %.2f TFLOP/s, frac %.5f" % (r["kernel"], r["achieved"], r["frac"]))


def frame(plan_json, trace_csv, bench=None):
    plan = json.load(open(plan_json))
    rows = []
    with open(trace_csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    stems = [i for i, r in enumerate(rows) if r[2].startswith("stem")]
    frames = [rows[a:b] for a, b in zip(stems[:-1], stems[1:])]
    frames = frames[len(frames) // 4:]                       # skip warm-up frames
    fam = {}
    used = 0
    for fr in frames:
        k = 0
        ok = True
        per = []
        for c in plan:
            if k >= len(fr):
                ok = False
                break
            dur = fr[k][1] - fr[k][0]
            k += 1
            if k < len(fr) and fr[k][2] == "splitk_reduce_kernel" and c["fn"].startswith("fs_conv2d_fwd"):
                dur += fr[k][1] - fr[k][0]
                k += 1
            per.append((c["family"], dur, c["flops"], c["bytes"]))
        if not ok or k != len(fr):
            continue
        used += 1
        for f_, dur, fl, by in per:
            a = fam.setdefault(f_, [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += dur; a[2] += fl; a[3] += by
    assert used, "no frame of the trace matches the plan (%d launches)" % len(plan)
    b = bench_line(bench) if bench else None
    print("%d frames matched the plan of %d launches" % (used, len(plan)))
    print("%-12s %9s %10s %10s %10s %9s %9s | %9s" % ("family", "launches", "GFLOP", "us/frame", "TFLOP/s", "frac", "GB/s", "bench"))
    for f_, (n, ns, fl, by) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        us = ns / used / 1e3
        tf = fl / used / (us * 1e-6) / 1e12
        bf = ""
        if b and f_ in b.get("kernel_families", {}):
            bf = "%9.4f" % (b["kernel_families"][f_]["TFLOPs"] / PEAK["bf16"])
        print("%-12s %9.0f %10.2f %10.1f %10.2f %9.4f %9.1f | %s" % (f_, n / used, fl / used / 1e9, us, tf, tf / PEAK["bf16"], by / used / (us * 1e-6) / 1e9, bf))


if __name__ == "__main__":
    if sys.argv[1] == "step":
        step(*sys.argv[2:5])
    else:
        frame(*sys.argv[2:])
