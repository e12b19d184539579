"""Launch census + in-step kernel timing: what a train step's dominant kernel achieves against its roofline.

The train steps replay most of their launches from hipGraphs, where individual kernels cannot be bracketed by events.  So
bench.py issues ONE more step eagerly (same kernels, one stream) with the C library's census at level 2: every kernel launch
carries a start/stop HIP event pair (hipExtLaunchKernelGGL - the dispatch's own begin/end timestamps, the quantity
rocprofv3's kernel trace reports) and the library accumulates launches and device time per kernel name and per convolution
geometry.  achieved = sum(algorithmic FLOPs of ALL launches of the family) / sum(their measured durations): nothing is
sampled, extrapolated or replayed warm (`roofline_timed`).  profiles/*_kernel_stats.csv (rocprofv3 --kernel-trace of the same
step, tools/prof_step.sh) hold the same table for cross-checking; tools/roofline_from_profile.py recomputes the fractions
from them.  The round-2 method (each geometry replayed alone, cache-warm, from a small hipGraph: `roofline`) is kept as the
"isolated" figure.
"""
import contextlib
import ctypes

import torch

from . import _lib
from . import kernels as K
from ._lib import FS_CONV_TRANSPOSED, CensusEntry, ConvDesc, KernelTime, call

IGEMM, HALO, WGRAD, STATS = 0, 1, 2, 0x100
FAMILY_NAMES = {IGEMM: "conv_igemm (fwd + dgrad)", HALO: "conv3x3_halo", WGRAD: "conv_wgrad"}
# The HBM-bound families of a train step, by kernel name (single and grouped forms): priced at their ALGORITHMIC bytes - one read or
# write of every operand map the pass needs (csrc: FS_NOTE_BYTES) - against the 8 TB/s HBM peak.  VERDICT r5 weak #6: in the round-5 C3
# step the BatchNorm family (30.9 ms) outweighed conv fwd + dgrad (26.2 ms) and the `roofline` object could not name it.
HBM_FAMILIES = {
    "batchnorm (train fwd + bwd)": ("bn_small_fwd", "bn_small_bwd", "bn_group_fwd", "bn_group_bwd", "chan_reduce", "bn_train_apply", "bn_bwd_apply",
                                    "bn_fwd_mixed", "bn_bwd_mixed"),          # (the mixed group launches of round 6 carry most of the family)
    "bilinear resample": ("bilinear_fwd", "bilinear_bwd"),
    "weighted sums / axpy": ("wsum", "wsum_bwd", "wsum_dot", "ew"),
}


def hbm_family_of(kernel_name):
    """Family of a census kernel name ('bn_small_fwd_group_kernel' -> 'batchnorm (train fwd + bwd)'), or None."""
    base = kernel_name[:-len("_kernel")] if kernel_name.endswith("_kernel") else kernel_name
    if base.endswith("_group"):
        base = base[:-len("_group")]
    for fam, names in HBM_FAMILIES.items():
        if base in names:
            return fam
    return None


@contextlib.contextmanager
def recording(level=1):
    """with recording() as rec: <issue one step>  ->  rec.entries = [(family, ConvDesc, count, ms)], rec.kernels = {name: (count, ms)},
    rec.kernel_bytes = {name: algorithmic HBM bytes of those launches}.  level 2 times every launch (ms is 0 at level 1)."""
    lib = _lib.lib()

    class _Rec:
        entries = []
        kernels = {}
        kernel_bytes = {}
    rec = _Rec()
    lib.fs_census_enable(level)
    try:
        yield rec
    finally:
        if level > 1:
            torch.cuda.synchronize()
        lib.fs_census_enable(0)
        n = lib.fs_census_read(None, 0)
        buf = (CensusEntry * max(n, 1))()
        n = min(n, lib.fs_census_read(ctypes.cast(buf, ctypes.c_void_p), n))
        out = []
        for i in range(n):
            d = ConvDesc()
            ctypes.memmove(ctypes.byref(d), ctypes.byref(buf[i].desc), ctypes.sizeof(ConvDesc))
            out.append((int(buf[i].family), d, int(buf[i].count), float(buf[i].ms)))
        rec.entries = out
        nk = lib.fs_census_read_kernels(None, 0)
        kb = (KernelTime * max(nk, 1))()
        nk = min(nk, lib.fs_census_read_kernels(ctypes.cast(kb, ctypes.c_void_p), nk))
        rec.kernels = {kb[i].name.decode(): (int(kb[i].count), float(kb[i].ms)) for i in range(nk)}
        rec.kernel_bytes = {kb[i].name.decode(): float(kb[i].bytes) for i in range(nk)}     # algorithmic HBM bytes (0: not priced)


def conv_flops(d):
    """ALGORITHMIC multiply-adds x 2.  A FS_CONV_TRANSPOSED descriptor is the data gradient of a stride-2 convolution written as a
    stride-1 convolution over the zero-inserted grid (N x Ho x Wo output pixels, (H, W) = the forward OUTPUT map): its algorithmic work
    is the forward convolution's, 2 N H W Cin Cout R S - not the 4x larger dense count of the zero-inserted geometry;
    conv_igemm2.hip executes exactly these (parity classes), conv_igemm.hip's zero-insertion form executes 4x as many on zeros."""
    if d.flags & FS_CONV_TRANSPOSED:
        return 2.0 * d.N * d.H * d.W * d.Cout * d.Cin * d.R * d.S
    return 2.0 * d.N * d.Ho * d.Wo * d.Cout * d.Cin * d.R * d.S


def conv_bytes(d):
    es = 2 if d.dtype == _lib.FS_BF16 else 4
    in_px = d.N * (d.vr_H * d.vr_W if d.vr_H > 0 else d.H * d.W)
    return es * (in_px * d.Cin + d.Cout * d.R * d.S * d.Cin + d.N * d.Ho * d.Wo * d.Cout)


def _graph_time_ms(fn, reps=20, rounds=3):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(ctypes.c_void_p(side.cuda_stream))
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(reps):
                fn(st)
        g.replay()
        best = None
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1) / reps
            best = t if best is None or t < best else best
    torch.cuda.current_stream().wait_stream(side)
    return best


def time_entry(family, d, device="cuda"):
    """Device time (ms) of one launch of this geometry on random operands."""
    dt = torch.bfloat16 if d.dtype == _lib.FS_BF16 else torch.float32
    fam = family & 0xff
    in_h, in_w = (d.vr_H, d.vr_W) if d.vr_H > 0 else (d.H, d.W)
    x = torch.randn(d.N * in_h * in_w * d.x_cs, device=device).to(dt)
    y = torch.empty(d.N * d.Ho * d.Wo * max(d.y_cs, d.Cout), dtype=dt, device=device)
    stats = torch.zeros(2 * d.Cout, dtype=torch.float32, device=device) if family & STATS else None
    dd = ConvDesc()
    ctypes.memmove(ctypes.byref(dd), ctypes.byref(d), ctypes.sizeof(ConvDesc))
    if fam == WGRAD:
        dy = torch.randn(d.N * d.Ho * d.Wo * d.y_cs, device=device).to(dt)
        dw = torch.zeros(d.Cout * d.R * d.S * d.Cin, dtype=torch.float32, device=device)        # [O][R][S][I]
        o_s, i_s, t_s = d.R * d.S * d.Cin, 1, d.Cin
        return _graph_time_ms(lambda st: call("fs_conv2d_wgrad_strided", st, ctypes.byref(dd), K._p(x), K._p(dy), K._p(dw), o_s, i_s, t_s))
    if fam == HALO:
        n = _lib.lib().fs_packed_weight_frag_elems(d.Cout, d.Cin, d.dtype)
        w = (torch.randn(n, device=device) * 0.05).to(dt)
        return _graph_time_ms(lambda st: call("fs_conv3x3_s1_fwd", st, ctypes.byref(dd), K._p(x), K._p(w), None, None, K._p(y), K._p(stats)))
    rows = max(d.w_os, d.R * d.S * max(d.w_ts, d.Cin)) if d.w_os else d.R * d.S * d.Cin
    # two-segment descriptors (fused pairs) read filter rows n_jump further / k_jump elements further: size the bank for them
    n_rows = d.Cout + (max(d.n_jump, 0) if d.n_seg > 0 else 0)
    w = (torch.randn(n_rows * rows + (abs(d.k_jump) if d.k_seg > 0 else 0), device=device) * 0.05).to(dt)
    ws = torch.empty(K.WORKSPACE_BYTES, dtype=torch.uint8, device=device)
    return _graph_time_ms(lambda st: call("fs_conv2d_fwd_ws", st, ctypes.byref(dd), K._p(x), K._p(w), None, None, K._p(y), K._p(stats),
                                          K._p(ws), K.WORKSPACE_BYTES))


def roofline(entries, dtype_name, peak_tflops, peak_hbm_gbs=8000.0, max_shapes=160, coverage=0.95):
    """Per family: launches/step, algorithmic FLOPs and bytes, measured time; returns (dominant-family roofline dict, families).
    Shapes are timed in descending order of their share of the step's FLOPs until `coverage` of all FLOPs (or max_shapes) is
    reached; the remainder is reported as un-timed (`flops_coverage`), never extrapolated."""
    fams = {}
    entries = [e[:3] for e in entries]
    order = sorted(entries, key=lambda e: -conv_flops(e[1]) * e[2])
    total_flops = sum(conv_flops(d) * c for _, d, c in order) or 1.0
    kept, acc = [], 0.0
    for e in order:
        if len(kept) >= max_shapes or acc >= coverage * total_flops:
            break
        kept.append(e)
        acc += conv_flops(e[1]) * e[2]
    skipped_launches = sum(c for _, _, c in order[len(kept):])
    for family, d, count in kept:
        ms = time_entry(family, d)
        f = fams.setdefault(family & 0xff, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, shapes=0))
        f["ms"] += ms * count
        f["flops"] += conv_flops(d) * count
        f["bytes"] += conv_bytes(d) * count
        f["launches"] += count
        f["shapes"] += 1
    if not fams:
        return None, {}
    dom_id, dom = max(fams.items(), key=lambda kv: kv[1]["ms"])
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    roof = {"kernel": FAMILY_NAMES[dom_id], "bound": "mfma", "achieved": round(ach, 2), "peak": peak_tflops, "unit": "TFLOP/s",
            "frac": round(ach / peak_tflops, 4), "launches_per_step": dom["launches"], "distinct_shapes": dom["shapes"],
            "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 3),
            "alg_flops_per_launch": dom["flops"] / dom["launches"], "alg_bytes_per_launch": dom["bytes"] / dom["launches"],
            "achieved_GBps": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1), "traffic": None,
            "flops_coverage": round(acc / total_flops, 4), "untimed_small_launches": skipped_launches,
            "method": "launch census of one step x per-shape HIP-event timing (fasterseg_amd/census.py)"}
    families = {FAMILY_NAMES[k]: {"ms_per_step": round(v["ms"], 3), "launches": v["launches"], "shapes": v["shapes"],
                                  "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in fams.items()}
    return roof, families


def roofline_timed(rec, dtype_name, peak_tflops, peak_hbm_gbs=8000.0, extra_entries=()):
    """`roofline` object + family table from a level-2 recording of one step: EVERY launch of every convolution family with
    its own measured duration.  extra_entries: (family, desc, count, ms) of launches that ran outside the recording (the frozen
    teacher's engine plan, timed by InferenceEngine.profile_in_frame)."""
    fams = {}
    for family, d, count, ms in list(rec.entries) + list(extra_entries):
        f = fams.setdefault(family & 0xff, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, shapes=0))
        f["ms"] += ms
        f["flops"] += conv_flops(d) * count
        f["bytes"] += conv_bytes(d) * count
        f["launches"] += count
        f["shapes"] += 1
    fams = {k: v for k, v in fams.items() if v["ms"] > 0}
    if not fams:
        return None, {}, {}
    kernel_ms = sum(ms for _, ms in rec.kernels.values()) + sum(e[3] for e in extra_entries)
    # the HBM-bound families: launches, device time and algorithmic bytes per kernel name, summed per family
    hbm = {}
    kbytes = getattr(rec, "kernel_bytes", {})
    for name, (count, ms) in rec.kernels.items():
        fam = hbm_family_of(name)
        if fam is None or ms <= 0:
            continue
        h = hbm.setdefault(fam, dict(ms=0.0, bytes=0.0, launches=0))
        h["ms"] += ms
        h["bytes"] += kbytes.get(name, 0.0)
        h["launches"] += count
    dom_id, dom = max(fams.items(), key=lambda kv: kv[1]["ms"])
    hbm_name, hbm_dom = max(hbm.items(), key=lambda kv: kv[1]["ms"]) if hbm else (None, None)
    method = ("one eager step at census level 2: every launch of the family timed by its own start/stop HIP event pair "
              "(hipExtLaunchKernelGGL) on the launch stream; achieved = sum of algorithmic work / sum of durations of ALL launches; the "
              "family is the largest by device time among ALL families of the step (convolutions AND BatchNorm / resample / weighted sums)")
    if hbm_dom is not None and hbm_dom["ms"] > dom["ms"] and hbm_dom["bytes"] > 0:
        ach = hbm_dom["bytes"] / (hbm_dom["ms"] * 1e-3) / 1e9
        roof = {"kernel": hbm_name, "bound": "hbm", "achieved": round(ach, 1), "peak": peak_hbm_gbs, "unit": "GB/s",
                "frac": round(ach / peak_hbm_gbs, 5), "launches_per_step": hbm_dom["launches"],
                "avg_launch_us": round(hbm_dom["ms"] / hbm_dom["launches"] * 1e3, 3),
                "share_of_kernel_time": round(hbm_dom["ms"] / max(kernel_ms, 1e-9), 4),
                "alg_bytes_per_launch": hbm_dom["bytes"] / hbm_dom["launches"], "traffic": None, "flops_coverage": 1.0, "method": method}
    else:
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        roof = {"kernel": FAMILY_NAMES[dom_id], "bound": "mfma", "achieved": round(ach, 2), "peak": peak_tflops, "unit": "TFLOP/s",
                "frac": round(ach / peak_tflops, 5), "launches_per_step": dom["launches"], "distinct_shapes": dom["shapes"],
                "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 3), "share_of_kernel_time": round(dom["ms"] / max(kernel_ms, 1e-9), 4),
                "alg_flops_per_launch": dom["flops"] / dom["launches"], "alg_bytes_per_launch": dom["bytes"] / dom["launches"],
                "achieved_GBps": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1), "traffic": None, "flops_coverage": 1.0, "method": method}
    families = {FAMILY_NAMES[k]: {"ms_per_step": round(v["ms"], 3), "launches": v["launches"], "shapes": v["shapes"],
                                  "avg_us": round(v["ms"] / v["launches"] * 1e3, 2),
                                  "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                  "frac_of_mfma_peak": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak_tflops, 5)} for k, v in fams.items()}
    for name, h in hbm.items():
        families[name] = {"ms_per_step": round(h["ms"], 3), "launches": h["launches"], "avg_us": round(h["ms"] / h["launches"] * 1e3, 2),
                          "alg_GB": round(h["bytes"] / 1e9, 4), "GBps": round(h["bytes"] / (h["ms"] * 1e-3) / 1e9, 1),
                          "frac_of_hbm_peak": round(h["bytes"] / (h["ms"] * 1e-3) / 1e9 / peak_hbm_gbs, 5)}
    # the step against the roofs: every family at its own roof (convolutions at the MFMA peak, the HBM-bound families at 8 TB/s)
    ideal_ms = sum(v["flops"] / (peak_tflops * 1e12) * 1e3 for v in fams.values()) + sum(h["bytes"] / (peak_hbm_gbs * 1e9) * 1e3 for h in hbm.values())
    roof["step_ideal_ms"] = round(ideal_ms, 4)
    kernels = {name: {"launches": c, "ms_per_step": round(ms, 3), "avg_us": round(ms / max(c, 1) * 1e3, 2)}
               for name, (c, ms) in sorted(rec.kernels.items(), key=lambda kv: -kv[1][1])}
    return roof, families, kernels
