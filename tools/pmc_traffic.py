"""Builds profiles/pmc_traffic.json (HBM bytes per launch per kernel family of the C2 inference frame) from two rocprofv3 --pmc
passes (FETCH_SIZE and WRITE_SIZE collected separately, as MI355X_MICROARCH.md prescribes: they do not fit one pass).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <dtype> [out.json]

Only the steady-state frames count: the engine build before them launches every tile / fusion candidate it times, so the
dispatches are cut into frames at the stem kernel and the LAST `FRAMES` complete frames of each pass are averaged.

Units / corrections (MI355X_MICROARCH.md section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide (16 B/lane) coalesced streaming read, so the read side is doubled for the conv / cell / resize kernels
(which only issue 16-byte loads); WRITE_SIZE is taken as is (it reproduces the known output sizes of the stem and the
logits kernels to 0.1 %).  Infinity-Cache hits are counted by these fabric-side counters, so at this working-set size
(everything < 256 MiB) the figure is an upper bound on true DRAM traffic.
"""
import collections
import csv
import json
import sys

FRAMES = 10
FAMILY = [("zoom_cell_kernel", "zoomcell"), ("conv3x3_halo_kernel", "conv"), ("conv_igemm_kernel", "conv"), ("stem_", "stem"),
          ("bilinear_fwd_nchw_kernel", "resize_nchw"), ("bilinear_argmax", "resize_argmax"), ("bilinear_fwd_kernel", "resize")]


def family(name):
    for key, fam in FAMILY:
        if key in name:
            return fam
    return None


def agg(path):
    """-> {family: [launches per frame, KiB per frame]} over the last FRAMES complete frames."""
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    stems = [i for i, r in enumerate(rows) if "stem_" in r[1]]
    if len(stems) < FRAMES + 2:
        raise SystemExit("not enough frames in %s" % path)
    a, b = stems[-FRAMES - 1], stems[-1]
    d = collections.defaultdict(lambda: [0, 0.0])
    for _, name, v in rows[a:b]:
        fam = family(name)
        if fam:
            d[fam][0] += 1
            d[fam][1] += v
    return {k: [v[0] / FRAMES, v[1] / FRAMES] for k, v in d.items()}, (b - a) / FRAMES


def main():
    (fetch, nf), (write, nw) = agg(sys.argv[1]), agg(sys.argv[2])
    dtype = sys.argv[3]
    out_path = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_traffic.json"
    res = {}
    for fam in fetch:
        n = fetch[fam][0]
        raw_f, raw_w = fetch[fam][1] / n * 1024, write.get(fam, [1, 0.0])[1] / max(1e-9, write.get(fam, [1, 0.0])[0]) * 1024
        res[fam] = {"launches_per_frame": n, "fetch_bytes_raw": round(raw_f), "write_bytes": round(raw_w),
                    "hbm_bytes_per_launch": round(2 * raw_f + raw_w)}
    try:
        with open(out_path) as f:
            doc = json.load(f)
    except Exception:
        doc = {}
    flat = {k: v["hbm_bytes_per_launch"] for k, v in res.items()}
    flat["conv3x3"] = flat.get("conv")          # bench.py family names (3x3 and 1x1 share the kernels)
    flat["conv1x1"] = flat.get("conv")
    doc[dtype] = flat
    doc[dtype + "_detail"] = res
    doc[dtype + "_frame"] = {"launches_per_frame": nf, "frames_averaged": FRAMES,
                             "hbm_bytes_per_frame": round(sum(v["hbm_bytes_per_launch"] * v["launches_per_frame"] for v in res.values()))}
    doc["_note"] = __doc__.strip().split("\n\n")[-1]
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc[dtype]), json.dumps(doc[dtype + "_frame"]))


if __name__ == "__main__":
    main()
