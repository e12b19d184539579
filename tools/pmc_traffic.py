"""Builds profiles/pmc_traffic.json (HBM bytes per launch per kernel family) from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE collected separately, as MI355X_MICROARCH.md prescribes: they do not fit one pass).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <dtype> [out.json]

Units / corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide (16 B/lane) coalesced streaming read, so the read side is doubled for the conv / resize kernels
(which only issue 16-byte loads); WRITE_SIZE is taken as is (it reproduces the known output sizes of the stem and the
logits kernels to 0.1 %).  Infinity-Cache hits are counted by these fabric-side counters, so at this working-set size
(everything < 256 MiB) the figure is an upper bound on true DRAM traffic.
"""
import collections
import csv
import json
import sys

FAMILY = [("conv3x3_halo_kernel", "conv"), ("conv_igemm_kernel", "conv"), ("stem_conv_kernel", "stem"), ("stem_lds_kernel", "stem"),
          ("bilinear_fwd_nchw_kernel", "resize_nchw"), ("bilinear_fwd_kernel", "resize")]


def family(name):
    for key, fam in FAMILY:
        if key in name:
            return fam
    return None


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        fam = family(r["Kernel_Name"])
        if fam:
            d[fam][0] += 1
            d[fam][1] += float(r["Counter_Value"])
    return d


def main():
    fetch, write, dtype = agg(sys.argv[1]), agg(sys.argv[2]), sys.argv[3]
    out_path = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_traffic.json"
    res = {}
    for fam in fetch:
        n = fetch[fam][0]
        raw_f, raw_w = fetch[fam][1] / n * 1024, write[fam][1] / max(1, write[fam][0]) * 1024
        res[fam] = {"launches_sampled": n, "fetch_bytes_raw": round(raw_f), "write_bytes": round(raw_w),
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
    doc["_note"] = __doc__.strip().split("\n\n")[-1]
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc[dtype]))


if __name__ == "__main__":
    main()
