"""HBM traffic and MFMA utilisation of the C2 frame PER KERNEL FAMILY OF THE PLAN (conv3x3 and conv1x1 separately), from rocprofv3 --pmc
passes over tools/profile_frame.py (frames issued in plan order on one stream: the k-th dispatch of a frame is the k-th launch of the plan).

    python tools/pmc_frame.py <plan.json> <out.json> fetch=<counter_collection.csv> write=<...> [mfma=<...>]

Per family: launches / frame, algorithmic bytes (plan), measured HBM bytes = 2 x FETCH_SIZE KiB (gfx950 half-count of 16-byte coalesced
loads, MI355X_MICROARCH.md section HBM) + WRITE_SIZE KiB, their ratio, and mfma_util = sum SQ_VALU_MFMA_BUSY_CYCLES / (max-per-XCD
GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs).  The fabric-side counters include Infinity-Cache hits: an upper bound on DRAM traffic at this size.
"""
import collections, csv, json, re, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"[<(].*$", "", name)
    return name.replace("fs::", "")


def dispatches(path):
    """-> [(start, kernel, {counter: value})] ordered by start; counters summed over instances, GRBM_GUI_ACTIVE as the max."""
    d = {}
    for r in csv.DictReader(open(path)):
        key = r["Dispatch_Id"]
        e = d.setdefault(key, [int(r["Start_Timestamp"]), short(r["Kernel_Name"]), collections.defaultdict(float)])
        v = float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            e[2][r["Counter_Name"]] = max(e[2][r["Counter_Name"]], v)
        else:
            e[2][r["Counter_Name"]] += v
    return sorted(d.values(), key=lambda e: e[0])


def per_family(plan, rows):
    stems = [i for i, r in enumerate(rows) if r[1].startswith("stem")]
    frames = [rows[a:b] for a, b in zip(stems[:-1], stems[1:])]
    frames = frames[len(frames) // 4:]
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    used = 0
    for fr in frames:
        k, per = 0, []
        for c in plan:
            if k >= len(fr):
                break
            acc = dict(fr[k][2]); k += 1
            if k < len(fr) and fr[k][1] == "splitk_reduce_kernel" and c["fn"].startswith("fs_conv2d_fwd"):
                for n, v in fr[k][2].items():
                    acc[n] = acc.get(n, 0.0) + v
                k += 1
            per.append((c["family"], acc))
        if len(per) != len(plan) or k != len(fr):
            continue
        used += 1
        for f_, acc in per:
            fam[f_]["launches"] += 1
            for n, v in acc.items():
                fam[f_][n] += v
    return fam, used


def main():
    plan = json.load(open(sys.argv[1]))
    out = sys.argv[2]
    res = collections.defaultdict(dict)
    for spec in sys.argv[3:]:
        tag, path = spec.split("=", 1)
        fam, used = per_family(plan, dispatches(path))
        assert used, "no frame of %s matches the plan" % path
        for f_, c in fam.items():
            res[f_]["launches_per_frame"] = c["launches"] / used
            for n, v in c.items():
                if n != "launches":
                    res[f_][n + "_per_frame"] = v / used
    alg = collections.defaultdict(float)
    for c in plan:
        alg[c["family"]] += c["bytes"]
    total_m = total_a = 0.0
    for f_, r in res.items():
        r["alg_bytes_per_frame"] = alg[f_]
        if "FETCH_SIZE_per_frame" in r and "WRITE_SIZE_per_frame" in r:
            r["hbm_bytes_per_frame"] = 1024 * (2 * r["FETCH_SIZE_per_frame"] + r["WRITE_SIZE_per_frame"])
            r["hbm_bytes_per_launch"] = r["hbm_bytes_per_frame"] / r["launches_per_frame"]
            r["traffic_ratio"] = r["hbm_bytes_per_frame"] / max(alg[f_], 1.0)
            total_m += r["hbm_bytes_per_frame"]; total_a += alg[f_]
        if r.get("GRBM_GUI_ACTIVE_per_frame", 0) > 0:
            r["mfma_util"] = r["SQ_VALU_MFMA_BUSY_CYCLES_per_frame"] / (r["GRBM_GUI_ACTIVE_per_frame"] * 256 * 4)
    res["frame"] = {"hbm_bytes": total_m, "alg_bytes": total_a, "traffic_ratio": total_m / max(total_a, 1.0)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for f_, r in res.items():
        print("%-12s %s" % (f_, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if not k.endswith("SIZE_per_frame") and "CYCLES" not in k and "GUI" not in k}))


if __name__ == "__main__":
    main()
