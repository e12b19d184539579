"""Per-kernel table out of rocprofv3 --pmc passes (one counter set per pass, collected with --kernel-trace only).

    python tools/pmc_table.py out.json [--window BEGIN_NS END_NS STEPS] name=counter_collection.csv ...

Every pass contributes the per-kernel SUM of its counters (Counter_Name column) over the dispatches inside the window (default: all),
divided by STEPS.  Derived columns (MI355X_MICROARCH.md, sections HBM / PMC slots):
  hbm_read_bytes  = 2 x FETCH_SIZE x 1024   (FETCH_SIZE is in KiB and on gfx950 reports half of the bytes of 16-byte coalesced loads -
                                             all loads of these kernels are 16-byte vectors)
  hbm_write_bytes = WRITE_SIZE x 1024
  mfma_util       = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs)   (gfx94x MfmaUtil formula; GRBM_GUI_ACTIVE is
                    reported once per XCD: the MAXIMUM over a dispatch's instances is taken, the SQ counter is summed over its instances)
Infinity-Cache hits are counted by the fabric-side FETCH/WRITE counters: with working sets < 256 MiB the byte figures are an upper bound
on DRAM traffic.
"""
import collections, csv, json, re, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"[<(].*$", "", name)
    return name.replace("fs::", "")


def main():
    args = sys.argv[1:]
    out = args.pop(0)
    lo, hi, steps = None, None, 1
    if args and args[0] == "--window":
        lo, hi, steps = int(args[1]), int(args[2]), int(args[3])
        args = args[4:]
    table = collections.defaultdict(lambda: collections.defaultdict(float))
    for spec in args:
        _, path = spec.split("=", 1)
        seen = collections.defaultdict(set)
        gui = {}                                     # (kernel, dispatch) -> max GRBM_GUI_ACTIVE over its per-XCD instances
        for r in csv.DictReader(open(path)):
            ts = int(r.get("Start_Timestamp", 0) or 0)
            if lo is not None and not (lo <= ts <= hi):
                continue
            k = short(r["Kernel_Name"])
            disp = r.get("Dispatch_Id", ts)
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                gui[(k, disp)] = max(gui.get((k, disp), 0.0), float(r["Counter_Value"]))
            else:
                table[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(disp)
        for (k, _), v in gui.items():
            table[k]["GRBM_GUI_ACTIVE"] += v
        for k, s in seen.items():
            table[k]["dispatches"] = max(table[k]["dispatches"], len(s))
    res = {}
    for k, c in table.items():
        n = c["dispatches"] / steps
        row = {"launches": round(n, 2)}
        for name, v in c.items():
            if name != "dispatches":
                row[name] = v / steps
        if "FETCH_SIZE" in c:
            row["hbm_read_bytes_per_launch"] = round(2 * 1024 * c["FETCH_SIZE"] / max(c["dispatches"], 1))
        if "WRITE_SIZE" in c:
            row["hbm_write_bytes_per_launch"] = round(1024 * c["WRITE_SIZE"] / max(c["dispatches"], 1))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            row["mfma_util"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 256 * 4), 5)
        res[k] = row
    with open(out, "w") as f:
        json.dump(dict(sorted(res.items(), key=lambda kv: -kv[1]["launches"])), f, indent=1)
    for k, row in sorted(res.items(), key=lambda kv: -kv[1]["launches"])[:24]:
        print("%-32s %s" % (k[:32], {a: (round(b, 4) if isinstance(b, float) else b) for a, b in row.items() if a not in ("FETCH_SIZE", "WRITE_SIZE")}))


if __name__ == "__main__":
    main()
