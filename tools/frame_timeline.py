"""One inference frame out of a rocprofv3 --kernel-trace CSV: every launch of the last complete frame (delimited by the stem
kernel) ordered by start time, with its offset, duration, hardware queue and grid.  Usage:
    python tools/frame_timeline.py <kernel_trace.csv> <out.csv>"""
import csv, re, sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"],
                     int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])),
                     int(r["VGPR_Count"]), int(r["LDS_Block_Size"])))
rows.sort()
stems = [i for i, r in enumerate(rows) if "stem" in r[2]]
a, b = stems[-3], stems[-2]
frame = rows[a:b]
t0 = frame[0][0]


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("fs::", "").replace("unsigned short", "bf16")


with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["launch", "start_us", "dur_us", "queue", "blocks", "vgpr", "lds_bytes", "kernel"])
    for i, (s, e, n, q, g, v, l) in enumerate(frame):
        w.writerow([i, round((s - t0) / 1e3, 2), round((e - s) / 1e3, 2), q, g, v, l, short(n)])
busy = sum(e - s for s, e, *_ in frame)
print("frame: %d launches, span %.1f us (next frame starts at %.1f us), kernel-time sum %.1f us" %
      (len(frame), (max(r[1] for r in frame) - t0) / 1e3, (rows[b][0] - t0) / 1e3, busy / 1e3))
for i, (s, e, n, q, g, v, l) in enumerate(frame):
    print("%3d  +%7.1f us  %6.1f us  q%-3s %6d blk  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, q, g, short(n)[:70]))
