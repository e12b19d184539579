"""Turns a rocprofv3 (--kernel-trace --stats) results database into the per-kernel summary table kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_xxx/bench_results.db [frames] > profiles/rNN_xxx_kernel_stats.md
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else None
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
                     "group by name order by 6 desc").fetchall()
    total = sum(r[5] for r in rows)
    print("| kernel | calls | avg us | min us | max us | total ms | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, avg, mn, mx, tot in rows:
        print("| `%s` | %d | %.2f | %.2f | %.2f | %.3f | %.1f |" % (name[:140], n, avg / 1e3, mn / 1e3, mx / 1e3, tot / 1e6, 100.0 * tot / total))
    print()
    print("total kernel time: %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    if frames:
        fs = [r for r in rows if "fs::" in r[0]]
        print("fasterseg kernels per frame (%d frames): %.1f us, %.1f launches" % (
            frames, sum(r[5] for r in fs) / frames / 1e3, sum(r[1] for r in fs) / frames))


if __name__ == "__main__":
    main()
