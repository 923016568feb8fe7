"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into per-kernel statistics:

    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db profiles/r01_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = float(sum(r[2] for r in rows))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, n, t, avg, mn, mx in rows:
            w.writerow([name, n, t, "%.1f" % avg, "%.2f" % (100.0 * t / tot), mn, mx])
    print("wrote %s: %d kernels, total %.3f ms" % (out, len(rows), tot / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
