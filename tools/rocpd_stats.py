#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats` on ROCm 7.2 writes
<name>_results.db) as a kernel-stats CSV: name, calls, total_ns, avg_ns, min_ns, max_ns, pct."""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for name, n, tot, avg, mn, mx in rows:
            if len(name) > 120:
                name = name[:117] + "..."
            w.writerow([name, n, int(tot), round(avg, 1), int(mn), int(mx), round(100.0 * tot / total, 3)])
    print(f"{len(rows)} kernels, total {total / 1e6:.1f} ms -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
