#!/usr/bin/env python
"""List the launches of one kernel from a rocprofv3 rocpd database, grouped by grid shape:
count, mean/total duration — to see WHICH call sites of a generic kernel cost the time."""
import sqlite3
import sys


def main(db, pattern):
    c = sqlite3.connect(db)
    rows = c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(duration), sum(duration) from kernels "
                     "where name like ? group by name, grid_x, grid_y, grid_z order by sum(duration) desc", (f"%{pattern}%",)).fetchall()
    tot = sum(r[7] for r in rows) or 1
    print(f"{'kernel':40s} {'grid(x,y,z)/wg':>28s} {'n':>5s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for name, gx, gy, gz, wx, n, avg, s in rows[:40]:
        print(f"{name[:40]:40s} {str((gx // wx, gy, gz)):>28s} {n:5d} {avg / 1e3:10.1f} {s / 1e6:10.2f} {100 * s / tot:6.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
