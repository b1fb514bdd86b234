#!/usr/bin/env python3
"""Turn the database `rocprofv3 --kernel-trace --stats` leaves into a per-kernel summary CSV.
usage: scripts/kt_stats.py <dir-or-db> <out.csv>"""
import csv
import glob
import os
import sqlite3
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    db = src if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[0]
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                     "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for name, n, tot, avg, mn, mx in rows:
            w.writerow([name, n, int(tot), f"{avg:.1f}", int(mn), int(mx), f"{100.0 * tot / total:.3f}"])
    print(open(out).read())


if __name__ == "__main__":
    main()
