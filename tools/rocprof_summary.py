#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel stats table,
the same columns `--stats` prints: calls, total, average, min, max, share."""
import sqlite3
import sys


def main(db_path):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    print("%-100s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for r in rows:
        print("%-100s %8d %12.3f %10.2f %10.2f %10.2f %6.1f" % (r[0][:100], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))


if __name__ == "__main__":
    main(sys.argv[1])
