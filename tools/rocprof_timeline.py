#!/usr/bin/env python3
"""Start/end times of consecutive kernel dispatches from a rocprofv3 rocpd database (--kernel-trace):
shows whether kernels of different streams overlap.  usage: rocprof_timeline.py DB [first] [count]"""
import sqlite3
import sys


def main(db_path, first=2000, count=30):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    extra = [c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols]
    q = "select name, start, end%s from kernels order by start limit %d offset %d" % ("".join(", " + c for c in extra), count, first)
    rows = list(db.execute(q))
    t0 = rows[0][1]
    print("%-44s %10s %10s %8s  %s" % ("kernel", "start_us", "end_us", "dur_us", " ".join(extra)))
    for r in rows:
        name = r[0].split("(")[0].split("::")[-1][:44]
        print("%-44s %10.2f %10.2f %8.2f  %s" % (name, (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, " ".join(str(x) for x in r[3:])))


if __name__ == "__main__":
    main(sys.argv[1], *[int(a) for a in sys.argv[2:4]])
