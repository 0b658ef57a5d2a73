#!/usr/bin/env python3
"""Where the GPU idles during one configs[2] batch: gaps > 15 us between consecutive kernel / copy dispatches of the LAST
infer_batch call in a rocprofv3 rocpd database (--kernel-trace --memory-copy-trace).  usage: gap_report.py DB"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
ev = [(s, e, n.split("(")[0].split("::")[-1][:40]) for n, s, e in db.execute("select name, start, end from kernels")]
if "memory_copies" in tabs:
    ev += [(s, e, "copy " + str(n)) for n, s, e in db.execute("select name, start, end from memory_copies")]
ev.sort()
# last batch = events after the last big gap (> 2 ms)
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - ev[i - 1][1] > 2_000_000:
        cut = i
ev = ev[cut:]
t0 = ev[0][0]
print("last call: %d events, %.2f ms from first start to last end" % (len(ev), (ev[-1][1] - t0) / 1e6))
busy = sum(e - s for s, e, _ in ev)
print("sum of durations %.2f ms" % (busy / 1e6))
tot = 0.0
for i in range(1, len(ev)):
    g = ev[i][0] - max(x[1] for x in ev[max(0, i - 4):i])
    if g > 15000:
        tot += g / 1e3
        print("gap %8.1f us  at %8.2f ms  after %-40s before %s" % (g / 1e3, (ev[i][0] - t0) / 1e6, ev[i - 1][2], ev[i][2]))
print("gaps > 15 us: %.1f us in total" % tot)
