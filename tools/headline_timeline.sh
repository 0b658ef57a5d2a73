#!/bin/bash
# Every kernel / copy of the LAST headline call with the idle time in front of it (rocprofv3 --kernel-trace --memory-copy-trace).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_h
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_h -o h -- python tools/headline_call.py > /tmp/prof_h.log 2>&1
grep "^call" /tmp/prof_h.log
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/prof_h/**/*.db", recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
ev = [(s, e, (n.split("(anonymous namespace)::")[-1] if "(anonymous namespace)::" in n else n).split("(")[0][:44]) for n, s, e in db.execute("select name, start, end from kernels")]
if "memory_copies" in tabs:
    ev += [(s, e, "copy " + str(n)) for n, s, e in db.execute("select name, start, end from memory_copies")]
ev.sort()
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - ev[i - 1][1] > 5_000_000:
        cut = i
ev = ev[cut:]
t0 = ev[0][0]
print("last call: %d events, %.3f ms first start to last end, busy %.3f ms" % (len(ev), (ev[-1][1] - t0) / 1e6, sum(e - s for s, e, _ in ev) / 1e6))
for i, (s, e, n) in enumerate(ev):
    gap = (s - max(x[1] for x in ev[max(0, i - 4):i])) / 1e3 if i else 0.0
    print("%9.1f us  gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n))
PY
