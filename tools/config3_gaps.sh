#!/bin/bash
# GPU idle gaps of the LAST configs[2] batch call (rocprofv3 --kernel-trace --memory-copy-trace + tools/gap_report.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_g
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_g -o g -- python tools/config3_batch.py 3 > /tmp/prof_g.log 2>&1
grep -A1 "^config3" /tmp/prof_g.log
python tools/gap_report.py $(find /tmp/prof_g -name "*.db" | head -1)
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/prof_g/**/*.db", recursive=True)[0])
ev = [(s, e, "copy " + str(n)) for n, s, e in db.execute("select name, start, end from memory_copies")]
ev += [(s, e, n.split("(")[0].split("::")[-1][:30]) for n, s, e in db.execute("select name, start, end from kernels")]
ev.sort()
tail = ev[-64:]
t0 = tail[0][0]
print("the last 64 events (end of the last call):")
for s, e, n in tail:
    print("%9.1f us  dur %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
