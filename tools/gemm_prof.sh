#!/bin/bash
# per-launch durations of the post-net GEMMs (tools/gemm_bench.py under rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F=${1:-800}
rm -rf /tmp/prof_g
timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_g -o g -- python tools/gemm_bench.py $F 6 > /tmp/gemm_prof.log 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/prof_g/**/*.db", recursive=True)[0])
rows = list(db.execute("select name, end-start from kernels where name like '%k_gemm_nt%' order by start"))
per = len(rows) // 6
last = rows[-per:]
print("post-net launches of the last call (us):", [(("64" if "<64" in n else "32"), round(t / 1e3, 1)) for n, t in last], "sum %.1f" % (sum(t for _, t in last) / 1e3))
PY
