#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per GEMM launch of the 52-chunk batch (configs[2]) with the row-tile-per-XCD mapping off / on
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for x in 0 1; do
  rm -rf /tmp/gf
  XDTTS_GEMM_XCD=$x timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/gf -o f -- python tools/config3_batch.py 0 > /tmp/gf.log 2>&1
  grep "^wall" /tmp/gf.log
  python - $x <<'PY'
import sqlite3, glob, sys
db = sqlite3.connect(glob.glob("/tmp/gf/**/*.db", recursive=True)[0])
rows = list(db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%k_gemm_nt%' group by kernel_name"))
for n, c, v in rows: print("XDTTS_GEMM_XCD=%s %s calls %d avg fetch %.1f MB (x2 corrected)" % (sys.argv[1], "64x64" if "<64" in n else "32x32", c, 2 * v * 1024 / 1e6))
PY
done
