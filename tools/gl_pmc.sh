#!/bin/bash
# HBM traffic of the persistent Griffin-Lim launch alone (BASELINE configs[4]: F = 1000, 60 iterations; 5 identical calls):
# two PMC passes (FETCH_SIZE, WRITE_SIZE), each with --kernel-trace only -> gpurun_out/rNN/gl_pmc.txt.   usage: tools/gl_pmc.sh NN GIT_HEAD
R=${1:-02}; HEAD=${2:-unknown}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r$R; mkdir -p $OUT
cat > /tmp/gl_one.py <<'PY'
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: F401
pkg = importlib.import_module("xd-tts_amd")
S = np.abs(np.random.default_rng(5).standard_normal((513, 1000))).astype(np.float32)
v = pkg.create_griffin_lim(seed=3)
for _ in range(5):
    v.infer_linear(S, iters=60)
print("device ms", v.last_timings())
PY
rm -rf /tmp/glf /tmp/glw
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/glf -o f -- python /tmp/gl_one.py > $OUT/gl_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/glw -o w -- python /tmp/gl_one.py > $OUT/gl_pmc_write.log 2>&1
python - "$(find /tmp/glf -name '*.db' | head -1)" "$(find /tmp/glw -name '*.db' | head -1)" "$HEAD" > $OUT/gl_pmc.txt <<'PY'
import sqlite3, sys
def avg(db, counter):
    rows = list(sqlite3.connect(db).execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (counter,)))
    return {r[0]: (r[1], r[2]) for r in rows if "k_gl_persistent" in r[0]}
f, w = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
alg = 12308.0 * 1000 * 60
print("Persistent Griffin-Lim launch alone: F = 1000 frames, 60 iterations + final ISTFT, one k_gl_persistent<4> dispatch per call (5 calls), git %s" % sys.argv[3])
print("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes); KB per dispatch; gfx950 correction: fetch x 2 (MI355X_MICROARCH.md)")
for k in f:
    n, fe = f[k]
    wr = w.get(k, (0, 0.0))[1]
    tr = 2 * fe * 1024 + wr * 1024
    print("k_gl_persistent<4>: dispatches %d  FETCH_SIZE %.1f KB  WRITE_SIZE %.1f KB  -> corrected traffic %.1f MB per dispatch" % (n, fe, wr, tr / 1e6))
    print("algorithmic bytes (SURVEY 8d: 12 308 B per frame per iteration) = %.1f MB  ->  traffic / algorithmic = %.2f" % (alg / 1e6, tr / alg))
print("(the state -- S, angles, previous spectrum -- is read once into LDS / registers and never written back; per iteration only the 768-sample overlaps")
print(" cross between neighbouring workgroups as 16-byte tagged granules {three samples, tag}, plus their polls)")
PY
cat $OUT/gl_pmc.txt
