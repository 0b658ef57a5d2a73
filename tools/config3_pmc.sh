#!/bin/bash
# HBM traffic per kernel of the 52-chunk batch (BASELINE configs[2]) alone: two PMC passes (FETCH_SIZE, WRITE_SIZE), each with
# --kernel-trace only -> gpurun_out/rNN/config3_pmc.txt.   usage: tools/config3_pmc.sh NN GIT_HEAD
R=${1:-02}; HEAD=${2:-unknown}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r$R; mkdir -p $OUT
rm -rf /tmp/c3f /tmp/c3w
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/c3f -o f -- python tools/config3_batch.py 0 > $OUT/config3_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/c3w -o w -- python tools/config3_batch.py 0 > $OUT/config3_pmc_write.log 2>&1
python - "$(find /tmp/c3f -name '*.db' | head -1)" "$(find /tmp/c3w -name '*.db' | head -1)" "$HEAD" > $OUT/config3_pmc.txt <<'PY'
import re, sqlite3, sys
def avg(db, counter):
    out = {}
    for name, n, a in sqlite3.connect(db).execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (counter,)):
        m = re.search(r"(k_\w+(<[^>]*>)?)", name)
        k = m.group(1) if m else name[:40]
        c, v = out.get(k, (0, 0.0))
        out[k] = (c + n, (v * c + a * n) / (c + n))
    return out
f, w = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
print("BASELINE configs[2]: 32 utterances -> 52 chunks in one lock-step batch (tools/config3_batch.py, one call), git %s" % sys.argv[3])
print("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes); averages per dispatch; gfx950 correction: fetch x 2 (MI355X_MICROARCH.md)")
print("%-28s %8s %14s %14s %16s" % ("kernel", "calls", "fetch KB", "write KB", "traffic MB (corr.)"))
tot = 0.0
STEP = ("k_lstm_mfma", "k_att_lstm_attention", "k_attention_b", "k_softmax_ctx", "k_prenet_b", "k_qenergy")
for k in sorted(f, key=lambda k: -(2 * f[k][1] + w.get(k, (0, 0.0))[1]) * f[k][0]):
    n, fe = f[k]
    wr = w.get(k, (0, 0.0))[1]
    tr = (2 * fe + wr) * 1024 / 1e6
    if k.startswith(STEP): tot += tr * n
    print("%-28s %8d %14.1f %14.1f %16.2f" % (k, n, fe, wr, tr))
it = max([n for k, (n, _) in f.items() if k.startswith("k_lstm_mfma<2560")] or [f.get("k_prenet_b", (2, 0))[0] - 1])  # one decoder-LSTM launch per iteration
print("the kernels of a decoder step: %.1f MB per lock-step iteration (%d iterations) against 73 MB algorithmic (71.3 MB of LSTM weights + per-chunk state)" % (tot / max(it, 1), it))
print("(FETCH_SIZE counts what leaves the L2s, Infinity-Cache hits included: the LSTM weights are re-read every step, the 0.4 - 0.65 MB activation operand once per XCD;")
print(" the small kernels' traffic is the partial-mel rows, the encoder memory and the location / energy arrays)")
PY
cat $OUT/config3_pmc.txt
