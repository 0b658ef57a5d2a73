#!/bin/bash
# Kernel timeline of one xdtts_synthesize_sequence call of 6 headline utterances (rocprofv3 --kernel-trace): per utterance, when its frame
# loop, its vocoder and the next utterance's encoder start and end -- do vocoder(u) and encoder(u + 1) run side by side?
# Repeated N times (separate processes): the sequence form has two modes (6.15 / 6.55 ms per utterance).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/seq_call.py <<'PY'
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
pkg = importlib.import_module("xd-tts_amd"); wl = importlib.import_module("xd-tts_amd.workloads")
m = pkg.Tacotron2.synthetic(); v = pkg.create_griffin_lim(iters=60, seed=0)
_i, chunks, _s = wl.config2(pkg)
sp = np.cumsum([len(c) for c in chunks]).astype(np.int64)
o = pkg.default_opts(fixed_frames_per_id=wl.FRAMES_PER_ID, dropout_seed=0)
utts = [wl.synth_ids(120, seed=1 + g) for g in range(8)]
pkg.synthesize_sequence(m, v, utts[:3], [sp] * 3, opts=o, want_mels=False)
t0 = time.perf_counter()
pkg.synthesize_sequence(m, v, utts, [sp] * 8, opts=o)
print("call: %.3f ms per utterance" % ((time.perf_counter() - t0) / 8 * 1e3), flush=True)
PY
for rep in $(seq 1 ${1:-4}); do
  rm -rf /tmp/prof_s
  timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_s -o s -- python /tmp/seq_call.py > /tmp/prof_s.log 2>&1
  grep "^call" /tmp/prof_s.log
  python - <<'PY'
import sqlite3, glob, re
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    return m.group(1)[:40] if m else n[:40]
db = sqlite3.connect(glob.glob("/tmp/prof_s/**/*.db", recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
ev = [(s, e, short(n), q) for n, s, e, q in db.execute("select name, start, end, %s from kernels order by start" % (qcol or "0"))]
# the last 8 two-chunk frame loops = the timed call
idx = [i for i, x in enumerate(ev) if x[2].startswith("k_decoder_persistent<2")][-8:]
if len(idx) < 6:
    print("kernel names:", sorted({x[2] for x in ev})[:30]); raise SystemExit
t0 = ev[idx[2]][0]
for u in (2, 3, 4):
    a, b = idx[u], idx[u + 1]
    print("utterance %d (queues: %s)" % (u, sorted({x[3] for x in ev[a:b]})))
    for s, e, n, q in ev[a:b]:
        if e - s > 20000 or n.startswith(("k_gl", "k_bilstm", "k_embed")):
            print("   %9.1f .. %9.1f us  %7.1f  q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
PY
done
