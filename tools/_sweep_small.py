import importlib, os, sys
sys.path.insert(0, "/root/repo")
import torch
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
m = pkg.Tacotron2.synthetic()
steps = 200
out = []
for B in [int(x) for x in (sys.argv[1:] or ["8","16","24","32"])]:
    chunks = [wl.synth_ids(60 + (7 * b) % 40, seed=10 + b) for b in range(B)]
    o = pkg.default_opts(dropout_seed=1)
    for _ in range(3):
        m.infer_batch(chunks, opts=o, fixed_steps=[steps] * B)
    out.append("%d:%.1f" % (B, m.last_timings()["decoder_ms"] * 1e3 / steps))
print(" ".join(out))
