"""Developer timing aid: us per decoder step of the persistent engine for 1 and 2 chunks in lock-step
(XDTTS_LIB selects the build)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
m = pkg.Tacotron2.synthetic()
o = pkg.default_opts(dropout_seed=1)
for B in (1, 2):
    chunks = [wl.synth_ids(95, seed=10 + b) for b in range(B)]
    best = 1e9
    for _ in range(5):
        m.infer_batch(chunks, opts=o, fixed_steps=[600] * B)
        t = m.last_timings()
        best = min(best, t["decoder_ms"] * 1e3 / t["steps"])
    print("B=%d: %.2f us per step (best of 5, 600 steps incl. launch set-up)" % (B, best), flush=True)
