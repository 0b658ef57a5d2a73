"""Developer timing aid: persistent Griffin-Lim per-iteration time against the frames a workgroup owns
(F = 256 workgroups x TF frames), i.e. frames per second per launch at TF = 4 .. 8."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
pkg = importlib.import_module("xd-tts_amd")
voc = pkg.create_griffin_lim(seed=3)
rng = np.random.default_rng(0)
for F in (1000, 1024, 1280, 1536, 1792, 2048):
    S = np.abs(rng.standard_normal((513, F))).astype(np.float32)
    for _ in range(3):
        voc.infer_linear(S, iters=60)
    t = voc.last_timings()
    print("F=%4d  device %.3f ms  %.2f us per iteration  %.1f frames per us of iteration" % (
        F, t["iterations_ms"], t["iterations_ms"] * 1e3 / 61, F / (t["iterations_ms"] * 1e3 / 61)), flush=True)
