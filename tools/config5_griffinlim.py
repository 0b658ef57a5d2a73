"""Measurement aid for BASELINE.json configs[4]: Griffin-Lim only, 1000-frame input, 30/60/120 iterations."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from test_gpu_griffinlim_more import chirps
pkg = importlib.import_module("xd-tts_amd")
orc = oracle.Oracle("f32")
F = 1000
sig = chirps(256 * (F - 1))
spec = orc.stft(sig)
S = np.hypot(spec[..., 0], spec[..., 1]).astype(np.float32)
voc = pkg.create_griffin_lim(seed=3)
for iters in (30, 60, 120):
    for _ in range(3):
        a = voc.infer_linear(S, iters=iters)
    t = voc.last_timings()
    per_iter_us = t["iterations_ms"] * 1e3 / (iters + 1)
    bytes_alg = 12308.0 * F * iters
    print("config5: F=%d iters=%3d  device %.3f ms (%.2f us per iteration)  %.0f samples/s  algorithmic %.1f GB/s (%.4f of 8 TB/s)" % (
        F, iters, t["iterations_ms"], per_iter_us, a.size / (t["iterations_ms"] * 1e-3), bytes_alg / (t["iterations_ms"] * 1e-3) / 1e9, bytes_alg / (t["iterations_ms"] * 1e-3) / 8e12))
t0 = time.perf_counter(); ref = orc.griffinlim(S, seed=3, iters=30); t1 = time.perf_counter()
print("cpu oracle (1 thread) 30 iterations: %.2f s" % (t1 - t0))
