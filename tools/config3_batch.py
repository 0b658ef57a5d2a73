"""Measurement aid for BASELINE.json configs[2]: batch of 32 variable-length utterances (40-200
phonemes -> chunked at the 100-id window like the reference), padded/masked lock-step decode."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (HIP runtime load order, see tests/conftest.py)
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
utts, chunks, steps, owner = wl.batch_utterances(pkg, seed=2)
m = pkg.Tacotron2.synthetic()
o = pkg.default_opts(dropout_seed=1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for _ in range(reps):
    m.infer_batch(chunks, opts=o, fixed_steps=steps)
t0 = time.perf_counter(); mels = m.infer_batch(chunks, opts=o, fixed_steps=steps); t1 = time.perf_counter()
tt = m.last_timings(); frames = sum(x.shape[1] for x in mels)
print("config3: %d utterances -> %d chunks, %d frames, lock-step iterations %d" % (len(utts), len(chunks), frames, tt["steps"]))
print("wall %.1f ms -> %.0f mel-frames/s; encoder %.2f decoder %.2f postnet %.2f ms; %.1f us per lock-step iteration" % ((t1 - t0) * 1e3, frames / (t1 - t0), tt["encoder_ms"], tt["decoder_ms"], tt["postnet_ms"], tt["decoder_ms"] * 1e3 / tt["steps"]))
