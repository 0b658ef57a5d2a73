"""Measurement aid for BASELINE.json configs[2]: batch of 32 variable-length chunks (40-200
phonemes -> chunked at the 100-id window like the reference), padded/masked lock-step decode."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_ids
pkg = importlib.import_module("xd-tts_amd")
rng = np.random.Generator(np.random.PCG64(2))
lens = rng.integers(40, 201, size=32)
chunks = []
for i, n in enumerate(lens):
    ids = synth_ids(int(n), seed=100 + i)
    sp = list(pkg.find_splits(ids, 100))
    if not sp or sp[-1] != len(ids):
        sp.append(len(ids))
    a = 0
    for e in sp:
        if e > a:
            chunks.append(ids[a:e]); a = e
steps = [int(np.floor(6.67 * len(c) + 0.5)) for c in chunks]
m = pkg.Tacotron2.synthetic()
o = pkg.default_opts(dropout_seed=1)
for _ in range(2):
    m.infer_batch(chunks, opts=o, fixed_steps=steps)
t0 = time.perf_counter(); mels = m.infer_batch(chunks, opts=o, fixed_steps=steps); t1 = time.perf_counter()
tt = m.last_timings(); frames = sum(x.shape[1] for x in mels)
print("config3: %d utterances -> %d chunks, %d frames, lock-step iterations %d" % (len(lens), len(chunks), frames, tt["steps"]))
print("wall %.1f ms -> %.0f mel-frames/s; encoder %.2f decoder %.2f postnet %.2f ms; %.1f us per lock-step iteration" % ((t1 - t0) * 1e3, frames / (t1 - t0), tt["encoder_ms"], tt["decoder_ms"], tt["postnet_ms"], tt["decoder_ms"] * 1e3 / tt["steps"]))
