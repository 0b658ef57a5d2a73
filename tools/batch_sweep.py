"""us per lock-step decoder iteration and mel-frames/s against the number of chunks B (which engine
serves which batch size): persistent kernel (B <= 2), persistent MFMA kernels (3..16; with XDTTS_P8=0 in the
environment: pairs of the persistent kernel for 3..4, the batched engine from 5), batched MFMA path (>= 17)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
m = pkg.Tacotron2.synthetic()
steps = 200
print("%4s %14s %16s %16s" % ("B", "us/iteration", "mel-frames/s", "engine"))
for B in ([int(a) for a in sys.argv[1:]] or (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 17, 24, 32, 48, 52, 64, 96)):  # optional: the batch sizes as arguments
    chunks = [wl.synth_ids(60 + (7 * b) % 40, seed=10 + b) for b in range(B)]
    o = pkg.default_opts(dropout_seed=1)
    for _ in range(2):
        m.infer_batch(chunks, opts=o, fixed_steps=[steps] * B)
    t = m.last_timings()
    us = t["decoder_ms"] * 1e3 / steps
    p8 = m.engine_state()["decoder_persistent8"] == 1 and 3 <= B <= 16
    eng = "persistent" if B <= 2 else ("persistent MFMA" if p8 else ("persistent pairs" if B <= 4 else "batched MFMA"))
    print("%4d %14.1f %16.0f %16s" % (B, us, B * steps / (t["decoder_ms"] * 1e-3), eng), flush=True)
