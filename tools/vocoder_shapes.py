"""Developer timing aid: the three workgroup shapes of the vocoder batch (XDTTS_GL_BATCH_FORCE) on the 32-utterance batch of
tools/vocoder_batch.py -> the relative costs in gl_batch_from_device's packing model (api.cpp)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
pkg = importlib.import_module("xd-tts_amd")
rng = np.random.default_rng(4)
for lo, hi in ((500, 1000), (200, 600)):
    Fs = [int(f) for f in rng.integers(lo, hi, size=32)]
    mels = [rng.uniform(-7.0, -1.0, size=(80, F)).astype(np.float32) for F in Fs]
    v = pkg.create_griffin_lim(iters=60, seed=3)
    for force in ("8", "41", "42", ""):
        if force:
            os.environ["XDTTS_GL_BATCH_FORCE"] = force
        else:
            os.environ.pop("XDTTS_GL_BATCH_FORCE", None)
        best = 1e9
        for _ in range(4):
            v.infer_batch(mels)
            best = min(best, v.last_timings()["iterations_ms"])
        print("frames %d..%d (%d total): force %-3s iterations %.3f ms" % (lo, hi, sum(Fs), force or "-", best), flush=True)
    v.close()
