"""Developer timing aid: launch set-up of the persistent decoder (weights into registers, context fold) from the
decoder time at two step counts (XDTTS_LIB selects the build)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
m = pkg.Tacotron2.synthetic()
o = pkg.default_opts(dropout_seed=1)
for B in (1, 2):
    chunks = [wl.synth_ids(95, seed=10 + b) for b in range(B)]
    t = {}
    for n in (50, 650):
        best = 1e9
        for _ in range(5):
            m.infer_batch(chunks, opts=o, fixed_steps=[n] * B)
            best = min(best, m.last_timings()["decoder_ms"] * 1e3)
        t[n] = best
    per = (t[650] - t[50]) / 600.0
    print("B=%d: %.2f us per step, set-up %.1f us per launch" % (B, per, t[50] - 50 * per), flush=True)
