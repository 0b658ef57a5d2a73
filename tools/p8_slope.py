"""Developer aid: in-loop step time of the 3..8-chunk engine = slope of the decoder time over the step count (the intercept is
the launch's set-up: seeding the exchange, the weights into registers, the roles' tables)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
m = pkg.Tacotron2.synthetic()
for B in [int(a) for a in sys.argv[1:]] or (3, 4, 5, 8):
    chunks = [wl.synth_ids(60 + (7 * b) % 40, seed=10 + b) for b in range(B)]
    o = pkg.default_opts(dropout_seed=1)
    t = {}
    for steps in (200, 1000):
        best = 1e9
        for _ in range(3):
            m.infer_batch(chunks, opts=o, fixed_steps=[steps] * B)
            best = min(best, m.last_timings()["decoder_ms"])
        t[steps] = best
    slope = (t[1000] - t[200]) * 1e3 / 800
    print("B=%d: 200 steps %.3f ms, 1000 steps %.3f ms -> %.2f us per step in the loop, %.0f us of set-up" % (B, t[200], t[1000], slope, t[200] * 1e3 - 200 * slope), flush=True)
