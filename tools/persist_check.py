"""Developer aid: persistent decoder vs launch-per-stage decoder on the same inputs (parity + step time)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("xd-tts_amd")
synth_ids = importlib.import_module("xd-tts_amd.workloads").synth_ids

def rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))

m = pkg.Tacotron2.synthetic()
cases = [("B=1 fixed 60", [synth_ids(95)], dict(fixed_steps=60)),
         ("B=1 fixed 400", [synth_ids(95)], dict(fixed_steps=400)),
         ("B=2 fixed/id", [synth_ids(95), synth_ids(25, seed=3)], dict(fixed_frames_per_id=6.667)),
         ("B=3 fixed 100", [synth_ids(40), synth_ids(77), synth_ids(100)], dict(fixed_steps=100)),
         ("B=1 gate", [synth_ids(30)], dict(max_steps=300)),
         ("B=2 both 200", [synth_ids(95), synth_ids(60, seed=3)], dict(fixed_steps=200)),
         ("B=4 all 100", [synth_ids(95), synth_ids(60, seed=3), synth_ids(33, seed=4), synth_ids(100, seed=5)], dict(fixed_steps=100)),
         ]
for name, ids_list, kw in cases:
    o = pkg.default_opts(**kw)
    res = {}
    for mode in ("launch", "persistent"):
        os.environ["XDTTS_DECODER"] = mode
        try:
            for _ in range(2):
                out = m.infer_batch(ids_list, opts=o)
            t = m.last_timings()
            res[mode] = (out, t)
        except Exception as e:  # noqa
            print(name, mode, "FAILED:", e)
            res[mode] = None
    if res["launch"] and res["persistent"]:
        (a, ta), (b, tb) = res["launch"], res["persistent"]
        fa = [x.shape for x in a]; fb = [x.shape for x in b]
        worst = max(rms(x, y) if x.shape == y.shape else float("inf") for x, y in zip(a, b))
        print("%-14s frames %s vs %s  mel rms %.3e | launch %.2f us/step  persistent %.2f us/step (steps %d / %d)" % (
            name, fa, fb, worst, ta["decoder_ms"] * 1e3 / max(ta["steps"], 1), tb["decoder_ms"] * 1e3 / max(tb["steps"], 1), ta["steps"], tb["steps"]))
