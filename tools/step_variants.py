"""Developer aid: in-loop step time of the persistent engines under option variants (which per-step work sits on the critical
path?): dropout on / off, stop rule on / off."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
m = pkg.Tacotron2.synthetic()
for B in [int(a) for a in sys.argv[1:]] or (1, 2, 4):
    chunks = [wl.synth_ids(60 + (7 * b) % 40, seed=10 + b) for b in range(B)]
    for name, kw, fixed in (("dropout seeded, fixed steps", dict(dropout_seed=1), True), ("dropout off, fixed steps", dict(dropout_mode=0), True),
                            ("dropout seeded, stop rule (never fires)", dict(dropout_seed=1), False)):
        t = {}
        for steps in (200, 1000):
            o = pkg.default_opts(max_steps=steps, **kw)
            best = 1e9
            for _ in range(3):
                m.infer_batch(chunks, opts=o, **(dict(fixed_steps=[steps] * B) if fixed else {}))
                best = min(best, m.last_timings()["decoder_ms"])
            t[steps] = best
        print("B=%d %-42s %.2f us per step in the loop" % (B, name, (t[1000] - t[200]) * 1e3 / 800), flush=True)
