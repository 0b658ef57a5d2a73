"""Developer aid: long sequences on the 3..16-chunk engines (its rings hold one slab per step): three runs each must give the same bits,
finite values, the engine still on."""
import importlib, sys, hashlib, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("xd-tts_amd"); wl = importlib.import_module("xd-tts_amd.workloads")
m = pkg.Tacotron2.synthetic()
for B, steps in [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]] or ((8, 3000), (5, 2500), (3, 4000), (16, 2500), (11, 3000)):  # optional arguments B:steps
    chunks = [wl.synth_ids(60 + (7 * b) % 40, seed=10 + b) for b in range(B)]
    o = pkg.default_opts(dropout_seed=1, max_steps=steps)
    hs = []
    for rep in range(3):
        mels = m.infer_batch(chunks, opts=o, fixed_steps=[steps - 17 * b for b in range(B)])
        hs.append(hashlib.sha1(b"".join(x.tobytes() for x in mels)).hexdigest()[:12])
    t = m.last_timings()
    print(B, steps, hs, "%.2f us/step" % (t["decoder_ms"] * 1e3 / steps), m.engine_state()["decoder_persistent8"], all(np.isfinite(x).all() for x in mels), flush=True)
