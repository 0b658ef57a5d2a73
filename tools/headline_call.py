"""Developer aid: the headline utterance (configs[1]) through xdtts_synthesize_ids a few times -- the command
tools/headline_timeline.sh traces."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
model = pkg.Tacotron2.synthetic(seed=wl.WEIGHT_SEED, rec_scale=1.0)
voc = pkg.create_griffin_lim(iters=60, seed=0)
_ids, chunks, _steps = wl.config2(pkg)
sp = np.cumsum([len(c) for c in chunks]).astype(np.int64)
opts = pkg.default_opts(fixed_frames_per_id=wl.FRAMES_PER_ID, dropout_seed=0, item_base=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for g in range(n):
    t0 = time.perf_counter()
    mel, audio = pkg.synthesize(model, voc, wl.synth_ids(120, seed=1 + g), splits=sp, opts=opts)
    t1 = time.perf_counter()
    print("call %d: %.3f ms wall, device %s %s" % (g, (t1 - t0) * 1e3, model.last_timings(), voc.last_timings()), flush=True)
    time.sleep(0.01)
