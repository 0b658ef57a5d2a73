"""Developer check of the 16-slot persistent MFMA decoder (csrc/decoder_persistent16.hip): ragged batches of 9..16 chunks against the
CPU oracle chunk by chunk, the same bits twice, and us per lock-step iteration beside the engines either side."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
import oracle
orc = oracle.Oracle("f32")

def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)))

blob = orc.weights_synthetic(seed=wl.WEIGHT_SEED, rec_scale=1.0)
m = pkg.Tacotron2.from_blob(blob)
LENS = [37, 100, 2, 64, 23, 81, 9, 55, 71, 14, 92, 48, 5, 66, 30, 87]
STEPS = [40, 25, 33, 12, 40, 18, 29, 37, 22, 40, 31, 8, 36, 27, 15, 39]
for B in [int(a) for a in sys.argv[1:]] or (9, 12, 16):
    ids = [wl.synth_ids(n, seed=40 + i) for i, n in enumerate(LENS[:B])]
    steps = np.asarray(STEPS[:B], dtype=np.int32)
    o = pkg.default_opts(dropout_seed=7, item_base=2)
    t0 = time.time()
    mels = m.infer_batch(ids, opts=o, fixed_steps=steps)
    st = m.engine_state()
    again = m.infer_batch(ids, opts=o, fixed_steps=steps)
    worst = 0.0
    if blob is not None:
        for b in range(B):
            ref = orc.infer_chunk(blob, ids[b], orc.default_opts(fixed_steps=int(steps[b]), dropout_seed=7, item=2 + b))
            assert mels[b].shape == ref.shape, (b, mels[b].shape, ref.shape)
            worst = max(worst, rms(mels[b], ref))
    same = all(np.array_equal(a, c) for a, c in zip(mels, again))
    print("B %2d: engine state %s  worst rms vs oracle %.2e  same bits twice %s  (%.1f s)" % (B, st, worst, same, time.time() - t0), flush=True)
