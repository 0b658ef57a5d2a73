"""Developer check of the persistent MFMA engine (decoder_persistent8.hip): free-running batches of 3..8 chunks through
infer_batch with XDTTS_P8=1 against the oracle, and timing against the engines it replaces."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
import oracle
orc = oracle.Oracle("f32")
blob = orc.weights_synthetic(seed=20240327, rec_scale=1.0)
rms = lambda a, b: float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))
steps_all = [40, 25, 33, 12, 40, 18, 29, 37]
for B in [int(a) for a in sys.argv[1:]] or (3, 4, 8):
    ids = [wl.synth_ids(30 + 9 * b, seed=40 + b) for b in range(B)]
    steps = steps_all[:B]
    o = pkg.default_opts(dropout_seed=7)
    os.environ["XDTTS_P8"] = "1"
    m = pkg.Tacotron2.from_blob(blob)
    t0 = time.time()
    mels = m.infer_batch(ids, opts=o, fixed_steps=np.array(steps, dtype=np.int32))
    print("B=%d: call %.2f s, decoder %.3f ms, steps %d" % (B, time.time() - t0, m.last_timings()["decoder_ms"], m.last_timings()["steps"]), flush=True)
    worst = 0.0
    for b in range(B):
        ref = orc.infer_chunk(blob, ids[b], orc.default_opts(fixed_steps=steps[b], dropout_seed=7, item=b))
        e = rms(mels[b], ref) if mels[b].shape == ref.shape else float("inf")
        worst = max(worst, e)
        print("   chunk %d: shape %s vs %s rms %.3e" % (b, mels[b].shape, ref.shape, e), flush=True)
    print("B=%d worst rms %.3e  engine_state %s" % (B, worst, m.engine_state()), flush=True)
    m.close()
