#!/usr/bin/env python3
"""Developer check of xd-tts_amd/gate_rig.py on the MI355X: rig the headline utterance's gate, then decode it
gate-ON through the pipeline and compare frame counts / frames with the fixed-steps run."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import torch  # noqa: F401
except ImportError:
    pass
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
rig = importlib.import_module("xd-tts_amd.gate_rig")
model = pkg.Tacotron2.synthetic(seed=wl.WEIGHT_SEED)
ids, chunks, steps = wl.config2(pkg)
sp = np.cumsum([len(c) for c in chunks]).astype(np.int64)
fixed = pkg.default_opts(fixed_frames_per_id=wl.FRAMES_PER_ID, dropout_seed=0, item_base=0)
ref = model.infer(ids, splits=sp, opts=fixed)
t0 = time.time()
m2, info = rig.rigged_gate_model(pkg, model, chunks, steps, pkg.default_opts(dropout_seed=0, item_base=0))
print("rig: %.1f s" % (time.time() - t0), info)
on = pkg.default_opts(dropout_seed=0, item_base=0)   # gate on, max_steps 1000
mel = m2.infer(ids, splits=sp, opts=on)
print("frames fixed", ref.shape, "gate-on", mel.shape, "steps", m2.last_timings())
if mel.shape == ref.shape:
    print("max |diff|", float(np.abs(mel - ref).max()))
for _ in range(3):
    t0 = time.time(); m2.infer(ids, splits=sp, opts=on); a = time.time() - t0
    t0 = time.time(); model.infer(ids, splits=sp, opts=fixed); b = time.time() - t0
    print("gate-on %.3f ms (decoder %.3f)  fixed %.3f ms" % (a * 1e3, m2.last_timings()["decoder_ms"], b * 1e3), model.last_timings()["decoder_ms"])
