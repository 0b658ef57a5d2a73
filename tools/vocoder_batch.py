"""Developer timing aid: the vocoder half of BASELINE.json configs[3] -- 32 utterances of 500-1000 frames
through GriffinLim.infer_batch, wall against device time, both workgroup shapes."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
pkg = importlib.import_module("xd-tts_amd")
rng = np.random.default_rng(4)
Fs = [int(f) for f in rng.integers(500, 1000, size=32)]
mels = [rng.uniform(-7.0, -1.0, size=(80, F)).astype(np.float32) for F in Fs]
v = pkg.create_griffin_lim(seed=3)
for shape in (0, 4):
    v.set_opts(batch_shape=shape)
    for _ in range(3):
        v.infer_batch(mels)
    t0 = time.perf_counter(); outs = v.infer_batch(mels); t1 = time.perf_counter()
    print("batch_shape=%d: %d frames, wall %.2f ms, device %s" % (shape, sum(Fs), (t1 - t0) * 1e3, v.last_timings()), flush=True)
# where the wall time goes: the C call alone, then the copies into numpy
import ctypes as C
v.set_opts(batch_shape=0)
ms = mels
n = len(ms)
for rep in range(3):
    ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in ms])
    nf = (C.c_size_t * n)(*[m.shape[1] for m in ms])
    audios = (pkg._PF * n)()
    ns = (C.c_size_t * n)()
    t0 = time.perf_counter()
    pkg._check(pkg.lib.xdtts_griffinlim_infer_batch(v._h, ptrs, 80, nf, n, audios, ns))
    t1 = time.perf_counter()
    outs = [pkg._take(audios[u], ns[u], (ns[u],)) for u in range(n)]
    t2 = time.perf_counter()
print("C call %.2f ms, numpy copies + free %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
