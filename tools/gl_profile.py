"""Per-phase wall clocks of the persistent Griffin-Lim kernel.  Needs a -DXDTTS_GL_PROFILE build of
libxdtts_hip.so:   make -C xd-tts_amd prof   (-> libxdtts_hip_prof.so; run with XDTTS_LIB=xd-tts_amd/libxdtts_hip_prof.so)
"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("xd-tts_amd")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(0)
S = np.abs(rng.standard_normal((513, F))).astype(np.float32)
voc = pkg.create_griffin_lim(seed=3)
path = "/tmp/gl_prof.txt"
voc.infer_linear(S, iters=iters)
os.environ["XDTTS_GL_PROFILE"] = path
voc.infer_linear(S, iters=iters)
t = voc.last_timings()
rows = open(path).read().split("\n")
nblk, n_iter = map(int, rows[0].split())
a = np.array([[int(x) for x in r.split()] for r in rows[1:1 + nblk]], dtype=np.float64) * 0.01 / (n_iter + 1)   # us per iteration
names = ["loop", "A inverse FFT", "barrier A", "B1 own overlap-add", "B2 middle samples", "B2 wait neighbours", "B2 finalise + barrier", "C gather + window", "C forward FFT", "C unpack + update", "B0 publish", "-"]
print("F=%d iters=%d: %d workgroups, device %.3f ms (%.2f us per iteration incl. profiling overhead)" % (F, iters, nblk, t["iterations_ms"], t["iterations_ms"] * 1e3 / (iters + 1)))
print("%-24s %8s %8s %8s   (us per iteration, over workgroups)" % ("phase", "mean", "min", "max"))
for i, n in enumerate(names):
    print("%-24s %8.3f %8.3f %8.3f" % (n, a[:, i].mean(), a[:, i].min(), a[:, i].max()))
print("%-24s %8.3f" % ("sum", a.sum(axis=1).mean()))
