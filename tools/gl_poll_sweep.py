"""Developer sweep: first-poll placement of the persistent Griffin-Lim kernel (XDTTS_GL_POLL_DELAY), us per iteration."""
import hashlib, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = wl.chirp_magnitude(F)
voc = pkg.create_griffin_lim(seed=3)
ref = None
for rep in range(2):
    for pd in [int(a) for a in sys.argv[2:]] or [0, 2, 4, 6, 8, 10, 14, -1, -3, -5, -9]:
        os.environ["XDTTS_GL_POLL_DELAY"] = str(pd)
        best = 1e9
        for _ in range(5):
            a = voc.infer_linear(S, iters=60)
            best = min(best, voc.last_timings()["iterations_ms"])
        if ref is None:
            ref = a
        print("poll_delay %3d: %.3f ms, %.3f us per iteration, same bits %s, sha1 %s" % (pd, best, best * 1e3 / 61, np.array_equal(a, ref), hashlib.sha1(a.tobytes()).hexdigest()[:12]), flush=True)
