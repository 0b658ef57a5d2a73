"""Developer check of the persistent Griffin-Lim engine: parity against the oracle and against the
launch-per-iteration engine (XDTTS_GL=launch) over frame counts that exercise every block shape,
then timings at the BASELINE sizes."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
orc = oracle.Oracle("f32")
rms = lambda a, b: float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))
voc = pkg.create_griffin_lim(seed=3)
rng = np.random.default_rng(0)
for F in (16, 17, 19, 23, 64, 100, 257, 800, 1000, 1024, 1025, 1030, 1800, 2304, 2305, 2400):
    spec = orc.stft(wl.chirps(256 * (F - 1)))
    S = np.hypot(spec[..., 0], spec[..., 1]).astype(np.float32)
    p0 = orc.phase_init(3, 513, F)
    it = 4
    os.environ.pop("XDTTS_GL", None)
    a = voc.infer_linear(S, phase0=p0, iters=it)
    os.environ["XDTTS_GL"] = "launch"
    b = voc.infer_linear(S, phase0=p0, iters=it)
    os.environ.pop("XDTTS_GL", None)
    ref = orc.griffinlim(S, phase0=p0, iters=it) if F <= 1100 else b
    ga, gr = voc.step(S, p0, np.zeros_like(p0), n_iter=2)
    oa, orr = orc.griffinlim_step(S, p0, np.zeros_like(p0), iters=2) if F <= 1100 else (ga, gr)
    print("F=%5d  persistent-vs-oracle %.2e  launch-vs-oracle %.2e  persistent-vs-launch %.2e  step rebuilt rel %.2e" % (
        F, rms(a, ref), rms(b, ref), rms(a, b), rms(gr, orr) / max(1e-30, float(np.sqrt(np.mean(orr.astype(np.float64) ** 2))))), flush=True)
for F in (800, 1000):
    spec = orc.stft(wl.chirps(256 * (F - 1)))
    S = np.hypot(spec[..., 0], spec[..., 1]).astype(np.float32)
    for mode in ("persistent", "launch"):
        if mode == "launch":
            os.environ["XDTTS_GL"] = "launch"
        else:
            os.environ.pop("XDTTS_GL", None)
        for iters in (30, 60, 120):
            for _ in range(3):
                voc.infer_linear(S, iters=iters)
            t = voc.last_timings()
            print("F=%d %-10s iters=%3d  device %.3f ms  %.2f us per iteration  algorithmic %.0f GB/s (%.3f of 8 TB/s)" % (
                F, mode, iters, t["iterations_ms"], t["iterations_ms"] * 1e3 / (iters + 1), 12308.0 * F * iters / (t["iterations_ms"] * 1e-3) / 1e9,
                12308.0 * F * iters / (t["iterations_ms"] * 1e-3) / 8e12), flush=True)
os.environ.pop("XDTTS_GL", None)
