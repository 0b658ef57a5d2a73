"""Parity margins of the GPU path against the CPU oracle (fp32 and fp64 builds) at the BASELINE sizes:
free-running mel drift over the frame index and Griffin-Lim audio after 30/60/120
iterations (8(c)(iii)).  Developer report; the pass/fail versions of these are in tests/."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
synth_ids = importlib.import_module("xd-tts_amd.workloads").synth_ids
from test_gpu_griffinlim_more import chirps
pkg = importlib.import_module("xd-tts_amd")

def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))

o32, o64 = oracle.Oracle("f32"), oracle.Oracle("f64")
blob = o32.weights_synthetic()
m = pkg.Tacotron2.from_blob(blob)
ids = synth_ids(95)
F = 633
gpu = m.infer(ids, opts=pkg.default_opts(fixed_steps=F, dropout_seed=0))
r32 = o32.infer_chunk(blob, ids, o32.default_opts(fixed_steps=F, dropout_seed=0))
r64 = o64.infer_chunk(blob, ids, o64.default_opts(fixed_steps=F, dropout_seed=0))
sig = float(np.sqrt(np.mean(np.asarray(r64, np.float64) ** 2)))
print("mel, one 95-id chunk, %d frames (signal RMS %.3f); RMS error over frames [a, b):" % (F, sig))
print("%-12s %12s %12s %12s" % ("frames", "gpu-f32orc", "gpu-f64orc", "f32orc-f64orc"))
for a, b in ((0, 50), (50, 200), (200, 400), (400, F), (0, F)):
    print("%-12s %12.2e %12.2e %12.2e" % ("[%d,%d)" % (a, b), rms(gpu[:, a:b], r32[:, a:b]), rms(gpu[:, a:b], r64[:, a:b]), rms(r32[:, a:b], r64[:, a:b])))
for mode in ("launch",):
    os.environ["XDTTS_DECODER"] = mode
    alt = m.infer(ids, opts=pkg.default_opts(fixed_steps=F, dropout_seed=0))
    del os.environ["XDTTS_DECODER"]
    print("launch-per-stage engine vs persistent engine: %.2e ; vs f32 oracle %.2e" % (rms(alt, gpu), rms(alt, r32)))

Fg = 1000
S = o32.stft(chirps(256 * (Fg - 1)))
S = np.hypot(S[..., 0], S[..., 1]).astype(np.float32)
voc = pkg.create_griffin_lim(seed=3)
print("Griffin-Lim, %d frames (config 5); audio RMS error (signal RMS in brackets):" % Fg)
for iters in (30, 60, 120):
    a = voc.infer_linear(S, iters=iters)
    b32 = o32.griffinlim(S, seed=3, iters=iters)
    b64 = o64.griffinlim(S, seed=3, iters=iters)
    print("  %3d iterations: gpu-f32orc %.2e  gpu-f64orc %.2e  f32orc-f64orc %.2e  [%.3f]" % (
        iters, rms(a, b32), rms(a, b64), rms(b32, b64), float(np.sqrt(np.mean(np.asarray(b64, np.float64) ** 2)))))
