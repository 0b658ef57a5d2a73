"""Post-net GEMM timing aid: runs Tacotron2 post-net (5 implicit-GEMM conv layers) on F frames a few times;
run under `rocprofv3 --kernel-trace --stats` and read the k_gemm_nt rows (tools/rocprof_summary.py)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("xd-tts_amd")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 800
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m = pkg.Tacotron2.synthetic()
fr = np.random.default_rng(0).standard_normal((F, 80)).astype(np.float32)
for _ in range(reps):
    out = m.postnet(fr)
print("postnet", F, out.shape, float(np.abs(out).mean()))
