"""Developer aid (profile build: make -C xd-tts_amd prof; XDTTS_LIB=xd-tts_amd/libxdtts_hip_prof.so): per-phase wall clock of the
persistent MFMA decoder (decoder_persistent8.hip), printed by the kernel itself for one workgroup of each role."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["XDTTS_P8"] = "1"
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
m = pkg.Tacotron2.synthetic()
steps = 200
for B in [int(a) for a in sys.argv[1:]] or (4, 8):
    chunks = [wl.synth_ids(60 + (7 * b) % 40, seed=10 + b) for b in range(B)]
    o = pkg.default_opts(dropout_seed=1)
    for _ in range(2):
        m.infer_batch(chunks, opts=o, fixed_steps=[steps] * B)
    print("B=%d: %.2f us/step" % (B, m.last_timings()["decoder_ms"] * 1e3 / steps), flush=True)
