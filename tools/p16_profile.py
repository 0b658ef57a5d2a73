"""Developer aid (profile build: make -C xd-tts_amd prof; XDTTS_LIB=xd-tts_amd/libxdtts_hip_prof.so): per-phase wall clock of the
16-slot persistent MFMA decoder (decoder_persistent16.hip), printed by the launch function for one workgroup of each role."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["XDTTS_P8"] = "1"
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
import subprocess
NAMES = {0: "x wait", 1: "x barrier", 2: "mfma x + att cell", 3: "h_att row wait", 4: "h_att row barrier", 5: "energies / nap", 6: "h_att stream + mfma dh, ah",
         20: "ep wait", 21: "ep barrier", 22: "softmax", 7: "reduce + publish ctx", 8: "ctx wait", 9: "mfma dc + barrier", 10: "dec cell",
         11: "mfma ac + location", 12: "h_dec row wait (+ W_p rows)", 13: "h_dec row barrier", 14: "projection / nap", 15: "h_dec stream + mfma dd",
         24: "mel wait", 25: "mel barrier", 26: "gate + layer 1", 27: "layer 2", 16: "publish x"}
ORDER = [0, 1, 2, 3, 4, 5, 6, 20, 21, 22, 7, 8, 9, 10, 11, 12, 13, 14, 15, 24, 25, 26, 27, 16]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    B = int(sys.argv[2])
    m = pkg.Tacotron2.synthetic()
    steps = 200
    chunks = [wl.synth_ids(60 + (7 * b) % 40, seed=10 + b) for b in range(B)]
    o = pkg.default_opts(dropout_seed=1)
    for _ in range(2):
        m.infer_batch(chunks, opts=o, fixed_steps=[steps] * B)
    print("B=%d: %.2f us/step" % (B, m.last_timings()["decoder_ms"] * 1e3 / steps), flush=True)
    sys.exit(0)
for B in [int(a) for a in sys.argv[1:]] or (12, 16):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(B)], capture_output=True, text=True).stdout
    vals = {}
    for ln in out.splitlines():
        f = ln.split()
        if f and f[0] == "P16PROF":
            vals[(int(f[1]), int(f[3]))] = float(f[4])   # the second call overwrites the first
        elif ln.startswith("B="):
            print(ln)
    wgs = sorted({k[0] for k in vals})
    role = {0: "attention", 8 * B: "projection+prenet", 255: "plain" if B < 16 else "projection+prenet"}
    print("%-30s" % "phase (us per step)" + "".join("%26s" % ("wg %d %s" % (w_, role.get(w_, ""))) for w_ in wgs))
    for i in ORDER:
        print("%-30s" % NAMES[i] + "".join("%26.2f" % vals.get((w_, i), float("nan")) for w_ in wgs))
    print("%-30s" % "sum" + "".join("%26.2f" % sum(vals.get((w_, i), 0.0) for i in ORDER) for w_ in wgs))
