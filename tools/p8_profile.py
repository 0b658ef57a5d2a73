"""Developer aid (profile build: make -C xd-tts_amd prof; XDTTS_LIB=xd-tts_amd/libxdtts_hip_prof.so): per-phase wall clock of the
persistent MFMA decoder (decoder_persistent8.hip), printed by the kernel itself for one workgroup of each role."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["XDTTS_P8"] = "1"
pkg = importlib.import_module("xd-tts_amd")
wl = importlib.import_module("xd-tts_amd.workloads")
import subprocess
NAMES = {0: "x wait", 1: "x barrier", 2: "att cell", 3: "h_att wait", 4: "h_att barrier", 5: "energies", 6: "mfma dh", 20: "ep wait", 21: "ep barrier",
         22: "softmax", 23: "reduce+publish ctx", 7: "mfma ah", 8: "ctx wait", 9: "ctx barrier", 10: "dec cell", 11: "mfma ac + location", 12: "h_dec wait",
         13: "h_dec barrier", 14: "projection", 15: "mfma dd", 24: "mel wait", 25: "mel barrier", 26: "gate + layer 1", 27: "layer 2", 16: "publish x",
         17: "failed rounds h_att", 18: "failed rounds ctx", 19: "failed rounds h_dec"}
ORDER = [0, 1, 2, 3, 4, 5, 6, 20, 21, 22, 23, 7, 8, 9, 10, 11, 12, 13, 14, 15, 24, 25, 26, 27, 16, 17, 18, 19]
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
for B in [int(a) for a in sys.argv[1:]] or (4, 8):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(B)], capture_output=True, text=True).stdout
    vals = {}
    for ln in out.splitlines():
        f = ln.split()
        if f and f[0] == "P8PROF":
            vals[(int(f[1]), int(f[3]))] = float(f[4])   # the second call overwrites the first
        elif ln.startswith("B="):
            print(ln)
    wgs = sorted({k[0] for k in vals})
    role = {0: "attention", 8 * B: "projection+prenet", 255: "plain"}
    print("%-22s" % "phase (us per step)" + "".join("%20s" % ("wg %d %s" % (w_, role.get(w_, ""))) for w_ in wgs))
    for i in ORDER:
        print("%-22s" % NAMES[i] + "".join("%20.2f" % vals.get((w_, i), float("nan")) for w_ in wgs))
    print("%-22s" % "sum" + "".join("%20.2f" % sum(vals.get((w_, i), 0.0) for i in ORDER[:-3]) for w_ in wgs))
