"""Developer timing aid: per-kernel in-chain cost via XDTTS_DEBUG_MIX (results are garbage)."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    os.environ["XDTTS_DEBUG_MIX"] = sys.argv[1]
    pkg = importlib.import_module("xd-tts_amd")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    synth_ids = importlib.import_module("xd-tts_amd.workloads").synth_ids
    ids = synth_ids(95)
    m = pkg.Tacotron2.synthetic()
    o = pkg.default_opts(fixed_steps=400)
    for _ in range(3):
        m.infer(ids, opts=o)
    t = m.last_timings()
    n = len(sys.argv[1])
    print("%-8s %.2f us/step (%d kernels/step -> %.2f us each)" % (sys.argv[1], t["decoder_ms"] * 1e3 / 400, n, t["decoder_ms"] * 1e3 / 400 / n))
else:
    for mix in ["paqsd", "ppppp", "aaaaa", "qqqqq", "sssss", "ddddd"]:
        subprocess.call([sys.executable, __file__, mix])
