import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa
pkg = importlib.import_module("xd-tts_amd"); wl = importlib.import_module("xd-tts_amd.workloads")
B = int(sys.argv[1])
m = pkg.Tacotron2.synthetic()
chunks = [wl.synth_ids(60 + (7 * b) % 40, seed=10 + b) for b in range(B)]
for _ in range(2):
    m.infer_batch(chunks, opts=pkg.default_opts(dropout_seed=1), fixed_steps=[120] * B)
print("B=%d: %.2f us per iteration" % (B, m.last_timings()["decoder_ms"] * 1e3 / 120), flush=True)
