"""Developer aid (profile build only: make CXXFLAGS='-O3 -std=c++17 -fPIC -DXDTTS_PERSIST_PROFILE'):
per-phase wall-clock of the persistent decoder, averaged per step, by workgroup role."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("xd-tts_amd")
synth_ids = importlib.import_module("xd-tts_amd.workloads").synth_ids
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = 400
path = "/tmp/persist_prof.txt"
os.environ["XDTTS_PERSIST_PROFILE"] = path
m = pkg.Tacotron2.synthetic()
ragged = len(sys.argv) > 2 and sys.argv[2] == "ragged"   # second chunk stops after ~13 steps
ids = [synth_ids(95, seed=1 + b) for b in range(B)]
o = pkg.default_opts(fixed_steps=steps)
if ragged:
    ids = [synth_ids(60), synth_ids(2, seed=2)]
    o = pkg.default_opts(fixed_frames_per_id=steps / 60.0)
for _ in range(2):
    m.infer_batch(ids, opts=o)
t = m.last_timings()
print("B=%d: %.2f us/step" % (B, t["decoder_ms"] * 1e3 / steps))
raw = np.loadtxt(path)
a = raw[:, :16] / 100.0 / steps  # 100 MHz clock -> us per step
names = ["loop", "wait x", "att tail", "wait h_att", "q+energies/bulk", "wait energies", "softmax", "dec tail+bulk", "wait h_dec", "proj/bulk/loc", "prenet: publish x", "pre: mel gathered", "pre: gate+store+L1", "pre: L1 barrier", "pre: L2", "attn: q rows"]
roles = {"attn c0": slice(0, 8), "pre c0": slice(8 * B, 8 * B + 16), "plain": slice(24 * B, 256)}
if B > 1:
    roles["attn c1"] = slice(8, 16)
    roles["pre c1"] = slice(8 * B + 16, 8 * B + 32)
print("%-18s" % "phase" + "".join("%14s" % r for r in roles))
for i, n in enumerate(names):
    print("%-18s" % n + "".join("%14.2f" % a[sl, i].mean() for sl in roles.values()))
print("%-18s" % "sum" + "".join("%14.2f" % a[sl].sum(axis=1).mean() for sl in roles.values()))
polls = raw[:, 16:20] / steps / 8.0  # failed poll rounds per step (slowest lane of a wave, mean over the 8 waves)
print("failed poll rounds per step")
for i, n in enumerate(["x", "h_att", "energies", "h_dec"]):
    print("%-18s" % n + "".join("%14.2f" % polls[sl, i].mean() for sl in roles.values()))
when = raw[:, 20:24] / 100.0 / steps  # mean wall clock (us) of each workgroup at four events
when = when - when.mean(axis=0, keepdims=True)
print("mean lateness (us) relative to the average workgroup")
for i, n in enumerate(["h_att published", "h_dec published(+bulk)", "x gathered", "energies gathered"]):
    print("%-24s" % n + "".join("%12.2f" % when[sl, i].mean() for sl in roles.values()) + "   max %.2f (wg %d)  min %.2f (wg %d)" % (when[:, i].max(), when[:, i].argmax(), when[:, i].min(), when[:, i].argmin()))
xcd = np.arange(256) % 8
print("by XCD (wg % 8), h_att published: " + " ".join("%.2f" % when[xcd == k, 0].mean() for k in range(8)))
