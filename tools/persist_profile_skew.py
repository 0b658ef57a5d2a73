"""Developer aid (profile build): per-phase wall clock of the persistent decoder's skewed pair loop, per step, by role.
A phase's time runs from the end of the previous phase (of the other chunk) to its own end: gather wait + arithmetic."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("xd-tts_amd")
synth_ids = importlib.import_module("xd-tts_amd.workloads").synth_ids
steps = 400
path = "/tmp/persist_prof.txt"
os.environ["XDTTS_PERSIST_PROFILE"] = path
m = pkg.Tacotron2.synthetic()
ids = [synth_ids(95, seed=1 + b) for b in range(2)]
o = pkg.default_opts(fixed_steps=steps)
for _ in range(2):
    m.infer_batch(ids, opts=o)
t = m.last_timings()
print("B=2 skewed: %.2f us/step (profile build)" % (t["decoder_ms"] * 1e3 / steps))
a = np.loadtxt(path)[:, :16] / 100.0 / steps
names = ["c0 ph1 x->h_att", "c0 ph2 h_att->(q,e)", "c0 ph3 e->h_dec", "c0 ph4 h_dec->(mel)", "c0 ph5 (prenet) end", "c1 ph1", "c1 ph2", "c1 ph3", "c1 ph4", "c1 ph5 end",
         "c0 ph5 role: entry", "c0 ph5 role: mel gathered", "-", "c1 ph5 role: entry", "c1 ph5 role: mel gathered", "-"]
roles = {"attn c0": slice(0, 8), "attn c1": slice(8, 16), "pre c0": slice(16, 32), "pre c1": slice(32, 48), "plain": slice(48, 256)}
print("%-28s" % "phase" + "".join("%10s" % r for r in roles))
for i, n in enumerate(names):
    print("%-28s" % n + "".join("%10.2f" % a[sl, i].mean() for sl in roles.values()))
print("%-28s" % "sum" + "".join("%10.2f" % a[sl].sum(axis=1).mean() for sl in roles.values()))
