import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
pkg = importlib.import_module("xd-tts_amd")
import oracle
from test_gpu_engine_hooks import chunks_for, snapshot, NAMES
orc = oracle.Oracle("f32")
blob = orc.weights_synthetic(seed=20240327, rec_scale=1.0)
tab = {t[0]: t for t in orc.tensor_table()}
def tensor(name, shape):
    _n, _s, off, n = tab[name]
    return blob[off:off + n].reshape(shape).astype(np.float64)
Wp = np.concatenate([tensor("linear_projection.weight", (80, 1536)), tensor("gate_layer.weight", (1, 1536))])
model = pkg.Tacotron2.from_blob(blob)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lens, mem, pm = chunks_for(orc, blob, B)
T = mem.shape[1]
opts = [orc.default_opts(dropout_seed=11, item=3 + b) for b in range(B)]
go = pkg.default_opts(dropout_seed=11, item_base=3)
sts = [orc.new_state() for _ in range(B)]
for step in range(3):
    snap = snapshot(sts, T)
    dec_in = np.stack([np.array(s.dec_in, dtype=np.float32) for s in sts])
    ref = [orc.decoder_step(blob, mem[b], pm[b], lens[b], sts[b], opts[b], step) for b in range(B)]
    after = snapshot(sts, T)
    for rep in range(2):
        out, gate, gst = model.decoder_steps("persistent8", mem, pm, lens, snap, dec_in, step, 1, opts=go)
        full = np.concatenate([out[:, 0], gate[:, :1]], axis=1)
        reff = np.stack([np.concatenate([r[0], [r[1]]]) for r in ref])
        d = full - reff
        for b in range(B):
            bad = np.nonzero(np.abs(d[b]) > 1e-6)[0]
            if len(bad) == 0:
                continue
            r = bad[0]
            hN, hO = after["decoder_hidden"][b].astype(np.float64), snap["decoder_hidden"][b].astype(np.float64)
            cN, cO = after["attention_context"][b].astype(np.float64), snap["attention_context"][b].astype(np.float64)
            cand = {"h_old": Wp[r, :1024] @ (hO - hN), "ctx_old": Wp[r, 1024:] @ (cO - cN), "no_ctx": -Wp[r, 1024:] @ cN, "no_h": -Wp[r, :1024] @ hN}
            for ob in range(B):
                if ob != b:
                    cand["h_of_chunk%d" % ob] = Wp[r, :1024] @ (after["decoder_hidden"][ob].astype(np.float64) - hN)
                    cand["ctx_of_chunk%d" % ob] = Wp[r, 1024:] @ (after["attention_context"][ob].astype(np.float64) - cN)
            print("step %d rep %d chunk %d rows %s: row %d err %.3e; candidates %s" % (step, rep, b, sorted(set(int(x) % 16 for x in bad)), r, d[b, r], {k: "%.3e" % v for k, v in cand.items()}))
