"""Measurement aid for bench.py's gate-ON variant (the reference's real mode: the stop rule of
src/tacotron2/mod.rs:319-324 decides the frame count on the device).  The synthetic weights' gate never
fires, so a gate-on run of them ends at max_decoder_steps.  `rigged_gate_model` returns a second handle
whose `gate_layer` is chosen such that sigmoid(gate) > 0.6 happens exactly at a requested step of every
chunk -- the rest of the network, hence every frame, is unchanged (the gate does not feed back).

How: the gate logit of step s is w . [decoder_hidden(s) ; attention_context(s)] + b, and
the trajectory of that 1536-vector does not depend on (w, b).  It is recorded with the library's own
parity hook (xdtts_tacotron2_decoder_steps, one call per step on the persistent engine, all chunks of the
utterance together -- the same dropout stream as the pipeline), then (w, b) = the minimum-norm solution
of  w . X_c(s) + b = ramp_c(s),  a straight line that crosses the threshold logit between the chunk's last two
steps.  Uses the GPU library only (no oracle)."""
import numpy as np

LOGIT_06 = float(np.log(0.6 / 0.4))   # sigmoid(x) > 0.6  <=>  x > ln 1.5 (gate_threshold, mod.rs:279)
NAMES = ("attention_hidden", "attention_cell", "decoder_hidden", "decoder_cell", "attention_weights", "attention_weights_cum", "attention_context")


def record_gate_inputs(pkg, model, chunks, steps, opts, window=100):
    """X[c] = (steps[c], 1536) rows [decoder_hidden ; attention_context] of chunk c, free-running."""
    B = len(chunks)
    mem = np.zeros((B, window, 512), dtype=np.float32)
    pm = np.zeros((B, window, 128), dtype=np.float32)
    for c, ids in enumerate(chunks):
        padded = np.zeros(window, dtype=np.int64)      # pad-to-100 / plen quirk of infer_chunk (mod.rs:369-375)
        padded[: len(ids)] = ids
        mem[c], pm[c] = model.encoder(padded)
    nv = np.array([len(c) for c in chunks], dtype=np.int32)
    dims = {"attention_hidden": 1024, "attention_cell": 1024, "decoder_hidden": 1024, "decoder_cell": 1024,
            "attention_weights": window, "attention_weights_cum": window, "attention_context": 512}
    st = {k: np.zeros((B, d), dtype=np.float32) for k, d in dims.items()}   # DecoderState::new (mod.rs:202-233)
    din = np.zeros((B, 80), dtype=np.float32)
    X = [np.zeros((n, 1536), dtype=np.float64) for n in steps]
    for s in range(max(steps)):
        out, _gate, st = model.decoder_steps("persistent" if B <= 2 else "batched", mem, pm, nv, st, din, s, 1, opts)
        din = out[:, 0, :]
        for c in range(B):
            if s < steps[c]:
                X[c][s, :1024] = st["decoder_hidden"][c]
                X[c][s, 1024:] = st["attention_context"][c]
    return X


def solve_gate(X, steps, slope=0.02):
    """(w[1536], b, slack, max |w|): per chunk the logit is the straight line through the threshold half-way between its
    last two steps, y_c(s) = LOGIT_06 + slope * (s - (n_c - 1.5)) -- below the threshold for every s < n_c - 1, above it at
    the last step.  800 equations, 1537 unknowns: the minimum-norm solution fits them exactly; slack = the distance (in
    logits) of the closest row from the wrong side of the threshold, slope / 2 when the fit is exact."""
    A = np.concatenate([np.concatenate([x, np.ones((len(x), 1))], axis=1) for x in X], axis=0)
    y = np.concatenate([LOGIT_06 + slope * (np.arange(n) - (n - 1.5)) for n in steps])
    sol = np.linalg.lstsq(A, y, rcond=None)[0]
    fit = A @ sol
    slack, at = np.inf, 0
    for n in steps:
        seg = fit[at : at + n]
        slack = min(slack, seg[-1] - LOGIT_06, LOGIT_06 - (seg[:-1].max() if n > 1 else -np.inf))
        at += n
    return sol[:-1], float(sol[-1]), float(slack), float(np.abs(sol[:-1]).max())


def rigged_gate_model(pkg, model, chunks, steps, opts, device_id=0):
    """A new Tacotron2 handle = `model`'s weights with gate_layer rigged to stop chunk c after exactly steps[c]
    frames under `opts`' dropout stream; returns (handle, info dict)."""
    X = record_gate_inputs(pkg, model, chunks, steps, opts)
    w, b, slack, wmax = solve_gate(X, steps)
    blob = model.blob()
    tab = {n: (shape, off) for n, shape, off in pkg.tensor_table()}
    woff, boff = tab["gate_layer.weight"][1], tab["gate_layer.bias"][1]
    blob[woff : woff + 1536] = w.astype(np.float32)
    blob[boff] = np.float32(b)
    return pkg.Tacotron2.from_blob(blob, device_id=device_id), {"slack_logits": slack, "max_abs_weight": wmax}
