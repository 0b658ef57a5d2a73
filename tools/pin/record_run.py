#!/usr/bin/env python3
"""Pin kit, step 2: record ONE run of the reference's three graphs with known dropout masks.

Runs OUTSIDE the build container, wherever the real artefacts and onnxruntime are (pip install onnxruntime numpy):

    python tools/pin/patch_decoder_iter.py models/tacotron2/decoder_iter.onnx /tmp/decoder_iter.pinned.onnx
    python tools/pin/record_run.py models/tacotron2 /tmp/decoder_iter.pinned.onnx tests/golden/reference_run.npz

What it does is what the reference does (file:line under /root/reference/src/tacotron2/mod.rs), with the randomness supplied:
  ids      the reference's own known-answer ids (mod.rs:470-483, 28 phoneme ids), padded with 0 to the 100-id window
           (:369-371), plen = [100] (:375)
  encoder  encoder.onnx (:379) -> memory, processed_memory
  state    DecoderState::new (:202-233): zeros; mask true for t >= 28 (:219-220)
  loop     N calls of decoder_iter (:304) threading the seven state tensors (:332-339); the prenet's two dropout layers get
           the recorded keep masks (PCG64(seed), p = 0.5) through the inputs patch_decoder_iter.py added
  postnet  postnet.onnx (:347) on the N frames -> (80, N) mel (:349-355)
and writes everything a comparison needs into one .npz (tests/test_gpu_reference_pinned.py reads it; the product side is
`Tacotron2::load(dir)` + `dropout_mode = 2` with the same masks).  N is fixed (default 48): the stop rule is not part of the
recording, the step where sigmoid(gate) first exceeds 0.6 is stored beside it.

`--backend torch` (build container, tests only) swaps onnxruntime for the torch modules of tests/nvidia_torch_export.py
loaded with the SAME weights the exported directory holds: a dry run of the whole kit on synthetic weights
(tests/test_gpu_pin_kit_dry_run.py) -- not a pin.
"""
import argparse
import json
import os
import sys

import numpy as np

KAT_IDS = [108, 119, 11, 88, 113, 108, 120, 11, 116, 73, 118, 129, 70, 130, 73, 133, 108, 143,
           117, 114, 11, 118, 66, 90, 97, 119, 11, 7]   # mod.rs:486-489 (correct_phoneme_id_output)
WINDOW = 100                                             # mod.rs:363,369-371
STATE = [("attention_hidden", 1024), ("attention_cell", 1024), ("decoder_hidden", 1024), ("decoder_cell", 1024),
         ("attention_weights", WINDOW), ("attention_weights_cum", WINDOW), ("attention_context", 512)]   # mod.rs:285-295
OUT_STATE = ["out_" + n for n, _ in STATE]                                                                 # mod.rs:332-339


class OrtBackend:
    def __init__(self, model_dir, pinned_decoder):
        import onnxruntime as ort

        so = ort.SessionOptions()
        so.graph_optimization_level = ort.GraphOptimizationLevel.ORT_ENABLE_ALL   # the reference: Level3 (mod.rs:246-259)
        mk = lambda p: ort.InferenceSession(p, so, providers=["CPUExecutionProvider"])  # noqa: E731
        self.enc, self.dec, self.post = mk(os.path.join(model_dir, "encoder.onnx")), mk(pinned_decoder), mk(os.path.join(model_dir, "postnet.onnx"))
        self.dec_inputs = [i.name for i in self.dec.get_inputs()]
        self.version = "onnxruntime " + ort.__version__

    def encoder(self, ids, lens):
        names = [i.name for i in self.enc.get_inputs()]
        out = self.enc.run(None, {names[0]: ids, names[1]: lens})          # positional, as the reference binds them (mod.rs:379-385)
        return out[0], out[1]

    def decoder(self, feed):
        names = [o.name for o in self.dec.get_outputs()]
        return dict(zip(names, self.dec.run(None, feed)))

    def postnet(self, mel):
        return self.post.run(None, {self.post.get_inputs()[0].name: mel})[0]


def dropout_feeds(new_inputs, keep_step):
    """keep_step (2, 256) uint8 -> the values of the inputs patch_decoder_iter.py added, in graph order (prenet layer 0, 1)."""
    feed, layer = {}, 0
    for ni in new_inputs:
        keep = keep_step[layer].astype(np.float32).reshape(ni["shape"])
        if ni["kind"] == "scale":
            feed[ni["name"]] = keep * np.float32(1.0 / (1.0 - ni.get("ratio", 0.5)))
        elif ni["kind"] == "uniform":
            feed[ni["name"]] = np.where(keep > 0, np.float32(0.25), np.float32(0.75))   # Less(u, 0.5): keep <=> u < 0.5
        else:
            raise SystemExit("input %s replaces a %s node: decide what to feed it" % (ni["name"], ni["kind"]))
        layer += 1
    return feed


def record(backend, new_inputs, steps, seed):
    ids = np.zeros((1, WINDOW), dtype=np.int64)
    ids[0, : len(KAT_IDS)] = KAT_IDS
    memory, pmem = backend.encoder(ids, np.array([WINDOW], dtype=np.int64))
    n_valid = len(KAT_IDS)
    mask = np.zeros((1, WINDOW), dtype=bool)
    mask[0, n_valid:] = True
    keep = (np.random.Generator(np.random.PCG64(seed)).random((steps, 2, 256)) < 0.5).astype(np.uint8)
    st = {n: np.zeros((1, d), dtype=np.float32) for n, d in STATE}
    dec_in = np.zeros((1, 80), dtype=np.float32)
    frames, gates, trace = [], [], {}
    for s in range(steps):
        feed = dict(st, decoder_input=dec_in, memory=memory, processed_memory=pmem, mask=mask)
        feed.update(dropout_feeds(new_inputs, keep[s]))
        out = backend.decoder(feed)
        frames.append(out["decoder_output"][0])
        gates.append(float(np.asarray(out["gate_prediction"]).reshape(-1)[0]))
        st = {n: np.asarray(out["out_" + n], dtype=np.float32) for n, _ in STATE}
        dec_in = out["decoder_output"].astype(np.float32)
        if s in (0, 1, 4, 5, steps - 2, steps - 1):   # (pairs: a teacher-forced call from the state BEFORE steps 1, 5 and the last)
            for n, _ in STATE:
                trace["state_step%d_%s" % (s, n)] = st[n][0]
    frames = np.stack(frames).astype(np.float32)
    gates = np.asarray(gates, dtype=np.float32)
    mel_post = backend.postnet(frames.T[None].copy())[0]
    sig = 1.0 / (1.0 + np.exp(-gates.astype(np.float64)))
    stop = int(np.argmax(sig > 0.6)) if (sig > 0.6).any() else -1
    return dict(ids=ids[0], n_valid=np.int32(n_valid), keep_masks=keep, memory=memory[0], processed_memory=pmem[0], frames=frames, gates=gates,
                mel_postnet=np.asarray(mel_post, dtype=np.float32), first_step_over_threshold=np.int32(stop), **trace)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("model_dir")
    ap.add_argument("pinned_decoder", help="output of patch_decoder_iter.py (its .json report must sit beside it)")
    ap.add_argument("out_npz")
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--seed", type=int, default=20240327)
    ap.add_argument("--backend", choices=["ort", "torch"], default="ort")
    a = ap.parse_args(argv)
    report = json.load(open(a.pinned_decoder + ".json"))
    if a.backend == "ort":
        backend = OrtBackend(a.model_dir, a.pinned_decoder)
    else:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
        from pin_torch_backend import TorchBackend

        backend = TorchBackend(a.model_dir)
    if len(report["new_inputs"]) not in (0, 2):
        raise SystemExit("expected the prenet's two dropout layers, the patcher found %d random nodes" % len(report["new_inputs"]))
    rec = record(backend, report["new_inputs"], a.steps, a.seed)
    np.savez_compressed(a.out_npz, recorded_with=np.array(backend.version), dropout_inputs=np.array(json.dumps(report["new_inputs"])), **rec)
    print("recorded %d steps with %s -> %s (first step with sigmoid(gate) > 0.6: %d)" % (a.steps, backend.version, a.out_npz, int(rec["first_step_over_threshold"])))
    return 0


if __name__ == "__main__":
    sys.exit(main())
