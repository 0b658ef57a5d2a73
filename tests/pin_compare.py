"""What a recorded run of the reference's graphs (tools/pin/record_run.py -> .npz) is compared with: the product, loaded
through Tacotron2::load(dir) from the SAME model directory, driven with the SAME dropout masks (dropout_mode 2).
Shared by tests/test_gpu_reference_pinned.py (the real pin, skipped until the artefacts exist) and
tests/test_gpu_pin_kit_dry_run.py (the kit end to end on synthetic weights).  Bound: the north star's 1e-4 RMS."""
import json

import numpy as np

STATE = ["attention_hidden", "attention_cell", "decoder_hidden", "decoder_cell", "attention_weights", "attention_weights_cum", "attention_context"]


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def compare_run(pkg, model_dir, rec, tol=1e-4):
    """Returns the margins; raises AssertionError on the first figure beyond `tol`."""
    m = pkg.Tacotron2.load(model_dir)
    out = {}
    try:
        ids = np.asarray(rec["ids"], dtype=np.int64)
        n_valid, steps = int(rec["n_valid"]), rec["frames"].shape[0]
        mem, pm = m.encoder(ids)                                                  # encoder.onnx, mod.rs:379
        out["memory_rms"], out["processed_memory_rms"] = rms(mem, rec["memory"]), rms(pm, rec["processed_memory"])
        assert out["memory_rms"] <= tol and out["processed_memory_rms"] <= tol, out
        keep = np.ascontiguousarray(rec["keep_masks"], dtype=np.uint8)[None]       # (1 chunk, steps, 2, 256)
        live = "dropout_inputs" not in rec or len(json.loads(str(rec["dropout_inputs"]))) > 0
        # (a graph whose exporter folded the dropout away draws nothing: the recording then ran without masks = dropout_mode 0)
        mk = (lambda **kw: pkg.default_opts(dropout_masks=keep, **kw)) if live else (lambda **kw: pkg.default_opts(dropout_mode=0, **kw))
        # the decoder loop from the RECORDED encoder output (so the decoder's figure is its own), mod.rs:302-342
        frames, gates = m.decoder(rec["memory"], rec["processed_memory"], n_valid, mk(fixed_steps=steps))
        out["frames_rms"], out["gate_max_abs"] = rms(frames, rec["frames"]), float(np.abs(gates - rec["gates"]).max())
        scale = max(1.0, float(np.abs(rec["frames"]).max()))
        assert frames.shape == rec["frames"].shape and out["frames_rms"] <= tol * scale and out["gate_max_abs"] <= tol * max(1.0, float(np.abs(rec["gates"]).max())), out
        # the seven state tensors after steps 0, 1, 5 and the last: one teacher-forced decoder_iter call each from the recorded state before
        for s in (1, 5, steps - 1):
            key = "state_step%d_" % (s - 1)
            if key + STATE[0] not in rec:
                continue
            st = {k: rec[key + k][None] for k in STATE}
            _o, _g, new = m.decoder_steps("persistent", rec["memory"][None], rec["processed_memory"][None], [n_valid], st, rec["frames"][s - 1][None], s, 1,
                                          opts=mk())
            if "state_step%d_%s" % (s, STATE[0]) in rec:
                worst = max(float(np.abs(new[k][0] - rec["state_step%d_%s" % (s, k)]).max()) / max(1.0, float(np.abs(rec["state_step%d_%s" % (s, k)]).max())) for k in STATE)
                out["state_step%d_worst_rel" % s] = worst
                assert worst <= tol, (s, worst)
        mel = m.postnet(rec["frames"])                                             # postnet.onnx on the RECORDED frames, mod.rs:347
        out["postnet_rms"] = rms(mel, rec["mel_postnet"])
        assert mel.shape == rec["mel_postnet"].shape and out["postnet_rms"] <= tol * max(1.0, float(np.abs(rec["mel_postnet"]).max())), out
        # and the whole chain the way Tacotron2::infer runs it (encoder -> loop -> post-net), mod.rs:361-393
        whole = m.infer_batch([ids[:n_valid]], opts=mk(), fixed_steps=np.array([steps], dtype=np.int32))[0]
        out["infer_rms"] = rms(whole, rec["mel_postnet"])
        assert out["infer_rms"] <= tol * max(1.0, float(np.abs(rec["mel_postnet"]).max())), out
    finally:
        m.close()
    return out
