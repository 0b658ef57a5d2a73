"""Pins the CPU oracle's Tacotron2 math against an independent torch (fp64) restatement, and its
structure against the reference's loop/state/stop rules (src/tacotron2/mod.rs:177-233,272-358)."""
import numpy as np
import pytest

import torch_ref
from conftest import rms, synth_ids


def test_weight_table_matches_reference_parameter_counts(orc):
    tab = orc.tensor_table()
    assert len(tab) == 76
    count = lambda pred: sum(n for name, _s, _o, n in tab if pred(name))
    dec = count(lambda n: n.split(".")[0] in ("prenet", "attention_rnn", "decoder_rnn", "linear_projection", "gate_layer") or (n.startswith("attention.") and "memory_layer" not in n))
    # decoder_iter.onnx is 72 766 349 B = 18 189 969 fp32 params + ~6.5 kB of graph
    assert dec == 18_189_969
    post = count(lambda n: n.startswith("postnet") and "bn" not in n)
    assert post == 4_343_888  # postnet.onnx: 17 414 016 B = (4 343 888 + 8 512 BN) fp32 + graph
    enc = count(lambda n: (n.startswith("encoder") or n.startswith("embedding") or "memory_layer" in n) and "bn" not in n)
    assert enc == 5_651_968  # encoder.onnx: 22 641 034 B = (5 651 968 + 6 144 BN) fp32 + graph


def test_symbol_range_and_state_init(orc):
    s = orc.new_state()
    for f in ("att_h", "att_c", "dec_h", "dec_c", "aw", "awc", "ctx", "dec_in"):
        assert not np.any(np.ctypeslib.as_array(getattr(s, f)))  # DecoderState::new: all zeros (mod.rs:202-233)


def test_gate_sigmoid_matches_reference_two_branch_form(orc64):
    # mod.rs:126-133
    for x in (-30.0, -1.5, 0.0, 0.4054651, 2.0, 30.0):
        ref = 1 / (1 + np.exp(-x)) if x >= 0 else np.exp(x) / (1 + np.exp(x))
        assert abs(orc64.sigmoid(x) - ref) < 1e-15
    assert orc64.sigmoid(0.4054652) > 0.6 > orc64.sigmoid(0.4054650)  # logit of the 0.6 threshold (mod.rs:279)


def test_encoder_vs_torch(orc64, blob):
    ids = np.zeros(24, dtype=np.int64)
    ids[:10] = [108, 119, 11, 88, 113, 108, 120, 11, 116, 7]
    mem, pm = orc64.encoder(blob, ids)
    tmem, tpm = torch_ref.encoder(orc64, blob, ids)
    assert np.abs(mem - tmem).max() < 1e-12
    assert np.abs(pm - tpm).max() < 1e-12


def test_decoder_steps_vs_torch_teacher_forced_and_free_running(orc64, blob):
    rng = np.random.default_rng(0)
    T, nv = 20, 13
    mem = rng.standard_normal((T, 512)) * 0.3
    pm = rng.standard_normal((T, 128)) * 0.3
    opts = orc64.default_opts(dropout_mode=1, dropout_seed=42, item=3)
    st = orc64.new_state()
    ts = torch_ref.DecoderState(T)
    for step in range(6):
        mel, gate = orc64.decoder_step(blob, mem, pm, nv, st, opts, step)
        keep0 = [orc64.dropout_keep(42, 3, step, 0, j) for j in range(256)]
        keep1 = [orc64.dropout_keep(42, 3, step, 1, j) for j in range(256)]
        tmel, tgate = torch_ref.decoder_step(orc64, blob, mem, pm, nv, ts, keep0, keep1)
        assert np.abs(mel - tmel).max() < 1e-12, step
        assert abs(gate - tgate) < 1e-12
        aw = np.ctypeslib.as_array(st.aw)[:T]
        assert np.all(aw[nv:] == 0) and abs(aw.sum() - 1) < 1e-12  # mask: mod.rs:219-220
        assert np.abs(aw - ts.aw.numpy()).max() < 1e-13
        assert np.abs(np.ctypeslib.as_array(st.awc)[:T] - ts.awc.numpy()).max() < 1e-13


def test_dropout_is_bernoulli_half_and_off_mode(orc64, blob):
    keeps = np.array([orc64.dropout_keep(1, 0, s, l, j) for s in range(40) for l in range(2) for j in range(256)])
    assert 0.47 < keeps.mean() < 0.53
    rng = np.random.default_rng(1)
    mem, pm = rng.standard_normal((8, 512)) * 0.3, rng.standard_normal((8, 128)) * 0.3
    st, ts = orc64.new_state(), torch_ref.DecoderState(8)
    for step in range(3):
        mel, _ = orc64.decoder_step(blob, mem, pm, 8, st, orc64.default_opts(dropout_mode=0), step)
        tmel, _ = torch_ref.decoder_step(orc64, blob, mem, pm, 8, ts)
        assert np.abs(mel - tmel).max() < 1e-12


def test_postnet_vs_torch(orc64, blob):
    rng = np.random.default_rng(2)
    frames = rng.standard_normal((23, 80))
    assert np.abs(orc64.postnet(blob, frames) - torch_ref.postnet(orc64, blob, frames)).max() < 1e-11


def test_run_decoder_stop_rule(orc64, orc, blob):
    """mod.rs:302-342: stop when sigmoid(gate) > 0.6 or i+1 == max_steps; the tripping frame is kept."""
    rng = np.random.default_rng(3)
    mem, pm = rng.standard_normal((12, 512)) * 0.3, rng.standard_normal((12, 128)) * 0.3
    o = orc64.default_opts(max_steps=9)
    frames, gates = orc64.run_decoder(blob, mem, pm, 12, o)
    assert len(frames) == 9 and np.all(gates < 0.4054651)  # synthetic gate never fires: capped at max_steps
    # rig the gate bias so the logit crosses the threshold at a known step
    rig = blob.copy()
    off = [t for t in orc64.tensor_table() if t[0] == "gate_layer.bias"][0][2]
    k = 5
    rig[off] += np.float32(0.4054651 - float(np.sort(gates)[-k]) + 1e-4)
    f2, g2 = orc64.run_decoder(rig, mem, pm, 12, orc64.default_opts(max_steps=9))
    first = int(np.argmax(g2 > 0.4054651))
    assert len(f2) == first + 1 and orc64.sigmoid(float(g2[-1])) > 0.6
    assert np.allclose(f2, frames[: first + 1])  # frames before the stop are unaffected by the gate bias
    # fixed_steps overrides the gate
    f3, _ = orc64.run_decoder(rig, mem, pm, 12, orc64.default_opts(fixed_steps=7))
    assert len(f3) == 7


def test_infer_chunk_pad_and_mask_quirk(orc64, blob):
    """mod.rs:361-393: ids are zero-padded to the window and the ENCODER sees the padded length
    (plen = 100), while the decoder mask uses the un-padded length."""
    ids = synth_ids(9)
    o = orc64.default_opts(fixed_steps=5, dropout_seed=2)
    a = orc64.infer_chunk(blob, ids, o, window=32)
    padded = np.zeros(32, dtype=np.int64)
    padded[:9] = ids
    mem, pm = orc64.encoder(blob, padded)
    frames, _ = orc64.run_decoder(blob, mem, pm, 9, o)
    assert np.array_equal(a, orc64.postnet(blob, frames))
    assert a.shape == (80, 5)


def test_f32_oracle_tracks_f64(orc, orc64, blob):
    ids = np.zeros(100, dtype=np.int64)
    ids[:30] = synth_ids(30)
    o32, o64 = orc.default_opts(fixed_steps=25, dropout_seed=9), orc64.default_opts(fixed_steps=25, dropout_seed=9)
    a, b = orc.infer_chunk(blob, ids[:30], o32), orc64.infer_chunk(blob, ids[:30], o64)
    assert rms(a, b) < 1e-6
