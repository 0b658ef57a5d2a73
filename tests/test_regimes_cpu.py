"""The "trained-like" weight regime of tests/regimes.py is what it claims to be (oracle only, no GPU): near-one-hot attention
that moves forward over a -inf-masked tail (mod.rs:219-220), softmax weights that underflow to exactly 0, LSTM
pre-activations on both rails, log-mel frames over the range of a real checkpoint, a gate that crosses 0.6 by itself
(mod.rs:319-324), and an f32 oracle that stays close to the f64 oracle there (so 1e-5 per step is a meaningful bar)."""
import numpy as np
import pytest

from conftest import synth_ids
from regimes import attention_lstm_preactivation_without_prenet, speech_like_mel, start_at_first_position, trained_like


@pytest.fixture(scope="module")
def hot(orc):
    return trained_like(orc, 20240327)


def test_attention_is_near_one_hot_moves_forward_and_underflows(orc, hot):
    n = 91
    ids = np.zeros(100, dtype=np.int64)
    ids[:n] = synth_ids(n, seed=32)
    mem, pm = orc.encoder(hot, ids)
    assert 0.3 < np.abs(mem).max() < 1.0 and mem.std() > 0.1
    st, o = start_at_first_position(orc.new_state(), mem), orc.default_opts(dropout_seed=11, item=3)
    peaks, tops, zeros, pre_max = [], [], [], 0.0
    for step in range(60):
        if step >= 5:
            pre_max = max(pre_max, float(np.abs(attention_lstm_preactivation_without_prenet(orc, hot, st)).max()))
        mel, _gate = orc.decoder_step(hot, mem, pm, n, st, o, step)
        aw = np.array(st.aw, dtype=np.float32)[:100]
        assert np.all(aw[n:] == 0.0)                         # the masked tail: exactly zero
        assert abs(float(aw.sum()) - 1.0) < 1e-5
        if step >= 4:
            peaks.append(int(aw.argmax()))
            tops.append(float(aw.max()))
            zeros.append(int(np.count_nonzero(aw[:n] == 0.0)))
    assert np.mean(np.array(tops) > 0.9) > 0.75, tops          # near one-hot most of the time, soft at the hand-overs
    assert min(tops) < 0.9
    d = np.diff(peaks)
    assert np.all(d >= 0) and 8 <= peaks[-1] - peaks[0] <= 40, peaks   # monotone, a few frames per position
    assert max(zeros) >= 5, zeros                              # valid positions far behind the peak underflow to exactly 0
    assert pre_max > 10.0, pre_max                             # gate pre-activations on the rails (before the prenet columns)
    assert np.abs(np.array(st.dec_c)).max() > 8.0             # cell states far outside the +-1 of the plain synthetic draw
    assert mel.min() < -8.0 and mel.max() > -1.0              # the log-mel range of a real checkpoint


def test_f32_oracle_stays_near_the_f64_oracle_in_the_trained_like_regime(orc, orc64, hot):
    n = 37
    ids = np.zeros(100, dtype=np.int64)
    ids[:n] = synth_ids(n, seed=31)
    mem, pm = orc.encoder(hot, ids)
    f32, g32 = orc.run_decoder(hot, mem, pm, n, orc.default_opts(fixed_steps=120, dropout_seed=11, item=3))
    f64, g64 = orc64.run_decoder(hot, mem.astype(np.float64), pm.astype(np.float64), n, orc64.default_opts(fixed_steps=120, dropout_seed=11, item=3))
    assert np.abs(f64).max() > 9.0
    assert np.sqrt(np.mean((f32 - f64) ** 2)) < 5e-6 and np.abs(g32 - g64).max() < 5e-6


def test_the_natural_gate_stops_by_itself(orc):
    blob = trained_like(orc, 7, natural_gate=True)
    counts = []
    for i, n in enumerate((37, 91, 12, 58)):
        ids = np.zeros(100, dtype=np.int64)
        ids[:n] = synth_ids(n, seed=40 + i)
        mem, pm = orc.encoder(blob, ids)
        frames, gates = orc.run_decoder(blob, mem, pm, n, orc.default_opts(max_steps=400, dropout_seed=3, item=i))
        counts.append(len(frames))
        assert gates[-1] > np.log(0.6 / 0.4) or len(frames) == 400
        assert np.all(gates[:-1] <= np.log(0.6 / 0.4) + 1e-6)
    assert len(set(counts)) > 1 and min(counts) >= 2 and sorted(counts)[1] < 400, counts


def test_speech_like_mel_has_a_floor_and_a_voice(orc):
    mel = speech_like_mel(120)
    assert mel.shape == (80, 120) and abs(mel.min() - np.log(1e-5)) < 1e-5 and 0.0 < mel.max() < 3.0
    assert np.mean(mel < -11.0) > 0.2
    basis = orc.mel_filter_bank()
    raw = orc.pinv(basis).astype(np.float64) @ np.exp(mel.astype(np.float64))
    assert np.mean(raw < 0) > 0.01      # the pseudo-inverse goes negative: the clip of mel->linear is exercised
