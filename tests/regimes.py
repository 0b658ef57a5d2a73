"""Weight regimes for the parity tests (TEST INFRASTRUCTURE: oracle-side blob transforms, the product is untouched).

The synthetic weights of BASELINE.md section 3 (`U(+-1/sqrt(fan_in))`, identity BatchNorm) put every kernel in ONE numeric
regime: diffuse attention, O(1) LSTM pre-activations, nothing saturates, no softmax weight underflows.  A trained
checkpoint (the reference loads NVIDIA's, /root/reference/src/tacotron2/mod.rs:137-138) lives in the opposite one.
`trained_like` rewrites a synthetic blob so that the decoder loop (mod.rs:302-342) sees

* location-sensitive attention that is near one-hot and MOVES: filter 0 of `location_conv` reads the previous alignment at
  t-1 and t, filter 1 the cumulative alignment at t; `location_dense` sends both along sign(v), so a position is held for a
  few frames (stay bonus), then pushed on (the cumulative penalty grows), positions behind the peak are pushed to
  -sum|v| (their softmax weight underflows to exactly 0) and the tail beyond `n_valid` is -inf (mod.rs:219-220);
* LSTM pre-activations of +-10...15: `weight_ih` x5, `weight_hh` x3, `bias_ih` += N(0, 3), forget-gate `bias_hh` += U(1, 3)
  (cell states grow to +-30, gates sit on both rails);
* log-mel frames spanning about -12...2: `linear_projection.bias` = linspace(-9, -1), `linear_projection.weight` x6;
* non-identity BatchNorm statistics in all eight BatchNorm layers (encoder x3, post-net x5);
* an encoder whose memory is O(0.5) instead of O(0.06) (`encoder.lstm.*.weight_ih` x4, `weight_hh` x2);
* optionally a gate layer that crosses the 0.6 threshold by itself (`natural_gate`): x40, the bias set from the logit's mean
  and spread on one probe chunk so that it hovers 1.5 standard deviations below the threshold -- the stop rule
  (mod.rs:319-324) fires at a step nobody solved for, a different one for every chunk.

tests/test_regimes_cpu.py asserts with the oracle alone that the regime is what this docstring says."""
import numpy as np


def trained_like(orc, seed=20240327, natural_gate=False):
    blob = orc.weights_synthetic(seed=seed, rec_scale=1.0).copy()
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    t = lambda name: orc.tensor(blob, name)  # noqa: E731  (views into blob)
    for name, _shape, off, numel in orc.tensor_table():
        v = blob[off : off + numel]
        if name.endswith("bn.weight"):
            v[:] = rng.uniform(0.5, 1.5, numel)
        elif name.endswith("bn.bias"):
            v[:] = rng.uniform(-0.3, 0.3, numel)
        elif name.endswith("bn.running_mean"):
            v[:] = rng.uniform(-0.2, 0.2, numel)
        elif name.endswith("bn.running_var"):
            v[:] = rng.uniform(0.5, 2.0, numel)
    for d in ("fwd", "bwd"):
        t("encoder.lstm.%s.weight_ih" % d)[:] *= 4.0
        t("encoder.lstm.%s.weight_hh" % d)[:] *= 2.0
    for cell in ("attention_rnn", "decoder_rnn"):
        t(cell + ".weight_ih")[:] *= 5.0
        t(cell + ".weight_hh")[:] *= 3.0
        t(cell + ".bias_ih")[:] += rng.normal(0.0, 3.0, 4096).astype(np.float32)
        t(cell + ".bias_hh")[1024:2048] += rng.uniform(1.0, 3.0, 1024).astype(np.float32)  # PyTorch gate order i, f, g, o
    v = t("attention.v.weight")
    v[:] *= 20.0
    t("attention.query_layer.weight")[:] *= 2.0
    t("attention.memory_layer.weight")[:] *= 3.0
    sg = np.sign(v).astype(np.float32)
    lc = t("attention.location_conv.weight")   # (32 filters, 2 channels = [previous ; cumulative], 31 taps, pad 15)
    ld = t("attention.location_dense.weight")  # (128, 32)
    lc[0] = 0.0
    lc[0, 0, 14] = 1.0    # previous alignment at t-1: move on
    lc[0, 0, 15] = 1.3    # previous alignment at t: stay
    ld[:, 0] = sg * 1.5
    lc[1] = 0.0
    lc[1, 1, 15] = 1.0    # cumulative alignment at t: penalty
    ld[:, 1] = -sg * 0.12
    t("linear_projection.bias")[:] = np.linspace(-9.0, -1.0, 80).astype(np.float32)
    t("linear_projection.weight")[:] *= 6.0
    if natural_gate:
        # calibrated on the STATISTICS of one probe chunk (not solved per step like xd-tts_amd/gate_rig.py): the logit's
        # mean over 120 free-running steps is put 1.5 of its standard deviations below the threshold's logit
        t("gate_layer.weight")[:] *= 40.0
        probe = np.zeros(100, dtype=np.int64)
        probe[:50] = 64 + (np.arange(50) * 7) % 84
        mem, pm = orc.encoder(blob, probe)
        _f, g = orc.run_decoder(blob, mem, pm, 50, orc.default_opts(fixed_steps=120, dropout_seed=1, item=0))
        t("gate_layer.bias")[:] += np.float32(np.log(0.6 / 0.4) - 1.5 * g[10:].std() - g[10:].mean())
    return blob


def start_at_first_position(state, memory):
    """DecoderState::new (mod.rs:202-233) starts from an all-zero alignment, and which position wins the first softmax of a
    random-weight model is arbitrary.  A trained model starts at position 0: put the alignment there (previous = cumulative =
    one-hot at 0, context = memory[0], which is what that alignment implies), everything else stays zero."""
    state.aw[0] = 1.0
    state.awc[0] = 1.0
    for j in range(512):
        state.ctx[j] = float(memory[0, j])
    return state


def attention_lstm_preactivation_without_prenet(orc, blob, state):
    """W_hh.h + W_ih[:, 256:].ctx + b_ih + b_hh of the attention LSTM (everything but the prenet columns), numpy."""
    h = np.array(state.att_h, dtype=np.float64)
    ctx = np.array(state.ctx, dtype=np.float64)
    w_ih = orc.tensor(blob, "attention_rnn.weight_ih").astype(np.float64)
    w_hh = orc.tensor(blob, "attention_rnn.weight_hh").astype(np.float64)
    return w_hh @ h + w_ih[:, 256:] @ ctx + orc.tensor(blob, "attention_rnn.bias_ih") + orc.tensor(blob, "attention_rnn.bias_hh")


def speech_like_mel(F, seed=5):
    """A log-mel (80, F) with the shape of a Tacotron2 output: a -11.5 floor (ln 1e-5, the compression's clamp) in pauses and
    above the voice band, harmonics-like ridges up to about +1.5, smooth in time."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(F)[None, :]
    m = np.arange(80)[:, None]
    f0 = 6.0 + 2.0 * np.sin(2 * np.pi * t / 97.0)
    ridge = sum(np.exp(-0.5 * ((m - k * f0) / 1.3) ** 2) for k in range(1, 9))
    envelope = np.exp(-m / 35.0)
    voiced = (np.sin(2 * np.pi * t / 61.0 + 0.7) > -0.4).astype(np.float64)
    lin = 1e-5 + 4.0 * voiced * envelope * ridge + 3e-4 * rng.random((80, F)) * voiced
    mel = np.log(np.maximum(lin, 1e-5))
    mel[:, : min(8, F)] = np.log(1e-5)      # leading silence: every band on the floor
    return mel.astype(np.float32)
