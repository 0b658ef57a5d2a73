"""GPU parity through the engines that serve the BASELINE configs.

* ONE decoder_iter call (mod.rs:304) from the oracle's own state, all nine outputs
  (mod.rs:306-307,332-339) to 1e-5 -- through the persistent weight-stationary kernel (configs[1]) and
  through the batched MFMA kernels (configs[2]/[3]), not only the launch-per-stage GEMV path.
* The engines' written-back state after a free run of 60 steps (what a later launch continues from).
* dropout_mode = 2: caller-supplied prenet keep masks (SURVEY 8(b) "explicit(mask ptr)", the hook that
  lets one recorded run of the real decoder_iter.onnx be compared) on all three engines.
* a batched decode re-run after an encoder exchange timeout attends over the fresh
  processed_memory transpose."""
import os

import numpy as np
import pytest

from conftest import rms, synth_ids

pytestmark = pytest.mark.gpu

NAMES = {"attention_hidden": "att_h", "attention_cell": "att_c", "decoder_hidden": "dec_h", "decoder_cell": "dec_c",
         "attention_weights": "aw", "attention_weights_cum": "awc", "attention_context": "ctx"}


class env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update({k: str(v) for k, v in self.kw.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def encode(orc, blob, n, seed):
    ids = np.zeros(100, dtype=np.int64)
    ids[:n] = synth_ids(n, seed=seed)
    return orc.encoder(blob, ids)


def snapshot(states, T):
    """The oracle's per-chunk states as (B, ...) arrays under the reference's tensor names."""
    out = {}
    for k, v in NAMES.items():
        out[k] = np.stack([np.array(getattr(s, v), dtype=np.float32)[: (T if v in ("aw", "awc") else None)] for s in states])
    return out


def chunks_for(orc, blob, B):
    lens = [37, 91, 12, 58, 23, 100, 64, 5, 77, 19, 46, 83, 30, 68, 9, 52][:B]
    enc = [encode(orc, blob, n, seed=31 + i) for i, n in enumerate(lens)]
    mem = np.stack([e[0] for e in enc])
    pm = np.stack([e[1] for e in enc])
    return lens, mem, pm


@pytest.mark.parametrize("engine,B", [("persistent", 1), ("persistent", 2), ("batched", 6), ("launch", 3), ("persistent8", 3), ("persistent8", 4), ("persistent8", 8), ("persistent8", 9), ("persistent8", 16)])
def test_teacher_forced_step_all_nine_outputs_per_engine(pkg, model, orc, blob, engine, B):
    lens, mem, pm = chunks_for(orc, blob, B)
    T = mem.shape[1]
    base = 3
    opts = [orc.default_opts(dropout_seed=11, item=base + b) for b in range(B)]
    go = pkg.default_opts(dropout_seed=11, item_base=base)
    sts = [orc.new_state() for _ in range(B)]
    worst = 0.0
    for step in range(12):
        snap = snapshot(sts, T)
        dec_in = np.stack([np.array(s.dec_in, dtype=np.float32) for s in sts])
        ref = [orc.decoder_step(blob, mem[b], pm[b], lens[b], sts[b], opts[b], step) for b in range(B)]  # advances the oracle
        if step not in (0, 1, 5, 11):
            continue
        out, gate, gst = model.decoder_steps(engine, mem, pm, lens, snap, dec_in, step, 1, opts=go)
        after = snapshot(sts, T)
        errs = [float(np.abs(out[:, 0] - np.stack([r[0] for r in ref])).max()), float(np.abs(gate[:, 0] - np.array([r[1] for r in ref])).max())]
        errs += [float(np.abs(gst[k] - after[k]).max()) for k in NAMES]
        worst = max(worst, max(errs))
        assert max(errs) <= 1e-5, (engine, B, step, errs)
    assert worst > 0  # two different implementations really were compared


@pytest.mark.parametrize("engine,B", [("persistent", 1), ("persistent", 2), ("batched", 6), ("batched", 1), ("launch", 2), ("persistent8", 3), ("persistent8", 8), ("persistent8", 1), ("persistent8", 12), ("persistent8", 16)])
def test_written_back_state_after_a_60_step_run(pkg, model, orc, blob, engine, B):
    """n_steps = 60 from DecoderState::new (mod.rs:202-233): every frame, every gate logit and the seven state
    tensors the engine leaves behind against the oracle's after the same 60 calls."""
    lens, mem, pm = chunks_for(orc, blob, B)
    T = mem.shape[1]
    n = 60
    opts = [orc.default_opts(dropout_seed=7, item=b) for b in range(B)]
    sts = [orc.new_state() for _ in range(B)]
    zero = snapshot(sts, T)
    frames = np.zeros((B, n, 80), dtype=np.float32)
    gates = np.zeros((B, n), dtype=np.float32)
    for step in range(n):
        for b in range(B):
            frames[b, step], gates[b, step] = orc.decoder_step(blob, mem[b], pm[b], lens[b], sts[b], opts[b], step)
    out, gate, gst = model.decoder_steps(engine, mem, pm, lens, zero, np.zeros((B, 80), dtype=np.float32), 0, n, opts=pkg.default_opts(dropout_seed=7))
    after = snapshot(sts, T)
    assert rms(out, frames) <= 1e-5 and np.abs(gate - gates).max() <= 1e-5
    for k in NAMES:
        assert np.abs(gst[k] - after[k]).max() <= 1e-5, (engine, k, float(np.abs(gst[k] - after[k]).max()))


@pytest.mark.parametrize("engine,B", [("persistent", 2), ("batched", 5), ("launch", 1), ("persistent8", 5), ("persistent8", 11)])
def test_step_hook_with_a_short_encoder_window_and_an_odd_step_count(pkg, model, orc, blob, engine, B):
    """T = 64 rows of encoder memory (not the reference's 100), 7 steps from step 3 of the oracle's run: odd counts end on the
    other ping-pong half of the launch-per-stage and batched engines."""
    T, n0, n = 64, 3, 7
    rng = np.random.Generator(np.random.PCG64(77))
    mem = (rng.standard_normal((B, T, 512)) * 0.5).astype(np.float32)
    pm = (rng.standard_normal((B, T, 128)) * 0.5).astype(np.float32)
    lens = [64, 41, 9, 57, 30, 22, 60, 13, 48, 35, 5][:B]
    opts = [orc.default_opts(dropout_seed=3, item=10 + b) for b in range(B)]
    sts = [orc.new_state() for _ in range(B)]
    for step in range(n0):
        for b in range(B):
            orc.decoder_step(blob, mem[b], pm[b], lens[b], sts[b], opts[b], step)
    snap = snapshot(sts, T)
    dec_in = np.stack([np.array(s.dec_in, dtype=np.float32) for s in sts])
    frames = np.zeros((B, n, 80), dtype=np.float32)
    gates = np.zeros((B, n), dtype=np.float32)
    for i in range(n):
        for b in range(B):
            frames[b, i], gates[b, i] = orc.decoder_step(blob, mem[b], pm[b], lens[b], sts[b], opts[b], n0 + i)
    out, gate, gst = model.decoder_steps(engine, mem, pm, lens, snap, dec_in, n0, n, opts=pkg.default_opts(dropout_seed=3, item_base=10))
    after = snapshot(sts, T)
    assert np.abs(out - frames).max() <= 1e-5 and np.abs(gate - gates).max() <= 1e-5
    for k in NAMES:
        assert np.abs(gst[k] - after[k]).max() <= 1e-5, (engine, k, float(np.abs(gst[k] - after[k]).max()))


def test_persistent_step_hook_refuses_an_inconsistent_context(pkg, model, orc, blob):
    lens, mem, pm = chunks_for(orc, blob, 1)
    st = snapshot([orc.new_state()], mem.shape[1])
    st["attention_context"][0, 3] = 0.5   # not attention_weights . memory (all-zero weights)
    with pytest.raises(pkg.XdttsError) as e:
        model.decoder_steps("persistent", mem, pm, lens, st, np.zeros((1, 80), dtype=np.float32), 0, 1)
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    with pytest.raises(pkg.XdttsError):
        model.decoder_steps("persistent", np.repeat(mem, 3, 0), np.repeat(pm, 3, 0), lens * 3, {k: np.repeat(v, 3, 0) for k, v in st.items()},
                            np.zeros((3, 80), dtype=np.float32), 0, 1)  # three chunks: not the persistent engine's shape


@pytest.mark.parametrize("engine,B", [("persistent", 1), ("persistent", 2), ("launch", 2), ("persistent8", 3), ("persistent8", 6), ("persistent8", 10), ("batched", 17)])
def test_explicit_dropout_masks_on_every_engine(pkg, orc, blob, engine, B):
    """dropout_mode 2: the caller's keep bytes [chunk][step][layer][unit] replace the seeded stream; every chunk must
    equal the oracle run with ITS masks (the batched engine sorts the chunks by length internally)."""
    lens = [37, 91, 12, 58, 23, 100, 45, 71, 8, 66, 29, 84, 17, 53, 95, 40, 62][:B]
    ids = [synth_ids(n, seed=61 + i) for i, n in enumerate(lens)]
    steps = [24, 40, 16, 33, 40, 9, 28, 36, 12, 31, 19, 40, 7, 26, 38, 14, 22][:B]
    rng = np.random.Generator(np.random.PCG64(99))
    masks = (rng.random((B, 40, 2, 256)) < 0.5).astype(np.uint8)
    masks[0, :, 1, :] *= 3  # any non-zero byte keeps
    o = pkg.default_opts(dropout_masks=masks, item_base=17, dropout_seed=1234)  # (seed and item_base must not matter)
    with env(**({"XDTTS_DECODER": "launch"} if engine == "launch" else {})):
        m = pkg.Tacotron2.from_blob(blob)
        mels = m.infer_batch(ids, opts=o, fixed_steps=np.array(steps, dtype=np.int32))
        st = m.engine_state()
        assert st["decoder_persistent8"] == (1 if engine == "persistent8" else -1), st  # (the batch ran on the engine the label names)
        m.close()
    for b in range(B):
        ref = orc.infer_chunk(blob, ids[b], orc.default_opts(fixed_steps=steps[b], masks=masks[b]))
        assert mels[b].shape == ref.shape == (80, steps[b])
        assert rms(mels[b], ref) <= 1e-5, (engine, b, rms(mels[b], ref))
    # and the masks really were used: the seeded stream's result lies well outside the tolerance (the synthetic weights
    # make the mel only weakly dependent on the prenet output: 7e-5 RMS between two dropout realisations)
    seeded = orc.infer_chunk(blob, ids[0], orc.default_opts(fixed_steps=steps[0], dropout_seed=1234, item=17))
    assert rms(mels[0], seeded) > 3e-5


def test_explicit_dropout_masks_with_the_gate_on(pkg, orc, blob):
    """The reference's real mode: the stop rule decides the frame count (mod.rs:319-324).  The masks must then cover max_steps;
    frame counts and frames equal the oracle's with the same masks (persistent engine, one chunk)."""
    from test_gpu_tacotron2_more import rigged_gate_blob

    n = 33
    ids = np.zeros(100, dtype=np.int64)
    ids[:n] = synth_ids(n, seed=1)
    mem, pm = orc.encoder(blob, ids)
    rig = rigged_gate_blob(orc, blob, mem, pm, n, 21, 30)
    rng = np.random.Generator(np.random.PCG64(4))
    masks = (rng.random((1, 90, 2, 256)) < 0.5).astype(np.uint8)
    rframes, rgates = orc.run_decoder(rig, mem, pm, n, orc.default_opts(max_steps=90, masks=masks[0]))
    assert 1 <= len(rframes) < 90
    m = pkg.Tacotron2.from_blob(rig)
    frames, gates = m.decoder(mem, pm, n, pkg.default_opts(max_steps=90, dropout_masks=masks))
    assert frames.shape == rframes.shape and rms(frames, rframes) <= 1e-5 and np.abs(gates - rgates).max() <= 1e-5
    with pytest.raises(pkg.XdttsError):   # masks shorter than max_steps cannot cover a gate-driven decode
        m.decoder(mem, pm, n, pkg.default_opts(max_steps=100, dropout_masks=masks))
    m.close()


def test_explicit_dropout_masks_argument_errors(pkg, model):
    ids = [synth_ids(20, seed=1)]
    masks = np.ones((1, 10, 2, 256), dtype=np.uint8)
    with pytest.raises(pkg.XdttsError) as e:   # 12 steps asked, 10 covered
        model.infer_batch(ids, opts=pkg.default_opts(dropout_masks=masks), fixed_steps=np.array([12], dtype=np.int32))
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    with pytest.raises(pkg.XdttsError) as e:   # mode 2 without masks
        model.infer_batch(ids, opts=pkg.default_opts(dropout_mode=2), fixed_steps=np.array([4], dtype=np.int32))
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    with pytest.raises(pkg.XdttsError) as e:
        model.infer_batch(ids, opts=pkg.default_opts(dropout_mode=3), fixed_steps=np.array([4], dtype=np.int32))
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    out = model.infer_batch(ids, opts=pkg.default_opts(dropout_masks=masks), fixed_steps=np.array([10], dtype=np.int32))
    assert out[0].shape == (80, 10)


def test_batched_decode_after_an_encoder_timeout_uses_the_fresh_memory(pkg, orc, blob, capfd):
    """One workgroup of the cooperative encoder BiLSTM never publishes (test hook): the exchange times out, the handle
    switches to the single-workgroup recurrence, encodes again and decodes again -- in batched mode over the NEW
    processed_memory transpose (it used to attend over the timed-out encoder's)."""
    B = 6
    ids = [synth_ids(n, seed=71 + i) for i, n in enumerate([40, 77, 15, 100, 63, 28])]
    steps = np.array([12, 20, 8, 20, 16, 10], dtype=np.int32)
    o = pkg.default_opts(dropout_seed=5)
    with env(XDTTS_ENC_FAULT=2, XDTTS_ENC_SPINS=20000):
        m = pkg.Tacotron2.from_blob(blob)
        mels = m.infer_batch(ids, opts=o, fixed_steps=steps)
    assert "encoder BiLSTM exchange timed out" in capfd.readouterr().err
    assert m.engine_state()["encoder_cooperative"] == 0
    for b in range(B):
        ref = orc.infer_chunk(blob, ids[b], orc.default_opts(fixed_steps=int(steps[b]), dropout_seed=5, item=b))
        assert rms(mels[b], ref) <= 1e-5, (b, rms(mels[b], ref))
    m.close()
