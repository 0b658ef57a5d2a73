"""GPU parity, decoder loop engines: the persistent weight-stationary kernel (B <= 2, T <= 128;
csrc/decoder_persistent.hip) and the launch-per-stage path (csrc/decoder.hip) must both reproduce
the CPU oracle -- same frame counts, frames within 1e-5 RMS -- and agree with each other.
XDTTS_DECODER=launch is the library's developer switch that forces the second engine."""
import os

import numpy as np
import pytest

from conftest import rms, synth_ids
from test_gpu_tacotron2_more import LOGIT_06, rigged_gate_blob

pytestmark = pytest.mark.gpu


class engine:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.old = os.environ.get("XDTTS_DECODER")
        os.environ["XDTTS_DECODER"] = self.name

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("XDTTS_DECODER", None)
        else:
            os.environ["XDTTS_DECODER"] = self.old


def encode(orc, blob, n, seed=1):
    ids = np.zeros(100, dtype=np.int64)
    ids[:n] = synth_ids(n, seed=seed)
    return orc.encoder(blob, ids)


@pytest.mark.parametrize("mode", ["persistent", "launch"])
def test_both_engines_match_oracle_with_stop_rule(pkg, orc, blob, mode):
    mem, pm = encode(orc, blob, 33)
    rig = rigged_gate_blob(orc, blob, mem, pm, 33, 21, 30)
    rframes, rgates = orc.run_decoder(rig, mem, pm, 33, orc.default_opts(dropout_seed=21))
    assert 1 <= len(rframes) < 60
    with engine(mode):
        m = pkg.Tacotron2.from_blob(rig)
        for _ in range(2):  # second call: state and exchange buffers are re-initialised
            frames, gates = m.decoder(mem, pm, 33, pkg.default_opts(dropout_seed=21))
            assert frames.shape == rframes.shape
            assert rms(frames, rframes) <= 1e-5 and np.abs(gates - rgates).max() <= 1e-5
            assert gates[-1] > LOGIT_06 and np.all(gates[:-1] <= LOGIT_06)
        m.close()


def test_engines_agree_on_ragged_pair_with_gate(pkg, orc, blob):
    """Two chunks in lock-step, each stopped by its own gate at a different step."""
    lens = [41, 18]
    mems = [encode(orc, blob, n, seed=5 + i) for i, n in enumerate(lens)]
    rig = rigged_gate_blob(orc, blob, mems[0][0], mems[0][1], lens[0], 100, 25)
    ids = [synth_ids(n, seed=5 + i) for i, n in enumerate(lens)]
    o = pkg.default_opts(dropout_seed=100, max_steps=90)
    out = {}
    for mode in ("persistent", "launch"):
        with engine(mode):
            m = pkg.Tacotron2.from_blob(rig)
            out[mode] = m.infer_batch(ids, opts=o)
            m.close()
    for a, b in zip(out["persistent"], out["launch"]):
        assert a.shape == b.shape and rms(a, b) <= 1e-6
    assert out["persistent"][0].shape[1] != out["persistent"][1].shape[1] or out["persistent"][0].shape[1] < 90


@pytest.mark.parametrize("steps", [(50, 17), (9, 64), (30, 30), (1, 40)])
def test_pair_survivor_continues_on_the_one_chunk_kernel(pkg, orc, blob, steps):
    """A 2-chunk launch ends when its first chunk stops; the other is continued by the 1-chunk kernel
    from the written-back state.  Either chunk may be the survivor; equal lengths need no hand-off."""
    lens = [44, 29]
    ids = [synth_ids(n, seed=21 + i) for i, n in enumerate(lens)]
    fs = np.asarray(steps, dtype=np.int32)
    m = pkg.Tacotron2.from_blob(blob)
    o = pkg.default_opts(dropout_seed=13)
    out = m.infer_batch(ids, opts=o, fixed_steps=fs)
    m.close()
    for b in range(2):
        padded = np.zeros(100, dtype=np.int64)
        padded[: lens[b]] = ids[b]
        mem, pm = orc.encoder(blob, padded)
        rframes, _ = orc.run_decoder(blob, mem, pm, lens[b], orc.default_opts(fixed_steps=int(steps[b]), dropout_seed=13, item=b))
        assert out[b].shape == (80, steps[b])
        assert rms(out[b], orc.postnet(blob, rframes)) <= 1e-5


def test_context_fold_table_agrees_with_the_in_kernel_fold(pkg, orc, blob):
    """The persistent kernel needs every LSTM / projection row's context columns folded into the encoder memory
    (W[row][ctx cols] . memory[t]).  One GEMM per request writes them as a table (views of it serve the launches of
    chunks 2..3 and the 1-chunk launch that continues a pair's survivor); XDTTS_NO_CTXFOLD (read per handle) makes every
    launch fold for itself.  Same products, summed in a different order: four ragged chunks (two launches of two, each
    with a survivor hand-off) within 1e-5 of the oracle and of each other in both forms."""
    lens = [44, 29, 7, 61]
    steps = np.asarray([50, 33, 21, 64], dtype=np.int32)
    ids = [synth_ids(n, seed=31 + i) for i, n in enumerate(lens)]
    out = {}
    for table in (True, False):
        os.environ["XDTTS_P8"] = "0"  # (this test is about the pair launches of the persistent engine)
        if not table:
            os.environ["XDTTS_NO_CTXFOLD"] = "1"
        try:
            m = pkg.Tacotron2.from_blob(blob)
        finally:
            os.environ.pop("XDTTS_NO_CTXFOLD", None)
            os.environ.pop("XDTTS_P8", None)
        out[table] = m.infer_batch(ids, opts=pkg.default_opts(dropout_seed=19), fixed_steps=steps)
        assert m.engine_state()["decoder_persistent"] == 1
        m.close()
    for b in range(4):
        padded = np.zeros(100, dtype=np.int64)
        padded[: lens[b]] = ids[b]
        mem, pm = orc.encoder(blob, padded)
        rframes, _ = orc.run_decoder(blob, mem, pm, lens[b], orc.default_opts(fixed_steps=int(steps[b]), dropout_seed=19, item=b))
        ref = orc.postnet(blob, rframes)
        for table in out:
            assert out[table][b].shape == (80, steps[b]) and rms(out[table][b], ref) <= 1e-5, (table, b)
    assert all(rms(a, c) <= 1e-5 for a, c in zip(out[True], out[False]))
    assert any(not np.array_equal(a, c) for a, c in zip(out[True], out[False]))  # (the two forms really are different code)


def test_skewed_pair_loop_is_bit_identical_to_the_lock_step_loop(pkg, orc, blob):
    """A pair of chunks runs the persistent kernel's skewed loop (the chunks two phases apart, a workgroup alternating
    between them, the next phase's first poll issued ahead); XDTTS_NO_SKEW (read per handle) keeps the lock-step loop.
    The arithmetic of a chunk is the same code in the same order in both, so the mels must be bit-identical -- with the
    gate deciding (either chunk may stop first: the rigged gate trips chunk 0 at 21 frames; with the chunks swapped it
    trips chunk 1), with fixed ragged step counts, and for four chunks (two launches of two)."""
    mem, pm = encode(orc, blob, 33)
    rig = rigged_gate_blob(orc, blob, mem, pm, 33, 21, 30)
    a, b = synth_ids(33, seed=1), synth_ids(57, seed=2)
    cases = [(rig, [a, b], dict(opts=pkg.default_opts(dropout_seed=21, max_steps=48))),
             (rig, [b, a], dict(opts=pkg.default_opts(dropout_seed=21, max_steps=48))),
             (blob, [a, b], dict(opts=pkg.default_opts(dropout_seed=5), fixed_steps=np.asarray([37, 52], dtype=np.int32))),
             (blob, [b, a, synth_ids(12, seed=3), synth_ids(70, seed=4)], dict(opts=pkg.default_opts(dropout_seed=9), fixed_steps=np.asarray([41, 17, 30, 55], dtype=np.int32)))]
    for wblob, ids, kw in cases:
        out = {}
        for skew in (True, False):
            os.environ["XDTTS_P8"] = "0"  # (four chunks: two pair launches, not the 3..8-chunk engine)
            if not skew:
                os.environ["XDTTS_NO_SKEW"] = "1"
            try:
                m = pkg.Tacotron2.from_blob(wblob)
            finally:
                os.environ.pop("XDTTS_NO_SKEW", None)
                os.environ.pop("XDTTS_P8", None)
            out[skew] = m.infer_batch(ids, **kw)
            assert m.engine_state()["decoder_persistent"] == 1
            m.close()
        assert len(out[True]) == len(out[False]) == len(ids)
        for x, y in zip(out[True], out[False]):
            assert x.shape == y.shape and np.array_equal(x, y)


@pytest.mark.parametrize("p8", ["0", "1"])
def test_three_and_four_chunks_run_as_two_persistent_launches(pkg, orc, blob, p8):
    """B = 3..4 with XDTTS_P8=0: the persistent engine takes the chunks two at a time over views of the state arrays
    (by default they run on decoder_persistent8.hip);
    every chunk must still equal its own single-chunk oracle run (item index = dropout stream)."""
    lens = [37, 12, 58, 23]
    ids = [synth_ids(n, seed=11 + i) for i, n in enumerate(lens)]
    for B in (3, 4):
        o = pkg.default_opts(fixed_steps=40, dropout_seed=77, item_base=5)
        os.environ["XDTTS_P8"] = p8
        try:
            m = pkg.Tacotron2.from_blob(blob)
        finally:
            del os.environ["XDTTS_P8"]
        out = m.infer_batch(ids[:B], opts=o)
        assert m.engine_state()["decoder_persistent8"] == int(p8)
        with engine("launch"):
            ref = m.infer_batch(ids[:B], opts=o)
        m.close()
        for a, b in zip(out, ref):
            assert a.shape == b.shape == (80, 40) and rms(a, b) <= 1e-6
        for b in range(B):
            padded = np.zeros(100, dtype=np.int64)
            padded[: lens[b]] = ids[b]
            mem, pm = orc.encoder(blob, padded)
            ro = orc.default_opts(fixed_steps=40, dropout_seed=77, item=5 + b)
            rframes, _ = orc.run_decoder(blob, mem, pm, lens[b], ro)
            post = orc.postnet(blob, rframes)
            assert rms(out[b], post) <= 1e-5


@pytest.mark.parametrize("T,n_valid", [(7, 5), (64, 64), (100, 1), (128, 120), (130, 97)])
def test_encoder_lengths_across_the_engine_boundary(pkg, model, orc, blob, T, n_valid):
    """T <= 128 runs the persistent kernel, longer memories the launch path; both mask t >= n_valid."""
    rng = np.random.Generator(np.random.PCG64(T))
    mem = rng.standard_normal((T, 512)).astype(np.float32) * 0.5
    pm = rng.standard_normal((T, 128)).astype(np.float32) * 0.5
    ro = orc.default_opts(fixed_steps=24, dropout_seed=9)
    rframes, rgates = orc.run_decoder(blob, mem, pm, n_valid, ro)
    frames, gates = model.decoder(mem, pm, n_valid, pkg.default_opts(fixed_steps=24, dropout_seed=9))
    assert frames.shape == rframes.shape == (24, 80)
    assert rms(frames, rframes) <= 1e-5 and np.abs(gates - rgates).max() <= 1e-5


def test_persistent_engine_long_sequence_and_dropout_off(pkg, model, orc, blob):
    mem, pm = encode(orc, blob, 95)
    for dm in (0, 1):
        ro = orc.default_opts(fixed_steps=300, dropout_seed=4, dropout_mode=dm)
        rframes, _ = orc.run_decoder(blob, mem, pm, 95, ro)
        frames, _ = model.decoder(mem, pm, 95, pkg.default_opts(fixed_steps=300, dropout_seed=4, dropout_mode=dm))
        assert frames.shape == rframes.shape
        assert rms(frames, rframes) <= 1e-4  # north_star tolerance at full length


def test_max_decoder_steps_1000_is_reached_without_a_stop(pkg, model, orc, blob):
    """max_decoder_steps = 1000 (mod.rs:280): the synthetic gate never fires, so the loop runs the
    full 1000 frames in one persistent launch; parity with the oracle at full length."""
    mem, pm = encode(orc, blob, 77)
    rframes, rgates = orc.run_decoder(blob, mem, pm, 77, orc.default_opts(dropout_seed=2))
    frames, gates = model.decoder(mem, pm, 77, pkg.default_opts(dropout_seed=2))
    assert frames.shape == rframes.shape == (1000, 80)
    assert rms(frames, rframes) <= 1e-4 and np.abs(gates - rgates).max() <= 1e-3
    assert model.last_timings()["steps"] == 1000


def test_lost_workgroup_drains_and_falls_back(pkg, orc, blob, capfd):
    """The persistent grid must be co-resident.  With one workgroup missing (test hook) every bounded
    spin runs out or sees the error word, the launch drains, and the handle re-decodes the request on
    the launch-per-stage engine -- correct frames, a message on stderr, no hang."""
    mem, pm = encode(orc, blob, 21)
    rframes, _ = orc.run_decoder(blob, mem, pm, 21, orc.default_opts(fixed_steps=30, dropout_seed=3))
    os.environ["XDTTS_PERSIST_FAULT"] = "201"   # workgroup 200 returns at once
    os.environ["XDTTS_PERSIST_SPINS"] = "20000"
    try:
        m = pkg.Tacotron2.from_blob(blob)
        frames, _ = m.decoder(mem, pm, 21, pkg.default_opts(fixed_steps=30, dropout_seed=3))
        assert "persistent decoder exchange timed out" in capfd.readouterr().err
        assert frames.shape == rframes.shape and rms(frames, rframes) <= 1e-5
    finally:
        del os.environ["XDTTS_PERSIST_FAULT"], os.environ["XDTTS_PERSIST_SPINS"]
    # the handle stays on the second engine; a fresh handle is persistent again
    frames2, _ = m.decoder(mem, pm, 21, pkg.default_opts(fixed_steps=30, dropout_seed=3))
    assert np.array_equal(frames2, frames)
    m.close()


def test_lost_workgroup_in_a_pair_drains_and_falls_back(pkg, orc, blob, capfd):
    """The same for a PAIR of chunks (the skewed loop: every phase's gather is a bounded spin that watches the error word,
    and a phase whose vector never comes ends the launch for both chunks): with a workgroup missing the request is decoded
    again on the launch-per-stage engine, both chunks within 1e-5 of the oracle, one message on stderr, no hang."""
    lens, steps = [31, 18], np.asarray([34, 22], dtype=np.int32)
    ids = [synth_ids(n, seed=61 + i) for i, n in enumerate(lens)]
    os.environ["XDTTS_PERSIST_FAULT"] = "77"   # workgroup 76 returns at once
    os.environ["XDTTS_PERSIST_SPINS"] = "20000"
    try:
        m = pkg.Tacotron2.from_blob(blob)
        out = m.infer_batch(ids, opts=pkg.default_opts(dropout_seed=23), fixed_steps=steps)
        assert "persistent decoder exchange timed out" in capfd.readouterr().err
    finally:
        del os.environ["XDTTS_PERSIST_FAULT"], os.environ["XDTTS_PERSIST_SPINS"]
    assert m.engine_state()["decoder_persistent"] == 0
    m.close()
    for b in range(2):
        padded = np.zeros(100, dtype=np.int64)
        padded[: lens[b]] = ids[b]
        mem, pm = orc.encoder(blob, padded)
        rframes, _ = orc.run_decoder(blob, mem, pm, lens[b], orc.default_opts(fixed_steps=int(steps[b]), dropout_seed=23, item=b))
        assert out[b].shape == (80, steps[b]) and rms(out[b], orc.postnet(blob, rframes)) <= 1e-5


def test_teacher_forced_single_step_all_nine_outputs(pkg, model, orc, blob):
    """ONE decoder_iter call (mod.rs:304) from an identical state -- the oracle's own, taken
    at several points of a free-running sequence -- must reproduce all nine outputs (mod.rs:306-307,
    332-339): decoder_output, gate_prediction and the seven out_* state tensors, to 1e-5."""
    nv = 37
    mem, pm = encode(orc, blob, nv)     # the encoder window: 100 rows, 37 of them un-padded (mask, mod.rs:219-220)
    T = mem.shape[0]
    o = orc.default_opts(dropout_seed=11, item=3)
    go = pkg.default_opts(dropout_seed=11, item_base=3)
    st = orc.new_state()
    names = {"attention_hidden": "att_h", "attention_cell": "att_c", "decoder_hidden": "dec_h", "decoder_cell": "dec_c",
             "attention_weights": "aw", "attention_weights_cum": "awc", "attention_context": "ctx"}
    worst = 0.0
    for step in range(12):
        snap = {k: np.array(getattr(st, v), dtype=np.float32)[: (T if v in ("aw", "awc") else None)] for k, v in names.items()}
        dec_in = np.array(st.dec_in, dtype=np.float32)
        mel, gate = orc.decoder_step(blob, mem, pm, nv, st, o, step)       # advances the oracle's state
        if step in (0, 1, 5, 11):
            gmel, ggate, gst = model.decoder_step(mem, pm, nv, snap, dec_in, step, opts=go)
            errs = [float(np.abs(gmel - mel).max()), abs(ggate - gate)]
            for k, v in names.items():
                ref = np.array(getattr(st, v), dtype=np.float32)[: (T if v in ("aw", "awc") else None)]
                errs.append(float(np.abs(gst[k] - ref).max()))
            worst = max(worst, max(errs))
            assert max(errs) <= 1e-5, (step, errs)
    assert worst > 0  # (the comparison really ran on different implementations)


def test_engine_state_and_reset(pkg, orc, blob):
    mem, pm = encode(orc, blob, 21)
    m = pkg.Tacotron2.from_blob(blob)
    assert m.engine_state()["decoder_persistent"] == -1
    m.decoder(mem, pm, 21, pkg.default_opts(fixed_steps=6, dropout_seed=3))
    assert m.engine_state() == {"decoder_persistent": 1, "encoder_cooperative": 1, "batched_attention": 2, "decoder_persistent8": -1}
    os.environ["XDTTS_PERSIST_FAULT"] = "201"
    os.environ["XDTTS_PERSIST_SPINS"] = "20000"
    try:
        f0, _ = m.decoder(mem, pm, 21, pkg.default_opts(fixed_steps=6, dropout_seed=3))
    finally:
        del os.environ["XDTTS_PERSIST_FAULT"], os.environ["XDTTS_PERSIST_SPINS"]
    assert m.engine_state()["decoder_persistent"] == 0            # demoted after the timed-out exchange
    m.engine_reset()
    f1, _ = m.decoder(mem, pm, 21, pkg.default_opts(fixed_steps=6, dropout_seed=3))
    assert m.engine_state()["decoder_persistent"] == 1 and rms(f0, f1) <= 1e-6
    m.close()
