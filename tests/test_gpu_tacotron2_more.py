"""GPU parity, second slice: stop rule, batches/chunks, error behaviour, weight container, golden
fixtures and the full-size BASELINE config -- all through the C ABI against the CPU oracle."""
import os

import numpy as np
import pytest

from conftest import rms, synth_ids

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOGIT_06 = 0.4054651  # ln(0.6 / 0.4): the gate threshold of src/tacotron2/mod.rs:279 as a logit


def rigged_gate_blob(orc, blob, mem, pm, n_valid, seed, kth, gain=-40.0):
    """Make the synthetic gate informative: scale (and flip) gate_layer.weight so the logit drifts
    upward from step to step, then set the bias so that the k-th largest of the first 60 logits
    crosses sigmoid(gate) > 0.6 by a 2e-3 margin -- the loop then ends well inside the 60 steps."""
    rig = blob.copy()
    tab = {t[0]: t for t in orc.tensor_table()}
    _n, _s, woff, wn = tab["gate_layer.weight"]
    rig[woff : woff + wn] *= np.float32(gain)
    boff = tab["gate_layer.bias"][2]
    rig[boff] = 0.0
    _f, gates = orc.run_decoder(rig, mem, pm, n_valid, orc.default_opts(fixed_steps=60, dropout_seed=seed))
    rig[boff] = np.float32(LOGIT_06 - float(np.sort(gates)[-kth]) + 2e-3)
    return rig


def test_gate_stop_rule_matches_oracle(pkg, orc, blob):
    """mod.rs:302-342: the frame whose sigmoid(gate) > 0.6 is kept and ends the loop, on the device."""
    ids = np.zeros(100, dtype=np.int64)
    ids[:33] = synth_ids(33)
    mem, pm = orc.encoder(blob, ids)
    for kth in (20, 45):
        rig = rigged_gate_blob(orc, blob, mem, pm, 33, 21, kth)
        rframes, rgates = orc.run_decoder(rig, mem, pm, 33, orc.default_opts(dropout_seed=21))
        assert 1 <= len(rframes) < 60
        m = pkg.Tacotron2.from_blob(rig)
        frames, gates = m.decoder(mem, pm, 33, pkg.default_opts(dropout_seed=21))
        assert frames.shape == rframes.shape  # identical frame count
        assert rms(frames, rframes) <= 1e-5 and np.abs(gates - rgates).max() <= 1e-5
        assert gates[-1] > LOGIT_06 and np.all(gates[:-1] <= LOGIT_06)
        m.close()


def test_stop_rule_next_to_the_threshold(pkg, model, orc, blob):
    """The kernels decide sigmoid(gate) > threshold without the sigmoid when the logit is clear of logit(threshold) (a band of
    1e-3 (1 + |logit|), device_utils.h: gate_fires) and with the reference's two-branch sigmoid inside it.  Thresholds placed
    2e-4 below / above the largest logit of a 30-step sequence put that step INSIDE the band on either side of the verdict;
    a threshold well away takes the short cut.  Frame counts must equal the oracle's every time."""
    ids = np.zeros(100, dtype=np.int64)
    ids[:27] = synth_ids(27, seed=6)
    mem, pm = orc.encoder(blob, ids)
    _, gates = orc.run_decoder(blob, mem, pm, 27, orc.default_opts(fixed_steps=30, dropout_seed=4))
    k = int(np.argmax(gates))
    top, second = float(gates[k]), float(np.sort(gates)[-2])
    assert top - second > 5e-4  # (no other step comes nearer to the two thresholds around `top` than the step itself)
    sig = lambda x: float(1.0 / (1.0 + np.exp(-np.float64(x))))
    first_above = lambda x: next((i + 1 for i, g in enumerate(gates) if g > x), 30)
    for logit in (top - 2e-4, top + 2e-4, top - 0.05):
        frames_expected = first_above(logit)
        thr = np.float32(sig(logit))
        rf, _ = orc.run_decoder(blob, mem, pm, 27, orc.default_opts(gate_threshold=float(thr), max_steps=30, dropout_seed=4))
        gf, gg = model.decoder(mem, pm, 27, pkg.default_opts(gate_threshold=float(thr), max_steps=30, dropout_seed=4))
        assert len(rf) == frames_expected and gf.shape == rf.shape, (logit, len(rf), gf.shape)
        assert rms(gf, rf) <= 1e-5
    # degenerate thresholds (no band: always the sigmoid itself): > 1 never fires, < 0 fires at once
    assert len(model.decoder(mem, pm, 27, pkg.default_opts(gate_threshold=1.5, max_steps=7, dropout_seed=4))[0]) == 7
    assert len(model.decoder(mem, pm, 27, pkg.default_opts(gate_threshold=-0.5, max_steps=7, dropout_seed=4))[0]) == 1
    assert len(model.decoder(mem, pm, 27, pkg.default_opts(gate_threshold=0.9999, max_steps=7, dropout_seed=4))[0]) == 7


def test_max_steps_cap_and_threshold_option(pkg, model, orc, blob):
    ids = np.zeros(100, dtype=np.int64)
    ids[:20] = synth_ids(20)
    mem, pm = orc.encoder(blob, ids)
    frames, gates = model.decoder(mem, pm, 20, pkg.default_opts(max_steps=37, dropout_seed=1))
    assert len(frames) == 37  # synthetic gate never fires: i + 1 == max_decoder_steps (mod.rs:320)
    # a low threshold stops at the first step (every logit's sigmoid is ~0.5 > 0.3)
    f1, _ = model.decoder(mem, pm, 20, pkg.default_opts(gate_threshold=0.3, dropout_seed=1))
    assert len(f1) == 1 and np.allclose(f1[0], frames[0], atol=1e-6)


def test_infer_ids_with_splits_equals_chunkwise_oracle(pkg, model, orc, blob):
    """Tacotron2::infer (mod.rs:398-437): chunks run independently and are concatenated in time."""
    ids = synth_ids(120)
    splits = pkg.find_splits(ids, 100)
    assert list(splits) == [95]
    o = pkg.default_opts(fixed_frames_per_id=0.5, dropout_seed=13)
    mel = model.infer(ids, splits=splits, opts=o)
    parts = []
    for item, (a, b) in enumerate(((0, 95), (95, 120))):
        steps = int(np.floor(0.5 * (b - a) + 0.5))  # lround, as the library rounds fixed_frames_per_id * n
        parts.append(orc.infer_chunk(blob, ids[a:b], orc.default_opts(fixed_steps=steps, dropout_seed=13, item=item)))
    ref = np.concatenate(parts, axis=1)
    assert mel.shape == ref.shape == (80, 48 + 13)
    assert rms(mel, ref) <= 1e-5
    # the Unit-level front door gives the same result
    toks = [pkg.generate_id_list()[i] for i in ids]
    mel2 = model.infer_units(toks, opts=o)
    assert np.array_equal(mel2, mel)


def test_batch_of_variable_length_chunks(pkg, model, orc, blob):
    """BASELINE.json configs[2] in small: variable-length chunks in one lock-step batch, per-chunk
    mask (mod.rs:219-220), per-chunk stop; each must equal the oracle run on its own."""
    rng = np.random.Generator(np.random.PCG64(2))
    lens = [int(x) for x in rng.integers(10, 101, size=5)]
    ids_list = [synth_ids(n, seed=10 + i) for i, n in enumerate(lens)]
    steps = [12, 30, 7, 22, 16]
    mels = model.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=3, item_base=40), fixed_steps=steps)
    for b, (ids, st) in enumerate(zip(ids_list, steps)):
        ref = orc.infer_chunk(blob, ids, orc.default_opts(fixed_steps=st, dropout_seed=3, item=40 + b))
        assert mels[b].shape == (80, st)
        assert rms(mels[b], ref) <= 1e-5, b


def test_batch_mels_are_pieces_of_one_slab_released_one_by_one(pkg, model):
    """xdtts_tacotron2_infer_batch hands out pieces of ONE pinned slab (xd-tts_amd/csrc/api.cpp: PinnedSlab): the slab
    returns to the pool only with the last piece, so a later call never writes into memory a live mel still refers to; a
    piece released twice is a no-op."""
    import ctypes as C
    ids_list = [synth_ids(n, seed=20 + i) for i, n in enumerate((40, 17, 63, 29, 88, 12))]
    steps = [9, 14, 5, 11, 8, 16]
    o = pkg.default_opts(dropout_seed=5)
    first = model.infer_batch(ids_list, opts=o, fixed_steps=steps)
    keep = [m.copy() for m in first]
    addr = [m.ctypes.data for m in first]
    assert all(addr[b + 1] - addr[b] == 4 * 80 * steps[b] for b in range(5))  # back to back in one slab
    del first[1], first[3]  # release two pieces (the views are the only owners)
    again = model.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=6), fixed_steps=steps)  # different masks
    assert not np.array_equal(again[0], keep[0])
    for m, k in zip(first, [keep[0], keep[2], keep[3], keep[5]]):
        assert np.array_equal(m, k)  # untouched by the second call
    # an explicit double release of a piece must not free anything twice
    B = len(ids_list)  # (the same engine as above: results bit-identical)
    lens = np.array([len(x) for x in ids_list[:B]], dtype=np.int32)
    ids = np.zeros((B, 100), dtype=np.int64)
    for b in range(B):
        ids[b, : lens[b]] = ids_list[b]
    fs = np.array(steps[:B], dtype=np.int32)
    mels, nf = (pkg._PF * B)(), (C.c_size_t * B)()
    pkg._check(pkg.lib.xdtts_tacotron2_infer_batch(model._h, pkg._ptr(ids), pkg._ptr(lens), B, 100, C.byref(o), pkg._ptr(fs), mels, nf))
    got = np.ctypeslib.as_array(mels[1], shape=(80, steps[1])).copy()
    assert np.array_equal(got, keep[1])
    pkg.lib.xdtts_free(mels[0])
    pkg.lib.xdtts_free(mels[0])
    assert np.array_equal(np.ctypeslib.as_array(mels[1], shape=(80, steps[1])), keep[1])  # the slab is still live
    for b in range(1, B):
        pkg.lib.xdtts_free(mels[b])
    third = model.infer_batch(ids_list, opts=o, fixed_steps=steps)
    assert all(np.array_equal(a, b) for a, b in zip(third, keep))


def test_batch_with_gate_stops_each_chunk_on_its_own(pkg, orc, blob):
    ids_list = [synth_ids(30, seed=5), synth_ids(55, seed=6), synth_ids(18, seed=7)]
    padded = np.zeros(100, dtype=np.int64)
    padded[:30] = ids_list[0]
    mem, pm = orc.encoder(blob, padded)
    rig = rigged_gate_blob(orc, blob, mem, pm, 30, 8, 40)
    m = pkg.Tacotron2.from_blob(rig)
    mels = m.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=8, max_steps=80))
    counts = []
    for b, ids in enumerate(ids_list):
        ref = orc.infer_chunk(rig, ids, orc.default_opts(dropout_seed=8, max_steps=80, item=b))
        counts.append(ref.shape[1])
        assert mels[b].shape == ref.shape, (b, mels[b].shape, ref.shape)
        assert rms(mels[b], ref) <= 1e-5
    assert len(set(counts)) > 1  # the chunks really stop at different steps
    m.close()


def test_error_behaviour(pkg, model):
    with pytest.raises(pkg.XdttsError) as e:  # the reference's assert!(units_len <= 100), mod.rs:363
        model.infer(synth_ids(101))
    assert e.value.status == pkg.XDTTS_ERR_TOO_LONG
    with pytest.raises(pkg.XdttsError) as e:
        model.infer(np.array([5, 148], dtype=np.int64))
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    with pytest.raises(pkg.XdttsError) as e:
        model.infer(np.array([], dtype=np.int64))
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    with pytest.raises(pkg.XdttsError) as e:
        pkg.Tacotron2.synthetic(device_id=99)
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    # a single 'a' works, like the reference's tacotron_sanity_test (mod.rs:511-522): 80 rows, >0 cols
    spec = model.infer_units(["a"], opts=pkg.default_opts(max_steps=6), as_character=True)
    assert spec.shape[0] == 80 and spec.shape[1] > 0


def test_weight_container_round_trip(pkg, model, orc, blob, tmp_path):
    model.save(str(tmp_path))
    assert os.path.getsize(tmp_path / "tacotron2.xdtw") > blob.nbytes
    m2 = pkg.Tacotron2.load(str(tmp_path))  # Tacotron2::load(path), mod.rs:242
    ids = synth_ids(17)
    o = pkg.default_opts(fixed_steps=9, dropout_seed=2)
    assert np.array_equal(m2.infer(ids, opts=o), model.infer(ids, opts=o))
    assert np.array_equal(m2.get_tensor("decoder_rnn.weight_hh"), orc.tensor(blob, "decoder_rnn.weight_hh"))
    m2.close()
    with pytest.raises(pkg.XdttsError) as e:  # the anyhow context of mod.rs:249
        pkg.Tacotron2.load(str(tmp_path / "missing"))
    assert e.value.status == pkg.XDTTS_ERR_IO
    raw = open(tmp_path / "tacotron2.xdtw", "rb").read()
    bad = tmp_path / "bad"
    bad.mkdir()
    open(bad / "tacotron2.xdtw", "wb").write(raw[: len(raw) // 2])
    with pytest.raises(pkg.XdttsError) as e:
        pkg.Tacotron2.load(str(bad))
    assert e.value.status == pkg.XDTTS_ERR_IO


def test_golden_fixture_on_gpu(pkg, model):
    g = np.load(os.path.join(G, "tacotron2_small.npz"))
    ids, steps, dseed = g["ids"], int(g["steps"]), int(g["dropout_seed"])
    padded = np.zeros(100, dtype=np.int64)
    padded[: len(ids)] = ids
    mem, pm = model.encoder(padded)
    assert np.abs(mem[:, ::32] - g["memory_cols"]).max() <= 1e-5
    assert np.abs(pm[:, ::8] - g["pmem_cols"]).max() <= 1e-5
    mel = model.infer(ids, opts=pkg.default_opts(fixed_steps=steps, dropout_seed=dseed))
    assert rms(mel, g["mel"]) <= 1e-5 and rms(mel, g["mel_f64"]) <= 1e-5


def test_full_size_config2_mel_parity(pkg, model, orc, blob):
    """BASELINE.json configs[1] at full size: 120 ids -> chunks 95 + 25 -> 800 frames, vs the oracle."""
    ids = synth_ids(120)
    splits = pkg.find_splits(ids, 100)
    o = pkg.default_opts(fixed_frames_per_id=800 / 120.0, dropout_seed=0)
    mel = model.infer(ids, splits=splits, opts=o)
    assert mel.shape == (80, 800)
    parts = [orc.infer_chunk(blob, ids[a:b], orc.default_opts(fixed_steps=int(round(800 / 120.0 * (b - a))), dropout_seed=0, item=i)) for i, (a, b) in enumerate(((0, 95), (95, 120)))]
    ref = np.concatenate(parts, axis=1)
    err = rms(mel, ref)
    assert err <= 1e-4, err  # the north-star tolerance
    assert err <= 1e-3 * float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))  # and relative to the signal
    t = model.last_timings()
    assert t["steps"] == 633  # lock-step: max(633, 167) iterations for 800 frames


def test_large_batch_runs_lstms_on_mfma(pkg, model, orc, blob):
    """B >= 8 chunks in lock-step switches the two LSTM kernels to the f32-MFMA GEMM form
    (k_lstm_mfma); 19 chunks also exercises a partial second 16-chunk tile.  Each chunk must still
    equal the oracle run on its own, including chunks that finish early."""
    rng = np.random.Generator(np.random.PCG64(9))
    lens = [int(x) for x in rng.integers(5, 101, size=19)]
    ids_list = [synth_ids(n, seed=50 + i) for i, n in enumerate(lens)]
    steps = [int(x) for x in rng.integers(3, 25, size=19)]
    mels = model.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=17, item_base=7), fixed_steps=steps)
    worst = 0.0
    for b, (ids, st) in enumerate(zip(ids_list, steps)):
        ref = orc.infer_chunk(blob, ids, orc.default_opts(fixed_steps=st, dropout_seed=17, item=7 + b))
        assert mels[b].shape == (80, st)
        worst = max(worst, rms(mels[b], ref))
    assert worst <= 1e-5, worst


def test_large_batch_with_gate(pkg, orc, blob):
    ids_list = [synth_ids(20 + 3 * i, seed=70 + i) for i in range(17)]
    padded = np.zeros(100, dtype=np.int64)
    padded[: len(ids_list[0])] = ids_list[0]
    mem, pm = orc.encoder(blob, padded)
    rig = rigged_gate_blob(orc, blob, mem, pm, len(ids_list[0]), 4, 40)
    m = pkg.Tacotron2.from_blob(rig)
    mels = m.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=4, max_steps=70))
    counts = set()
    for b, ids in enumerate(ids_list):
        ref = orc.infer_chunk(rig, ids, orc.default_opts(dropout_seed=4, max_steps=70, item=b))
        assert mels[b].shape == ref.shape, (b, mels[b].shape, ref.shape)
        assert rms(mels[b], ref) <= 1e-5
        counts.add(ref.shape[1])
    assert len(counts) > 1
    m.close()


def test_batch_beyond_one_mfma_pass(pkg, model, orc, blob):
    """70 chunks: more than the 64 chunks one pass of k_lstm_mfma covers, so the kernels loop over two
    chunk super-tiles with the weights held in registers; spot-check a handful against the oracle."""
    rng = np.random.Generator(np.random.PCG64(11))
    lens = [int(x) for x in rng.integers(4, 60, size=70)]
    ids_list = [synth_ids(n, seed=200 + i) for i, n in enumerate(lens)]
    steps = [int(x) for x in rng.integers(2, 9, size=70)]
    mels = model.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=23), fixed_steps=steps)
    assert [m.shape for m in mels] == [(80, s) for s in steps]
    for b in (0, 15, 16, 47, 63, 64, 69):
        ref = orc.infer_chunk(blob, ids_list[b], orc.default_opts(fixed_steps=steps[b], dropout_seed=23, item=b))
        assert rms(mels[b], ref) <= 1e-5, b


def _handle(pkg, blob, p8):
    """a handle with the 3..8-chunk persistent engine on ("1") or off ("0": such batches then take the engines either side)"""
    os.environ["XDTTS_P8"] = p8  # read when the handle is created
    try:
        return pkg.Tacotron2.from_blob(blob)
    finally:
        del os.environ["XDTTS_P8"]


@pytest.mark.parametrize("n", [5, 17, 64])
def test_one_launch_attention_at_the_batch_size_limits(pkg, orc, blob, n):
    """5 chunks is the smallest lock-step batch of the MFMA path (20 of the 256 blocks of k_att_lstm_attention turn
    into attention blocks; with the 3..16-chunk engines off -- by default 17 is the smallest), 64 the largest that keeps the
    attention LSTM and the attention in one launch (all 256 do)."""
    rng = np.random.Generator(np.random.PCG64(40 + n))
    ids_list = [synth_ids(int(x), seed=700 + i) for i, x in enumerate(rng.integers(4, 90, size=n))]
    steps = [int(x) for x in rng.integers(3, 10, size=n)]
    model = _handle(pkg, blob, "0" if n == 5 else "1")
    mels = model.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=37), fixed_steps=steps)
    st = model.engine_state()
    model.close()
    assert st["batched_attention"] == 2 and st["decoder_persistent8"] == (0 if n == 5 else -1)
    assert [m.shape for m in mels] == [(80, s) for s in steps]
    for b in sorted({0, 1, n // 2, n - 2, n - 1}):
        ref = orc.infer_chunk(blob, ids_list[b], orc.default_opts(fixed_steps=steps[b], dropout_seed=37, item=b))
        assert rms(mels[b], ref) <= 1e-5, b


def test_onnx_export_converts_and_loads(pkg, model, blob, tmp_path):
    """ONNX graphs (synthetic, written the way torch.onnx.export names/packs the
    parameters) -> tools/onnx_to_xdtw.py -> Tacotron2::load(dir) gives the same mel as the blob."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(G), "..", "tools"))
    import onnx_to_xdtw as conv
    import onnx_writer

    T = {name: blob[off : off + int(np.prod(shape))].reshape(shape) for name, shape, off in pkg.tensor_table()}
    assert [n for n, _ in conv.tensor_table()] == [t[0] for t in pkg.tensor_table()]
    onnx_writer.write_models(str(tmp_path), T, "named")
    ids = synth_ids(40)
    o = pkg.default_opts(fixed_steps=30, dropout_seed=8)
    want = model.infer(ids, opts=o)
    m = pkg.Tacotron2.load(str(tmp_path))                      # straight from the three .onnx files (csrc/onnx_load.cpp)
    got = m.infer(ids, opts=o)
    m.close()
    assert got.shape == want.shape == (80, 30) and np.array_equal(got, want)
    (tmp_path / "c").mkdir()
    conv.write_container(str(tmp_path / "c"), conv.collect(str(tmp_path)))   # and through the offline converter
    m = pkg.Tacotron2.load(str(tmp_path / "c"))
    got = m.infer(ids, opts=o)
    m.close()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("p8", ["0", "1"])
def test_batched_path_with_a_smaller_window(pkg, orc, blob, p8):
    """Seven chunks with max_chunk = 50 on the batched MFMA path (XDTTS_P8=0) and on the 3..8-chunk persistent engine: encoder
    window, mask, location tiles and the operand-order activation copies all follow T; every chunk still equals its own oracle run."""
    lens = [50, 7, 33, 48, 21, 50, 12]
    ids_list = [synth_ids(n, seed=90 + i) for i, n in enumerate(lens)]
    steps = [18, 5, 11, 20, 9, 14, 16]
    o = pkg.default_opts(dropout_seed=23, item_base=2, max_chunk=50)
    model = _handle(pkg, blob, p8)
    mels = model.infer_batch(ids_list, opts=o, fixed_steps=steps)
    assert model.engine_state()["decoder_persistent8"] == int(p8)
    model.close()
    for b, (ids, st) in enumerate(zip(ids_list, steps)):
        ref = orc.infer_chunk(blob, ids, orc.default_opts(fixed_steps=st, dropout_seed=23, item=2 + b), window=50)
        assert mels[b].shape == (80, st) and rms(mels[b], ref) <= 1e-5, b


def test_batched_path_with_a_larger_window(pkg, model, orc, blob):
    """max_chunk = 140 (> 128 encoder steps): the batched path computes the location features with the FMA
    blocks instead of the MFMA ones, and the persistent single-chunk engine gives way to the launch path;
    every chunk still equals its own oracle run."""
    lens = [140, 9, 77, 131, 64, 140]
    ids_list = [synth_ids(n, seed=120 + i) for i, n in enumerate(lens)]
    steps = [12, 5, 9, 14, 7, 10]
    o = pkg.default_opts(dropout_seed=29, item_base=1, max_chunk=140)
    mels = model.infer_batch(ids_list, opts=o, fixed_steps=steps)
    for b, (ids, st) in enumerate(zip(ids_list, steps)):
        ref = orc.infer_chunk(blob, ids, orc.default_opts(fixed_steps=st, dropout_seed=29, item=1 + b), window=140)
        assert mels[b].shape == (80, st) and rms(mels[b], ref) <= 1e-5, b


def _batch_case(n=17):  # (from 17 chunks on a lock-step batch runs on the two-launch batched engine)
    ids_list = [synth_ids(18 + 5 * i, seed=500 + i) for i in range(n)]
    steps = [5 + (3 * i) % 7 for i in range(n)]
    return ids_list, steps


def test_batched_attention_forms_agree(pkg, orc, blob):
    """Lock-step batches run the attention LSTM, the energies, the softmax and the context as ONE launch whose blocks
    exchange tagged granules (k_att_lstm_attention), as the LSTM plus a one-launch attention (k_attention_b; also what
    batches beyond 64 chunks use), or as three kernels with grid boundaries in between.  The three forms differ only
    in how the 32 partial energies of a time step are grouped before they are summed: each within 1e-5 of the oracle,
    and of each other."""
    ids_list, steps = _batch_case()
    out = {}
    for form in ("2", "1", "0"):
        os.environ["XDTTS_ATT_FUSED"] = form
        try:
            m = pkg.Tacotron2.from_blob(blob)
        finally:
            del os.environ["XDTTS_ATT_FUSED"]
        out[form] = m.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=31), fixed_steps=steps)
        assert m.engine_state()["batched_attention"] == int(form)
        m.close()
    for b, (ids, st) in enumerate(zip(ids_list, steps)):
        ref = orc.infer_chunk(blob, ids, orc.default_opts(fixed_steps=st, dropout_seed=31, item=b))
        for form in out:
            assert out[form][b].shape == (80, st) and rms(out[form][b], ref) <= 1e-5, (form, b)
        assert rms(out["2"][b], out["0"][b]) <= 1e-5 and rms(out["1"][b], out["0"][b]) <= 1e-5


def test_early_attention_lstm_partial_agrees_with_the_whole_pass(pkg, orc, blob):
    """In the one-launch attention form the attention LSTM's 1536 columns over [ctx ; h_att] are multiplied one launch
    early (extra blocks of the previous step's decoder-LSTM launch) and the attention launch adds that partial to its
    256 prenet columns; XDTTS_NO_EARLY (read per handle) makes the attention launch multiply the whole K.  Same
    products, grouped differently: each within 1e-5 of the oracle over a 60-step decode of 21 chunks (two MFMA tiles, a
    ragged second one, chunks stopping at different steps), and of each other."""
    ids_list = [synth_ids(20 + 3 * i, seed=700 + i) for i in range(21)]
    steps = [60 - 2 * i for i in range(21)]
    out = {}
    for early in (True, False):
        if not early:
            os.environ["XDTTS_NO_EARLY"] = "1"
        try:
            m = pkg.Tacotron2.from_blob(blob)
        finally:
            os.environ.pop("XDTTS_NO_EARLY", None)
        out[early] = m.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=37), fixed_steps=steps)
        assert m.engine_state()["batched_attention"] == 2
        m.close()
    for b in (0, 7, 15, 16, 20):
        ref = orc.infer_chunk(blob, ids_list[b], orc.default_opts(fixed_steps=steps[b], dropout_seed=37, item=b))
        for early in out:
            assert out[early][b].shape == (80, steps[b]) and rms(out[early][b], ref) <= 1e-5, (early, b)
    assert all(rms(a, c) <= 1e-5 for a, c in zip(out[True], out[False]))
    assert any(not np.array_equal(a, c) for a, c in zip(out[True], out[False]))  # (the two forms really are different code)


def test_two_launch_form_agrees_with_the_prenet_launch(pkg, orc, blob):
    """Batches of 5..64 chunks run TWO launches per lock-step iteration: the decoder-LSTM blocks publish h_dec as granules and
    blocks 4b..4b+3 run chunk b's projection, stop rule and prenet as their tail, the location features ride on other blocks
    of that launch; XDTTS_NO_TAIL (read per handle) keeps the prenet launch that sums partial-mel rows.  Same arithmetic,
    sums grouped differently: with fixed step counts (21 chunks, ragged second tile, chunks stopping at different steps, and 56
    chunks: no decoder-LSTM block is free for a location unit) and with the gate deciding (frame counts equal), every chunk
    within 1e-5 of the oracle and of the other form."""
    def both(blob_, ids_list, **kw):
        out = {}
        for tail in (True, False):
            if not tail:
                os.environ["XDTTS_NO_TAIL"] = "1"
            try:
                m = pkg.Tacotron2.from_blob(blob_)
            finally:
                os.environ.pop("XDTTS_NO_TAIL", None)
            out[tail] = m.infer_batch(ids_list, **kw)
            assert m.engine_state()["batched_attention"] == 2
            m.close()
        return out
    for n in (21, 56):
        ids_list = [synth_ids(20 + (3 * i) % 70, seed=800 + i) for i in range(n)]
        steps = [40 - (2 * i) % 31 for i in range(n)]
        out = both(blob, ids_list, opts=pkg.default_opts(dropout_seed=41), fixed_steps=steps)
        for b in (0, 7, 15, 16, n - 1):
            ref = orc.infer_chunk(blob, ids_list[b], orc.default_opts(fixed_steps=steps[b], dropout_seed=41, item=b))
            for tail in out:
                assert out[tail][b].shape == (80, steps[b]) and rms(out[tail][b], ref) <= 1e-5, (n, tail, b)
        assert all(rms(a, c) <= 1e-5 for a, c in zip(out[True], out[False]))
        assert any(not np.array_equal(a, c) for a, c in zip(out[True], out[False]))  # (the two forms really are different code)
    ids_list = [synth_ids(20 + 3 * i, seed=70 + i) for i in range(17)]
    padded = np.zeros(100, dtype=np.int64)
    padded[: len(ids_list[0])] = ids_list[0]
    mem, pm = orc.encoder(blob, padded)
    rig = rigged_gate_blob(orc, blob, mem, pm, len(ids_list[0]), 4, 40)
    out = both(rig, ids_list, opts=pkg.default_opts(dropout_seed=4, max_steps=70))
    counts = set()
    for b, ids in enumerate(ids_list):
        ref = orc.infer_chunk(rig, ids, orc.default_opts(dropout_seed=4, max_steps=70, item=b))
        for tail in out:
            assert out[tail][b].shape == ref.shape and rms(out[tail][b], ref) <= 1e-5, (tail, b)
        counts.add(ref.shape[1])
    assert len(counts) > 1


@pytest.mark.parametrize("form", ["2", "1"])
def test_lost_attention_block_falls_back(pkg, orc, blob, capfd, form):
    """The blocks of a chunk wait for each other's partial energies inside one launch.  With one block never
    publishing (test hook) the bounded spins run out, the error word is set, and the handle decodes the request
    again with separate kernels: correct frames, a message on stderr, no hang; engine_reset restores the fast form."""
    ids_list, steps = _batch_case(17)  # (3..16 chunks have engines of their own: decoder_persistent8.hip / decoder_persistent16.hip)
    os.environ["XDTTS_ATT_FUSED"] = form
    os.environ["XDTTS_ATT_FAULT"] = "3"     # block 2 = chunk 0's third attention block in either form
    os.environ["XDTTS_ATT_SPINS"] = "20000"
    try:
        m = pkg.Tacotron2.from_blob(blob)
        mels = m.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=31), fixed_steps=steps)
        assert "batched attention exchange timed out" in capfd.readouterr().err
    finally:
        del os.environ["XDTTS_ATT_FAULT"], os.environ["XDTTS_ATT_SPINS"]
    assert m.engine_state()["batched_attention"] == 0
    for b, (ids, st) in enumerate(zip(ids_list, steps)):
        ref = orc.infer_chunk(blob, ids, orc.default_opts(fixed_steps=st, dropout_seed=31, item=b))
        assert rms(mels[b], ref) <= 1e-5, b
    try:
        m.engine_reset()
    finally:
        del os.environ["XDTTS_ATT_FUSED"]
    assert m.engine_state()["batched_attention"] == int(form)
    again = m.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=31), fixed_steps=steps)
    assert all(rms(a, b) <= 1e-5 for a, b in zip(again, mels))
    m.close()


def test_lost_h_dec_in_the_two_launch_form_falls_back(pkg, orc, blob, capfd):
    """In the two-launch form the four tail blocks of a chunk wait, inside the decoder-LSTM launch, for the h_dec granules of all
    256 LSTM blocks.  With one block never publishing (test hook) the bounded spins run out, the error word is set and the
    handle decodes the request again with separate kernels: correct frames, a message on stderr, no hang."""
    ids_list, steps = _batch_case(17)
    os.environ["XDTTS_TAIL_FAULT"] = "7"
    os.environ["XDTTS_ATT_SPINS"] = "20000"
    try:
        m = pkg.Tacotron2.from_blob(blob)
        mels = m.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=31), fixed_steps=steps)
        assert "batched attention exchange timed out" in capfd.readouterr().err
    finally:
        del os.environ["XDTTS_TAIL_FAULT"], os.environ["XDTTS_ATT_SPINS"]
    assert m.engine_state()["batched_attention"] == 0
    for b, (ids, st) in enumerate(zip(ids_list, steps)):
        ref = orc.infer_chunk(blob, ids, orc.default_opts(fixed_steps=st, dropout_seed=31, item=b))
        assert rms(mels[b], ref) <= 1e-5, b
    m.engine_reset()
    again = m.infer_batch(ids_list, opts=pkg.default_opts(dropout_seed=31), fixed_steps=steps)
    assert m.engine_state()["batched_attention"] == 2 and all(rms(a, b) <= 1e-5 for a, b in zip(again, mels))
    m.close()


def test_gemm_tile_shapes_give_identical_results(tmp_path):
    """k_gemm_nt picks 32x32 tiles for the single-utterance shapes and 64x64 once a grid fills the chip twice;
    both accumulate every output element's K products in ascending order, so a batch decoded with either
    shape forced (XDTTS_GEMM_TILE is read once per process: two child processes) is bit-identical.  (Split-K -- the single-utterance
    shapes -- groups a tile's K products by slice: off here, XDTTS_GEMM_SPLITK=1; its own parity is every single-utterance test.)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import importlib, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import torch\n"
        "from conftest import synth_ids\n"
        "pkg = importlib.import_module('xd-tts_amd')\n"
        "m = pkg.Tacotron2.synthetic()\n"
        "ids = [synth_ids(20 + 3 * i, seed=300 + i) for i in range(24)]\n"
        "mels = m.infer_batch(ids, opts=pkg.default_opts(dropout_seed=5), fixed_steps=[6 + i %% 5 for i in range(24)])\n"
        "np.savez(sys.argv[1], *mels)\n"
    ) % (root, os.path.join(root, "tests"))
    out = {}
    for tile in ("32", "64"):
        path = str(tmp_path / ("mels_%s.npz" % tile))
        env = dict(os.environ, XDTTS_GEMM_TILE=tile, XDTTS_GEMM_SPLITK="1")
        r = subprocess.run([sys.executable, "-c", script, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        with np.load(path) as z:
            out[tile] = [z[k] for k in z.files]
    assert len(out["32"]) == 24
    assert all(np.array_equal(a, b) for a, b in zip(out["32"], out["64"]))


@pytest.mark.parametrize("switch", ["XDTTS_HRING", "XDTTS_HRING=1r", "XDTTS_HFIRST"])
def test_round6_rebuilds_of_the_two_launch_engine_stay_parity_green(switch, tmp_path):
    """The two forms of the batched iteration that round 6 built, measured and did not make the default (h_att(s) as an
    in-launch operand ring for the decoder LSTM's h_att columns; the attention launch multiplying its own h_att(s-1) columns ahead of
    x(s)) are still in the library behind an environment switch (read once per process: a child process).  Each must keep producing the
    oracle's frames: 21 chunks of different lengths, fixed and different step counts, every chunk against its own oracle run."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import importlib, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import torch, oracle\n"
        "from conftest import synth_ids, rms\n"
        "pkg = importlib.import_module('xd-tts_amd')\n"
        "orc = oracle.Oracle('f32')\n"
        "blob = orc.weights_synthetic(seed=20240327, rec_scale=1.0)\n"
        "m = pkg.Tacotron2.from_blob(blob)\n"
        "n = 21\n"
        "ids = [synth_ids(18 + 4 * i, seed=800 + i) for i in range(n)]\n"
        "steps = [9 + (5 * i) %% 11 for i in range(n)]\n"
        "mels = m.infer_batch(ids, opts=pkg.default_opts(dropout_seed=23), fixed_steps=steps)\n"
        "assert m.engine_state()['batched_attention'] == 2, m.engine_state()\n"
        "worst = 0.0\n"
        "for b in range(n):\n"
        "    ref = orc.infer_chunk(blob, ids[b], orc.default_opts(fixed_steps=steps[b], dropout_seed=23, item=b))\n"
        "    assert mels[b].shape == ref.shape, (b, mels[b].shape, ref.shape)\n"
        "    worst = max(worst, rms(mels[b], ref))\n"
        "print('WORST %%.3e' %% worst)\n"
    ) % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, **{switch.split("=")[0]: (switch.split("=") + ["1"])[1]}), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    worst = float([ln for ln in r.stdout.splitlines() if ln.startswith("WORST")][-1].split()[1])
    assert worst <= 1e-5, (switch, worst)


@pytest.mark.parametrize("env", [{"XDTTS_GEMM_SPLITK": "2"}, {"XDTTS_GEMM_SPLITK": "3"}, {"XDTTS_GEMM_SPLITK": "4"},
                                 {"XDTTS_GEMM_SPLITK": "4", "XDTTS_GEMM_SPLIT_TILE": "64"}], ids=lambda e: "-".join(e.values()))
def test_split_k_gemm_at_forced_slice_counts(env, tmp_path):
    """k_gemm_nt's split-K (gemm.hip: grid z = item x K-slice, partial tiles through a workspace, the last arriver sums them in slice
    order) is chosen by gemm_splitk_plan from the shape; here the slice count and the tile are forced (read once per process: a child)
    on every single-utterance GEMM that can take them -- encoder convolutions, BiLSTM input projection, memory layer, post-net -- for
    chunk lengths whose row counts are not multiples of the tile (1, 7, 37 ids) and the full window (100 ids): the frames must stay the
    oracle's to 1e-5 (a slice boundary that dropped or doubled a K slab would show at 1e-2)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import importlib, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import torch, oracle\n"
        "from conftest import synth_ids, rms\n"
        "pkg = importlib.import_module('xd-tts_amd')\n"
        "orc = oracle.Oracle('f32')\n"
        "blob = orc.weights_synthetic(seed=20240327, rec_scale=1.0)\n"
        "m = pkg.Tacotron2.from_blob(blob)\n"
        "worst = 0.0\n"
        "for n, steps in ((1, 9), (7, 33), (37, 70), (100, 131)):\n"
        "    ids = synth_ids(n, seed=900 + n)\n"
        "    mel = m.infer_batch([ids], opts=pkg.default_opts(dropout_seed=5), fixed_steps=[steps])[0]\n"
        "    ref = orc.infer_chunk(blob, ids, orc.default_opts(fixed_steps=steps, dropout_seed=5, item=0))\n"
        "    assert mel.shape == ref.shape, (n, mel.shape, ref.shape)\n"
        "    worst = max(worst, rms(mel, ref))\n"
        "print('WORST %%.3e' %% worst)\n"
    ) % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    worst = float([ln for ln in r.stdout.splitlines() if ln.startswith("WORST")][-1].split()[1])
    assert worst <= 1e-5, (env, worst)
