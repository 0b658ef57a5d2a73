"""GPU parity, the engine of lock-step batches of 3..16 chunks (csrc/decoder_persistent8.hip: 4 / 8 chunk slots,
csrc/decoder_persistent16.hip: 16 -- one persistent launch for the whole decoder loop of mod.rs:302-342, the LSTMs of all chunks on
the matrix cores, the state crossing CUs through write-once rings).  Free-running batches against the CPU oracle chunk by chunk, the stop rule on the device, the engines either side of
it as a second opinion, the lost-workgroup path.  (One decoder_iter from an imported state, engine by engine:
tests/test_gpu_engine_hooks.py.)"""
import os

import numpy as np
import pytest

from conftest import rms, synth_ids
from test_gpu_tacotron2_more import LOGIT_06, rigged_gate_blob

pytestmark = pytest.mark.gpu


def handle(pkg, blob, p8="1"):
    os.environ["XDTTS_P8"] = p8  # read when the handle is created
    try:
        return pkg.Tacotron2.from_blob(blob)
    finally:
        del os.environ["XDTTS_P8"]


@pytest.mark.parametrize("B", [3, 4, 5, 7, 8, 9, 12, 13, 16])
def test_ragged_batches_match_the_oracle_chunk_by_chunk(pkg, orc, blob, B):
    """Chunks of different lengths (2..100 ids: the full 100-id window too) that stop at different steps: every chunk must
    equal its own single-chunk oracle run (dropout stream = index of the chunk), to 1e-5 RMS, frame counts identical."""
    lens = [37, 100, 2, 64, 23, 81, 9, 55, 71, 14, 92, 48, 5, 66, 30, 87][:B]
    steps = np.asarray([40, 25, 33, 12, 40, 18, 29, 37, 22, 40, 31, 8, 36, 27, 15, 39][:B], dtype=np.int32)
    ids = [synth_ids(n, seed=40 + i) for i, n in enumerate(lens)]
    m = handle(pkg, blob)
    o = pkg.default_opts(dropout_seed=7, item_base=2)
    mels = m.infer_batch(ids, opts=o, fixed_steps=steps)
    assert m.engine_state()["decoder_persistent8"] == 1
    again = m.infer_batch(ids, opts=o, fixed_steps=steps)
    for b in range(B):
        ref = orc.infer_chunk(blob, ids[b], orc.default_opts(fixed_steps=int(steps[b]), dropout_seed=7, item=2 + b))
        assert mels[b].shape == ref.shape == (80, steps[b]) and rms(mels[b], ref) <= 1e-5, b
        assert np.array_equal(mels[b], again[b])  # the same bits every time
    # the engines that served these sizes before (pairs of the persistent decoder / the batched engine) agree
    m0 = handle(pkg, blob, "0")
    other = m0.infer_batch(ids, opts=o, fixed_steps=steps)
    assert m0.engine_state()["decoder_persistent8"] == 0
    assert all(rms(a, c) <= 1e-5 for a, c in zip(mels, other))
    m.close()
    m0.close()


@pytest.mark.parametrize("B", [5, 11])
def test_the_stop_rule_ends_every_chunk_on_its_own(pkg, orc, blob, B):
    """mod.rs:319-324 on the device, per chunk: with a rigged gate the chunks of a batch stop at different frames (the tripping
    frame kept), the launch ends when the last one has -- frame counts and frames equal to the oracle's."""
    ids = np.zeros(100, dtype=np.int64)
    ids[:33] = synth_ids(33)
    mem, pm = orc.encoder(blob, ids)
    rig = rigged_gate_blob(orc, blob, mem, pm, 33, 21, 30)
    lens = [33, 57, 12, 70, 45, 26, 88, 19, 61, 40, 95][:B]
    chunks = [synth_ids(n, seed=1 + i) for i, n in enumerate(lens)]
    m = handle(pkg, rig)
    mels = m.infer_batch(chunks, opts=pkg.default_opts(dropout_seed=21, max_steps=48))
    assert m.engine_state()["decoder_persistent8"] == 1
    counts = set()
    for b, c in enumerate(chunks):
        ref = orc.infer_chunk(rig, c, orc.default_opts(dropout_seed=21, max_steps=48, item=b))
        assert mels[b].shape == ref.shape and rms(mels[b], ref) <= 1e-5, (b, mels[b].shape, ref.shape)
        counts.add(ref.shape[1])
    assert len(counts) > 1 and min(counts) < 48  # (the gate, not the cap, ended some of them)
    m.close()


@pytest.mark.parametrize("B", [5, 12])
def test_a_lost_workgroup_times_out_and_the_request_is_decoded_again(pkg, orc, blob, capfd, B):
    """Every spin of the kernel is bounded.  With one workgroup never showing up (test hook) the others run out of polls, set
    the error word and leave; the handle decodes the request again on its other engines: correct frames, a message on
    stderr, no hang; engine_reset puts the engine back."""
    lens = [30, 41, 18, 66, 25, 52, 9, 77, 36, 60, 21, 44][:B]
    ids = [synth_ids(n, seed=70 + i) for i, n in enumerate(lens)]
    o = pkg.default_opts(dropout_seed=3)
    steps = np.asarray([12, 9, 14, 7, 11, 8, 13, 10, 6, 12, 9, 14][:B], dtype=np.int32)
    m = handle(pkg, blob)
    good = m.infer_batch(ids, opts=o, fixed_steps=steps)
    assert m.engine_state()["decoder_persistent8"] == 1
    os.environ["XDTTS_PERSIST_FAULT"] = "131"
    os.environ["XDTTS_PERSIST_SPINS"] = "20000"
    try:
        mels = m.infer_batch(ids, opts=o, fixed_steps=steps)
        assert "persistent MFMA decoder exchange timed out" in capfd.readouterr().err
    finally:
        del os.environ["XDTTS_PERSIST_FAULT"], os.environ["XDTTS_PERSIST_SPINS"]
    assert m.engine_state()["decoder_persistent8"] == 0
    assert all(a.shape == c.shape and rms(a, c) <= 1e-5 for a, c in zip(mels, good))
    m.engine_reset()
    assert m.engine_state()["decoder_persistent8"] == -1
    again = m.infer_batch(ids, opts=o, fixed_steps=steps)
    assert m.engine_state()["decoder_persistent8"] == 1 and all(np.array_equal(a, c) for a, c in zip(again, good))
    m.close()


def test_long_sequences_and_the_sizes_either_side(pkg, orc, blob):
    """400 steps of 6 chunks (the rings hold one slab per step): the last frames still within 1e-5 of the oracle for a probe
    chunk; 2 and 17 chunks do not take this engine."""
    lens = [60, 95, 33, 71, 48, 88]
    ids = [synth_ids(n, seed=10 + i) for i, n in enumerate(lens)]
    m = handle(pkg, blob)
    o = pkg.default_opts(dropout_seed=1)
    mels = m.infer_batch(ids, opts=o, fixed_steps=[400] * 6)
    assert m.engine_state()["decoder_persistent8"] == 1
    ref = orc.infer_chunk(blob, ids[4], orc.default_opts(fixed_steps=400, dropout_seed=1, item=4))
    assert mels[4].shape == (80, 400) and rms(mels[4], ref) <= 1e-5 and rms(mels[4][:, -20:], ref[:, -20:]) <= 1e-5
    m.close()
    m = handle(pkg, blob)
    m.infer_batch(ids[:2], opts=o, fixed_steps=[5, 5])
    m.infer_batch([ids[i % 6] for i in range(17)], opts=o, fixed_steps=[5] * 17)
    assert m.engine_state()["decoder_persistent8"] == -1  # never probed
    # ... nor does a request capped beyond 16384 steps (the rings hold one slab per step): the synthetic gate never fires, so
    # this one runs to its cap on the batched engine
    long = m.infer_batch(ids[:5], opts=pkg.default_opts(dropout_seed=1, max_steps=16500))
    assert all(x.shape == (80, 16500) for x in long) and m.engine_state()["decoder_persistent8"] == -1
    assert all(np.isfinite(x).all() for x in long)
    m.close()
