"""xdtts_synthesize_sequence: a sequence of utterances, each decoded alone as the reference does it (src/lib.rs:110-159, one
`infer` after the other), the vocoder of utterance u overlapped with the encoder of utterance u + 1 ("what's
missing" 5: the per-GPU lock serialised every call).  Same bits as one xdtts_synthesize_ids call per utterance; the engines stay
on their fast paths (two grids that want the chip co-resident -- Griffin-Lim and the cooperative encoder BiLSTM -- are in flight
together, the frame loop waits for the vocoder); and it is faster."""
import importlib
import time

import numpy as np
import pytest

from conftest import synth_ids

pytestmark = pytest.mark.gpu
wl = importlib.import_module("xd-tts_amd.workloads")


def _utterances(pkg):
    lens = [120, 40, 95, 28, 130, 77]
    ids = [synth_ids(n, seed=11 + i) for i, n in enumerate(lens)]
    splits = [np.asarray(pkg.find_splits(x, 100), dtype=np.uintp) for x in ids]
    return ids, splits


def test_sequence_equals_one_call_per_utterance(pkg, model, capfd):
    voc = pkg.create_griffin_lim(iters=30, seed=9)
    ids, splits = _utterances(pkg)
    o = pkg.default_opts(fixed_frames_per_id=3.0, dropout_seed=4, item_base=0)
    one = [pkg.synthesize(model, voc, x, splits=s, opts=o) for x, s in zip(ids, splits)]
    mels, audios = pkg.synthesize_sequence(model, voc, ids, splits, opts=o)
    for (m1, a1), m2, a2 in zip(one, mels, audios):
        assert m1.shape == m2.shape and a1.shape == a2.shape
        assert np.array_equal(m1, m2) and np.array_equal(a1, a2)
    # without the mels, and a single utterance
    _none, audios2 = pkg.synthesize_sequence(model, voc, ids[:1], splits[:1], opts=o, want_mels=False)
    assert _none is None and np.array_equal(audios2[0], one[0][1])
    st = model.engine_state()
    assert st["decoder_persistent"] == 1 and st["encoder_cooperative"] == 1
    err = capfd.readouterr().err
    assert "timed out" not in err and "refused" not in err, err
    # argument errors: nothing is returned
    with pytest.raises(pkg.XdttsError):
        pkg.synthesize_sequence(model, voc, [ids[0], np.zeros(0, dtype=np.int64)], None, opts=o)
    with pytest.raises(pkg.XdttsError):   # 130 ids without splits: longer than the window (mod.rs:363)
        pkg.synthesize_sequence(model, voc, [ids[1], ids[4]], None, opts=o)
    voc.close()


def test_sequence_with_the_stop_rule_deciding(pkg, orc, blob):
    """gate on (the reference's real mode): the frame counts come from the device, nothing can be enqueued ahead of them"""
    from test_gpu_tacotron2_more import rigged_gate_blob

    n = 30
    ids0 = np.zeros(100, dtype=np.int64)
    ids0[:n] = synth_ids(n, seed=5)
    mem, pm = orc.encoder(blob, ids0)
    m = pkg.Tacotron2.from_blob(rigged_gate_blob(orc, blob, mem, pm, n, 8, 40))
    voc = pkg.create_griffin_lim(iters=10, seed=2)
    ids = [synth_ids(30, seed=5), synth_ids(55, seed=6), synth_ids(18, seed=7)]
    o = pkg.default_opts(dropout_seed=8, max_steps=60)
    one = [pkg.synthesize(m, voc, x, opts=o) for x in ids]
    mels, audios = pkg.synthesize_sequence(m, voc, ids, None, opts=o)
    assert len({x.shape[1] for x in mels}) > 1
    for (m1, a1), m2, a2 in zip(one, mels, audios):
        assert np.array_equal(m1, m2) and np.array_equal(a1, a2)
    voc.close()
    m.close()


def test_sequence_is_faster_than_the_calls_one_by_one(pkg, model):
    voc = pkg.create_griffin_lim(iters=60, seed=0)
    _ids, chunks, _steps = wl.config2(pkg)
    sp = np.cumsum([len(c) for c in chunks]).astype(np.uintp)
    utts = [wl.synth_ids(120, seed=1 + g) for g in range(8)]
    o = pkg.default_opts(fixed_frames_per_id=wl.FRAMES_PER_ID, dropout_seed=0, item_base=0)
    for _ in range(2):
        pkg.synthesize(model, voc, utts[0], splits=sp, opts=o)
        pkg.synthesize_sequence(model, voc, utts[:2], [sp, sp], opts=o, want_mels=False)
    best_loop = best_seq = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for x in utts:
            pkg.synthesize(model, voc, x, splits=sp, opts=o)
        best_loop = min(best_loop, time.perf_counter() - t0)
        t0 = time.perf_counter()
        pkg.synthesize_sequence(model, voc, utts, [sp] * len(utts), opts=o)
        best_seq = min(best_seq, time.perf_counter() - t0)
    print("8 headline utterances: one by one %.3f ms, as a sequence %.3f ms" % (best_loop * 1e3, best_seq * 1e3))
    assert best_seq < 0.99 * best_loop, (best_seq, best_loop)
    voc.close()


def test_sequence_survives_a_timed_out_exchange_in_either_half(pkg, model, capfd):
    """Inside a sequence the co-resident engines keep their fault policy: a Griffin-Lim launch whose exchange times out (test hook:
    a tiny poll limit + a straggler workgroup) is run again on the fallback engine when its audio is collected -- S is still intact,
    the next vocoder has not been enqueued -- and a frame loop that loses a workgroup is decoded again on the launch-per-stage
    engine; every utterance still comes out right, and the engines come back after a reset."""
    import os

    voc = pkg.create_griffin_lim(iters=12, seed=3)
    ids = [synth_ids(n, seed=70 + i) for i, n in enumerate([40, 33, 52])]
    o = pkg.default_opts(fixed_frames_per_id=2.0, dropout_seed=2)
    want = [pkg.synthesize(model, voc, x, opts=o) for x in ids]
    os.environ["XDTTS_GL_SPINS"] = "50"
    os.environ["XDTTS_GL_SLOW"] = "2"
    try:
        mels, audios = pkg.synthesize_sequence(model, voc, ids, None, opts=o)
    finally:
        del os.environ["XDTTS_GL_SPINS"], os.environ["XDTTS_GL_SLOW"]
    assert "persistent Griffin-Lim exchange timed out" in capfd.readouterr().err
    for (m1, a1), m2, a2 in zip(want, mels, audios):
        assert np.array_equal(m1, m2)
        assert a1.shape == a2.shape and float(np.sqrt(np.mean((a1 - a2) ** 2))) <= 1e-4   # (the fallback engine: same audio to fp32 drift)
    voc.close()
    voc = pkg.create_griffin_lim(iters=12, seed=3)
    os.environ["XDTTS_PERSIST_FAULT"] = "131"
    os.environ["XDTTS_PERSIST_SPINS"] = "20000"
    try:
        mels, audios = pkg.synthesize_sequence(model, voc, ids, None, opts=o)
    finally:
        del os.environ["XDTTS_PERSIST_FAULT"], os.environ["XDTTS_PERSIST_SPINS"]
    assert "persistent decoder exchange timed out" in capfd.readouterr().err
    assert model.engine_state()["decoder_persistent"] == 0
    for (m1, _a1), m2 in zip(want, mels):
        assert m1.shape == m2.shape and float(np.sqrt(np.mean((m1 - m2) ** 2))) <= 1e-5
    model.engine_reset()
    mels, audios = pkg.synthesize_sequence(model, voc, ids, None, opts=o)
    assert model.engine_state()["decoder_persistent"] == 1
    for (m1, a1), m2, a2 in zip(want, mels, audios):
        assert np.array_equal(m1, m2) and np.array_equal(a1, a2)
    voc.close()


def test_a_sequence_beside_other_handles_on_the_same_gpu(pkg, blob):
    """One thread streams a sequence through its handles while another thread calls xdtts_synthesize_ids on a second pair of handles of
    the same GPU: the sequence keeps the per-GPU lock from its first launch to its last collect, the other thread's co-resident
    launches wait their turn -- same bits as the serial runs, nobody falls off the fast engines."""
    import threading

    m1, m2 = pkg.Tacotron2.from_blob(blob), pkg.Tacotron2.from_blob(blob)
    v1, v2 = pkg.create_griffin_lim(iters=20, seed=1), pkg.create_griffin_lim(iters=20, seed=1)
    ids = [synth_ids(n, seed=90 + i) for i, n in enumerate([70, 45, 88, 31])]
    o = pkg.default_opts(fixed_frames_per_id=2.5, dropout_seed=6)
    want = [pkg.synthesize(m1, v1, x, opts=o) for x in ids]
    out, errs = {}, []

    def seq():
        try:
            for _ in range(3):
                out["seq"] = pkg.synthesize_sequence(m1, v1, ids, None, opts=o)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    def single():
        try:
            for _ in range(3):
                out["single"] = [pkg.synthesize(m2, v2, x, opts=o) for x in ids]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=seq), threading.Thread(target=single)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for u, (m, a) in enumerate(want):
        assert np.array_equal(out["seq"][0][u], m) and np.array_equal(out["seq"][1][u], a)
        assert np.array_equal(out["single"][u][0], m) and np.array_equal(out["single"][u][1], a)
    assert m1.engine_state()["decoder_persistent"] == 1 and m2.engine_state()["decoder_persistent"] == 1
    for h in (v1, v2, m1, m2):
        h.close()
