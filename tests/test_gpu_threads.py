"""The `&self` / `Send + Sync` surface (SURVEY 8(b) "Threading"; src/lib.rs:110-159 takes &self, the Rust shim declares
`unsafe impl Sync`): one handle called from two threads, and two mel-gen handles plus one
vocoder handle on one GPU with interleaved calls.  Every launch that needs the whole chip co-resident (persistent decoder,
cooperative encoder BiLSTM, persistent Griffin-Lim) is serialised by the per-GPU lock in api.cpp: results must be bit-equal
to the serial run, nothing may deadlock, and the handles must still be on their fast engines afterwards.
(ctypes releases the GIL around every foreign call, so the calls really overlap.)"""
import threading

import numpy as np
import pytest

from conftest import synth_ids

pytestmark = pytest.mark.gpu


def run_threads(fns, timeout=300):
    out, err = [None] * len(fns), []

    def wrap(i, f):
        try:
            out[i] = f()
        except Exception as e:  # noqa: BLE001
            err.append(e)

    ts = [threading.Thread(target=wrap, args=(i, f)) for i, f in enumerate(fns)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout)
    assert not any(t.is_alive() for t in ts), "deadlock: a thread did not return"
    assert not err, err
    return out


def test_two_threads_on_one_tacotron2_handle(pkg, model):
    reqs = [(synth_ids(30 + 7 * i, seed=80 + i), pkg.default_opts(fixed_steps=20 + 3 * i, dropout_seed=i)) for i in range(6)]
    serial = [model.infer(ids, opts=o).copy() for ids, o in reqs]

    def worker(k):
        return lambda: [model.infer(reqs[j][0], opts=reqs[j][1]).copy() for j in range(k, len(reqs), 2)] * 1

    a, b = run_threads([worker(0), worker(1)])
    got = [None] * len(reqs)
    got[0::2], got[1::2] = a, b
    for g, s in zip(got, serial):
        assert np.array_equal(g, s)
    assert model.engine_state()["decoder_persistent"] == 1


def test_two_handles_and_one_vocoder_interleaved(pkg, blob):
    m1, m2 = pkg.Tacotron2.from_blob(blob), pkg.Tacotron2.from_blob(blob)
    voc = pkg.create_griffin_lim(iters=12, seed=3)
    reqs = [(synth_ids(25 + 11 * i, seed=90 + i), pkg.default_opts(fixed_steps=24 + 4 * i, dropout_seed=10 + i)) for i in range(5)]
    serial = [[x.copy() for x in pkg.synthesize(m1, voc, ids, opts=o)] for ids, o in reqs]

    def worker(m, order):
        return lambda: {j: [x.copy() for x in pkg.synthesize(m, voc, reqs[j][0], opts=reqs[j][1])] for j in order}

    a, b = run_threads([worker(m1, [0, 1, 2, 3, 4]), worker(m2, [4, 3, 2, 1, 0])])
    for res in (a, b):
        for j, (mel, audio) in res.items():
            assert np.array_equal(mel, serial[j][0]) and np.array_equal(audio, serial[j][1]), j
    # a batched decode (cooperative attention launch) on one handle beside persistent decodes on the other
    ids6 = [synth_ids(20 + 4 * i, seed=70 + i) for i in range(17)]  # (17 chunks: the two-launch batched engine; up to 16 run on the persistent MFMA engines)
    ob = pkg.default_opts(fixed_steps=18, dropout_seed=4)
    want_b = [x.copy() for x in m1.infer_batch(ids6, opts=ob)]
    got_b, got_s = run_threads([lambda: [x.copy() for x in m1.infer_batch(ids6, opts=ob)],
                                lambda: [m2.infer(reqs[j][0], opts=reqs[j][1]).copy() for j in range(5)]])
    assert all(np.array_equal(x, y) for x, y in zip(got_b, want_b))
    assert all(np.array_equal(x, serial[j][0]) for j, x in enumerate(got_s))
    # small batches (the 3..8-chunk persistent engine: its grid owns the chip too) on both handles at once, beside each other and
    # beside single decodes: the chip lock makes the launches take turns, every result keeps its bits, nothing times out
    ids4 = [synth_ids(22 + 13 * i, seed=30 + i) for i in range(4)]
    o4 = pkg.default_opts(fixed_steps=21, dropout_seed=8)
    want4 = [x.copy() for x in m1.infer_batch(ids4, opts=o4)]
    want5 = [x.copy() for x in m2.infer_batch(ids6[:5], opts=ob)]
    got4, got5, got1 = run_threads([lambda: [[x.copy() for x in m1.infer_batch(ids4, opts=o4)] for _ in range(3)],
                                    lambda: [[x.copy() for x in m2.infer_batch(ids6[:5], opts=ob)] for _ in range(3)],
                                    lambda: [voc.infer(serial[j][0]).copy() for j in range(5)]])
    assert all(np.array_equal(x, y) for rep in got4 for x, y in zip(rep, want4))
    assert all(np.array_equal(x, y) for rep in got5 for x, y in zip(rep, want5))
    assert all(np.array_equal(x, serial[j][1]) for j, x in enumerate(got1))
    for m in (m1, m2):
        st = m.engine_state()
        assert st["decoder_persistent"] == 1 and st["encoder_cooperative"] == 1 and st["batched_attention"] == 2 and st["decoder_persistent8"] == 1
        m.close()
    voc.close()
