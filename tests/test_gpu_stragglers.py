"""The in-launch exchanges under UNEVEN load (MI355X_MICROARCH.md: "test every hand-off under uneven load, checking every
word"): one workgroup of the persistent decoder / the persistent Griffin-Lim kernel stalls for ~7 us at a different point of
every step (test hooks XDTTS_PERSIST_SLOW / XDTTS_GL_SLOW).  The tagged two-slot granule protocol must hold every other
workgroup to at most one step ahead of the straggler and hand it untorn, un-overwritten data: results bit-identical to the
undisturbed run, whichever role the straggler has."""
import os

import numpy as np
import pytest

from conftest import synth_ids

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("B", [1, 2])
def test_persistent_decoder_with_a_straggler_workgroup(pkg, model, B):
    ids = [synth_ids(n, seed=50 + i) for i, n in enumerate([61, 33][:B])]
    steps = np.array([48, 29][:B], dtype=np.int32)   # (a pair: the survivor continues on the 1-chunk kernel)
    o = pkg.default_opts(dropout_seed=8)
    want = [m.copy() for m in model.infer_batch(ids, opts=o, fixed_steps=steps)]
    # attention role (workgroup 0), projection / prenet role (8 B + 1), a plain LSTM slice (200), the last workgroup
    for wg in (1, 8 * B + 2, 201, 256):
        with env(XDTTS_PERSIST_SLOW=wg):
            got = model.infer_batch(ids, opts=o, fixed_steps=steps)
        for g, w in zip(got, want):
            assert np.array_equal(g, w), wg
    assert model.engine_state()["decoder_persistent"] == 1   # no exchange timed out


def test_persistent_griffinlim_with_a_straggler_workgroup(pkg, orc):
    import importlib

    wl = importlib.import_module("xd-tts_amd.workloads")
    F = 203   # 51 workgroups, the last one with 3 frames
    S = wl.chirp_magnitude(F)
    voc = pkg.create_griffin_lim(iters=12, seed=5)
    want = voc.infer_linear(S).copy()
    for wg in (1, 2, 26, 51):
        with env(XDTTS_GL_SLOW=wg):
            assert np.array_equal(voc.infer_linear(S), want), wg
    voc.close()


def test_batched_decoder_with_a_straggler_block(pkg, model):
    """Lock-step batches of 5..64 chunks: two launches per step whose blocks hand h_att, the partial energies, h_dec and the mel
    rows to each other as tagged granules.  One block of BOTH launches stalls ~7 us between its LSTM pass and its exchange role, every
    other step (test hook XDTTS_ATT_SLOW): an attention / tail block of chunk 0, one of another chunk, a decoder-LSTM block that is no chunk's tail, the
    last block.  Results bit-identical to the undisturbed run."""
    n = 21
    ids = [synth_ids(20 + 3 * i, seed=900 + i) for i in range(n)]
    steps = np.array([30 - (i % 9) for i in range(n)], dtype=np.int32)
    o = pkg.default_opts(dropout_seed=12)
    m = pkg.Tacotron2.synthetic()
    want = [x.copy() for x in m.infer_batch(ids, opts=o, fixed_steps=steps)]
    for blk in (1, 2, 47, 201, 256):
        with env(XDTTS_ATT_SLOW=blk):
            got = m.infer_batch(ids, opts=o, fixed_steps=steps)
        for g, w in zip(got, want):
            assert np.array_equal(g, w), blk
    assert m.engine_state()["batched_attention"] == 2   # no exchange timed out
    m.close()
