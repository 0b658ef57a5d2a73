"""G6 on the device: xdtts_griffinlim_opts.output_normalise / rms_target (include/xdtts.h) against the
oracle's orc_output_normalise, through every entry that ends GriffinLim::infer (src/lib.rs:141): the
single call, the vocoder batch (utterances normalised per launch group) and the two pipelines."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, dtype=np.float64) ** 2)))


def _mel(rng, F):
    return (rng.uniform(-7.0, -1.0, size=(80, F)) + 1.5 * np.sin(np.arange(F) / 4.0)[None, :]).astype(np.float32)


def test_every_mode_of_the_single_call_matches_the_oracle(pkg, orc):
    v = pkg.create_griffin_lim(iters=8, seed=4)
    rng = np.random.default_rng(0)
    for F in (2, 7, 40, 333):  # fallback engine (F < 16) and the persistent one
        mel = _mel(rng, F)
        v.set_opts(output_normalise=0)
        raw = v.infer(mel)
        assert raw.shape == (256 * (F - 1),) and _rms(raw) > 0
        for mode, target in ((1, 0.1), (2, 0.1), (2, 0.25), (3, 0.1), (3, 0.6)):
            v.set_opts(output_normalise=mode, rms_target=target)
            got = v.infer(mel)
            ref = orc.output_normalise(raw, mode=mode, target=target)
            assert np.abs(got - ref).max() <= 4e-7 * float(np.abs(ref).max()), (F, mode)
            if mode == 2 or (mode == 3 and target == 0.1):
                assert abs(_rms(got) - target) <= 2e-7 * target / 0.1 + 1e-7
            elif mode == 3:   # rms 0.6 would push the peaks past 1: the scale stops at 1 / max|y| (no sample for the i16 cast to clip)
                assert abs(float(np.abs(got).max()) - 1.0) <= 2e-7 and _rms(got) < target and float(np.abs(orc.output_normalise(raw, mode=2, target=target)).max()) > 1.0
            else:
                assert float(np.abs(got).max()) == 1.0
        v.set_opts(output_normalise=2, rms_target=0.1)
        assert np.array_equal(v.infer(mel), v.infer(mel))  # fixed reduction order: the same bits every time
        # xdtts_griffinlim_infer_linear is the loop alone (G2..G5): never normalised
        S = v.mel_to_linear(mel)
        assert np.array_equal(v.infer_linear(S), raw)
    for bad in (dict(output_normalise=4), dict(output_normalise=-1), dict(rms_target=0.0), dict(rms_target=float("nan"))):
        with pytest.raises(pkg.XdttsError) as e:
            v.set_opts(**bad)
        assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    # an all-zero signal stays as it is (mel -> exp -> ... cannot make one; the linear entry's S = 0 can, but that entry
    # does not normalise; so the rule is exercised through the oracle restatement in tests/test_reference_audio_cpu.py)
    v.close()


def test_a_vocoder_batch_normalises_each_utterance_on_its_own(pkg, orc):
    v = pkg.create_griffin_lim(iters=6, seed=2)
    rng = np.random.default_rng(1)
    Fs = [120, 9, 333, 40, 1100, 64, 17, 250, 5, 800]  # tiny ones and a long one ride alone, the rest share launches
    mels = [_mel(rng, F) * (0.6 + 0.1 * i) for i, F in enumerate(Fs)]  # different levels before normalisation
    v.set_opts(batch_shape=4, output_normalise=0)
    raw = v.infer_batch(mels)
    levels = [_rms(a) for a in raw]
    assert max(levels) > 1.5 * min(levels)
    for mode in (2, 1, 3):
        v.set_opts(batch_shape=4, output_normalise=mode)
        got = v.infer_batch(mels)
        single = [v.infer(m) for m in mels]
        for a, b, r in zip(got, single, raw):
            assert np.array_equal(a, b)  # batch_shape 4: bit for bit the single call, normalisation included
            ref = orc.output_normalise(r, mode=mode)
            assert np.abs(a - ref).max() <= 4e-7 * float(np.abs(ref).max())
        v.set_opts(batch_shape=0)
        for a, r in zip(v.infer_batch(mels), raw):  # (8-frame workgroups: another summation order inside Griffin-Lim)
            pk = float(np.abs(a).max())
            if mode == 3:  # rms 0.1, or -- a crest factor above 20 dB (the 4-frame utterance) -- a peak of exactly 1 below that level
                assert abs(_rms(a) - 0.1) <= 1e-6 or (abs(pk - 1.0) <= 1e-6 and _rms(a) < 0.1)
            else:
                assert abs((_rms(a) if mode == 2 else pk) - (0.1 if mode == 2 else 1.0)) <= 1e-6
    v.close()


def test_the_pipelines_return_normalised_audio(pkg, model, orc):
    v = pkg.create_griffin_lim(iters=5, seed=1)
    ids = np.array([108, 119, 11, 88, 113, 108, 120, 11, 116, 7], dtype=np.int64)
    o = pkg.default_opts(fixed_steps=24, dropout_seed=5)
    mel, audio = pkg.synthesize(model, v, ids, opts=o)
    assert abs(_rms(audio) - 0.1) <= 1e-6
    assert np.array_equal(audio, v.infer(mel))
    v.set_opts(output_normalise=0)
    _, raw = pkg.synthesize(model, v, ids, opts=o)
    assert np.abs(audio - orc.output_normalise(raw, mode=3)).max() <= 4e-7 * float(np.abs(audio).max())
    v.set_opts(output_normalise=3)
    groups = [[ids], [ids[:6], ids[3:]], [ids[::-1].copy()]]
    steps = [[20], [18, 30], [41]]
    mels, audios = pkg.synthesize_batch(model, v, groups, opts=pkg.default_opts(dropout_seed=5), fixed_steps=steps)
    for m, a in zip(mels, audios):
        assert a.shape == (256 * (m.shape[1] - 1),) and abs(_rms(a) - 0.1) <= 1e-6
    # the i16 stage (src/lib.rs:155) on normalised audio: nothing saturates at this level
    pcm = pkg.audio_to_i16(audios[1])
    assert int(np.abs(pcm.astype(np.int32)).max()) < 32767
    v.close()
