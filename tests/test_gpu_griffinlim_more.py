"""GPU parity, vocoder: golden fixture, full GriffinLim::infer path (mel -> linear -> phase
recovery), edge sizes, the BASELINE Griffin-Lim-only config and size-independent properties."""
import os

import numpy as np
import pytest

from conftest import rms

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def voc(pkg):
    if pkg.device_count() < 1:
        pytest.skip("no HIP device")
    v = pkg.create_griffin_lim(iters=30, seed=3)  # create_griffin_lim(), src/tacotron2/mod.rs:441-458
    yield v
    v.close()


def chirps(n):
    """BASELINE.md config-5 signal: five linear chirps 100 Hz - 7 kHz plus a little noise."""
    t = np.arange(n) / 22050.0
    rng = np.random.default_rng(3)
    y = sum(0.15 * np.sin(2 * np.pi * (f0 + 0.5 * (f1 - f0) * t / t[-1]) * t) for f0, f1 in ((100, 900), (400, 2500), (1200, 4000), (3000, 5500), (5000, 7000)))
    return (y + 0.01 * rng.standard_normal(n)).astype(np.float32)


def test_golden_fixture_on_gpu(voc):
    g = np.load(os.path.join(G, "griffinlim_small.npz"))
    voc.set_seed(int(g["phase_seed"]))
    a = voc.infer_linear(g["S"], iters=int(g["iters"]))
    assert rms(a, g["audio"]) <= 1e-4 and rms(a, g["audio_f64"]) <= 1e-4


def test_full_infer_from_mel(pkg, voc, orc):
    """GriffinLim::infer(&mel) (src/lib.rs:141): ln-mel -> exp -> NNLS(=clipped pinv) -> ^(1/1.7)
    -> 30 iterations, against the oracle chain."""
    rng = np.random.default_rng(11)
    F = 40
    mel = (rng.uniform(-7.0, -1.0, size=(80, F)) + 2.0 * np.sin(np.arange(F) / 5.0)[None, :]).astype(np.float32)
    voc.set_seed(5)
    audio = voc.infer(mel)
    S = orc.mel_to_linear(orc.pinv(orc.mel_filter_bank()), mel, power=1.7)
    raw = orc.griffinlim(S, seed=5, iters=30)
    ref = orc.output_normalise(raw, mode=3, target=0.1)  # G6: the handle's default (rms 0.1, never past +-1)
    d = voc.get_opts()
    assert d.output_normalise == 3 and d.rms_target == np.float32(0.1)
    assert audio.shape == ref.shape == (256 * (F - 1),)
    assert rms(audio, ref) <= 1e-4 * 0.1 / float(np.sqrt(np.mean(raw.astype(np.float64) ** 2))) and rms(audio, ref) <= 1e-4
    assert abs(float(np.sqrt(np.mean(audio.astype(np.float64) ** 2))) - 0.1) <= 1e-6
    t = voc.last_timings()
    assert t["total_ms"] > 0 and t["iterations_ms"] > 0


def test_edge_sizes(voc, orc):
    rng = np.random.default_rng(2)
    for F in (2, 3, 4, 5, 9):
        S = np.abs(rng.standard_normal((513, F))).astype(np.float32)
        voc.set_seed(1)
        a = voc.infer_linear(S, iters=4)
        ref = orc.griffinlim(S, seed=1, iters=4)
        assert a.shape == ref.shape == (256 * (F - 1),)
        assert rms(a, ref) <= 1e-4 * max(1.0, float(np.abs(ref).max())), F


def test_errors(pkg, voc):
    with pytest.raises(pkg.XdttsError) as e:
        voc.infer(np.zeros((79, 10), dtype=np.float32))
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    with pytest.raises(pkg.XdttsError) as e:
        voc.infer(np.zeros((80, 1), dtype=np.float32))
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    with pytest.raises(pkg.XdttsError) as e:  # n_fft inferred from the basis must be 1024
        pkg.GriffinLim(np.ones((80, 257), dtype=np.float32), 192, 1.7, 30, 0.99)
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG


def test_parity_200_frames_60_iterations(voc, orc, orc64):
    """60 free-running iterations amplify fp32 rounding noise (the fp32 oracle itself ends ~1e-4 from the
    fp64 oracle here), so "within 1e-4 of the fp32 oracle" is a coin toss for ANY fp32 implementation
    whose roundings differ; the well-defined statement is that the GPU is no further from the fp64
    oracle than the fp32 oracle is (x2), and still within 1e-4-class distance of both."""
    F = 200
    sig = chirps(256 * (F - 1))
    spec = orc.stft(sig)
    S = np.hypot(spec[..., 0], spec[..., 1]).astype(np.float32)
    p0 = orc.phase_init(3, 513, F)
    a = voc.infer_linear(S, phase0=p0, iters=60)
    f32 = orc.griffinlim(S, phase0=p0, iters=60)
    f64 = orc64.griffinlim(S, phase0=p0, iters=60)
    eg, ef = rms(a, f64), rms(f32, f64)
    assert eg <= 2.0 * ef + 1e-6, (eg, ef)
    assert rms(a, f32) <= 5e-4 and eg <= 5e-4
    # and the first 10 iterations, before the noise has grown, meet the 1e-4 bar directly
    a10 = voc.infer_linear(S, phase0=p0, iters=10)
    assert rms(a10, orc.griffinlim(S, phase0=p0, iters=10)) <= 1e-4


def test_config5_full_size_properties(voc, orc):
    """BASELINE.json configs[4]: 1000-frame input, 30/60/120 iterations.  Checked through
    size-independent properties: length, finiteness, scale equivariance, determinism, and
    monotone improvement of spectral consistency with the iteration count."""
    F = 1000
    sig = chirps(256 * (F - 1))
    spec = orc.stft(sig)
    S = np.hypot(spec[..., 0], spec[..., 1]).astype(np.float32)
    voc.set_seed(3)
    out = {it: voc.infer_linear(S, iters=it) for it in (2, 30, 60, 120)}
    for it, a in out.items():
        assert a.shape == (255744,) and np.all(np.isfinite(a))

    def inconsistency(y):
        r = orc.stft(y)
        return float(np.linalg.norm(np.hypot(r[..., 0], r[..., 1]) - S) / np.linalg.norm(S))

    errs = [inconsistency(out[it]) for it in (2, 30, 60, 120)]
    assert errs[0] > errs[1] > errs[2] > errs[3]
    assert np.array_equal(voc.infer_linear(S, iters=30), out[30])  # deterministic
    scaled = voc.infer_linear(2.0 * S, iters=30)
    assert rms(scaled, 2.0 * out[30]) <= 1e-5 * max(1.0, float(np.abs(out[30]).max()))


def test_mel_to_linear_options_match_the_oracle(pkg, orc):
    """xdtts_griffinlim_opts: each convention switch of GriffinLim::infer's first step, and the NNLS
    refinement (projected gradient, two MFMA GEMMs per step), against the oracle's restatement."""
    v = pkg.create_griffin_lim(iters=8, seed=2)
    A = orc.mel_filter_bank()
    P = orc.pinv(A)
    rng = np.random.default_rng(4)
    F = 70
    mel = (rng.uniform(-7.0, -1.0, size=(80, F)) + 1.5 * np.sin(np.arange(F) / 4.0)[None, :]).astype(np.float32)
    d = v.get_opts()
    assert (d.nnls_iters, d.power_mode, d.mel_decompress, d.output_normalise) == (0, 0, 0, 3)
    base = v.mel_to_linear(mel)
    scale = float(np.sqrt(np.mean(base.astype(np.float64) ** 2)))
    for kw in (dict(), dict(power_mode=1), dict(power_mode=2), dict(mel_decompress=1), dict(mel_decompress=2),
               dict(nnls_iters=1), dict(nnls_iters=7, power_mode=2), dict(nnls_iters=60)):
        full = dict(nnls_iters=0, power_mode=0, mel_decompress=0, output_normalise=0)
        full.update(kw)
        v.set_opts(**full)
        S = v.mel_to_linear(mel)
        ref = orc.mel_to_linear_opts(P, A, mel, power=1.7, nnls_iters=full["nnls_iters"], power_mode=full["power_mode"], decompress=full["mel_decompress"])
        sc = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        assert np.all(np.isfinite(S)) and rms(S, ref) <= 2e-5 * sc, (kw, rms(S, ref), sc)
    # the refinement really lowers the mel-domain residual
    v.set_opts(nnls_iters=0, power_mode=2, mel_decompress=0)
    x0 = v.mel_to_linear(mel)
    v.set_opts(nnls_iters=40)
    x1 = v.mel_to_linear(mel)
    m = np.exp(mel.astype(np.float64))
    r0, r1 = np.linalg.norm(A.astype(np.float64) @ x0 - m), np.linalg.norm(A.astype(np.float64) @ x1 - m)
    assert r1 < 0.8 * r0 and x1.min() >= 0
    # defaults restored -> the documented reading again; peak normalisation scales the same audio to |y| <= 1
    v.set_opts(nnls_iters=0, power_mode=0, mel_decompress=0, output_normalise=0)
    assert np.array_equal(v.mel_to_linear(mel), base) and scale > 0
    a0 = v.infer(mel)
    v.set_opts(output_normalise=1)
    a1 = v.infer(mel)
    peak = float(np.abs(a0).max())
    assert abs(float(np.abs(a1).max()) - 1.0) <= 1e-6 and rms(a1, a0 / peak) <= 1e-6
    assert rms(a1, orc.output_normalise(a0, mode=1)) <= 1e-7
    with pytest.raises(pkg.XdttsError) as e:
        v.set_opts(power_mode=7)
    assert e.value.status == pkg.XDTTS_ERR_BAD_ARG
    v.close()


def test_vocoder_batch_equals_one_by_one(pkg):
    """xdtts_griffinlim_infer_batch: several utterances per persistent launch (workgroups never span two),
    tiny (< 16 frames) and long (> 1024) ones on their own path.  With batch_shape = 4 every audio is
    bit-identical to the single-utterance call, whatever the mix and the order; the default picks 8-frame
    workgroups when they save launches, and the audio then agrees within the fp32 drift of 12 iterations."""
    v = pkg.create_griffin_lim(iters=12, seed=9)
    v.set_opts(batch_shape=4)
    rng = np.random.default_rng(8)
    Fs = [37, 16, 5, 400, 1100, 19, 2, 257, 64, 1024, 333]
    mels = [(rng.uniform(-7.0, -1.0, size=(80, F)) + 1.5 * np.sin(np.arange(F) / 6.0)[None, :]).astype(np.float32) for F in Fs]
    one = [v.infer(m) for m in mels]
    got = v.infer_batch(mels)
    assert len(got) == len(Fs)
    for F, a, b in zip(Fs, got, one):
        assert a.shape == (256 * (F - 1),) and np.array_equal(a, b), F
    got2 = v.infer_batch(mels[::-1])[::-1]
    assert all(np.array_equal(a, b) for a, b in zip(got2, one))
    # many short utterances: more workgroups than CUs -> several launches
    many = [mels[0]] * 40 + [mels[3]] * 5
    outs = v.infer_batch(many)
    assert all(np.array_equal(o, one[0]) for o in outs[:40]) and all(np.array_equal(o, one[3]) for o in outs[40:])
    # default shape: whatever the packing model prices lowest -- 8 frames per workgroup where that saves launches (forced here
    # with the developer switch: since round 4 two 4-frame workgroups per CU win this case): same audio within drift, and
    # the result does not depend on the call (deterministic)
    v.set_opts(batch_shape=0)
    ref = [one[0]] * 40 + [one[3]] * 5
    os.environ["XDTTS_GL_BATCH_FORCE"] = "8"
    try:
        auto = v.infer_batch(many)
        rel = [float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)) / np.sqrt(np.mean(b.astype(np.float64) ** 2))) for a, b in zip(auto, ref)]
        assert max(rel) <= 5e-5, max(rel)  # measured 5e-6
        assert not all(np.array_equal(a, b) for a, b in zip(auto, ref))  # (the 8-frame shape really ran)
        assert all(np.array_equal(a, b) for a, b in zip(v.infer_batch(many), auto))
    finally:
        del os.environ["XDTTS_GL_BATCH_FORCE"]
    free = v.infer_batch(many)  # the model's own choice: one of the two, the same on every call
    rel = [float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)) / np.sqrt(np.mean(b.astype(np.float64) ** 2))) for a, b in zip(free, ref)]
    assert max(rel) <= 5e-5 and all(np.array_equal(a, b) for a, b in zip(v.infer_batch(many), free))
    v.set_opts(batch_shape=4)
    # options apply per utterance in a batch too
    v.set_opts(output_normalise=1, nnls_iters=3)
    single = [v.infer(m) for m in mels[:4]]
    for a, b in zip(v.infer_batch(mels[:4]), single):
        assert np.array_equal(a, b) and abs(float(np.abs(a).max()) - 1.0) <= 1e-6
    v.close()
