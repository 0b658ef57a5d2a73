"""Pins the CPU oracle's vocoder math (librosa 0.9 restatement, the algorithm the griffin-lim crate
ports -- slides/vocoding.typ:50) against torch.stft/istft, numpy and scipy's L-BFGS-B."""
import numpy as np
import pytest
import scipy.optimize
import torch

import torch_ref
from conftest import rms


def slaney_mel_bank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(htk=False, norm='slaney') written with numpy array ops."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    h2m = lambda f: np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-9) / min_log_hz) / logstep, f / f_sp)
    m2h = lambda m: np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)
    fft = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_f = m2h(np.linspace(h2m(np.float64(fmin)), h2m(np.float64(fmax)), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fft)
    W = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        W[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    W *= (2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels]))[:, None]
    return W.astype(np.float32)


def test_mel_filter_bank_vs_numpy(orc):
    # create_mel_filter_bank(22050.0, 1024, 80, 0.0, Some(8000.0)) -- src/tacotron2/mod.rs:453
    B = orc.mel_filter_bank(22050.0, 1024, 80, 0.0, 8000.0)
    assert B.shape == (80, 513)
    assert np.array_equal(B, slaney_mel_bank(22050, 1024, 80, 0.0, 8000.0))
    assert np.all(B.sum(1) > 0) and np.all(B[:, 372:] == 0)  # nothing above 8 kHz (bin 371.5)
    assert np.array_equal(orc.mel_filter_bank(16000.0, 512, 40, 50.0, 7600.0), slaney_mel_bank(16000, 512, 40, 50.0, 7600.0))


def test_pinv_vs_numpy(orc):
    B = orc.mel_filter_bank()
    P = orc.pinv(B)
    ref = np.linalg.pinv(B.astype(np.float64))
    assert np.abs(P - ref).max() < 1e-5 * np.abs(ref).max()
    assert np.abs(B.astype(np.float64) @ P.astype(np.float64) - np.eye(80)).max() < 1e-5


def test_nnls_lbfgsb_stops_at_the_clipped_least_squares_point(orc64):
    """librosa.util.nnls = L-BFGS-B from clip(lstsq(A, M), 0) on 0.5||Ax-M||^2 / M.size.  With that
    1/M.size scaling the projected gradient is already below pgtol at the start, so the solver
    returns its initial point: mel->linear IS clip(pinv @ M, 0).  Re-derived here with scipy."""
    A = orc64.mel_filter_bank().astype(np.float64)
    rng = np.random.default_rng(0)
    mel = rng.uniform(-9.0, 0.7, size=(80, 40))  # Tacotron2 ln-mel range
    M = np.exp(mel)
    x0 = np.clip(np.linalg.lstsq(A, M, rcond=None)[0], 0, None)

    def obj(x):
        diff = A @ x.reshape(x0.shape) - M
        return 0.5 * np.sum(diff**2) / M.size, (A.T @ diff).ravel() / M.size

    x, _f, info = scipy.optimize.fmin_l_bfgs_b(obj, x0.ravel(), bounds=[(0, None)] * x0.size, m=A.shape[1])
    assert info["nit"] == 0 and np.array_equal(x.reshape(x0.shape), x0)
    S = orc64.mel_to_linear(orc64.pinv(A), mel, power=1.7)
    assert np.abs(S - x0 ** (1 / 1.7)).max() < 2e-5 * max(1.0, np.abs(x0).max() ** (1 / 1.7))


def test_stft_istft_vs_torch(orc64):
    rng = np.random.default_rng(1)
    y = rng.standard_normal(256 * 37)
    S = orc64.stft(y)
    St = torch_ref.stft(y).numpy()
    assert S.shape == (513, 38, 2)
    assert np.abs(S[..., 0] + 1j * S[..., 1] - St).max() < 1e-11
    yi = orc64.istft(S)
    assert np.abs(yi - y).max() < 1e-12  # perfect reconstruction with hann, hop = n_fft/4
    Sr = rng.standard_normal((513, 21, 2))
    yo = orc64.istft(Sr)
    yt = torch_ref.istft(torch.as_tensor(Sr[..., 0] + 1j * Sr[..., 1]), 256 * 20).numpy()
    assert yo.shape == (256 * 20,) and np.abs(yo - yt).max() < 1e-11


def test_griffinlim_vs_torch_restatement(orc64, orc):
    F = 24
    t = np.arange(256 * (F - 1)) / 22050.0
    sig = 0.5 * np.sin(2 * np.pi * 440 * t) + 0.3 * np.sin(2 * np.pi * (1000 + 2000 * t) * t)
    spec = orc64.stft(sig)
    S = np.hypot(spec[..., 0], spec[..., 1])
    p0 = orc64.phase_init(3, 513, F)
    assert np.abs(np.hypot(p0[..., 0], p0[..., 1]) - 1).max() < 1e-12
    a = orc64.griffinlim(S, phase0=p0, iters=12)
    b = torch_ref.griffinlim(S, p0, 12)
    assert a.shape == b.shape == (256 * (F - 1),)
    assert np.abs(a - b).max() < 1e-9
    # seeded path == explicit phase0 path; f32 flavour tracks f64
    assert np.array_equal(orc64.griffinlim(S, seed=3, iters=12), a)
    assert rms(orc.griffinlim(S, seed=3, iters=12), a) < 1e-4
    # the iteration reduces spectral inconsistency
    def err(y):
        r = orc64.stft(y)
        return np.linalg.norm(np.hypot(r[..., 0], r[..., 1]) - S) / np.linalg.norm(S)
    assert err(orc64.griffinlim(S, seed=3, iters=30)) < err(orc64.griffinlim(S, seed=3, iters=2))


def test_griffinlim_scale_equivariance(orc64):
    rng = np.random.default_rng(5)
    S = np.abs(rng.standard_normal((513, 10)))
    a = orc64.griffinlim(S, seed=1, iters=5)
    b = orc64.griffinlim(3.0 * S, seed=1, iters=5)
    assert np.abs(b - 3.0 * a).max() < 1e-10 * max(1.0, np.abs(a).max())


def test_nnls_refinement_descends_to_the_scipy_solution(orc64):
    """mel -> linear as bounded least squares.  The projected-gradient steps behind
    xdtts_griffinlim_opts.nnls_iters (restated in the oracle) must lower 1/2 |A x - m|^2 monotonically from the
    clipped least-squares start and approach scipy's exact NNLS optimum (the fixed point the crate's L-BFGS-B
    refinement -- lbfgsb 0.1.0, Cargo.lock:888-895 -- looks for)."""
    from scipy.optimize import nnls

    A = orc64.mel_filter_bank()
    P = orc64.pinv(A)
    A64 = A.astype(np.float64)
    L = orc64.nnls_lipschitz(A)
    assert abs(L - np.linalg.eigvalsh(A64 @ A64.T).max()) <= 1e-9 * L
    rng = np.random.default_rng(0)
    mel = rng.uniform(-7, -1, size=(80, 5)).astype(np.float32)
    m = np.exp(mel.astype(np.float64))
    best = np.array([0.5 * nnls(A64, m[:, t])[1] ** 2 for t in range(5)])
    prev = first = None
    for it in (0, 1, 10, 100, 2000):
        X = orc64.mel_to_linear_opts(P, A, mel, nnls_iters=it, power_mode=2)
        assert X.min() >= 0
        obj = np.array([0.5 * np.sum((A64 @ X[:, t] - m[:, t]) ** 2) for t in range(5)])
        assert np.all(obj >= best * (1 - 1e-9))
        if prev is not None:
            assert np.all(obj <= prev * (1 + 1e-12))
        prev = obj
        first = obj if first is None else first
    assert np.all((first - prev) >= 0.95 * (first - best))   # 2000 steps close >= 95 % of the gap to the optimum


def test_mel_to_linear_switches(orc64):
    A = orc64.mel_filter_bank()
    P = orc64.pinv(A)
    rng = np.random.default_rng(1)
    mel = rng.uniform(-5, -1, size=(80, 3)).astype(np.float32)
    base = orc64.mel_to_linear(P, mel, power=1.7)
    assert np.array_equal(orc64.mel_to_linear_opts(P, A, mel, power=1.7), base)            # defaults = the documented reading
    x = orc64.mel_to_linear_opts(P, A, mel, power=1.7, power_mode=2)
    assert np.allclose(base, x ** (1 / 1.7), rtol=1e-12) and np.allclose(orc64.mel_to_linear_opts(P, A, mel, power=1.7, power_mode=1), x ** 1.7, rtol=1e-12)
    lin = np.exp(mel.astype(np.float64)).astype(np.float32)                                  # already-linear input, decompress = none
    assert np.allclose(orc64.mel_to_linear_opts(P, A, lin, power_mode=2, decompress=1), x, rtol=1e-5, atol=1e-9)
    log10 = (mel.astype(np.float64) / np.log(10)).astype(np.float32)                         # 10^x de-compression of the same magnitudes
    assert np.allclose(orc64.mel_to_linear_opts(P, A, log10, power_mode=2, decompress=2), x, rtol=1e-4, atol=1e-8)
