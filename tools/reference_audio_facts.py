#!/usr/bin/env python3
"""Forensics on the only outputs of the hot path the reference holds: slides/audio/*.wav.

BUILD-CONTAINER ONLY: reads /root/reference/slides/audio (which does not exist on the GPU box) and
writes tests/golden/reference_audio_facts.json -- statistics, not audio; no sample of the reference's
files is stored.  numpy + scipy only (no oracle, no product library): every figure can be re-derived
by anyone holding the reference checkout.

Two of the five files (goodbye.wav, capital_nonsense.wav) are mono / 22 050 Hz / int16 = WAV_SPEC
(src/lib.rs:25-30), i.e. what `app` writes at src/lib.rs:153-157 from GriffinLim::infer's Vec<f32>
(src/lib.rs:141).  The script answers, per such file:

  format       does the header equal WAV_SPEC?
  level        RMS / peak of full scale; the f32 RMS BEFORE the `(s * i16::MAX) as i16` cast under
               the two candidate casts (truncation toward zero = Rust `as`; round-to-nearest)
  length       samples mod 256 -> frame-count convention 256*(F-1) (librosa istft, center) or 256*F
  band         share of energy above 8 kHz -> create_mel_filter_bank(.., fmax = Some(8000.0))
               (src/tacotron2/mod.rs:453) and "Griffin-Lim from an 80-band mel", not WaveGlow
  mel range    Griffin-Lim's output has |STFT(y)| ~ c*S with S = clip(pinv(B) M, 0)^e, e = 1/1.7
               (librosa mel_to_stft), 1 or 1.7 -- the three `power_mode`s of xdtts_griffinlim_opts.
               For the right e, Z = |STFT(y)|^(1/e) is (up to the clip and Griffin-Lim's residual
               inconsistency) in the 80-dimensional column space of pinv(B); the relative residual
               of |STFT(y)| against clip(pinv(B) (B Z), 0)^e is scale-free, so the unknown gain c
               drops out.  It separates "vocoded from an 80-band mel through pinv(B)" (0.09) from
               the float32 files of other pipelines (0.23-0.33) -- but it can NOT tell the three
               exponents apart: `calibration` re-vocodes a mel of the file itself under each
               exponent with a numpy Griffin-Lim (30 iterations, momentum 0.99) and the residual is
               lowest at e = 1 whichever exponent made the audio.  power_mode stays undetermined.
  alignment    the residual (e = 1) as a function of the analysis frame offset (0..255 samples):
               its minimum says where the synthesis frames sat -> centre / trim convention of the ISTFT
"""
import json
import os
import struct
import sys

import numpy as np

AUDIO_DIR = "/root/reference/slides/audio"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_audio_facts.json")
WAV_SPEC = {"channels": 1, "sample_rate": 22050, "bits_per_sample": 16, "sample_format": "int"}  # src/lib.rs:25-30
N_FFT, HOP, N_MELS, SR, FMAX, POWER = 1024, 256, 80, 22050, 8000.0, 1.7  # src/tacotron2/mod.rs:453-456


def read_wav(path):
    """RIFF reader for PCM int16 (format 1) and IEEE float32 (format 3) files."""
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE"
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        tag, size = raw[pos : pos + 4], struct.unpack("<I", raw[pos + 4 : pos + 8])[0]
        body = raw[pos + 8 : pos + 8 + size]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif tag == b"data":
            data = body
        pos += 8 + size + (size & 1)
    code, ch, sr, _, _, bits = fmt
    hdr = {"channels": ch, "sample_rate": sr, "bits_per_sample": bits, "sample_format": {1: "int", 3: "float"}.get(code, str(code))}
    x = np.frombuffer(data, dtype={(1, 16): "<i2", (3, 32): "<f4"}[(code, bits)])
    return hdr, x


def slaney_mel_bank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(htk=False, norm='slaney') -- what create_mel_filter_bank ports."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    h2m = lambda f: np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-9) / min_log_hz) / logstep, f / f_sp)
    m2h = lambda m: np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)
    fft = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_f = m2h(np.linspace(h2m(np.float64(fmin)), h2m(np.float64(fmax)), n_mels + 2))
    fdiff, ramps = np.diff(mel_f), np.subtract.outer(mel_f, fft)
    W = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        W[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    return W * (2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels]))[:, None]


def stft_mag(y, offset=0, window="periodic"):
    """|STFT| with librosa's conventions (reflect pad n_fft/2, hann, hop 256); `offset` shifts the
    frame grid to the right by that many samples."""
    n = np.arange(N_FFT)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / (N_FFT if window == "periodic" else N_FFT - 1))
    yp = np.pad(y, N_FFT // 2, mode="reflect")[offset:]
    nfr = 1 + (len(yp) - N_FFT) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(nfr)[:, None]
    return np.abs(np.fft.rfft(yp[idx] * w, axis=1)).T  # (513, F)


def range_residual(mag, B, Bp, e):
    """relative distance of mag from clip(pinv(B) B mag^(1/e), 0)^e (always in the magnitude domain)."""
    R = np.clip(Bp @ (B @ mag ** (1.0 / e)), 0, None) ** e
    return float(np.linalg.norm(mag - R) / np.linalg.norm(mag))


def numpy_griffinlim(S, iters=30, momentum=0.99, seed=0):
    """librosa 0.9 griffinlim restated with numpy (calibration of the residual test only)."""
    n = np.arange(N_FFT)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / N_FFT)

    def istft(X):
        F = X.shape[1]
        fr = np.fft.irfft(X.T, axis=1) * w
        y, ws = np.zeros(N_FFT + HOP * (F - 1)), np.zeros(N_FFT + HOP * (F - 1))
        for f in range(F):
            y[f * HOP : f * HOP + N_FFT] += fr[f]
            ws[f * HOP : f * HOP + N_FFT] += w * w
        return (y / np.maximum(ws, 1e-10))[N_FFT // 2 : -N_FFT // 2]

    def stft(y):
        yp = np.pad(y, N_FFT // 2, mode="reflect")
        idx = np.arange(N_FFT)[None, :] + HOP * np.arange(1 + (len(yp) - N_FFT) // HOP)[:, None]
        return np.fft.rfft(yp[idx] * w, axis=1).T

    ang = np.exp(2j * np.pi * np.random.default_rng(seed).random(S.shape))
    tprev = 0
    for _ in range(iters):
        reb = stft(istft(S * ang))
        a = reb - (momentum / (1 + momentum)) * tprev
        tprev, ang = reb, a / (np.abs(a) + 1e-16)
    return istft(S * ang)


EXPONENTS = {"inverse (S = (pinv M)^(1/1.7), librosa mel_to_stft)": 1.0 / POWER, "none (S = pinv M)": 1.0, "direct (S = (pinv M)^1.7)": POWER}


def calibrate(mag, B, Bp):
    """vocode a mel of this very file under each exponent, then ask the residual test which it was."""
    table = {}
    for made_with, et in EXPONENTS.items():
        S = np.clip(Bp @ (B @ mag ** (1.0 / et)), 0, None) ** et
        m = stft_mag(numpy_griffinlim(S))
        table[made_with] = {k: range_residual(m, B, Bp, e) for k, e in EXPONENTS.items()}
    return table


def level_facts(q):
    """q: the int16 samples.  RMS as stored, and the f32 RMS implied before the cast."""
    v = q.astype(np.float64)
    a = np.abs(v)
    stored = np.sqrt(np.mean(v * v)) / 32767.0
    # truncation toward zero (Rust `as i16`): |s|*32767 uniform in [|q|, |q|+1) (q = 0: (-1, 1))
    trunc = np.sqrt(np.mean(np.where(a > 0, (a + 0.5) ** 2 + 1.0 / 12, 1.0 / 3))) / 32767.0
    # round to nearest: uniform in [q-1/2, q+1/2)
    rnd = np.sqrt(np.mean(v * v + 1.0 / 12)) / 32767.0
    return {
        "rms_stored": stored,
        "rms_before_cast_if_truncating": trunc,
        "rms_before_cast_if_rounding": rnd,
        "peak_stored": float(a.max() / 32767.0),
        "mean_stored": float(v.mean() / 32767.0),
        "crest_factor_db": float(20 * np.log10(a.max() / np.sqrt(np.mean(v * v)))),
        "zeros": int((q == 0).sum()),
        "plus_ones": int((q == 1).sum()),
        "minus_ones": int((q == -1).sum()),
    }


def analyse(name, B, Bp):
    hdr, x = read_wav(os.path.join(AUDIO_DIR, name))
    facts = {"header": hdr, "matches_WAV_SPEC": hdr == WAV_SPEC, "samples": int(len(x)), "samples_mod_256": int(len(x) % HOP),
             "samples_div_256": len(x) / HOP, "seconds": len(x) / hdr["sample_rate"]}
    y = x.astype(np.float64) / (32767.0 if x.dtype.kind == "i" else 1.0)
    spec = np.abs(np.fft.rfft(y)) ** 2
    f = np.fft.rfftfreq(len(y), 1.0 / hdr["sample_rate"])
    facts["energy_share_above_8kHz"] = float(spec[f > FMAX].sum() / spec.sum())
    facts["energy_share_above_8_2kHz"] = float(spec[f > 8200.0].sum() / spec.sum())
    if x.dtype.kind == "i":
        facts["level"] = level_facts(x)
    else:
        facts["level"] = {"rms_stored": float(np.sqrt(np.mean(y * y))), "peak_stored": float(np.abs(y).max())}
    mag = stft_mag(y)
    facts["frames_at_hop_256_centered"] = int(mag.shape[1])
    facts["mel_range_residual"] = {k: range_residual(mag, B, Bp, e) for k, e in EXPONENTS.items()}
    facts["mel_range_residual_symmetric_window"] = range_residual(stft_mag(y, 0, "symmetric"), B, Bp, 1.0)
    offs = list(range(0, HOP, 16))
    res = [range_residual(stft_mag(y, o), B, Bp, 1.0) for o in offs]
    facts["alignment"] = {"offsets": offs, "residual": res, "best_offset": offs[int(np.argmin(res))],
                          "contrast": float(max(res) / min(res))}
    if facts["matches_WAV_SPEC"]:
        facts["calibration"] = calibrate(mag, B, Bp)
        facts["calibration_verdict"] = "residual lowest at e = 1 for audio made with every exponent: power_mode undetermined"
    return facts


def main():
    if not os.path.isdir(AUDIO_DIR):
        sys.exit("needs the reference checkout at /root/reference (build container only)")
    B = slaney_mel_bank(SR, N_FFT, N_MELS, 0.0, FMAX)
    Bp = np.linalg.pinv(B)
    out = {"source": "slides/audio/*.wav of xd009642/xd-tts, analysed by tools/reference_audio_facts.py",
           "WAV_SPEC": WAV_SPEC, "files": {}}
    for name in sorted(os.listdir(AUDIO_DIR)):
        if name.endswith(".wav"):
            out["files"][name] = analyse(name, B, Bp)
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for name, f in out["files"].items():
        lv = f["level"]
        print("%-28s spec=%s n=%d (%.3f x256) rms=%.6f peak=%.4f >8k=%.1e best_off=%d" % (
            name, f["matches_WAV_SPEC"], f["samples"], f["samples_div_256"], lv["rms_stored"], lv["peak_stored"],
            f["energy_share_above_8kHz"], f["alignment"]["best_offset"]))
        if "rms_before_cast_if_truncating" in lv:
            print("   f32 rms before the cast: truncating %.7f, rounding %.7f" % (lv["rms_before_cast_if_truncating"], lv["rms_before_cast_if_rounding"]))
        print("   mel-range residuals:", {k.split(" ")[0]: round(v, 4) for k, v in f["mel_range_residual"].items()},
              "symmetric window:", round(f["mel_range_residual_symmetric_window"], 4), "alignment contrast %.3f" % f["alignment"]["contrast"])
        for made, row in f.get("calibration", {}).items():
            print("   calibration, made with %-8s ->" % made.split(" ")[0], {k.split(" ")[0]: round(v, 4) for k, v in row.items()})


if __name__ == "__main__":
    main()
