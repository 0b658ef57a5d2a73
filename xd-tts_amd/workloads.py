"""Synthetic workloads of BASELINE.json's configs (definitions: BASELINE.md section 3, SURVEY.md
section 8(d)).  Shared by bench.py, tools/ and the parity tests so they all measure and check the
same inputs.  Pure numpy + the host-side chunker; nothing here touches the GPU."""
import numpy as np

WEIGHT_SEED = 20240327
T_ENC = 100                 # encoder window, src/tacotron2/mod.rs:363,369-371,399
FRAMES_PER_ID = 800 / 120.0  # config 2: 120 ids -> 800 frames (gate disabled)
FRAMES_PER_ID_BATCH = 6.67   # configs 3/4: fixed_steps_i = round(6.67 * len_i)


def synth_ids(n, seed=1):
    """ARPAbet ids 64..147 (PCG64), space (11) at every 6th position, '.' (7) last."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = 64 + rng.integers(0, 84, size=n)
    ids[5::6] = 11
    ids[-1] = 7
    return ids.astype(np.int64)


def lround(x):
    return int(np.floor(x + 0.5))


def chunk_utterance(pkg, ids, window=T_ENC):
    """find_splits + the trailing split of src/tacotron2/mod.rs:399,412-414 -> list of chunks."""
    sp = [int(s) for s in pkg.find_splits(ids, window)]
    if not sp or sp[-1] != len(ids):
        sp.append(len(ids))
    out, a = [], 0
    for e in sp:
        if e > a:
            out.append(ids[a:e])
            a = e
    return out


def config2(pkg):
    """One 120-id utterance -> chunks 95 + 25 -> 633 + 167 frames."""
    ids = synth_ids(120)
    chunks = chunk_utterance(pkg, ids)
    steps = [lround(FRAMES_PER_ID * len(c)) for c in chunks]
    return ids, chunks, steps


def batch_utterances(pkg, seed=2, n_utt=32):
    """configs[2]: `n_utt` utterances of 40..200 ids (PCG64(seed)), chunked at the 100-id window like
    the reference; utterance u uses id seed 100 + u + 1000*(seed-2).  Returns (utterances, chunks,
    steps, owner) where owner[c] is the utterance index of chunk c."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = rng.integers(40, 201, size=n_utt)
    utts, chunks, steps, owner = [], [], [], []
    for u, n in enumerate(lens):
        ids = synth_ids(int(n), seed=100 + u + 1000 * (seed - 2))
        utts.append(ids)
        for c in chunk_utterance(pkg, ids):
            chunks.append(c)
            steps.append(lround(FRAMES_PER_ID_BATCH * len(c)))
            owner.append(u)
    return utts, chunks, steps, owner


def config4(pkg, n_batches=8):
    """configs[3]: 256 utterances = 8 config-3 batches (seeds 2..9)."""
    utts = []
    for s in range(2, 2 + n_batches):
        utts.extend(batch_utterances(pkg, seed=s)[0])
    return utts


def chirps(n):
    """config 5 signal: five linear chirps 100 Hz - 7 kHz plus white noise sigma 0.01."""
    t = np.arange(n) / 22050.0
    rng = np.random.default_rng(3)
    y = sum(0.15 * np.sin(2 * np.pi * (f0 + 0.5 * (f1 - f0) * t / t[-1]) * t) for f0, f1 in ((100, 900), (400, 2500), (1200, 4000), (3000, 5500), (5000, 7000)))
    return (y + 0.01 * rng.standard_normal(n)).astype(np.float32)


def chirp_magnitude(F, n_fft=1024, hop=256):
    """config 5 input: S = |STFT| of the chirp signal, (n_fft/2+1, F) fp32 -- centred frames, reflect padding,
    periodic hann, the vocoder's own analysis parameters (numpy; workload generation only)."""
    y = chirps(hop * (F - 1)).astype(np.float64)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)
    idx = np.arange(n_fft)[None, :] + hop * np.arange(F)[:, None]
    return np.abs(np.fft.rfft(yp[idx] * win[None, :], axis=1)).T.astype(np.float32).copy()
