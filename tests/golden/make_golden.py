#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (oracle/xdtts_oracle.c).

The reference holds no numeric fixtures for this path (parity unpinned), so
these vectors pin (a) the oracle against accidental change and (b) the HIP library against the same
numbers on the GPU box, where neither /root/reference nor this script's inputs are needed.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 20240327


def main():
    o32, o64 = oracle.Oracle("f32"), oracle.Oracle("f64")
    blob = o32.weights_synthetic(seed=SEED, rec_scale=1.0)

    # --- Tacotron2: the reference's phoneme KAT ids (mod.rs:470-483), first 12, 16 decoder steps
    ids = np.array([108, 119, 11, 88, 113, 108, 120, 11, 116, 73, 118, 7], dtype=np.int64)
    steps, dseed = 16, 5
    padded = np.zeros(100, dtype=np.int64)
    padded[: len(ids)] = ids
    mem, pm = o32.encoder(blob, padded)
    frames, gates = o32.run_decoder(blob, mem, pm, len(ids), o32.default_opts(fixed_steps=steps, dropout_seed=dseed))
    mel = o32.postnet(blob, frames)
    mel64 = o64.infer_chunk(blob, ids, o64.default_opts(fixed_steps=steps, dropout_seed=dseed))
    np.savez_compressed(
        os.path.join(HERE, "tacotron2_small.npz"),
        weight_seed=SEED, ids=ids, steps=steps, dropout_seed=dseed,
        memory_cols=mem[:, ::32].astype(np.float32), pmem_cols=pm[:, ::8].astype(np.float32),
        frames=frames.astype(np.float32), gates=gates.astype(np.float32), mel=mel.astype(np.float32), mel_f64=mel64,
        blob_checksum=np.float64(blob.astype(np.float64).sum()), blob_abs_checksum=np.float64(np.abs(blob.astype(np.float64)).sum()),
    )

    # --- Griffin-Lim: 48 frames of a two-tone + chirp signal, 30 iterations (mod.rs:456)
    F = 48
    t = np.arange(256 * (F - 1)) / 22050.0
    sig = 0.5 * np.sin(2 * np.pi * 440 * t) + 0.3 * np.sin(2 * np.pi * (1000 + 2000 * t) * t)
    spec = o64.stft(sig)
    S = np.hypot(spec[..., 0], spec[..., 1]).astype(np.float32)
    audio32 = o32.griffinlim(S, seed=3, iters=30)
    audio64 = o64.griffinlim(S, seed=3, iters=30)
    np.savez_compressed(os.path.join(HERE, "griffinlim_small.npz"), S=S, phase_seed=3, iters=30, audio=audio32.astype(np.float32), audio_f64=audio64)

    # --- mel filter bank + mel->linear
    B = o32.mel_filter_bank()
    rng = np.random.default_rng(7)
    melin = rng.uniform(-8.0, 0.5, size=(80, 6)).astype(np.float32)
    lin = o32.mel_to_linear(o32.pinv(B), melin, power=1.7)
    np.savez_compressed(os.path.join(HERE, "mel_basis.npz"), rows=B[[0, 1, 40, 79]], row_index=np.array([0, 1, 40, 79]), checksum=np.float64(B.astype(np.float64).sum()),
                        nnz=np.int64((B > 0).sum()), mel_in=melin, linear=lin.astype(np.float32))
    for f in sorted(os.listdir(HERE)):
        print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
