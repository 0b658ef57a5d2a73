"""The oracle against the committed golden fixtures (tests/golden/, made by make_golden.py)."""
import os

import numpy as np

from conftest import rms

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_synthetic_weights_are_reproducible(blob):
    g = np.load(os.path.join(G, "tacotron2_small.npz"))
    assert blob.size == 28_200_481
    assert float(blob.astype(np.float64).sum()) == float(g["blob_checksum"])
    assert float(np.abs(blob.astype(np.float64)).sum()) == float(g["blob_abs_checksum"])


def test_oracle_tacotron2_matches_golden(orc, orc64, blob):
    g = np.load(os.path.join(G, "tacotron2_small.npz"))
    ids, steps, dseed = g["ids"], int(g["steps"]), int(g["dropout_seed"])
    padded = np.zeros(100, dtype=np.int64)
    padded[: len(ids)] = ids
    mem, pm = orc.encoder(blob, padded)
    assert np.abs(mem[:, ::32] - g["memory_cols"]).max() < 1e-6
    assert np.abs(pm[:, ::8] - g["pmem_cols"]).max() < 1e-6
    frames, gates = orc.run_decoder(blob, mem, pm, len(ids), orc.default_opts(fixed_steps=steps, dropout_seed=dseed))
    assert np.abs(frames - g["frames"]).max() < 1e-6 and np.abs(gates - g["gates"]).max() < 1e-6
    assert rms(orc.postnet(blob, frames), g["mel"]) < 1e-6
    assert rms(orc64.infer_chunk(blob, ids, orc64.default_opts(fixed_steps=steps, dropout_seed=dseed)), g["mel_f64"]) < 1e-12


def test_oracle_griffinlim_matches_golden(orc, orc64):
    g = np.load(os.path.join(G, "griffinlim_small.npz"))
    a = orc.griffinlim(g["S"], seed=int(g["phase_seed"]), iters=int(g["iters"]))
    assert rms(a, g["audio"]) < 1e-6
    assert rms(orc64.griffinlim(g["S"], seed=int(g["phase_seed"]), iters=int(g["iters"])), g["audio_f64"]) < 1e-12
    assert rms(g["audio"], g["audio_f64"]) < 1e-4  # fp32 path stays within the north-star tolerance of fp64


def test_oracle_mel_basis_matches_golden(orc):
    g = np.load(os.path.join(G, "mel_basis.npz"))
    B = orc.mel_filter_bank()
    assert np.array_equal(B[g["row_index"]], g["rows"])
    assert float(B.astype(np.float64).sum()) == float(g["checksum"]) and int((B > 0).sum()) == int(g["nnz"])
    lin = orc.mel_to_linear(orc.pinv(B), g["mel_in"], power=1.7)
    assert np.abs(lin - g["linear"]).max() <= 1e-5 * max(1.0, float(np.abs(g["linear"]).max()))
