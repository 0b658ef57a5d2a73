"""Tacotron2::load(dir) end to end on files written by torch's own ONNX exporter: the mel of the
handle loaded from `encoder.onnx` / `decoder_iter.onnx` / `postnet.onnx` (src/tacotron2/mod.rs:246-259) against the ORACLE
on the same weights -- not against another HIP handle.  A wrong gate order, bias split, MatMul orientation or BatchNorm
reading in csrc/onnx_load.cpp would fail here."""
import numpy as np
import pytest

from conftest import rms, synth_ids

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("fuse_bn", [False, True])
def test_handle_loaded_from_a_torch_export_matches_the_oracle(pkg, orc, blob, tmp_path, fuse_bn):
    import nvidia_torch_export as nx

    rng = np.random.Generator(np.random.PCG64(21))
    b2 = blob.copy()
    T = {}
    for name, shape, off in pkg.tensor_table():
        a = b2[off: off + int(np.prod(shape))].reshape(shape)
        if ".bn." in name:  # non-trivial BatchNorm statistics (the seeded blob's are the identity)
            kind = name.rsplit(".", 1)[1]
            a[...] = {"weight": 0.75 + 0.5 * rng.random(shape), "bias": 0.1 * rng.standard_normal(shape),
                      "running_mean": 0.1 * rng.standard_normal(shape), "running_var": 0.5 + rng.random(shape)}[kind].astype(np.float32)
        T[name] = a
    nx.export_model_dir(str(tmp_path), T, fuse_bn=fuse_bn)
    m = pkg.Tacotron2.load(str(tmp_path))
    ids = synth_ids(43, seed=9)
    mel = m.infer(ids, opts=pkg.default_opts(fixed_steps=48, dropout_seed=6))
    ref = orc.infer_chunk(b2, ids, orc.default_opts(fixed_steps=48, dropout_seed=6))
    assert mel.shape == ref.shape == (80, 48)
    assert rms(mel, ref) <= 1e-5, rms(mel, ref)
    # the graph-by-graph hooks too: encoder and post-net of the loaded handle
    padded = np.zeros(100, dtype=np.int64)
    padded[:43] = ids
    mem, pm = m.encoder(padded)
    omem, opm = orc.encoder(b2, padded)
    assert np.abs(mem - omem).max() <= 1e-5 and np.abs(pm - opm).max() <= 1e-5
    m.close()
