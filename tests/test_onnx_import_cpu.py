"""tools/onnx_to_xdtw.py (the reference's on-disk weight format): synthetic
ONNX graphs laid out the two ways torch.onnx.export writes this model must convert to the exact
canonical tensors -- LSTM gate re-order, MatMul/Gemm orientation, conv order, BN statistics."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import onnx_to_xdtw as conv  # noqa: E402
import onnx_writer  # noqa: E402


def random_tensors(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    T = {}
    for name, shape in conv.tensor_table():
        a = rng.standard_normal(shape).astype(np.float32)
        if name.endswith("running_var"):
            a = np.abs(a) + 0.5
        T[name] = a
    return T


def read_container(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"XDTW0001"
    (n,) = struct.unpack_from("<I", raw, 8)
    out, data0 = {}, 12 + 96 * n
    for i in range(n):
        name, ndim, d0, d1, d2, off, numel = struct.unpack_from("<64sI3IQQ", raw, 12 + 96 * i)
        shape = (d0, d1, d2)[:ndim]
        out[name.rstrip(b"\0").decode()] = np.frombuffer(raw, "<f4", numel, data0 + 4 * off).reshape(shape)
    return out


@pytest.mark.parametrize("style", ["named", "folded"])
def test_synthetic_export_round_trips(tmp_path, style):
    T = random_tensors(3 if style == "named" else 4)
    onnx_writer.write_models(str(tmp_path), T, style)
    path, total = conv.write_container(str(tmp_path), conv.collect(str(tmp_path)))
    assert total == 28200481
    got = read_container(path)
    assert list(got) == [n for n, _ in conv.tensor_table()]
    for name, _shape in conv.tensor_table():
        if style == "folded" and ".bn." in name:  # the exporter folded conv+BN: identity statistics
            want = {"weight": 1.0, "bias": 0.0, "running_mean": 0.0, "running_var": np.float32(1.0 - 1e-5)}[name.rsplit(".", 1)[1]]
            assert np.all(got[name] == np.float32(want)), name
        else:
            assert np.array_equal(got[name], T[name]), name


def test_lstm_gate_reorder_is_iofc_to_ifgo():
    H = 3
    a = np.arange(4 * H, dtype=np.float32).reshape(4 * H, 1)        # PyTorch rows i,f,g,o
    onnx = onnx_writer.onnx_gates(a, H)                               # i,o,f,c
    assert onnx.ravel().tolist() == [0, 1, 2, 9, 10, 11, 3, 4, 5, 6, 7, 8]
    assert np.array_equal(conv._pt_gates(onnx, H), a)


def test_lfs_pointer_and_missing_pieces_are_reported(tmp_path):
    for f in ("encoder", "decoder_iter", "postnet"):
        open(str(tmp_path / (f + ".onnx")), "wb").write(b"version https://git-lfs.github.com/spec/v1\noid sha256:00\nsize 1\n")
    with pytest.raises(ValueError, match="git-LFS pointer"):
        conv.collect(str(tmp_path))
    T = random_tensors(5)
    onnx_writer.write_models(str(tmp_path), T, "folded")
    open(str(tmp_path / "postnet.onnx"), "wb").write(onnx_writer.model([], []))
    with pytest.raises(ValueError, match="expected 5 conv layers"):
        conv.collect(str(tmp_path))


def test_reference_checkout_holds_pointers_only():
    """Why the converter is unverified against the real graphs (documented in its header)."""
    ref = "/root/reference/models/tacotron2/encoder.onnx"
    if not os.path.exists(ref):
        pytest.skip("reference checkout not present on this box")
    assert os.path.getsize(ref) < 1024 and open(ref, "rb").read(24).startswith(b"version https://git-lfs")


# ---- the same import inside the library: Tacotron2::load(dir) on the reference's model directory ----

@pytest.mark.parametrize("style", ["named", "folded"])
def test_library_reads_the_onnx_model_dir(pkg, tmp_path, style):
    """xdtts_model_dir_read (csrc/onnx_load.cpp, the host half of xdtts_tacotron2_load) on the three graphs."""
    T = random_tensors(5 if style == "named" else 6)
    onnx_writer.write_models(str(tmp_path), T, style)
    got = pkg.read_model_dir(str(tmp_path))
    assert list(got) == [n for n, _ in conv.tensor_table()]
    for name, _shape in conv.tensor_table():
        if style == "folded" and ".bn." in name:
            want = {"weight": 1.0, "bias": 0.0, "running_mean": 0.0, "running_var": np.float32(1.0 - 1e-5)}[name.rsplit(".", 1)[1]]
            assert np.all(got[name] == np.float32(want)), name
        else:
            assert np.array_equal(got[name], T[name]), name
    # identical to the offline converter's container, and the container wins when both are present
    conv.write_container(str(tmp_path), conv.collect(str(tmp_path)))
    again = pkg.read_model_dir(str(tmp_path))
    assert all(np.array_equal(again[n], got[n]) for n in got)


def test_library_reports_lfs_pointers_and_broken_files(pkg, tmp_path):
    """The reference checkout ships git-LFS pointers (models/tacotron2/*.onnx are 133-byte text files): the
    load must say so; truncated or garbage files must fail cleanly (XDTTS_ERR_IO), never crash."""
    with pytest.raises(pkg.XdttsError) as e:
        pkg.read_model_dir(str(tmp_path))                       # empty directory
    assert e.value.status == pkg.XDTTS_ERR_IO and "encoder.onnx" in str(e.value)
    for f in ("encoder", "decoder_iter", "postnet"):
        (tmp_path / (f + ".onnx")).write_bytes(b"version https://git-lfs.github.com/spec/v1\noid sha256:c16355ad\nsize 22641034\n")
    with pytest.raises(pkg.XdttsError) as e:
        pkg.read_model_dir(str(tmp_path))
    assert e.value.status == pkg.XDTTS_ERR_IO and "git lfs pull" in str(e.value)
    T = random_tensors(7)
    onnx_writer.write_models(str(tmp_path), T, "named")
    good = (tmp_path / "decoder_iter.onnx").read_bytes()
    rng = np.random.Generator(np.random.PCG64(1))
    for cut in (10, 1000, len(good) // 2, len(good) - 5):
        (tmp_path / "decoder_iter.onnx").write_bytes(good[:cut])
        with pytest.raises(pkg.XdttsError) as e:
            pkg.read_model_dir(str(tmp_path))
        assert e.value.status == pkg.XDTTS_ERR_IO
    (tmp_path / "decoder_iter.onnx").write_bytes(bytes(rng.integers(0, 256, 4096, dtype=np.uint8)))
    with pytest.raises(pkg.XdttsError) as e:
        pkg.read_model_dir(str(tmp_path))
    assert e.value.status == pkg.XDTTS_ERR_IO
    (tmp_path / "decoder_iter.onnx").write_bytes(good)
    bad = dict(T)
    bad["prenet.0.weight"] = np.full_like(T["prenet.0.weight"], np.nan)   # a NaN weight is refused, not loaded
    onnx_writer.write_models(str(tmp_path), bad, "named")
    with pytest.raises(pkg.XdttsError) as e:
        pkg.read_model_dir(str(tmp_path))
    assert e.value.status == pkg.XDTTS_ERR_IO and "non-finite" in str(e.value)
