"""Tacotron2::load(dir) (src/tacotron2/mod.rs:242-267) pinned against an INDEPENDENT writer (VERDICT round 2, item 2;
SURVEY 8(f) rank 1): the three graphs are written by torch.onnx.export (TorchScript exporter, the family NVIDIA's
export_tacotron2_onnx.py used) from NVIDIA-structured torch modules (tests/nvidia_torch_export.py), and
xdtts_model_dir_read must recover every canonical tensor -- gate order, [Wb|Rb] split, MatMul/Gemm orientation, conv order
and BatchNorm handling are the exporter's conventions here, not this repo's.  The same modules' own forward passes
(torch's conv1d / batch_norm / LSTM kernels) pin the C oracle once more, end to end per graph."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import nvidia_torch_export as nx  # noqa: E402

# graph I/O names the reference binds -- src/tacotron2/mod.rs:284-296 (inputs!), :306-307 and :332-339 (outputs by name), :349
REF_DEC_INPUTS = ["decoder_input", "attention_hidden", "attention_cell", "decoder_hidden", "decoder_cell", "attention_weights",
                  "attention_weights_cum", "attention_context", "memory", "processed_memory", "mask"]
REF_DEC_OUTPUTS = ["decoder_output", "gate_prediction", "out_attention_hidden", "out_attention_cell", "out_decoder_hidden",
                   "out_decoder_cell", "out_attention_weights", "out_attention_weights_cum", "out_attention_context"]
# `size` lines of the git-LFS pointers models/tacotron2/{encoder,decoder_iter,postnet}.onnx
LFS_SIZES = {"encoder.onnx": 22641034, "decoder_iter.onnx": 72766349, "postnet.onnx": 17414016}


def random_tensors(pkg, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    T = {}
    for name, shape, _off in pkg.tensor_table():
        a = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        if name.endswith("running_var"):
            a = np.abs(a) + 0.5
        if name.endswith("bn.weight"):
            a = a + 1.0
        T[name] = a
    return T


@pytest.fixture(scope="module")
def exported(pkg, tmp_path_factory):
    out = {}
    for fuse in (False, True):
        d = tmp_path_factory.mktemp("fused" if fuse else "unfused")
        T = random_tensors(pkg, 11 if fuse else 12)
        nx.export_model_dir(str(d), T, fuse_bn=fuse)
        out[fuse] = (str(d), T)
    return out


def test_unfused_export_recovers_every_tensor_bit_exactly(pkg, exported):
    d, T = exported[False]
    got = pkg.read_model_dir(d)
    assert list(got) == [n for n, _s, _o in pkg.tensor_table()] and len(got) == 76
    for name in got:
        assert np.array_equal(got[name], T[name]), name


def test_unfused_export_has_the_size_of_the_reference_files(exported):
    """The exporter's files for this architecture are within 0.01 % of the sizes in the reference's LFS pointers: the layer
    inventory, tensor shapes and the un-fused BatchNorm form are those of the real artefacts."""
    import os

    d, _T = exported[False]
    for f, want in LFS_SIZES.items():
        size = os.path.getsize(os.path.join(d, f))
        assert abs(size - want) <= 2048, (f, size, want)


@pytest.mark.parametrize("opset", [9, 11, 12, 17])
def test_exports_of_other_opsets_load_bit_exactly(pkg, tmp_path, opset):
    """The exporter's graph details change with the opset (Squeeze / Unsqueeze axes as attributes or inputs, Dropout's form, the
    LSTM node's layout attribute): opsets 9 to 17 -- the range an export of the reference's vintage can have -- all load, every
    tensor bit for bit; the decoder_iter.onnx of an opset-12 export is 72 766 097 bytes, the reference's 72 766 349."""
    import os

    T = random_tensors(pkg, 20 + opset)
    nx.export_model_dir(str(tmp_path), T, opset=opset, fuse_bn=False)
    got = pkg.read_model_dir(str(tmp_path))
    for name in got:
        assert np.array_equal(got[name], T[name]), (opset, name)
    for f, want in LFS_SIZES.items():
        assert abs(os.path.getsize(os.path.join(str(tmp_path), f)) - want) <= 2048, (opset, f)


def test_fused_export_recovers_the_folded_convolutions(pkg, exported):
    d, T = exported[True]
    got = pkg.read_model_dir(d)
    ident = {"weight": 1.0, "bias": 0.0, "running_mean": 0.0, "running_var": np.float32(1.0 - 1e-5)}
    for name in got:
        if ".bn." in name:  # the exporter folded BN into the conv: the library writes identity statistics
            assert np.all(got[name] == np.float32(ident[name.rsplit(".", 1)[1]])), name
        elif name.endswith(".conv.weight") or name.endswith(".conv.bias"):
            w, b = nx.fused_conv_bn(T, name.rsplit(".conv.", 1)[0])
            want = w if name.endswith("weight") else b
            # within 2 ulp: the fold's own rounding (the exporter's pass multiplies in a different association)
            ulps = np.abs(got[name] - want) / np.spacing(np.maximum(np.abs(want), np.float32(1e-30)))
            assert float(ulps.max()) <= 2.0, (name, float(ulps.max()))
        else:
            assert np.array_equal(got[name], T[name]), name


def test_graph_io_names_are_the_ones_the_reference_binds(pkg, exported):
    d, _T = exported[False]
    io = pkg.describe_model_dir(d)
    assert io["decoder_iter.onnx"] == (REF_DEC_INPUTS, REF_DEC_OUTPUTS)
    assert io["postnet.onnx"][1] == ["mel_outputs_postnet"] and len(io["postnet.onnx"][0]) == 1
    assert len(io["encoder.onnx"][0]) == 2 and len(io["encoder.onnx"][1]) == 3


def test_a_graph_the_reference_could_not_bind_is_refused(pkg, tmp_path):
    T = random_tensors(pkg, 13)
    bad = list(REF_DEC_OUTPUTS)
    bad[2] = "attention_hidden_out"   # mod.rs:333 asks for "out_attention_hidden"
    nx.export_model_dir(str(tmp_path), T, dec_outputs=bad)
    with pytest.raises(pkg.XdttsError) as e:
        pkg.read_model_dir(str(tmp_path))
    assert e.value.status == pkg.XDTTS_ERR_IO and "out_attention_hidden" in str(e.value)
    blob = np.empty(pkg.lib.xdtts_tensor_total(), dtype=np.float32)
    assert pkg.lib.xdtts_model_dir_read(str(tmp_path).encode(), blob.ctypes.data_as(C.c_void_p), blob.size) == pkg.XDTTS_ERR_IO


def _blob_from(pkg, T):
    blob = np.zeros(pkg.lib.xdtts_tensor_total(), dtype=np.float32)
    for name, shape, off in pkg.tensor_table():
        blob[off: off + int(np.prod(shape))] = T[name].ravel()
    return blob


def test_the_exporting_modules_pin_the_oracle(pkg, orc64, exported):
    """The modules whose exports have the reference's file sizes, run by torch itself in float64 (its own conv1d,
    batch_norm and LSTM kernels), against the float64 build of the C oracle on the same weights: encoder, one decoder_iter
    call (dropout off on both sides), post-net -- agreement to 1e-9, i.e. the same function up to summation order."""
    orc = orc64
    _d, T = exported[False]
    enc, dec, post = [m.double() for m in nx.build_modules(T)]
    blob = _blob_from(pkg, T)
    from conftest import synth_ids

    ids = np.zeros(100, dtype=np.int64)
    ids[:37] = synth_ids(37, seed=3)
    with torch.no_grad():
        mem, pmem, _ = enc(torch.from_numpy(ids)[None], torch.tensor([100]))
    omem, opm = orc.encoder(blob, ids)
    assert np.abs(mem[0].numpy() - omem).max() <= 1e-9 and np.abs(pmem[0].numpy() - opm).max() <= 1e-9
    # one decoder step from a non-trivial state
    rng = np.random.Generator(np.random.PCG64(2))
    st = orc.new_state()
    vals = {}
    for f, n in (("att_h", 1024), ("att_c", 1024), ("dec_h", 1024), ("dec_c", 1024), ("ctx", 512), ("dec_in", 80)):
        vals[f] = rng.standard_normal(n) * 0.3
    aw = rng.random(100)
    aw[37:] = 0
    aw /= aw.sum()
    vals["aw"], vals["awc"] = aw, aw * 3
    for f, v in vals.items():
        getattr(st, f)[: len(v)] = v.tolist()
    import torch.nn.functional as F

    real_dropout = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x   # dropout off (the oracle's dropout_mode 0)
    try:
        mask = torch.zeros(1, 100, dtype=torch.bool)
        mask[0, 37:] = True
        t = lambda v: torch.from_numpy(np.asarray(v, dtype=np.float64))[None]
        with torch.no_grad():
            out = dec(t(vals["dec_in"]), t(vals["att_h"]), t(vals["att_c"]), t(vals["dec_h"]), t(vals["dec_c"]), t(vals["aw"]), t(vals["awc"]),
                      t(vals["ctx"]), t(omem), t(opm), mask)
    finally:
        F.dropout = real_dropout
    mel, gate = orc.decoder_step(blob, omem, opm, 37, st, orc.default_opts(dropout_mode=0), 0)
    got = [o[0].numpy() for o in out]
    want = [mel, np.array([gate]), np.array(st.att_h), np.array(st.att_c), np.array(st.dec_h), np.array(st.dec_c),
            np.array(st.aw)[:100], np.array(st.awc)[:100], np.array(st.ctx)]
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.abs(g - np.asarray(w, dtype=np.float64)).max() <= 1e-9, (nx.DEC_OUTPUTS[i], float(np.abs(g - w).max()))
    frames = rng.standard_normal((40, 80)) * 0.5
    with torch.no_grad():
        pm = post(torch.from_numpy(frames.T.copy())[None])[0].numpy()
    assert np.abs(pm - orc.postnet(blob, frames)).max() <= 1e-9
