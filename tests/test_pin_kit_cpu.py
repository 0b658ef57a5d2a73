"""The CPU half of the pin kit (tools/pin/) without a GPU: torch's ONNX exporter writes a model directory
whose decoder_iter.onnx draws its prenet dropout at run time, tools/pin/patch_decoder_iter.py turns the two random nodes into
graph inputs, tools/pin/record_run.py records a run with seeded masks (torch backend: tests/pin_torch_backend.py) -- and the ORACLE,
given the same masks (dropout_mode 2), reproduces that run: torch's own LSTM / conv / softmax kernels against the C restatement,
in the trained-like regime, with the dropout live."""
import importlib.util
import json
import os
import sys

import numpy as np

from regimes import trained_like

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_patch_record_and_oracle(orc, tmp_path):
    import nvidia_torch_export as nte

    blob = trained_like(orc, 7)
    T = {n: blob[off : off + numel].reshape(shape) for n, shape, off, numel in orc.tensor_table()}
    model_dir = str(tmp_path / "model")
    os.makedirs(model_dir)
    nte.export_model_dir(model_dir, T)
    patch = _load(os.path.join(ROOT, "tools", "pin", "patch_decoder_iter.py"), "patch_decoder_iter")
    pinned = str(tmp_path / "decoder_iter.pinned.onnx")
    assert patch.main(["patch", os.path.join(model_dir, "decoder_iter.onnx"), pinned]) == 0
    report = json.load(open(pinned + ".json"))
    assert [i["name"] for i in report["new_inputs"]] == ["dropout_scale_0", "dropout_scale_1"]
    # a git-LFS pointer (what the build container holds at /root/reference/models/tacotron2) is refused with a message
    ptr = tmp_path / "pointer.onnx"
    ptr.write_bytes(b"version https://git-lfs.github.com/spec/v1\noid sha256:0\nsize 1\n")
    assert patch.main(["patch", str(ptr), str(tmp_path / "x.onnx")]) == 1
    # the RandomUniformLike form (torch.bernoulli: NVIDIA's inference prenet) on a hand-written graph: node removed, output fed
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import onnx_writer as ow

    g = ow.model([ow.node("Relu", ["x"], ["r"]), ow.node("RandomUniformLike", ["r"], ["u"], name="rnd"), ow.node("Less", ["u", "p"], ["m"]),
                  ow.node("Mul", ["r", "m"], ["y"])], [], inputs=["x"], outputs=["y"])
    out, rep = patch.patch(g)
    assert [r["op"] for r in rep["random_nodes"]] == ["RandomUniformLike"] and rep["new_inputs"][0]["name"] == "dropout_uniform_0" and rep["new_inputs"][0]["kind"] == "uniform"
    nodes = [patch.parse_node(v) for f, w, v in patch.fields([v for f, w, v in patch.fields(out) if f == 7][0]) if f == 1]
    assert [n["op"] for n in nodes] == ["Relu", "Identity", "Less", "Mul"] and nodes[1]["inputs"] == ["dropout_uniform_0"] and nodes[1]["outputs"] == ["u"]

    rec_mod = _load(os.path.join(ROOT, "tools", "pin", "record_run.py"), "record_run")
    npz = str(tmp_path / "run.npz")
    assert rec_mod.main([model_dir, pinned, npz, "--steps", "32", "--backend", "torch"]) == 0
    r = np.load(npz)
    mem, pm = orc.encoder(blob, r["ids"])
    assert np.sqrt(np.mean((mem - r["memory"]) ** 2)) <= 1e-6 and np.sqrt(np.mean((pm - r["processed_memory"]) ** 2)) <= 1e-6
    fr, gt = orc.run_decoder(blob, r["memory"], r["processed_memory"], 28, orc.default_opts(fixed_steps=32, masks=r["keep_masks"]))
    assert float(np.abs(r["frames"]).max()) > 5.0
    assert np.sqrt(np.mean((fr - r["frames"]) ** 2)) <= 1e-5 and np.abs(gt - r["gates"]).max() <= 1e-5
    assert np.sqrt(np.mean((orc.postnet(blob, r["frames"]) - r["mel_postnet"]) ** 2)) <= 1e-5
    # other masks give another run (the recorded masks matter)
    fr2, _ = orc.run_decoder(blob, r["memory"], r["processed_memory"], 28, orc.default_opts(fixed_steps=32, masks=1 - r["keep_masks"]))
    assert np.sqrt(np.mean((fr2 - r["frames"]) ** 2)) > 1e-3
