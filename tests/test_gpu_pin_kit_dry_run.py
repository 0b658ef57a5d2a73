"""The pin kit (tools/pin/) end to end on SYNTHETIC weights -- a dry run, not a pin:

  1. torch's own ONNX exporter writes the reference's model directory from NVIDIA-structured modules whose prenet dropout is live
     (tests/nvidia_torch_export.py: F.dropout(training=True) -> two `Dropout` nodes in decoder_iter.onnx);
  2. tools/pin/patch_decoder_iter.py turns the two random nodes into graph inputs (checked structurally: the patched file is
     read back with the repo's own protobuf reader);
  3. tools/pin/record_run.py records a run on the reference's known-answer ids with seeded masks -- here with the torch backend
     standing where onnxruntime stands in a real pin;
  4. tests/pin_compare.py loads the SAME directory through Tacotron2::load(dir) (mod.rs:242), feeds the SAME masks
     (dropout_mode 2) and compares encoder, decoder loop, per-step state, post-net and the whole infer chain at 1e-4.
Trained-like weights (tests/regimes.py), so the numbers are in the regime of a real checkpoint."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

from pin_compare import compare_run
from regimes import trained_like

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pin_kit_end_to_end_on_synthetic_weights(pkg, orc, tmp_path):
    import nvidia_torch_export as nte

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import onnx_to_xdtw

    blob = trained_like(orc, 7)
    T = {n: blob[off : off + numel].reshape(shape) for n, shape, off, numel in orc.tensor_table()}
    model_dir = str(tmp_path / "model")
    os.makedirs(model_dir)
    nte.export_model_dir(model_dir, T, fuse_bn=False)        # BatchNormalization nodes kept, as exporters of the reference's vintage wrote

    patch = _load(os.path.join(ROOT, "tools", "pin", "patch_decoder_iter.py"), "patch_decoder_iter")
    pinned = str(tmp_path / "decoder_iter.pinned.onnx")
    assert patch.main(["patch", os.path.join(model_dir, "decoder_iter.onnx"), pinned]) == 0
    report = json.load(open(pinned + ".json"))
    assert [r["op"] for r in report["random_nodes"]] == ["Dropout", "Dropout"] and not any(r["mask_output_used"] for r in report["random_nodes"])
    assert [(i["name"], i["kind"], i["shape"], i["ratio"]) for i in report["new_inputs"]] == [("dropout_scale_0", "scale", [1, 256], 0.5), ("dropout_scale_1", "scale", [1, 256], 0.5)]
    g0, g1 = onnx_to_xdtw.Graph(os.path.join(model_dir, "decoder_iter.onnx")), onnx_to_xdtw.Graph(pinned)
    ops0, ops1 = [n.op for n in g0.nodes], [n.op for n in g1.nodes]
    assert ops0.count("Dropout") == 2 and ops1.count("Dropout") == 0 and ops1.count("Mul") == ops0.count("Mul") + 2 and len(ops0) == len(ops1)
    muls = [n for n in g1.nodes if n.op == "Mul" and n.inputs[1].startswith("dropout_scale_")]
    drops = [n for n in g0.nodes if n.op == "Dropout"]
    assert [(m.inputs[0], m.outputs[0]) for m in muls] == [(d.inputs[0], d.outputs[0]) for d in drops]   # same wires in and out
    # the weights survive the rewrite bit for bit
    import shutil

    twin = str(tmp_path / "model_pinned")
    os.makedirs(twin)
    for f in ("encoder.onnx", "postnet.onnx"):
        shutil.copy(os.path.join(model_dir, f), os.path.join(twin, f))
    shutil.copy(pinned, os.path.join(twin, "decoder_iter.onnx"))
    a, b = onnx_to_xdtw.collect(model_dir), onnx_to_xdtw.collect(twin)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    # a file without random nodes reports none
    assert patch.patch(open(os.path.join(model_dir, "postnet.onnx"), "rb").read())[1] == {"random_nodes": [], "new_inputs": []}

    rec_mod = _load(os.path.join(ROOT, "tools", "pin", "record_run.py"), "record_run")
    npz = str(tmp_path / "run.npz")
    assert rec_mod.main([model_dir, pinned, npz, "--steps", "40", "--backend", "torch"]) == 0
    rec = dict(np.load(npz, allow_pickle=False))
    assert rec["frames"].shape == (40, 80) and rec["keep_masks"].shape == (40, 2, 256) and "dry run" in str(rec["recorded_with"])
    assert list(rec["ids"][:28]) == rec_mod.KAT_IDS and int(rec["n_valid"]) == 28
    assert float(np.abs(rec["frames"]).max()) > 5.0          # trained-like weights: log-mel magnitudes, not the +-0.1 of the plain draw

    margins = compare_run(pkg, model_dir, rec, tol=1e-4)
    assert margins["frames_rms"] > 0                          # two implementations really were compared
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "pin_kit_dry_run.json"), "w") as fh:
        json.dump(margins, fh, indent=1, sort_keys=True)
    # ... and masks that differ from the recorded ones do NOT reproduce the run (the comparison has teeth)
    rec2 = dict(rec, keep_masks=1 - rec["keep_masks"])
    with pytest.raises(AssertionError):
        compare_run(pkg, model_dir, rec2, tol=1e-4)
