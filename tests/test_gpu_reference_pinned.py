"""The pin: skipped until someone who holds the reference's artefacts has run tools/pin/ --

    tests/golden/reference_run.npz            one recorded run of encoder.onnx / decoder_iter.onnx (dropout masks supplied) /
                                              postnet.onnx under onnxruntime on the reference's own known-answer ids
                                              (tools/pin/patch_decoder_iter.py + tools/pin/record_run.py)
    tests/golden/reference_mel_to_linear.npy  the griffin-lim crate's mel -> linear stage on a fixed mel (tools/pin/dump_mel_to_linear.rs)
    $XDTTS_REFERENCE_MODEL_DIR                the directory with the three real .onnx files (default: /root/reference/models/tacotron2,
                                              which holds git-LFS pointers in the build container)

With them present this is the test that turns "parity unpinned" (oracle/xdtts_oracle.h) into a pin: Tacotron2::load(dir)
(mod.rs:242) + dropout_mode 2 against the recorded numbers at the north star's 1e-4, and the vocoder's first stage against the
crate's own under the option set that matches."""
import json
import os

import numpy as np
import pytest

from pin_compare import compare_run

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
MODEL_DIR = os.environ.get("XDTTS_REFERENCE_MODEL_DIR", "/root/reference/models/tacotron2")


def _real_models():
    try:
        return all(os.path.getsize(os.path.join(MODEL_DIR, f + ".onnx")) > 1 << 20 for f in ("encoder", "decoder_iter", "postnet"))
    except OSError:
        return False


@pytest.mark.skipif(not (os.path.exists(os.path.join(G, "reference_run.npz")) and _real_models()),
                    reason="no recorded reference run / no real model files: see INTEGRATION.md section 5 (tools/pin/)")
def test_recorded_reference_run(pkg):
    rec = dict(np.load(os.path.join(G, "reference_run.npz"), allow_pickle=False))
    assert "onnxruntime" in str(rec["recorded_with"]), "reference_run.npz must come from onnxruntime on the real artefacts (record_run.py --backend ort)"
    margins = compare_run(pkg, MODEL_DIR, rec, tol=1e-4)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "reference_pinned.json"), "w") as fh:
        json.dump(margins, fh, indent=1)


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "reference_mel_to_linear.npy")),
                    reason="no crate-side mel -> linear dump: see INTEGRATION.md section 5 (tools/pin/dump_mel_to_linear.rs)")
def test_crate_mel_to_linear(pkg):
    """The crate's deterministic first stage on the fixed mel of tools/pin/fixed_mel.py: exactly one setting of
    (mel_decompress, power_mode, nnls_iters) of xdtts_griffinlim_opts must reproduce it; the winner is what the defaults should be."""
    import ctypes
    import importlib.util

    spec = importlib.util.spec_from_file_location("fixed_mel", os.path.join(ROOT, "tools", "pin", "fixed_mel.py"))
    fixed = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fixed)
    mel = fixed.fixed_mel()
    want = np.load(os.path.join(G, "reference_mel_to_linear.npy")).astype(np.float32)
    if want.shape == (mel.shape[1], 513):
        want = want.T.copy()
    assert want.shape == (513, mel.shape[1]), want.shape
    voc = pkg.create_griffin_lim()
    hits = []
    for dec in (0, 1, 2):
        for pm in (0, 1, 2):
            for it in (0, 32, 256):
                voc.set_opts(mel_decompress=dec, power_mode=pm, nnls_iters=it)
                S = voc.mel_to_linear(mel)
                rel = float(np.sqrt(np.mean((S.astype(np.float64) - want) ** 2)) / max(1e-30, np.sqrt(np.mean(want.astype(np.float64) ** 2))))
                if rel <= 1e-4:
                    hits.append((dec, pm, it, rel))
    voc.close()
    assert hits, "no setting of (mel_decompress, power_mode, nnls_iters) reproduces the crate's mel -> linear stage"
    d = pkg.GriffinLimOpts()
    pkg.lib.xdtts_griffinlim_opts_default(ctypes.byref(d))
    assert any((d.mel_decompress, d.power_mode) == h[:2] for h in hits), ("the defaults are not the crate's convention", hits)
