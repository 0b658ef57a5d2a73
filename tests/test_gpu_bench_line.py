"""The driver's contract for `python bench.py` (one GPU, default flags but fewer steps): ONE JSON line on stdout with the
headline metric of BASELINE.json configs[1], the `roofline` object of the dominant kernel and the `cpu_baseline` object
(the oracle timed on a bounded sample) -- checked field by field, so that a change to bench.py that breaks the line is caught
here and not at round end."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    with open(os.path.join(ROOT, "BASELINE.json")) as fh:
        base = json.load(fh)
    assert "mel-frames/s" in out["metric"] and "mel-frames/s" in base["metric"]
    assert out["unit"] == "mel-frames/s" and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert (out["n_gpus"], out["steps"], out["warmup"]) == (1, 3, 1)
    assert out["value"] > 50_000 and abs(out["value"] * out["ms_per_step"] * 1e-3 - 800) < 1.0   # 800 frames per step (utterance)
    assert out["vs_baseline"] is None and base["published"] == {}                                 # no published number for this metric
    assert out["dtype"] == "f32" and out["data"].startswith("synthetic")
    assert "configs[1]" in out["config"]["workload"] and "model" not in out["config"]
    rl = out["roofline"]
    assert rl["bound"] in ("hbm", "mfma") and rl["unit"] in ("GB/s", "TFLOP/s")
    assert rl["achieved"] > 0 and rl["peak"] > 0 and math.isclose(rl["frac"], rl["achieved"] / rl["peak"], rel_tol=1e-9)   # the contract's definition, unclamped
    assert rl["traffic"] is None or rl["traffic"] > 0
    assert 0.0 < rl["frac_of_floor"] < 1.2 and rl["latency_floor_us"] > 0   # this kernel's ruler, under its own key
    cb = out["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["unit"] == "mel-frames/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert out["value"] > 20 * cb["value"]   # (north_star: >= 20x real time is a far lower bar; the CPU port runs ~1.5x real time per core)
    # the line certifies itself: the first TIMED utterance's mel and its 30- / 60-iteration audio against the
    # oracle on the same inputs, computed after the timed region by the cpu_baseline machinery
    par = out["parity"]
    assert par["mel_shape_equal"] and par["frames"] == 800
    assert par["mel_rms"] <= 1e-4 and par["audio_rms_30it"] <= 1e-4 and par["mel_to_linear_rel_rms"] <= 1e-5   # the north star's tolerance, literally
    assert par["audio_rms_60it"] > 0 and par["audio_f32_vs_f64_60it"] > 0
    assert par["audio_gpu_vs_f64_60it"] <= 2 * par["audio_f32_vs_f64_60it"] + 1e-6                           # 60 iterations: bounded by fp32's own drift
    assert par["timed_audio_vs_scaled_60it_audio_rms"] <= 1e-6                                                # the timed call's audio IS that audio (x the output level)
    assert par["north_star_1e-4"]["mel"] and par["north_star_1e-4"]["audio_30it"]
    # both headline forms are first-class keys (filled when the extras run)
    assert "ms_per_step_one_call" in out and "one-call-per-utterance" in out["config"]["submission"]
