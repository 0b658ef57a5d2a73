"""CPU baselines of bench.py, one leg per process (so a leg can be given its own thread environment
and a hard timeout):  python oracle/cpu_baseline.py {c1|omp|torch} [threads]   -> one JSON line.

Every leg runs the full BASELINE.json configs[1] utterance once (120 ids -> chunks 95 + 25 -> 800
frames -> 60-iteration Griffin-Lim -> 204 544 samples) the way the reference executes it: batch 1, one
frame per decoder call (src/tacotron2/mod.rs:302-342), chunks in sequence (mod.rs:422-434), then the
vocoder (src/lib.rs:141).  TEST/BASELINE INFRASTRUCTURE ONLY.
  c1     oracle/xdtts_oracle.c, one thread            (BASELINE.md section 3, row C1); with a directory as third argument it also
         leaves its mel there (mel_oracle.npy) for the parity leg
  parity <dir>: the CHECKER beside the timed number -- <dir> holds what the GPU produced for the first timed utterance (mel_gpu.npy:
         its mel; S_gpu.npy: the GPU's mel -> linear of it; a30_gpu.npy / a60_gpu.npy: the GPU's un-normalised 30- / 60-iteration
         Griffin-Lim audio from that S and the phase of p0.npy, which the "phase" leg wrote there beforehand); prints their RMS distances from the oracle's outputs on the same
         inputs (bench.py's "parity" object)
  omp    the same source built with -fopenmp          (row C2; bit-identical results)
  torch  oracle/torch_cpu.py: MKL/oneDNN + pocketfft  (row C3)
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GL_ITERS, T_ENC, SR = 60, 100, 22050.0


def rms(a, b):
    import numpy as np

    return float(np.sqrt(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)))


def parity(d):
    """bench.py's self-check: the mel and audio of the first TIMED utterance against the oracle on the same ids / S / phase."""
    import numpy as np
    import oracle

    pkg = importlib.import_module("xd-tts_amd")          # host-side chunker only
    wl = importlib.import_module("xd-tts_amd.workloads")
    ids, chunks, steps = wl.config2(pkg)
    orc, orc64 = oracle.Oracle("f32"), oracle.Oracle("f64")
    mel_gpu = np.load(os.path.join(d, "mel_gpu.npy"))
    pm = os.path.join(d, "mel_oracle.npy")
    if os.path.exists(pm):
        mel = np.load(pm)                                  # (the c1 leg's: same call, same weights)
    else:
        blob = orc.weights_synthetic(seed=wl.WEIGHT_SEED)
        mel = np.concatenate([orc.infer_chunk(blob, c, orc.default_opts(fixed_steps=int(s), dropout_seed=0, item=i), window=T_ENC)
                              for i, (c, s) in enumerate(zip(chunks, steps))], axis=1)
    out = {"what": "first timed utterance of the headline against oracle/xdtts_oracle.c: mel on the same ids / weights / dropout stream; audio = "
                   "Griffin-Lim from the SAME S (the GPU's mel -> linear) and the same seeded phase, un-normalised, RMS of the difference in signal units",
           "frames": int(mel_gpu.shape[1]), "mel_shape_equal": bool(mel_gpu.shape == mel.shape)}
    out["mel_rms"] = rms(mel_gpu, mel) if mel_gpu.shape == mel.shape else None
    S = np.load(os.path.join(d, "S_gpu.npy"))
    pinv = orc.pinv(orc.mel_filter_bank())
    S32 = orc.mel_to_linear(pinv, mel_gpu, power=1.7)
    out["mel_to_linear_rel_rms"] = rms(S, S32) / float(np.sqrt(np.mean(S32.astype(np.float64) ** 2)))
    p0 = np.load(os.path.join(d, "p0.npy"))               # (the "phase" leg's: what the GPU was handed)
    a30, a60 = np.load(os.path.join(d, "a30_gpu.npy")), np.load(os.path.join(d, "a60_gpu.npy"))
    f32_30, f32_60 = orc.griffinlim(S, phase0=p0, iters=30), orc.griffinlim(S, phase0=p0, iters=60)
    f64_60 = orc64.griffinlim(S, phase0=p0, iters=60)
    out["audio_rms_30it"] = rms(a30, f32_30)
    out["audio_rms_60it"] = rms(a60, f32_60)
    out["audio_gpu_vs_f64_60it"] = rms(a60, f64_60)
    out["audio_f32_vs_f64_60it"] = rms(f32_60, f64_60)
    out["audio_signal_rms"] = float(np.sqrt(np.mean(f64_60.astype(np.float64) ** 2)))
    out["north_star_1e-4"] = {"mel": bool(out["mel_rms"] is not None and out["mel_rms"] <= 1e-4), "audio_30it": bool(out["audio_rms_30it"] <= 1e-4),
                              "audio_60it": bool(out["audio_rms_60it"] <= 1e-4),
                              "note": "60 iterations: the f32 oracle itself is audio_f32_vs_f64_60it from the f64 oracle (Griffin-Lim amplifies one ulp of phase to ~5e-4); "
                                      "the tests bound GPU - f64 by 2 x (f32 - f64) there and hold the 30-iteration setting of the reference (mod.rs:456) to 1e-4 literally"}
    print(json.dumps(out))


def main():
    leg = sys.argv[1]
    if leg == "phase":  # the seeded initial phase (seed 0) of an F-frame utterance as the oracle draws it -> <dir>/p0.npy, so that GPU and oracle start from the same bits
        import numpy as np
        import oracle

        np.save(os.path.join(sys.argv[2], "p0.npy"), oracle.Oracle("f32").phase_init(0, 513, int(sys.argv[3])))
        print(json.dumps({"ok": True}))
        return None
    if leg == "parity":
        return parity(sys.argv[2])
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dump = sys.argv[3] if len(sys.argv) > 3 else None
    if leg == "omp":
        os.environ["OMP_NUM_THREADS"] = str(threads)
        os.environ.setdefault("OMP_PROC_BIND", "close")
    import numpy as np
    import oracle

    pkg = importlib.import_module("xd-tts_amd")          # host-side chunker only
    wl = importlib.import_module("xd-tts_amd.workloads")
    ids, chunks, steps = wl.config2(pkg)
    orc = oracle.Oracle("f32", omp=(leg == "omp"))
    blob = orc.weights_synthetic(seed=wl.WEIGHT_SEED)
    pinv = orc.pinv(orc.mel_filter_bank())
    if leg in ("c1", "omp"):
        t0 = time.perf_counter()
        mel = np.concatenate([orc.infer_chunk(blob, c, orc.default_opts(fixed_steps=int(s), dropout_seed=0, item=i), window=T_ENC)
                              for i, (c, s) in enumerate(zip(chunks, steps))], axis=1)
        t1 = time.perf_counter()
        audio = orc.griffinlim(orc.mel_to_linear(pinv, mel, power=1.7), seed=0, iters=GL_ITERS)
        tm, tv = t1 - t0, time.perf_counter() - t1
        if dump:
            np.save(os.path.join(dump, "mel_oracle.npy"), mel)
        what = "C port (oracle/xdtts_oracle.c), one thread" if leg == "c1" else "the same C port built with -fopenmp (bit-identical results)"
    else:
        import torch
        from oracle import torch_cpu

        tt = torch_cpu.TorchTacotron2(orc, blob, threads=threads)
        threads = tt.threads
        t0 = time.perf_counter()
        mel = np.concatenate([tt.infer_chunk(c, int(s), 0, i, window=T_ENC) for i, (c, s) in enumerate(zip(chunks, steps))], axis=1)
        tm = time.perf_counter() - t0
        S = orc.mel_to_linear(pinv, mel, power=1.7)
        p0 = orc.phase_init(0, 513, mel.shape[1])
        t2 = time.perf_counter()
        audio = torch_cpu.griffinlim(S, p0, GL_ITERS)
        tv = time.perf_counter() - t2
        what = "torch-CPU restatement (oracle/torch_cpu.py): MKL/oneDNN GEMV + pocketfft, fp32"
    print(json.dumps({"value": mel.shape[1] / (tm + tv), "unit": "mel-frames/s", "cores": threads, "what": what, "mel_gen_s": tm, "vocoder_s": tv,
                      "rtf": (tm + tv) / (audio.size / SR), "frames": int(mel.shape[1]), "samples": int(audio.size)}))


if __name__ == "__main__":
    main()
