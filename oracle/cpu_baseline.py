"""CPU baselines of bench.py, one leg per process (so a leg can be given its own thread environment
and a hard timeout):  python oracle/cpu_baseline.py {c1|omp|torch} [threads]   -> one JSON line.

Every leg runs the full BASELINE.json configs[1] utterance once (120 ids -> chunks 95 + 25 -> 800
frames -> 60-iteration Griffin-Lim -> 204 544 samples) the way the reference executes it: batch 1, one
frame per decoder call (src/tacotron2/mod.rs:302-342), chunks in sequence (mod.rs:422-434), then the
vocoder (src/lib.rs:141).  TEST/BASELINE INFRASTRUCTURE ONLY.
  c1     oracle/xdtts_oracle.c, one thread            
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


def main():
    leg = sys.argv[1]
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
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
