#!/usr/bin/env python3
"""bench.py -- headline benchmark of the xd-tts hot path on MI355X.

Metric (BASELINE.json): mel-frames/s (+ audio samples/s, RTF) on a 120-phoneme synthetic utterance,
Tacotron2 decoder loop + post-net + 60-iteration Griffin-Lim (BASELINE.json configs[1]); one
"step" = one full utterance through XdTts::infer's sequence (src/lib.rs:110-159): ids -> chunks ->
encoder -> decoder loop -> post-net -> mel->linear -> Griffin-Lim -> audio.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL); utterances
are independent, so ranks shard them with no data-path collective ("scaling": "weak").
Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLE_RATE = 22050.0
HOP = 256
N_IDS = 120
TOTAL_FRAMES = 800            # gate disabled, 800 frames for 120 phonemes
GL_ITERS = 60                 # BASELINE.json configs[1]
WEIGHT_SEED = 20240327
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec

# Algorithmic bytes of one decoder step: the fp32 decoder_iter parameters are
# streamed once per lock-step iteration, plus per ACTIVE chunk the encoder memory + processed memory
# (T x (512+128) x 4) and the recurrent state read+write.
DECODER_PARAM_BYTES = 18_189_969 * 4
T_ENC = 100


def per_item_bytes(T):
    return T * (512 + 128) * 4 + 2 * (4 * 1024 + 2 * T + 512 + 80) * 4


def pmc_traffic():
    """HBM bytes of the decoder launches of one utterance (its whole frame loop) from the committed rocprofv3
    PMC passes of this same command (profiles/): counters cannot be read from inside the process,
    so the figure is the profiled one."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not files:
        return None
    return json.load(open(files[-1])).get("decoder_launch_traffic_bytes")


def synth_ids(n, seed=1):
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = 64 + rng.integers(0, 84, size=n)
    ids[5::6] = 11
    ids[-1] = 7
    return ids.astype(np.int64)


def cpu_baseline(ids, splits, chunk_steps):
    """The CPU oracle (single-thread C port of the same algorithm) on the same utterance, once."""
    import oracle

    orc = oracle.Oracle("f32")
    blob = orc.weights_synthetic(seed=WEIGHT_SEED)
    basis = orc.mel_filter_bank()
    pinv = orc.pinv(basis)
    t0 = time.perf_counter()
    mels = []
    start = 0
    for item, (end, steps) in enumerate(zip(splits, chunk_steps)):
        o = orc.default_opts(fixed_steps=int(steps), dropout_seed=0, item=item)
        mels.append(orc.infer_chunk(blob, ids[start:end], o, window=T_ENC))
        start = end
    mel = np.concatenate(mels, axis=1)
    t1 = time.perf_counter()
    S = orc.mel_to_linear(pinv, mel, power=1.7)
    audio = orc.griffinlim(S, seed=0, iters=GL_ITERS)
    t2 = time.perf_counter()
    frames = mel.shape[1]
    return {
        "value": frames / (t2 - t0),
        "unit": "mel-frames/s",
        "cores": 1,
        "kind": "port",
        "sample": "the full config-2 utterance once: %d ids -> %d frames, %d-iteration Griffin-Lim, %d samples"
        % (len(ids), frames, GL_ITERS, audio.size),
        "mel_gen_s": t1 - t0,
        "vocoder_s": t2 - t1,
        "rtf": (t2 - t0) / (audio.size / SAMPLE_RATE),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    pkg = importlib.import_module("xd-tts_amd")
    if pkg.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libxdtts_hip has no CPU path")

    ids = synth_ids(N_IDS)
    splits = list(pkg.find_splits(ids, T_ENC))
    if not splits or splits[-1] != len(ids):
        splits.append(len(ids))          # src/tacotron2/mod.rs:412-414
    lens = np.diff([0] + splits)
    fpi = TOTAL_FRAMES / float(N_IDS)
    chunk_steps = [int(np.floor(fpi * n + 0.5)) for n in lens]   # lround, as the library does

    model = pkg.Tacotron2.synthetic(seed=WEIGHT_SEED, rec_scale=1.0, device_id=local_rank)
    vocoder = pkg.create_griffin_lim(device_id=local_rank, iters=GL_ITERS, seed=0)
    opts = pkg.default_opts(fixed_frames_per_id=fpi, dropout_seed=0, item_base=0)
    sp = np.asarray(splits, dtype=np.int64)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pkg.synthesize(model, vocoder, ids, splits=sp, opts=opts)

    dec_ms = gl_ms = enc_ms = post_ms = m2l_ms = 0.0
    dec_steps = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mel, audio = pkg.synthesize(model, vocoder, ids, splits=sp, opts=opts)   # synchronous: returns host buffers
        tt, tg = model.last_timings(), vocoder.last_timings()
        enc_ms += tt["encoder_ms"]
        dec_ms += tt["decoder_ms"]
        post_ms += tt["postnet_ms"]
        dec_steps += tt["steps"]
        m2l_ms += tg["mel_to_linear_ms"]
        gl_ms += tg["iterations_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    frames = mel.shape[1]
    samples = audio.size
    K = args.steps
    total_frames = frames * K * world
    value = total_frames / elapsed

    # roofline of the dominant kernel: the persistent decoder (k_decoder_persistent<2> while both chunks
    # run, continued by k_decoder_persistent<1> for the longer one; HIP events around the pair on the
    # library's stream, summed over the timed region).  Its algorithmic bytes are SURVEY 8(d)'s
    # per-step figure x the steps executed: every decoder parameter once per step + per-chunk
    # memory/state for the chunks still active at that step.
    steps_per_utt = dec_steps / K
    active = sum(min(s, int(steps_per_utt)) for s in chunk_steps)          # sum over steps of #active chunks
    bytes_per_utt = steps_per_utt * DECODER_PARAM_BYTES + active * per_item_bytes(T_ENC)
    step_us = dec_ms / dec_steps * 1e3
    achieved = bytes_per_utt / (dec_ms / K * 1e-3) / 1e9
    out = {
        "metric": "mel-frames/s per GPU, 120-phoneme utterance (Tacotron2 decoder+postnet + %d-iter Griffin-Lim), end-to-end" % GL_ITERS,
        "value": value,
        "unit": "mel-frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seeded fp32 weights of the NVIDIA Tacotron2 shapes, seeded 120-id utterance)",
        "config": {
            "workload": "BASELINE.json configs[1]: batch=1 utterance, 120 phoneme ids -> chunks %s (window 100) -> %d mel frames (gate disabled) -> %d-iter Griffin-Lim -> %d samples"
            % (list(map(int, lens)), frames, GL_ITERS, samples),
            "utterances_per_gpu_per_step": 1,
            "parallelism": "utterance-shard x%d" % world,
        },
        "audio_samples_per_s": samples * K * world / elapsed,
        "rtf": (elapsed / K) / (samples / SAMPLE_RATE),
        "x_realtime": (samples / SAMPLE_RATE) / (elapsed / K),
        "phase_ms_per_utterance": {
            "encoder": enc_ms / K,
            "decoder_loop": dec_ms / K,
            "postnet": post_ms / K,
            "mel_to_linear": m2l_ms / K,
            "griffinlim_iterations": gl_ms / K,
        },
        "roofline": {
            "kernel": "k_decoder_persistent (<2> for the %d steps both chunks run, then <1>: %d lock-step decoder steps per utterance in two launches; weights stay in registers, so HBM traffic is far below the algorithmic bytes and the bound in practice is the 6 state-exchange edges per step, DESIGN.md section 4)" % (min(chunk_steps), int(steps_per_utt)),
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(),
            "us_per_launch": dec_ms / K * 1e3,
            "algorithmic_bytes_per_launch": bytes_per_utt,
            "launches_per_utterance": 2,
            "steps_per_launch": steps_per_utt,
            "us_per_step": step_us,
            "algorithmic_bytes_per_step": bytes_per_utt / steps_per_utt,
        },
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ids, splits, chunk_steps)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
