#!/usr/bin/env python3
"""bench.py -- headline benchmark of the xd-tts hot path on MI355X.

Metric (BASELINE.json): mel-frames/s (+ audio samples/s, RTF) on a 120-phoneme synthetic utterance,
Tacotron2 decoder loop + post-net + 60-iteration Griffin-Lim (BASELINE.json configs[1]); one
"step" = one full utterance through XdTts::infer's sequence (src/lib.rs:110-159): ids -> chunks ->
encoder -> decoder loop -> post-net -> mel->linear -> Griffin-Lim -> audio.  The K utterances of a rank are
submitted as ONE xdtts_synthesize_sequence call (each decoded alone, batch 1; the vocoder of one runs beside the
encoder of the next); extra.headline_one_call_per_utterance times K separate synchronous calls.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL).  Utterances
are independent (src/tacotron2/mod.rs:422-434), so the N*K utterances of the
timed region are assigned to ranks by xd-tts_amd/shard.py with no data-path collective
("scaling": "weak": K utterances per GPU whatever N is).  Rank 0 prints ONE JSON line.

Besides the headline the line carries `extra` blocks for the other BASELINE.json configs, measured
after the timed region (each a few tens of ms of GPU time):
  extra.config3  configs[2]: 32 variable-length utterances (40-200 ids) -> <=100-id chunks decoded as one
                 padded/masked lock-step batch (decoder-only figures + roofline of the batched LSTM kernels)
  extra.config4  configs[3]: 32 utterances per GPU (256 at N = 8), length-sorted round-robin over the ranks,
                 end to end (batched mel-gen + per-utterance vocoder); counters all-gathered over RCCL
  extra.config5  configs[4]: Griffin-Lim only, 1000 frames, 30/60/120 iterations, with its HBM roofline
and `cpu_baseline` (N = 1 only): the CPU port of the same algorithm on the host cores.
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
GL_ITERS = 60                 # BASELINE.json configs[1]
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
VALU_F32_PEAK_TF = 157.3      # MI355X_MICROARCH.md spec table: "Peak FP32 (vector) 157.3 TFLOPS" (= the fp32 matrix rate, 64 FLOP/clk/SIMD); the persistent decoder computes on the VALU
POSTNET_FLOP_PER_FRAME = 8683520.0  # 5 x conv1d k5 (80->512->512->512->512->80), 2 flops per MAC

# Algorithmic work of one decoder step: the fp32 decoder_iter parameters are
# streamed once per lock-step iteration, plus per ACTIVE chunk the encoder memory + processed memory
# (T x (512+128) x 4) and the recurrent state read+write; 2 (18 167 296 + 6 720 T) flops per chunk.
DECODER_PARAM_BYTES = 18_189_969 * 4
T_ENC = 100
GL_BYTES_PER_FRAME_ITER = 12308.0   # fused minimum per frame per iteration


def per_item_bytes(T):
    return T * (512 + 128) * 4 + 2 * (4 * 1024 + 2 * T + 512 + 80) * 4


def per_item_flops(T):
    return 2.0 * (18_167_296 + 6_720 * T)


def pmc_traffic():
    """HBM bytes of the decoder launches of one utterance (its whole frame loop) from the committed
    rocprofv3 PMC passes of this same command (profiles/): counters cannot be read from inside the
    process, so the figure is the profiled one, stamped with the commit it was profiled at."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    return d.get("decoder_launch_traffic_bytes"), {"file": os.path.relpath(files[-1], ROOT), "profiled_at": d.get("git_head", "round 1")}


def edge_floor(pkg, device):
    """The latency floor of the persistent decoder's step, measured NOW on the benched device: the step's five dependent
    all-gather exchanges with no arithmetic in between (same grid, same granule transport, the consumers' first-poll
    delays as in the engine; csrc/edge_floor.hip through xdtts_edge_floor_us, best of five launches of 2000 steps, after
    the timed region).  The step time is a fraction of this, not of an HBM figure: nothing streams."""
    try:
        us = pkg.edge_floor_us(device, 2000, T_ENC, True)
        return us, {"measured": "live on the benched device in this run (xdtts_edge_floor_us: 2000 steps, T = %d, best of 5)" % T_ENC,
                    "kernel": "xd-tts_amd/csrc/edge_floor.hip", "without_poll_delays_us": pkg.edge_floor_us(device, 2000, T_ENC, False)}
    except Exception as e:  # noqa: BLE001
        return None, {"error": repr(e)}


def cpu_baseline(parity_dir=None):
    """The CPU ports of the same algorithm on this box's host cores, each on the full config-2 utterance
    once (oracle/cpu_baseline.py, one process per leg with a hard timeout): the single-thread C oracle
    (the reference's execution model), the same source with OpenMP, and a torch-CPU restatement.  The
    multi-threaded legs use min(host cores, 32) threads: the step is a 72 MB GEMV stream, which a few
    memory channels' worth of cores saturate."""
    import subprocess

    ncpu = os.cpu_count() or 1
    nthr = min(ncpu, 32)

    def leg(name, threads, extra_arg=None):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), name, str(threads)] + ([extra_arg] if extra_arg else []),
                               capture_output=True, text=True, timeout=150)
            if r.returncode != 0:
                return {"error": r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "rc %d" % r.returncode}
            return json.loads(r.stdout.strip().splitlines()[-1])
        except subprocess.TimeoutExpired:
            return {"error": "timed out after 150 s"}
        except Exception as e:   # a baseline leg must never take the bench line down
            return {"error": repr(e)}

    out = leg("c1", 1, parity_dir)
    log("cpu baseline: 1 thread done")
    out["kind"] = "port"
    out["host_cores"] = ncpu
    if "frames" in out:
        out["sample"] = "the full config-2 utterance once: 120 ids -> %d frames, %d-iteration Griffin-Lim, %d samples" % (out["frames"], GL_ITERS, out["samples"])
    out["all_cores_openmp"] = leg("omp", nthr)
    log("cpu baseline: OpenMP done")
    out["all_cores_torch"] = leg("torch", nthr)
    log("cpu baseline: torch done")
    return out


def parity_check(parity_dir):
    """The checker beside the number (oracle/cpu_baseline.py parity, its own process, after the timed region): the mel and the
    30- / 60-iteration audio of the FIRST TIMED utterance against the oracle on the same inputs."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "parity", parity_dir], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "rc %d" % r.returncode}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def log(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--check-shared-utterance", action="store_true",
                    help="every rank also synthesizes utterance 0 (outside the timed region) and the line carries the sha256 of "
                         "its mel and audio per rank: identical bits on every rank and in a single-process run")
    ap.add_argument("--broadcast-weights", action="store_true",
                    help="SURVEY 8(e)'s load-time collective: rank 0 builds the weight blob (113 MB) and broadcasts it (RCCL over xGMI, "
                         "GPU 0 -> all); every rank loads from the received blob.  Default: each rank generates the same seeded weights")
    ap.add_argument("--load-from-dir", default=None, metavar="DIR",
                    help="every rank loads through Tacotron2::load(DIR) (mod.rs:242) -- the path a deployment uses -- from a "
                         "tacotron2.xdtw that rank 0 writes there first (synthetic weights); default: each rank generates them")
    args = ap.parse_args()

    import torch   # before the product library: see tests/conftest.py on the load order of the HIP runtime

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # developer aids for exercising the N > 1 control flow on a 1-GPU box: XDTTS_BENCH_DEVICE pins every rank to
    # one device, XDTTS_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU)
    dev_override = os.environ.get("XDTTS_BENCH_DEVICE")
    if dev_override is not None:
        local_rank = int(dev_override)
        # several ranks on ONE GPU: their co-resident launches (persistent decoder, cooperative encoder, persistent
        # Griffin-Lim) must take turns across processes too -- the library's flock next to its per-process lock
        import tempfile

        os.environ.setdefault("XDTTS_CHIP_LOCK_DIR", tempfile.gettempdir())
    backend = os.environ.get("XDTTS_BENCH_BACKEND", "nccl")
    red_dev = "cuda" if backend == "nccl" else "cpu"
    # XDTTS_BENCH_FORCE_DIST=1: take the distributed branch at world_size 1 too (under `torch.distributed.run --nproc-per-node 1`),
    # so that RCCL's init, the device-tensor all_reduce / all_gather / broadcast and the barriers run on a 1-GPU box
    # (tests/test_gpu_bench_rccl_one_rank.py) -- the code an 8-GPU node executes, not a gloo stand-in
    if world > 1 or os.environ.get("XDTTS_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)                      # first collective: RCCL's communicator is really up before the library loads
            log("rccl: init_process_group(nccl) ok, world %d, all_reduce -> %g" % (world, float(probe.item())))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local_rank)

    pkg = importlib.import_module("xd-tts_amd")
    wl = importlib.import_module("xd-tts_amd.workloads")
    shard = importlib.import_module("xd-tts_amd.shard")
    if pkg.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libxdtts_hip has no CPU path")

    if args.broadcast_weights:
        n_floats = int(pkg.lib.xdtts_tensor_total())
        if rank == 0:
            seedling = pkg.Tacotron2.synthetic(seed=wl.WEIGHT_SEED, rec_scale=1.0, device_id=local_rank)
            blob_t = torch.from_numpy(seedling.blob()).to(red_dev)
            seedling.close()
        else:
            blob_t = torch.empty(n_floats, dtype=torch.float32, device=red_dev)
        if dist is not None:
            dist.broadcast(blob_t, src=0)          # the one load-time collective (no data-path collective follows)
        model = pkg.Tacotron2.from_blob(blob_t.cpu().numpy(), device_id=local_rank)
        del blob_t
    elif args.load_from_dir:
        if rank == 0:
            os.makedirs(args.load_from_dir, exist_ok=True)
            seedling = pkg.Tacotron2.synthetic(seed=wl.WEIGHT_SEED, rec_scale=1.0, device_id=local_rank)
            seedling.save(args.load_from_dir)
            seedling.close()
        if dist is not None:
            dist.barrier()
        model = pkg.Tacotron2.load(args.load_from_dir, device_id=local_rank)
    else:
        model = pkg.Tacotron2.synthetic(seed=wl.WEIGHT_SEED, rec_scale=1.0, device_id=local_rank)
    vocoder = pkg.create_griffin_lim(device_id=local_rank, iters=GL_ITERS, seed=0)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- headline: configs[1], K utterances per rank -------------------------------------------------
    K = args.steps
    utterances = [wl.synth_ids(120, seed=1 + g) for g in range(world * K)]        # N*K distinct utterances
    mine = shard.shard_utterances([len(u) for u in utterances], rank, world)       # K of them for this rank
    _ids0, chunks0, chunk_steps = wl.config2(pkg)
    lens = [len(c) for c in chunks0]
    sp = np.cumsum(lens).astype(np.int64)       # spaces sit at fixed positions: every utterance splits 95 + 25
    opts = pkg.default_opts(fixed_frames_per_id=wl.FRAMES_PER_ID, dropout_seed=0, item_base=0)
    # The K utterances of a rank are submitted the way a server streams sentences: ONE xdtts_synthesize_sequence call.  Every
    # utterance is still decoded alone (batch 1, the reference's loop over src/lib.rs:122-141) and comes back with the bits of
    # a single xdtts_synthesize_ids call; what the sequence adds is that the vocoder of utterance u runs beside the encoder of
    # utterance u + 1 (the frame loop between them owns every CU).  extra.headline_one_call_per_utterance times the K single calls.
    if args.warmup > 0:
        pkg.synthesize_sequence(model, vocoder, [utterances[mine[0]]] * args.warmup, [sp] * args.warmup, opts=opts, want_mels=False)
    seq = [utterances[g] for g in mine]
    barrier()
    t0 = time.perf_counter()
    mels, audios = pkg.synthesize_sequence(model, vocoder, seq, [sp] * K, opts=opts)   # synchronous: returns host buffers (K mels, K audios)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    mel, audio = mels[-1], audios[-1]
    tt, tg = model.last_timings(), vocoder.last_timings()      # HIP-event sums over the K utterances of the timed region
    enc_ms, dec_ms, post_ms, dec_steps = tt["encoder_ms"], tt["decoder_ms"], tt["postnet_ms"], tt["steps"]
    m2l_ms, gl_ms = tg["mel_to_linear_ms"], tg["iterations_ms"]

    frames = mel.shape[1]
    samples = audio.size
    value = frames * K * world / elapsed

    # roofline of the dominant kernel: the persistent decoder (k_decoder_persistent<2> while both chunks
    # run, continued by k_decoder_persistent<1> for the longer one; HIP events around the pair on the
    # library's stream, summed over the timed region).  Algorithmic bytes = SURVEY 8(d)'s per-step
    # figure x the steps executed.
    steps_per_utt = dec_steps / K
    active = sum(min(s, int(steps_per_utt)) for s in chunk_steps)          # sum over steps of #active chunks
    bytes_per_utt = steps_per_utt * DECODER_PARAM_BYTES + active * per_item_bytes(T_ENC)
    achieved = bytes_per_utt / (dec_ms / K * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic()
    floor_us, floor_src = edge_floor(pkg, local_rank)
    flops_per_step = active * per_item_flops(T_ENC) / steps_per_utt           # 37.7 MFLOP per active chunk
    us_per_step = dec_ms / dec_steps * 1e3
    out = {
        "metric": "mel-frames/s per GPU, 120-phoneme utterance (Tacotron2 decoder+postnet + %d-iter Griffin-Lim), end-to-end" % GL_ITERS,
        "value": value,
        "unit": "mel-frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": elapsed / K * 1e3,
        "ms_per_step_one_call": None,   # (filled below: the same K utterances as K synchronous calls, the headline form of rounds 1-4)
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seeded fp32 weights of the NVIDIA Tacotron2 shapes, seeded 120-id utterances)",
        "config": {
            "workload": "BASELINE.json configs[1]: batch=1 utterance, 120 phoneme ids -> chunks %s (window 100) -> %d mel frames (gate disabled) -> %d-iter Griffin-Lim -> %d samples"
            % (lens, frames, GL_ITERS, samples),
            "utterances_per_gpu_per_step": 1,
            "submission": "the K utterances of a rank as one xdtts_synthesize_sequence call: batch 1 per utterance, vocoder(u) beside encoder(u + 1) -- "
                          "`value` / `ms_per_step` (rounds 5-6).  The round-over-round series of rounds 1-4 is the one-call-per-utterance form: top-level "
                          "`ms_per_step_one_call` / `value_one_call` (= extra.headline_one_call_per_utterance), K synchronous xdtts_synthesize_ids calls",
            "parallelism": "utterance-shard x%d (xd-tts_amd/shard.py, no data-path collective)" % world,
        },
        "audio_samples_per_s": samples * K * world / elapsed,
        "rtf": (elapsed / K) / (samples / SAMPLE_RATE),
        "x_realtime": (samples / SAMPLE_RATE) / (elapsed / K),
        "phase_note": "HIP-event times per utterance; in a sequence the encoder of utterance u + 1 and the vocoder of utterance u run side by side, "
                      "so the phases sum to more than ms_per_step",
        "phase_ms_per_utterance": {
            "encoder": enc_ms / K,
            "decoder_loop": dec_ms / K,
            "postnet": post_ms / K,
            "mel_to_linear": m2l_ms / K,
            "griffinlim_iterations": gl_ms / K,
        },
        "postnet_roofline": {"kernel": "k_gemm_nt<32,32> x 5 (implicit-GEMM conv1d k5 over the utterance's %d frames)" % frames, "bound": "mfma",
                             "achieved": POSTNET_FLOP_PER_FRAME * frames / (post_ms / K * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                             "frac": POSTNET_FLOP_PER_FRAME * frames / (post_ms / K * 1e-3) / 1e12 / MFMA_F32_PEAK_TF},
        "roofline": {
            "kernel": "k_decoder_persistent (<2> for the %d steps both chunks run, then <1>: %d lock-step decoder steps per utterance in two launches)" % (min(chunk_steps), int(steps_per_utt)),
            # `achieved` / `peak` / `frac` are the contract's: SURVEY 8(d)'s ALGORITHMIC bytes (every decoder parameter counted
            # once per step) per second against the HBM peak, frac = achieved / peak, unclamped.  The weights stay in the
            # register files, so nothing streams (`traffic`, PMC, is < 1 % of the algorithmic bytes) and that ratio can pass
            # 1.0; what the step is really bound by is the LATENCY of its five dependent inter-CU exchanges, reported next
            # to it under its own keys: `latency_floor_us` = the same five exchanges with no arithmetic between them
            # (csrc/edge_floor.hip, measured live on this chip), `frac_of_floor` = floor / step time.
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "latency_floor_us": floor_us,
            "latency_floor_source": floor_src,
            "frac_of_floor": (floor_us / us_per_step) if floor_us else None,
            "vector_fma_frac": flops_per_step / (us_per_step * 1e-6) / 1e12 / VALU_F32_PEAK_TF,
            "frac_note": "frac = achieved / peak as the bench contract defines it (algorithmic bytes per second against 8 TB/s); the "
                         "weights are resident in registers (traffic < 1 % of the algorithmic bytes), so it can pass 1.0 and says "
                         "little about this kernel.  frac_of_floor = latency floor / step time (unclamped): the kernel's real bound.  "
                         "vector_fma_frac = SURVEY 8(d)'s flops per step / step time against the fp32 vector peak",
            "traffic": traffic,
            "traffic_source": traffic_src,
            "limiter": "inter-CU exchange latency (5 dependent all-gather edges per step), not HBM bandwidth",
            "hbm_traffic_GBs": (traffic / (dec_ms / K * 1e-3) / 1e9) if traffic else None,
            "us_per_launch": dec_ms / K * 1e3,
            "algorithmic_bytes_per_launch": bytes_per_utt,
            "launches_per_utterance": 2,
            "steps_per_launch": steps_per_utt,
            "us_per_step": us_per_step,
            "algorithmic_bytes_per_step": bytes_per_utt / steps_per_utt,
        },
    }

    log("headline done: %.0f frames/s" % value)
    # What the parity leg checks (after everything timed): the FIRST TIMED utterance's mel as it came back from the timed call, the
    # GPU's mel -> linear of it and the un-normalised 30- / 60-iteration audio from that S and the oracle's seeded phase table (the audio
    # of the timed call itself is the 60-iteration one from the in-kernel stream behind the output normalisation: compared with that below)
    parity_dir = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and np.array_equal(seq[0], _ids0):
        import tempfile

        parity_dir = tempfile.mkdtemp(prefix="xdtts_parity_")
        np.save(os.path.join(parity_dir, "mel_gpu.npy"), mels[0])
        S_gpu = vocoder.mel_to_linear(mels[0])
        np.save(os.path.join(parity_dir, "S_gpu.npy"), S_gpu)
        import subprocess

        subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "phase", parity_dir, str(frames)], capture_output=True, timeout=120)
        p0 = np.load(os.path.join(parity_dir, "p0.npy"))   # the oracle's seeded phase table (seed 0): GPU and oracle start from the same bits
        np.save(os.path.join(parity_dir, "a30_gpu.npy"), vocoder.infer_linear(S_gpu, phase0=p0, iters=30))
        np.save(os.path.join(parity_dir, "a60_gpu.npy"), vocoder.infer_linear(S_gpu, phase0=p0, iters=60))
        a60 = vocoder.infer_linear(S_gpu, iters=60)        # ... and from the in-kernel seeded stream, as the timed call ran it
        scale = float(np.dot(audios[0].astype(np.float64), a60.astype(np.float64)) / max(np.dot(a60.astype(np.float64), a60.astype(np.float64)), 1e-30))
        timed_audio_is_that_audio = float(np.sqrt(np.mean((audios[0].astype(np.float64) - scale * a60.astype(np.float64)) ** 2)))
    extra = {}
    if args.check_shared_utterance:
        import hashlib

        m0, a0 = pkg.synthesize(model, vocoder, utterances[0], splits=sp, opts=opts)
        mine_sha = {"rank": rank, "mel_sha256": hashlib.sha256(np.ascontiguousarray(m0).tobytes()).hexdigest(),
                    "audio_sha256": hashlib.sha256(np.ascontiguousarray(a0).tobytes()).hexdigest(), "frames": int(m0.shape[1]), "samples": int(a0.size)}
        if dist is not None:
            every = [None] * world
            dist.all_gather_object(every, mine_sha)
        else:
            every = [mine_sha]
        extra["shared_utterance"] = {"what": "utterance 0 of the headline set synthesized by every rank", "per_rank": every}
    if not args.no_extras:
        # ---- the same utterances with the REFERENCE's vocoder setting: 30 iterations (GriffinLim::new(.., 30, 0.99),
        # src/tacotron2/mod.rs:456; SURVEY 8(d) "reference default 30 also reported"), outside the timed region above
        try:
            voc30 = pkg.create_griffin_lim(device_id=local_rank, iters=30, seed=0)
            pkg.synthesize(model, voc30, utterances[mine[0]], splits=sp, opts=opts)
            barrier()
            t30 = time.perf_counter()
            gl30 = 0.0
            for g in mine:
                _m, a30 = pkg.synthesize(model, voc30, utterances[g], splits=sp, opts=opts)
                gl30 += voc30.last_timings()["iterations_ms"]
            barrier()
            e30 = max_over_ranks(time.perf_counter() - t30)
            extra["headline_30_iterations"] = {
                "workload": "configs[1] with the reference's 30 Griffin-Lim iterations (mod.rs:456) instead of BASELINE.json's 60",
                "mel_frames_per_s": frames * K * world / e30, "audio_samples_per_s": a30.size * K * world / e30,
                "ms_per_utterance": e30 / K * 1e3, "x_realtime": (a30.size / SAMPLE_RATE) / (e30 / K), "griffinlim_iterations_ms": gl30 / K}
            voc30.close()
        except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
            extra["headline_30_iterations"] = {"error": repr(e)}
    if not args.no_extras:
        # ---- the same K utterances as K synchronous xdtts_synthesize_ids calls, one after the other (the headline of rounds 1-4: nothing
        # of utterance u + 1 starts before utterance u's audio is on the host)
        try:
            pkg.synthesize(model, vocoder, utterances[mine[0]], splits=sp, opts=opts)
            barrier()
            ts0 = time.perf_counter()
            for g in mine:
                m1, a1 = pkg.synthesize(model, vocoder, utterances[g], splits=sp, opts=opts)
            barrier()
            es = max_over_ranks(time.perf_counter() - ts0)
            out["ms_per_step_one_call"] = es / K * 1e3
            out["value_one_call"] = frames * K * world / es
            extra["headline_one_call_per_utterance"] = {
                "workload": "configs[1]'s K utterances through K xdtts_synthesize_ids calls (strictly sequential, the reference's loop)",
                "mel_frames_per_s": frames * K * world / es, "ms_per_utterance": es / K * 1e3, "x_realtime": (samples / SAMPLE_RATE) / (es / K),
                "same_bits_as_the_sequence_call": bool(np.array_equal(a1, audio) and np.array_equal(m1, mel))}
        except Exception as e:  # noqa: BLE001
            extra["headline_one_call_per_utterance"] = {"error": repr(e)}
    if not args.no_extras:
        # ---- the same utterance with the gate ON (the reference's real mode, mod.rs:319-324): gate_layer rigged so that
        # sigmoid(gate) > 0.6 fires exactly at 633 / 167 frames (xd-tts_amd/gate_rig.py; every frame is unchanged), so the
        # device-side stop rule, the survivor hand-over of the pair and the frame-count round trip are TIMED, not only tested
        try:
            rig = importlib.import_module("xd-tts_amd.gate_rig")
            gopts = pkg.default_opts(dropout_seed=0, item_base=0)       # gate_threshold 0.6, max_steps 1000, no fixed steps
            utt_g = utterances[mine[0]]
            chunks_g = [utt_g[: lens[0]], utt_g[lens[0] :]]
            rigged, rinfo = rig.rigged_gate_model(pkg, model, chunks_g, chunk_steps, gopts, device_id=local_rank)
            mel_fix, _a = pkg.synthesize(model, vocoder, utt_g, splits=sp, opts=opts)
            for _ in range(2):
                mel_on, a_on = pkg.synthesize(rigged, vocoder, utt_g, splits=sp, opts=gopts)
            barrier()
            tg0 = time.perf_counter()
            dec_on = 0.0
            for _ in range(K):
                mel_on, a_on = pkg.synthesize(rigged, vocoder, utt_g, splits=sp, opts=gopts)
                dec_on += rigged.last_timings()["decoder_ms"]
            barrier()
            eg = max_over_ranks(time.perf_counter() - tg0)
            extra["headline_gate_on"] = {
                "workload": "configs[1]'s utterance with the stop rule deciding (gate_layer rigged to fire at %s frames; gate_threshold 0.6, max_decoder_steps 1000)" % chunk_steps,
                "frames": int(mel_on.shape[1]), "frames_equal_fixed_steps_run": bool(mel_on.shape == mel_fix.shape and np.array_equal(mel_on, mel_fix)),
                "mel_frames_per_s": mel_on.shape[1] * K * world / eg, "ms_per_utterance": eg / K * 1e3, "decoder_loop_ms": dec_on / K,
                "decoder_us_per_step": dec_on / K * 1e3 / max(chunk_steps), "rig": rinfo}
            rigged.close()
            log("gate-on variant done")
        except Exception as e:  # noqa: BLE001
            extra["headline_gate_on"] = {"error": repr(e)}
    if not args.no_extras:
        # ---- configs[2] / configs[3]: this rank's share of 32*N utterances as one lock-step batch -------
        all_utts = wl.config4(pkg, n_batches=world)
        share, chunks, steps, owner = shard.plan_share(lambda ids: wl.chunk_utterance(pkg, ids), all_utts, rank, world, wl.FRAMES_PER_ID_BATCH)
        bo = pkg.default_opts(dropout_seed=1, item_base=0)
        # A rank that fails here must still meet the others at the barriers and in the all-gather (the headline
        # line is already measured and has to be printed): its share then counts as failed, seconds = -1.
        res, share_err = None, None
        try:
            for _ in range(2):
                shard.run_share(model, vocoder, share, chunks, steps, owner, bo)      # warm-up
        except Exception as e:  # noqa: BLE001
            share_err = repr(e)
        barrier()
        if share_err is None:
            try:
                res = shard.run_share(model, vocoder, share, chunks, steps, owner, bo)
            except Exception as e:  # noqa: BLE001
                share_err = repr(e)
        barrier()
        # configs[3] is timed on the one-call form (xdtts_synthesize_batch: the mel stays in HBM between mel-gen and vocoder);
        # configs[2] above on the mel-gen entry point alone, the mel delivered to the host
        res_f = None
        if share_err is None:
            try:
                for _ in range(2):
                    shard.run_share(model, vocoder, share, chunks, steps, owner, bo, fused=pkg.synthesize_batch)  # warm-up
            except Exception as e:  # noqa: BLE001
                share_err = repr(e)
        barrier()
        if share_err is None:
            try:
                res_f = shard.run_share(model, vocoder, share, chunks, steps, owner, bo, fused=pkg.synthesize_batch)
                voc_dev_ms = vocoder.last_timings()["total_ms"]
            except Exception as e:  # noqa: BLE001
                share_err = repr(e)
        barrier()
        log("config3/4 share done" if share_err is None else "config3/4 share FAILED: " + share_err)
        counters = res_f if res_f is not None else {"frames": 0, "samples": 0, "seconds": -1.0}
        totals, max_s, per_rank = shard.gather_counters(counters, dist, device=red_dev if dist else "cpu")
        share_ok = all(r["seconds"] >= 0 for r in per_rank)
    if not args.no_extras and not share_ok:
        extra["config3"] = extra["config4"] = {"error": "a rank's share failed", "this_rank": share_err, "per_rank": per_rank}
    if not args.no_extras and share_ok:
        fr, t_mel, tm, voc_ms = res["frames"], res["mel_gen_seconds"], res["timings"], res["vocoder_seconds"] * 1e3
        it = tm["steps"]
        act = float(sum(steps))                                                    # active chunk-steps of this rank
        dsec = tm["decoder_ms"] * 1e-3
        c3_bytes = it * DECODER_PARAM_BYTES + act * per_item_bytes(T_ENC)
        c3_flops = act * per_item_flops(T_ENC)
        extra["config3"] = {
            "workload": "BASELINE.json configs[2]: %d utterances (40-200 ids) -> %d chunks in one padded/masked lock-step batch, %d frames, %d lock-step iterations (this rank)"
            % (len(share), len(chunks), fr, it),
            "mel_frames_per_s_mel_gen": fr / t_mel,
            "ms": {"encoder": tm["encoder_ms"], "decoder_loop": tm["decoder_ms"], "postnet": tm["postnet_ms"], "wall_mel_gen": t_mel * 1e3},
            "postnet_roofline": {"kernel": "k_gemm_nt<64,64> x 5 (implicit-GEMM conv1d k5, SURVEY 8d: 8 683 520 FLOP per frame)", "bound": "mfma",
                                 "achieved": POSTNET_FLOP_PER_FRAME * fr / (tm["postnet_ms"] * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                 "frac": POSTNET_FLOP_PER_FRAME * fr / (tm["postnet_ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF},
            "us_per_lockstep_iteration": tm["decoder_ms"] * 1e3 / it,
            "roofline": {
                "kernel": "the decoder iteration of the batched path (two launches: k_att_lstm_attention, k_lstm_mfma<DEC> with the next iteration's early attention-LSTM blocks and the projection / stop rule / prenet tail)",
                "bound": "mfma",
                "achieved": c3_flops / dsec / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": c3_flops / dsec / 1e12 / MFMA_F32_PEAK_TF,
                "hbm_achieved_GBs": c3_bytes / dsec / 1e9, "hbm_frac": c3_bytes / dsec / 1e9 / HBM_PEAK_GBS,
                "note": "at 52 chunks the step is balanced between the two roofs (1.0 GFLOP and 73 MB per iteration: 6.6 us of fp32 MFMA, 9.1 us of HBM); traffic: profiles/",
            },
        }
        extra["config4"] = {
            "workload": "BASELINE.json configs[3]: %d utterances, %d per GPU (length-sorted round-robin, xd-tts_amd/shard.py), batched mel-gen + batched %d-iter Griffin-Lim in one call (xdtts_synthesize_batch)"
            % (len(all_utts), len(share), GL_ITERS),
            "utterances_per_s": len(all_utts) / max_s,
            "mel_frames_per_s": totals["frames"] / max_s,
            "audio_samples_per_s": totals["samples"] / max_s,
            "rtf": max_s / (totals["samples"] / SAMPLE_RATE),
            "seconds_max_over_ranks": max_s,
            "vocoder_ms_this_rank": res_f["vocoder_seconds"] * 1e3,
            "device_ms_this_rank": {"mel_gen": res_f["timings"]["total_ms"], "vocoder": voc_dev_ms, "wall": res_f["seconds"] * 1e3},
            "two_call_form": {"seconds_this_rank": res["seconds"], "vocoder_ms_this_rank": voc_ms,
                              "note": "xdtts_tacotron2_infer_batch then xdtts_griffinlim_infer_batch, the mel crossing the host (the reference's own call sequence per utterance, src/lib.rs:123,141)"},
            "per_rank": per_rank,
        }
    if not args.no_extras:
        # ---- configs[4]: Griffin-Lim only, 1000 frames ------------------------------------------------
        if rank == 0:
          try:
            F5 = 1000
            S5 = wl.chirp_magnitude(F5)   # |STFT| of five linear chirps 100 Hz - 7 kHz + white noise, 255 744 samples
            vocoder.set_seed(3)           # initial phase: the counter stream, seed 3
            c5 = {"workload": "BASELINE.json configs[4]: Griffin-Lim only, 513 x %d magnitude of the chirp signal (255 744 samples), phase seed 3" % F5, "runs": []}
            for iters in (30, 60, 120):
                for _ in range(3):
                    a5 = vocoder.infer_linear(S5, iters=iters)
                ms = vocoder.last_timings()["iterations_ms"]
                gbs = GL_BYTES_PER_FRAME_ITER * F5 * iters / (ms * 1e-3) / 1e9
                c5["runs"].append({"iterations": iters, "device_ms": ms, "us_per_iteration": ms * 1e3 / (iters + 1), "audio_samples_per_s": a5.size / (ms * 1e-3),
                                   "roofline": {"kernel": "k_gl_persistent<4> (one launch: all iterations + final ISTFT)", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                                "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}})
            c5["note"] = "state (S, angles, previous spectrum) lives in LDS for the whole call, so `achieved` is algorithmic bytes per second (12 308 B per frame per iteration, SURVEY 8d); measured HBM traffic: profiles/"
            extra["config5"] = c5
            log("config5 done")
          except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
            extra["config5"] = {"error": repr(e)}
        # ---- a handful of concurrent sentences: lock-step batches of 3..16 chunks on their own engines (17: the batched engine beside them) ----
        if rank == 0:
          try:
            sb = {"workload": "lock-step decoder loop of B chunks of 60..99 ids, 200 frames each (src/phonemes.rs:677-680: the batched / parallel sentences at the size "
                              "a server with a few concurrent utterances has), decoder loop alone", "runs": []}
            for B in (3, 4, 8, 9, 12, 16, 17):
                chunks = [wl.synth_ids(60 + (7 * b) % 40, seed=10 + b) for b in range(B)]
                o = pkg.default_opts(dropout_seed=1)
                for _ in range(2):
                    model.infer_batch(chunks, opts=o, fixed_steps=[200] * B)
                ms = model.last_timings()["decoder_ms"]
                sb["runs"].append({"chunks": B, "us_per_iteration": ms * 1e3 / 200, "mel_frames_per_s": B * 200 / (ms * 1e-3)})
            sb["engine"] = ("3..8 chunks: k_decoder_persistent8, 9..16: k_decoder_persistent16 (one persistent launch, LSTMs of all chunks on the matrix cores); "
                            "17: the two-launch batched engine") if model.engine_state()["decoder_persistent8"] == 1 else "fallback engines"
            # four sentences that arrive together: one xdtts_synthesize_batch call against four xdtts_synthesize_ids calls
            import time as _t
            four = [wl.synth_ids(95, seed=30 + u) for u in range(4)]
            o4 = pkg.default_opts(fixed_frames_per_id=wl.FRAMES_PER_ID, dropout_seed=0)
            st4 = [[int(round(wl.FRAMES_PER_ID * 95))]] * 4
            for _ in range(2):
                pkg.synthesize_batch(model, vocoder, [[u] for u in four], opts=o4, fixed_steps=st4, want_mels=False)
            t0 = _t.perf_counter()
            for _ in range(3):
                _, au = pkg.synthesize_batch(model, vocoder, [[u] for u in four], opts=o4, fixed_steps=st4, want_mels=False)
            tb = (_t.perf_counter() - t0) / 3
            for u in four:
                pkg.synthesize(model, vocoder, u, opts=o4)
            t0 = _t.perf_counter()
            for _ in range(3):
                for u in four:
                    pkg.synthesize(model, vocoder, u, opts=o4)
            ts = (_t.perf_counter() - t0) / 3
            fr = 4 * st4[0][0]
            sb["four_sentences"] = {"what": "4 utterances of 95 ids (633 frames each), 60-iteration Griffin-Lim: one batched call against four single calls, host wall clock",
                                    "batched_call_ms": tb * 1e3, "four_single_calls_ms": ts * 1e3, "mel_frames_per_s_batched": fr / tb, "mel_frames_per_s_single_calls": fr / ts}
            extra["small_batches"] = sb
            log("small batches done")
          except Exception as e:  # noqa: BLE001
            extra["small_batches"] = {"error": repr(e)}
    if extra:
        out["extra"] = extra

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(parity_dir)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)}
            if parity_dir:
                out["parity"] = parity_check(parity_dir)
                out["parity"]["timed_audio_vs_scaled_60it_audio_rms"] = timed_audio_is_that_audio   # (the timed call's audio = that audio x the G6 level)
                log("parity leg done")
                import shutil

                shutil.rmtree(parity_dir, ignore_errors=True)
            else:
                out["parity"] = None
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
