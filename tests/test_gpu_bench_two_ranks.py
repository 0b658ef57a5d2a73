"""The N > 1 control flow of bench.py with REAL handles, before an 8-GPU node ever sees it (SURVEY.md 8(e),
BASELINE.json configs[3]): two ranks launched by torch.distributed.run exactly as the driver does, both pinned
to GPU 0 through bench.py's developer override (XDTTS_BENCH_DEVICE=0; gloo replaces RCCL, which refuses two
ranks on one GPU; the library's cross-process chip lock makes their co-resident launches take turns).  Every
rank loads through Tacotron2::load(dir) (mod.rs:242) from the container rank 0 wrote."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_on_one_gpu(pkg, tmp_path):
    import importlib

    wl = importlib.import_module("xd-tts_amd.workloads")
    shard = importlib.import_module("xd-tts_amd.shard")
    env = dict(os.environ, XDTTS_BENCH_DEVICE="0", XDTTS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--check-shared-utterance", "--load-from-dir", str(tmp_path / "model")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["cpu_baseline"] is None
    assert out["value"] > 0 and out["roofline"]["frac"] > 0 and out["roofline"]["frac_of_floor"] > 0   # (the floor is measured while the other rank shares the GPU: no upper bound here)
    assert os.path.exists(tmp_path / "model" / "tacotron2.xdtw")
    # the engines stayed on their fast paths (no exchange timed out into a fallback while the two ranks shared the GPU)
    assert "timed out" not in r.stderr and "refused" not in r.stderr, r.stderr[-3000:]

    # configs[3]'s share pipeline: both ranks reported, and the counts are the plan's
    c4 = out["extra"]["config4"]
    assert "error" not in c4, c4
    per_rank = c4["per_rank"]
    assert len(per_rank) == 2 and all(p["seconds"] > 0 for p in per_rank)
    all_utts = wl.config4(pkg, n_batches=2)
    want_frames = want_samples = 0
    for rank in range(2):
        share, chunks, steps, owner = shard.plan_share(lambda ids: wl.chunk_utterance(pkg, ids), all_utts, rank, 2, wl.FRAMES_PER_ID_BATCH)
        assert len(share) == 32
        fr = sum(steps)
        per_utt = {}
        for o, st in zip(owner, steps):
            per_utt[o] = per_utt.get(o, 0) + st
        sm = sum(256 * (f - 1) for f in per_utt.values())
        assert per_rank[rank]["frames"] == fr and per_rank[rank]["samples"] == sm, (rank, per_rank[rank], fr, sm)
        want_frames += fr
        want_samples += sm
    assert abs(c4["mel_frames_per_s"] * c4["seconds_max_over_ranks"] - want_frames) < 1e-3 * want_frames
    assert abs(c4["audio_samples_per_s"] * c4["seconds_max_over_ranks"] - want_samples) < 1e-3 * want_samples

    # the shared utterance: the same bits on both ranks and in this (third) process
    sh = out["extra"]["shared_utterance"]["per_rank"]
    assert [p["rank"] for p in sh] == [0, 1]
    assert sh[0]["mel_sha256"] == sh[1]["mel_sha256"] and sh[0]["audio_sha256"] == sh[1]["audio_sha256"]
    model = pkg.Tacotron2.synthetic(seed=wl.WEIGHT_SEED, rec_scale=1.0)
    voc = pkg.create_griffin_lim(iters=60, seed=0)
    ids = wl.synth_ids(120, seed=1)
    sp = np.cumsum([len(c) for c in wl.config2(pkg)[1]]).astype(np.int64)
    mel, audio = pkg.synthesize(model, voc, ids, splits=sp, opts=pkg.default_opts(fixed_frames_per_id=wl.FRAMES_PER_ID, dropout_seed=0, item_base=0))
    assert (sh[0]["frames"], sh[0]["samples"]) == (mel.shape[1], audio.size) == (800, 204544)
    assert hashlib.sha256(np.ascontiguousarray(mel).tobytes()).hexdigest() == sh[0]["mel_sha256"]
    assert hashlib.sha256(np.ascontiguousarray(audio).tobytes()).hexdigest() == sh[0]["audio_sha256"]
    # gate-on variant ran on both ranks too
    g = out["extra"]["headline_gate_on"]
    assert "error" not in g and g["frames"] == 800 and g["frames_equal_fixed_steps_run"]
    voc.close()
    model.close()


def test_bench_two_ranks_with_broadcast_weights(pkg):
    """SURVEY 8(e)'s one load-time collective: rank 0 builds the 113 MB weight blob and broadcasts it, every rank loads from what it
    received (gloo here; RCCL GPU 0 -> all on a multi-GPU node).  The shared utterance must come out with the bits of a handle
    that generated the weights itself."""
    import importlib

    wl = importlib.import_module("xd-tts_amd.workloads")
    env = dict(os.environ, XDTTS_BENCH_DEVICE="0", XDTTS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--check-shared-utterance", "--broadcast-weights"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    sh = out["extra"]["shared_utterance"]["per_rank"]
    assert out["n_gpus"] == 2 and len(sh) == 2 and sh[0]["audio_sha256"] == sh[1]["audio_sha256"] and sh[0]["mel_sha256"] == sh[1]["mel_sha256"]
    model = pkg.Tacotron2.synthetic(seed=wl.WEIGHT_SEED, rec_scale=1.0)
    voc = pkg.create_griffin_lim(iters=60, seed=0)
    sp = np.cumsum([len(c) for c in wl.config2(pkg)[1]]).astype(np.int64)
    mel, audio = pkg.synthesize(model, voc, wl.synth_ids(120, seed=1), splits=sp, opts=pkg.default_opts(fixed_frames_per_id=wl.FRAMES_PER_ID, dropout_seed=0, item_base=0))
    assert hashlib.sha256(np.ascontiguousarray(mel).tobytes()).hexdigest() == sh[0]["mel_sha256"]
    assert hashlib.sha256(np.ascontiguousarray(audio).tobytes()).hexdigest() == sh[0]["audio_sha256"]
    assert np.array_equal(model.blob(), pkg.Tacotron2.from_blob(model.blob()).blob())
    voc.close()
    model.close()
