"""Utterance sharding for multi-GPU runs.

Utterances -- and the <=100-id chunks inside one (src/tacotron2/mod.rs:422-434) -- share no state,
so the N GPUs of a node each take a static share and no collective sits on the data path.  The
only communication is a barrier and one all-gather of per-rank counters at the end (RCCL on GPUs,
gloo in the CPU tests).
"""
import numpy as np


def shard_utterances(lengths, rank, world):
    """Static round-robin over the length-sorted order (longest first), so every rank gets a
    similar mix of long and short utterances; each rank's share is returned longest-first, which
    keeps the padding waste of its lock-step batches low.  Returns indices into `lengths`."""
    lengths = np.asarray(lengths)
    order = np.argsort(-lengths, kind="stable")
    return [int(i) for i in order[rank::world]]


def gather_counters(local, dist=None, device="cpu"):
    """All-gathers {frames, samples, seconds} and returns (totals, max_seconds, per_rank).
    `dist` is torch.distributed (already initialised) or None for a single process."""
    keys = ("frames", "samples", "seconds")
    vec = [float(local[k]) for k in keys]
    if dist is None or not dist.is_initialized():      # (a world of one rank still goes through the collective: bench.py XDTTS_BENCH_FORCE_DIST)
        rows = [vec]
    else:
        import torch

        t = torch.tensor(vec, dtype=torch.float64, device=device)
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        rows = [[float(x) for x in o.cpu()] for o in out]
    per_rank = [dict(zip(keys, r)) for r in rows]
    totals = {k: sum(r[k] for r in per_rank) for k in ("frames", "samples")}
    max_seconds = max(r["seconds"] for r in per_rank)
    return totals, max_seconds, per_rank


def plan_share(chunker, utterances, rank, world, frames_per_id):
    """This rank's share of `utterances` (id arrays) as a lock-step batch: returns (share, chunks, steps,
    owner) -- utterance indices (longest first), their <=window-id chunks in order (`chunker(ids)` is
    find_splits + the trailing split, src/tacotron2/mod.rs:399,412-414), the fixed frame count of each
    chunk (round(frames_per_id * len)), and the owning utterance of each chunk."""
    share = shard_utterances([len(u) for u in utterances], rank, world)
    chunks, steps, owner = [], [], []
    for u in share:
        for c in chunker(utterances[u]):
            chunks.append(c)
            steps.append(int(np.floor(frames_per_id * len(c) + 0.5)))
            owner.append(u)
    return share, chunks, steps, owner


def run_share(model, vocoder, share, chunks, steps, owner, opts, fused=None):
    """XdTts::infer (src/lib.rs:110-159) for every utterance of a rank's share: all chunks through ONE
    batched mel-gen call (chunks are independent, src/tacotron2/mod.rs:422-434), then per utterance the
    chunk mels concatenated on the time axis (mod.rs:430) and the vocoder (lib.rs:141).  `fused` is the
    one-call form of the same (xdtts_synthesize_batch through the package's synthesize_batch: the mel
    stays in HBM between the halves); None = the two batch entry points with the mel crossing the host.
    Returns the counters gather_counters() takes plus the mel-gen / vocoder split."""
    import time

    t0 = time.perf_counter()
    if fused is not None:
        groups, gsteps = [], []
        for u in share:
            idx = [i for i in range(len(chunks)) if owner[i] == u]
            assert idx == list(range(idx[0], idx[0] + len(idx))), "an utterance's chunks are consecutive"
            groups.append([chunks[i] for i in idx])
            gsteps.append([steps[i] for i in idx])
        umels, outs = fused(model, vocoder, groups, opts=opts, fixed_steps=gsteps)
        t2 = time.perf_counter()
        timings = model.last_timings()
        mel_s = min(timings.get("total_ms", 0.0) * 1e-3, t2 - t0)  # device time of the mel-gen half (HIP events)
        audio = dict(zip(share, outs))
        return {"frames": int(sum(m.shape[1] for m in umels)), "samples": int(sum(a.size for a in outs)), "seconds": t2 - t0,
                "mel_gen_seconds": mel_s, "vocoder_seconds": (t2 - t0) - mel_s, "timings": timings, "mels": umels, "audio": audio}
    mels = model.infer_batch(chunks, opts=opts, fixed_steps=steps)
    timings = model.last_timings()
    t1 = time.perf_counter()
    samples, audio = 0, {}
    umels = [np.concatenate([mels[i] for i in range(len(chunks)) if owner[i] == u], axis=1) for u in share]
    if hasattr(vocoder, "infer_batch"):   # utterances share the vocoder's persistent launches
        outs = vocoder.infer_batch(umels)
    else:
        outs = [vocoder.infer(m) for m in umels]
    for u, a in zip(share, outs):
        audio[u] = a
        samples += a.size
    t2 = time.perf_counter()
    return {"frames": int(sum(m.shape[1] for m in mels)), "samples": int(samples), "seconds": t2 - t0, "mel_gen_seconds": t1 - t0,
            "vocoder_seconds": t2 - t1, "timings": timings, "mels": mels, "audio": audio}
