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
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
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
