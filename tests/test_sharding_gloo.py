"""world_size-2 gloo test of the N>1 path: static utterance sharding + the end-of-run counter
all-gather (the only communication; there is no data-path collective)."""
import importlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lengths, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = importlib.import_module("xd-tts_amd.shard")
    mine = shard.shard_utterances(lengths, rank, world)
    frames = sum(int(round(6.67 * lengths[i])) for i in mine)
    local = {"frames": frames, "samples": 256 * (frames - len(mine)), "seconds": 0.5 + 0.25 * rank}
    totals, max_s, per_rank = shard.gather_counters(local, dist)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array(mine))
    if rank == 0:
        np.save(os.path.join(out_dir, "tot.npy"), np.array([totals["frames"], totals["samples"], max_s, len(per_rank)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_covers_every_utterance_once(tmp_path):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ["PYTHONPATH"] = root + os.pathsep + os.environ.get("PYTHONPATH", "")
    rng = np.random.Generator(np.random.PCG64(2))
    lengths = rng.integers(40, 201, size=64)  # BASELINE.json configs[2]/[3]: 40-200 phonemes
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), lengths, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert sorted(list(a) + list(b)) == list(range(64))  # every utterance exactly once
    assert abs(len(a) - len(b)) <= 1
    assert abs(int(lengths[a].sum()) - int(lengths[b].sum())) <= int(lengths.max())  # balanced work
    assert list(lengths[a]) == sorted(lengths[a], reverse=True)  # longest first within a rank
    tot = np.load(tmp_path / "tot.npy")
    frames = sum(int(round(6.67 * l)) for l in lengths)
    assert tot[0] == frames and tot[1] == 256 * (frames - 64)
    assert tot[2] == 0.75 and tot[3] == 2  # time = MAX over ranks


def test_single_process_path():
    shard = importlib.import_module("xd-tts_amd.shard")
    assert shard.shard_utterances([5, 9, 7], 0, 1) == [1, 2, 0]
    totals, max_s, per = shard.gather_counters({"frames": 10, "samples": 2304, "seconds": 1.5})
    assert totals == {"frames": 10.0, "samples": 2304.0} and max_s == 1.5 and len(per) == 1
