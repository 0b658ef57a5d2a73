"""bench.py's RCCL branch on real hardware.  The two-rank test replaces RCCL by gloo (RCCL refuses
two ranks on one GPU), so `init_process_group("nccl", device_id=...)`, the device-tensor all_reduce(MAX), gather_counters on
`cuda`, the `--broadcast-weights` broadcast of a `cuda` blob and the barriers had never executed anywhere.  With
XDTTS_BENCH_FORCE_DIST=1 a world_size-1 run under torch.distributed.run takes exactly that branch: RCCL initialises next to
libxdtts_hip.so in one process (the load-order hazard of tests/conftest.py), every collective runs on the GPU, and the
numbers must be the plain run's.  Utterance independence: /root/reference/src/tacotron2/mod.rs:422-434; SURVEY 8(e)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_rccl_branch_with_one_rank(pkg):
    args = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--check-shared-utterance"]
    env = dict(os.environ, XDTTS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("XDTTS_BENCH_BACKEND", None)
    env.pop("XDTTS_BENCH_DEVICE", None)
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")]
    rccl = subprocess.run(launch + args + ["--broadcast-weights"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    out = _line(rccl)
    assert "[bench" in rccl.stderr and "rccl: init_process_group(nccl) ok" in rccl.stderr, rccl.stderr[-3000:]   # the branch really ran
    assert "timed out" not in rccl.stderr and "refused" not in rccl.stderr, rccl.stderr[-3000:]
    plain = _line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600,
                                 env={k: v for k, v in os.environ.items() if not k.startswith("XDTTS_BENCH")}, cwd=ROOT))
    for o in (out, plain):
        assert o["n_gpus"] == 1 and o["steps"] == 3 and o["scaling"] == "weak" and o["value"] > 50_000
    assert 0.8 < out["value"] / plain["value"] < 1.25, (out["value"], plain["value"])            # the same ballpark
    a, b = out["extra"]["shared_utterance"]["per_rank"], plain["extra"]["shared_utterance"]["per_rank"]
    assert len(a) == len(b) == 1 and a[0]["mel_sha256"] == b[0]["mel_sha256"] and a[0]["audio_sha256"] == b[0]["audio_sha256"]   # broadcast blob = own blob
    c4 = out["extra"]["config4"]                                                                    # gather_counters over RCCL on cuda tensors
    assert "error" not in c4 and len(c4["per_rank"]) == 1 and c4["per_rank"][0]["seconds"] > 0
    p4 = plain["extra"]["config4"]["per_rank"][0]
    assert (c4["per_rank"][0]["frames"], c4["per_rank"][0]["samples"]) == (p4["frames"], p4["samples"])
    assert out["extra"]["headline_gate_on"]["frames_equal_fixed_steps_run"]
