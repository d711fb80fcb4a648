"""Rehearsal of the data-parallel GPU path on a one-GPU box: bench.py started the way the driver starts it for N > 1
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), two ranks sharing the one card, with gloo
standing in for RCCL (XPT_DIST_BACKEND).  Everything else is the real distributed trainer: parameter broadcast, per-rank
data, captured forward+backward per rank, all-reduce of the flat gradient buffer, fused Adam, max-over-ranks timing and
rank 0's JSON line.  (The reference: model/model_util/distributer.py:5-44 MirroredStrategy, losses.py:49 global-batch
averaging.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_card_train_in_step(gpu_device):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, XPT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--batch", "2", "--height", "64", "--width", "192", "--no-cpu-baseline", "--no-roofline"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, run.stdout[-2000:]                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["global_batch"] == 4
    assert out["config"]["mode"] == "distributed" and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and 0.0 < out["config"]["final_loss"] < 10.0
    # the decoder / PoseNet gradients (the larger bucket) are exchanged while the encoder's backward runs
    exchange = out["config"]["grad_exchange"]
    assert exchange["buckets"] == 2 and 0.5 * exchange["bytes"] < exchange["overlapped_bytes"] < exchange["bytes"]


def test_bench_launches_its_own_ranks(gpu_device):
    """`python bench.py --gpus 2` with no launcher around it: the parent must start the ranks itself before touching the GPU
    and relay rank 0's line (the round-1 scaling run died here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(XPT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "2",
           "--height", "64", "--width", "192", "--no-cpu-baseline", "--no-roofline"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["steps"] == 3
