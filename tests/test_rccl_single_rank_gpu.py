"""RCCL on the one card of the test box: the data-parallel trainer with a REAL `nccl` (= RCCL on ROCm) process group of
world size 1.  The sequence that only RCCL can exercise -- graph replay, an `async_op` all-reduce on RCCL's own stream,
second graph replay out of the same memory pool, second all-reduce, waits, fused Adam (train_val.ModelTrainerDistrib,
reference: model/model_util/distributer.py:5-44, model/train_val.py:105-118, losses.py:49) -- runs for eight training
steps in a fresh child process and must reproduce the one-graph trainer bit for bit; two asynchronous all-reduce calls
per step must have been issued to RCCL."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, steps, **env):
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                              "LOCAL_WORLD_SIZE", "XPT_DIST_BACKEND")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "determinism_train.py"), mode, "noaug", str(steps)],
                         capture_output=True, text=True, timeout=900, env=dict(clean, **env), cwd=ROOT)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    return {l.split()[0]: l.split()[1:] for l in run.stdout.splitlines() if l and l.split()[0].isupper()}


def test_two_graph_step_over_a_one_rank_rccl_group_equals_the_single_graph_step(gpu_device):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    single = _run("graph", 8)
    rccl = _run("distributed", 8, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(port), XPT_DP_OVERLAP="1", XPT_DP_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    assert rccl["PROCESS_GROUP"][:2] == ["nccl", "1"], rccl.get("PROCESS_GROUP")
    assert rccl["TWO_PHASE"][0] == "True" and rccl["TWO_PHASE"][1] == "_GraphPair", rccl["TWO_PHASE"]
    # every step: the decoder / PoseNet bucket between the two replays and the encoder bucket after the second one, both async
    # (the first step also captures and replay-checks the graph pair: more than one exchange there)
    assert rccl["ALLREDUCE_PER_STEP"][1:] == ["2+0"] * 7, rccl["ALLREDUCE_PER_STEP"]
    assert int(rccl["ALLREDUCE_PER_STEP"][0].split("+")[0]) >= 2
    assert rccl["LOSSES"][2:] == single["LOSSES"][2:], f"one graph {single['LOSSES']}\\nRCCL two graphs {rccl['LOSSES']}"
    assert rccl["PARAMSUM"] == single["PARAMSUM"]
