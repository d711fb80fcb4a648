#!/usr/bin/env python3
"""Probe for the intermittent "non-finite state on graph replay" failure: captures the training step in a fresh process
(fresh MIOpen user db, so the fast find chooses solvers anew) and, when the replay check fires, names the parameters
whose values / moments went non-finite.

    python tools/replay_flake_probe.py [fp32|bf16] [H W B] [rigid|flow|joint]     exit code 0 = clean, 3 = check fired
"""
import os
import sys
import tempfile

if os.environ.get("PROBE_DEFAULT_DB") != "1":
    os.environ.setdefault("MIOPEN_USER_DB_PATH", tempfile.mkdtemp(prefix="miopen_udb_"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.config import opts  # noqa: E402
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
H, W, B = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (64, 192, 2)
nets = sys.argv[5] if len(sys.argv) > 5 else "rigid"
net_names, loss_weights = {"rigid": (opts.RIGID_NET, opts.LOSS_RIGID_T1), "flow": (opts.FLOW_NET, opts.LOSS_FLOW),
                           "joint": (opts.JOINT_NET, {"cmbL1": 5.0, "cmbSSIM": 0.5, "smoothe": 1.0})}[nets]
opts.CONV_DTYPE = dtype
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = B
opts.IMAGE_SIZES["kitti_raw"] = (H, W)
opts.TRAIN_MODE = "graph"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, loss_weights, opts.SCALE_WEIGHT_T1,
                                                              net_names, ckpt_name="__probe__")
trainer, _ = tv.train_val_factory("graph", model, loss_object, 0, False, aug if os.environ.get("AUG") == "1" else None,
                                  optimizer)
fired = []


def probing_check(self, state, saved, replays=4):
    ok = True
    print('[probe] replay check running', flush=True)
    for rep in range(replays):
        self.graph.replay()
        torch.cuda.synchronize()
        flat = optimizer.flat
        names = {id(q): f"{net}.{n}" for net, m in model.models.items() for n, q in m.named_parameters()}
        for p, off in zip(flat.params, flat.offsets):
            for label, buf in (("data", flat.data), ("m", optimizer.m), ("v", optimizer.v)):
                t = buf[off:off + p.numel()]
                if not bool(torch.isfinite(t).all()) or float(t.abs().max()) > 1e8:
                    fired.append((rep, f"{names[id(p)]}[{label}]", tuple(p.shape)))
                    ok = False
        for t, s in zip(state, saved):
            t.copy_(s)
    torch.cuda.synchronize()
    return None if ok else 'probe: non-finite state'


tv._StepGraph._replay_report = probing_check
try:
    for i in range(3):
        out = trainer.run_a_batch(dataset.batches[0])
    torch.cuda.synchronize()
    print(f"[probe {dtype} {H}x{W} b{B}] loss {float(out[1]):.6f}", flush=True)
except RuntimeError as e:
    print(f"[probe] {e}", flush=True)
if fired:
    reps = sorted({r for r, _, _ in fired})
    names = sorted({(n, s) for _, n, s in fired})
    print(f"[probe] REPLAY CHECK FIRED at replays {reps}: {len(names)} parameters, first: {names[:24]}", flush=True)
    sys.exit(3)
print("[probe] clean", flush=True)
