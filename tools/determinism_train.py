#!/usr/bin/env python3
"""Loss sequence of K full training steps (forward, backward, fused Adam) from the seeded initial weights: two processes
must print the same numbers.   python tools/determinism_train.py [eager|graph|distributed] [aug|noaug] [K]
(distributed with XPT_DP_OVERLAP=1 and no other rank: the two-graph step of the data-parallel trainer)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.config import opts  # noqa: E402
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv  # noqa: E402

import torch.distributed as dist  # noqa: E402

# (launched with RANK / WORLD_SIZE in the environment -- e.g. WORLD_SIZE=1 on a one-GPU box -- the distributed trainer
#  creates its process group: backend nccl = RCCL; every collective it issues is counted)
_calls = {"async": 0, "sync": 0}
_all_reduce = dist.all_reduce


def _counting_all_reduce(tensor, *a, **kw):
    _calls["async" if kw.get("async_op") else "sync"] += 1
    return _all_reduce(tensor, *a, **kw)


dist.all_reduce = _counting_all_reduce
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
use_aug = (sys.argv[2] if len(sys.argv) > 2 else "aug") == "aug"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 12
opts.CONV_DTYPE = os.environ.get("XPT_DET_DTYPE", "bf16")
if opts.CONV_DTYPE in ("bf16", "fp16"):
    from xpt_mde_2021_amd.hip import lib as _xlib
    _xlib.set_half_format(opts.CONV_DTYPE)
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.TRAIN_MODE = mode
torch.manual_seed(0)
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1,
                                                              opts.RIGID_NET, ckpt_name="__det__")
trainer, _ = tv.train_val_factory(mode, model, loss_object, 0, False, aug if use_aug else None, optimizer)
losses, per_step = [], []
for i in range(K):
    before = dict(_calls)
    out = trainer.run_a_batch(dataset.batches[i % len(dataset.batches)])
    losses.append(float(out[1]))
    per_step.append((_calls["async"] - before["async"], _calls["sync"] - before["sync"]))
torch.cuda.synchronize()
flat = optimizer.flat
if dist.is_initialized():
    print("PROCESS_GROUP", dist.get_backend(), dist.get_world_size(),
          ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else "-")
    print("ALLREDUCE_PER_STEP", " ".join(f"{a}+{s}" for a, s in per_step))
graph = getattr(trainer, "_graph", None)
print("CAPTURED", graph is not None and graph.graph is not None, "library" if getattr(graph, "library_path", False) else "own-kernels",
      getattr(graph, "census", None))
if mode == "distributed":
    print("TWO_PHASE", trainer._early_start is not None, type(trainer._graph.graph).__name__)
print("LOSSES", mode, "aug" if use_aug else "noaug", " ".join(f"{v:.9f}" for v in losses))
print("PARAMSUM", f"{float(flat.data.double().abs().sum()):.9f}" if hasattr(flat, "data") else "")

# ---- structurally-zero entries (pretrained_nets.NASNetMobileEncoder.structural_pads, depth_net.DepthNetPretrained): after K
# captured bf16 steps every padded entry must still be EXACTLY 0.0 in the fp32 master weights, their bf16 shadow and both Adam
# moments (running variances: 1.0) -- the "same function as the 11 / 22-filter cells" invariant on the GPU path
def _pad_mask(t, o, i):
    keep = torch.zeros(t.shape[:2] if (t.dim() > 1 and i is not None) else t.shape[:1], dtype=torch.bool)
    oo = o if o is not None else torch.arange(t.shape[0])
    if keep.dim() == 2:
        keep[oo[:, None], i[None, :]] = True
    else:
        keep[oo] = True
    return (~keep).to(t.device).view(*keep.shape, *([1] * (t.dim() - keep.dim()))).expand_as(t)


pads, bad, entries = [], [], 0
for net in model.models.values():
    for m in net.modules():
        fn = getattr(m, "structural_pads", None)
        if callable(fn):
            pads.extend(fn())
index = {id(p): k for k, p in enumerate(flat.params)} if hasattr(flat, "params") else {}
for t, o, i in pads:
    mask = _pad_mask(t, o, i)
    if not bool(mask.any()):
        continue
    is_var = t.dim() == 1 and float(t.detach()[mask].float().min()) == 1.0 and float(t.detach()[mask].float().max()) == 1.0
    views = {"master": t.detach()}
    if id(t) in index:
        k = index[id(t)]
        off = flat.offsets[k]
        views["adam_m"] = flat._view(optimizer.m, t, off)
        views["adam_v"] = flat._view(optimizer.v, t, off)
        if getattr(t, "shadow_bf16", None) is not None:
            views["shadow_bf16"] = t.shadow_bf16
        if t.grad is not None:
            views["grad"] = t.grad
    for what, v in views.items():
        vals = v[mask].float()
        entries += vals.numel()
        if what == "master" and is_var:
            continue
        if float(vals.abs().max()) != 0.0:
            bad.append((what, tuple(t.shape), float(vals.abs().max())))
print("STRUCTZERO", f"tensors {len(pads)} entries {entries} nonzero {len(bad)}", bad[:4])
if os.environ.get("XPT_DET_CHECKPOINT"):
    import tempfile
    from xpt_mde_2021_amd.model.build_model import pretrained_nets as pn
    with tempfile.TemporaryDirectory() as tmp:
        model.save_weights(tmp, "det")
        disk = torch.load(os.path.join(tmp, "depthnet_det.pt"))
        enc = model.models["depthnet"].encoder
        exported = pn.export_keras_weights(enc)
        print("CHECKPOINT", "stem1", tuple(disk["encoder.cells.0.conv.weight"].shape), "stem2", tuple(disk["encoder.cells.1.conv.weight"].shape),
              "up2.conv2", tuple(disk["up2.conv2.conv.weight"].shape), "keras_elements", sum(v.numel() for v in exported.values()))
