#!/usr/bin/env python3
"""NaN-poisoning run of the training step: every torch.empty / empty_like / new_empty float buffer that the host side
of the HIP path allocates is filled with NaN before use, so a kernel that reads memory nobody wrote (pitch padding,
workspace tails, partial slots of an unused split) turns the loss / gradients non-finite deterministically instead of
once in a while, depending on what the allocator hands out.

    python tools/nan_poison.py [eager|graph] [fp32|bf16] [H W B] [rigid|flow|joint]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

_empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty
POISON = float(os.environ.get("POISON", "nan"))


def _poison(t):
    if t.is_floating_point() and t.is_cuda and t.numel():
        t.fill_(POISON)
    return t


torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: _poison(_new_empty(self, *a, **k))

from xpt_mde_2021_amd.config import opts  # noqa: E402
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
H, W, B = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (64, 192, 2)
nets = sys.argv[6] if len(sys.argv) > 6 else "rigid"
net_names, loss_weights = {"rigid": (opts.RIGID_NET, opts.LOSS_RIGID_T1), "flow": (opts.FLOW_NET, opts.LOSS_FLOW),
                           "joint": (opts.JOINT_NET, {"cmbL1": 5.0, "cmbSSIM": 0.5, "smoothe": 1.0})}[nets]
opts.CONV_DTYPE = dtype
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = B
opts.IMAGE_SIZES["kitti_raw"] = (H, W)
opts.TRAIN_MODE = mode
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, loss_weights, opts.SCALE_WEIGHT_T1,
                                                              net_names, ckpt_name="__poison__")
trainer, _ = tv.train_val_factory(mode, model, loss_object, 0, False, None, optimizer)
bad = 0
for i in range(6):
    out = trainer.run_a_batch(dataset.batches[i % len(dataset.batches)])
    torch.cuda.synchronize()
    flat = optimizer.flat
    fin = bool(torch.isfinite(flat.data).all()) and bool(torch.isfinite(optimizer.m).all())
    loss = float(out[1])
    print(f"[{mode} {dtype} {H}x{W} b{B}] step {i} loss {loss:.6f} params finite {fin}", flush=True)
    bad += (not fin) or (loss != loss)
print("POISON RESULT:", "READS UNWRITTEN MEMORY" if bad else "clean", flush=True)
sys.exit(1 if bad else 0)
