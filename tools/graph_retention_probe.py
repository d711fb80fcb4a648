#!/usr/bin/env python3
"""Which Python objects keep the autograd graph of a finished training step alive?  (A surviving graph keeps its
AccumulateGrad nodes -- and the stream they were created on -- alive, which autograd warns about during hipGraph
capture.)  Runs two eager steps, then lists the live tensors that still carry a grad_fn, before and after gc.collect().

    python tools/graph_retention_probe.py [fp32|bf16]
"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.config import opts  # noqa: E402
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
opts.CONV_DTYPE = dtype
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 2
opts.IMAGE_SIZES["kitti_raw"] = (64, 192)
opts.TRAIN_MODE = "eager"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1,
                                                              opts.RIGID_NET, ckpt_name="__probe__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, None, optimizer)
gc.collect()
gc.disable()


def live_graph_tensors():
    found = []
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.grad_fn is not None:
                found.append((tuple(o.shape), type(o.grad_fn).__name__))
        except Exception:
            pass
    return found


for step in range(2):
    out = trainer.run_a_batch(dataset.batches[0])
    torch.cuda.synchronize()
    del out
    before = live_graph_tensors()
    unreachable = gc.collect()
    after = live_graph_tensors()
    print(f"[retention] step {step}: {len(before)} tensors with grad_fn alive after the step "
          f"(gc found {unreachable} unreachable objects; {len(after)} left after gc)", flush=True)
    for shape, fn in before[:15]:
        print(f"    {fn:40s} {shape}", flush=True)
