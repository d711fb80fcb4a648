#!/usr/bin/env python3
"""Deferred parameter-gradient partials of one training step: bytes the producing launches write and xpt_reduce_partials reads back,
per destination (largest first).   python tools/sink_census.py [TOP]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.config import opts  # noqa: E402
from xpt_mde_2021_amd.hip import ops  # noqa: E402
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv  # noqa: E402

TOP = int(sys.argv[1]) if len(sys.argv) > 1 else 40
opts.CONV_DTYPE = "bf16"
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.TRAIN_MODE = "eager"
torch.manual_seed(0)
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1,
                                                              opts.RIGID_NET, ckpt_name="__census__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, aug, optimizer)
names = {}
for net_name, net in model.models.items():
    for pname, p in net.named_parameters():
        fg = getattr(p, "flat_grad", None)
        if fg is not None:
            names[fg.data_ptr()] = f"{net_name}.{pname} {tuple(p.shape)}"
rows = []
sink_cls = ops.GradSink
orig = sink_cls.flush


def flush(self):
    for d, s, off, n, ns, st in self.pending:
        rows.append((n * ns * 4, n, ns, names.get(d.data_ptr(), hex(d.data_ptr()))))
    return orig(self)


sink_cls.flush = flush
trainer.run_a_batch(dataset.batches[0])
rows.clear()
trainer.run_a_batch(dataset.batches[1])
torch.cuda.synchronize()
total = sum(r[0] for r in rows)
print(f"jobs {len(rows)}  partial bytes {total / 1e6:.1f} MB  gradient bytes {sum(r[1] for r in rows) * 4 / 1e6:.1f} MB")
rows.sort(reverse=True)
acc = 0
for b, n, ns, name in rows[:TOP]:
    acc += b
    print(f"{b / 1e6:8.2f} MB  n {n:8d}  splits {ns:5d}  cum {acc / total:5.2f}  {name}")
