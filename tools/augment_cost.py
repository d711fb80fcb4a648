#!/usr/bin/env python3
"""Cost of the in-step augmentation: the bench configuration captured with and without the augmenter."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
opts.CONV_DTYPE = "bf16"
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
for use_aug in (True, False):
    torch.manual_seed(0)
    dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
    model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1,
                                                                  opts.RIGID_NET, ckpt_name="__aug__")
    trainer, _ = tv.train_val_factory("graph", model, loss_object, 0, False, aug if use_aug else None, optimizer)
    for i in range(6):
        trainer.run_a_batch(dataset.batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        trainer.run_a_batch(dataset.batches[i % 4])
    torch.cuda.synchronize()
    print(f"AUG {'on ' if use_aug else 'off'}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step", flush=True)
