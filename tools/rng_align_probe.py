#!/usr/bin/env python3
"""Does a hipGraph replay draw the same device random numbers as eager execution from the same generator state?
(The training step contains the device-side augmentation.)  Prints the loss of eager runs and replays started from
one saved CUDA generator state."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.config import opts  # noqa: E402
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv  # noqa: E402

opts.CONV_DTYPE = "bf16"
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.AUGMENT_PROBS = {"CropAndResize": 1.0, "HorizontalFlip": 0.5, "ColorJitter": 1.0}      # every draw matters
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1,
                                                              opts.RIGID_NET, ckpt_name="__rng__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, aug, optimizer)
feats = dataset.batches[0]
for _ in range(2):
    trainer.forward_backward(feats)
torch.cuda.synchronize()
rng = torch.cuda.get_rng_state()
eager = []
for _ in range(2):
    torch.cuda.set_rng_state(rng)
    eager.append(float(trainer.forward_backward(feats)[1]))
other = float(trainer.forward_backward(feats)[1])          # next draws: a different augmentation
graph = tv._StepGraph(trainer.forward_backward)
graph(feats)
replays = []
for _ in range(3):
    torch.cuda.set_rng_state(rng)
    replays.append(float(graph(feats)[1]))
print(f"[rng] eager from the saved state: {eager}; eager with the next draws: {other:.6f}; replays from the saved state: {replays}",
      flush=True)
