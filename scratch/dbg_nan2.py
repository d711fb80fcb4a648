import sys, torch
sys.path.insert(0, "/root/repo")
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opts.CONV_DTYPE = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("graph", model, loss_object, 0, False, aug, optimizer)
for i in range(8):
    preds, loss, by = trainer.run_a_batch(dataset.batches[i % 4])
    torch.cuda.synchronize()
    print(i, float(loss), {k: round(float(v), 5) for k, v in by.items()}, "param finite", bool(torch.isfinite(optimizer.flat.data).all()),
          "grad finite", bool(torch.isfinite(optimizer.flat.grad).all()), "m finite", bool(torch.isfinite(optimizer.m).all()),
          "depth finite", bool(torch.isfinite(preds["depth_ms"][0]).all()), "pose", preds["pose"][0,0].tolist())
