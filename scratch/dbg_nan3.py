import sys, torch
sys.path.insert(0, "/root/repo")
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.CONV_DTYPE = "bf16"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("graph", model, loss_object, 0, False, aug, optimizer)
names = []
for net, m in model.models.items():
    for n, p in m.named_parameters():
        if p.requires_grad: names.append((net + "." + n, p))
for i in range(4):
    preds, loss, by = trainer.run_a_batch(dataset.batches[i % 4])
    torch.cuda.synchronize()
    badm = ~torch.isfinite(optimizer.m)
    print("step", i, float(loss), "bad m count", int(badm.sum()))
    if badm.any():
        for (n, p), off in zip(names, optimizer.flat.offsets):
            seg = badm[off:off + p.numel()]
            if seg.any():
                print("  ", n, tuple(p.shape), "bad", int(seg.sum()), "of", p.numel(), "first idx", int(seg.nonzero()[0]))
        break
