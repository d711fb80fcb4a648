import sys, collections, torch
sys.path.insert(0, "/root/repo")
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
from torch.profiler import profile, ProfilerActivity
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.CONV_DTYPE = "bf16"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, None, optimizer)
feats = dataset.batches[0]
preds, loss, by = trainer.run_a_batch(feats)
for _ in range(2): tv.merge_results(feats, preds, loss, by, False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    tv.merge_results(feats, preds, loss, by, False)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    for k in ev.kernels:
        agg[(k.name[:70], ev.name)][0] += 1; agg[(k.name[:70], ev.name)][1] += k.duration
tot_n = sum(v[0] for v in agg.values()); tot_t = sum(v[1] for v in agg.values())
print("metrics launches", tot_n, "kernel us", round(tot_t, 1))
for (kn, op), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{n:4d} {us:8.1f}us {kn:70s} {op}")
