import sys, torch
sys.path.insert(0, "/root/repo")
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
from torch.profiler import profile, ProfilerActivity
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.CONV_DTYPE = "bf16"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, aug, optimizer)
for i in range(3): trainer.run_a_batch(dataset.batches[i % 4])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for i in range(2): trainer.run_a_batch(dataset.batches[i % 4])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
