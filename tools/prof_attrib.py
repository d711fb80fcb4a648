"""Attribute the small elementwise launches of one eager training step to source lines."""
import sys, collections, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
from torch.profiler import profile, ProfilerActivity
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.CONV_DTYPE = "bf16"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, None, optimizer)
for i in range(3): trainer.run_a_batch(dataset.batches[i % 4])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    trainer.run_a_batch(dataset.batches[0])
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.kernels:
        continue
    stack = [s for s in (ev.stack or []) if "xpt_mde_2021_amd" in s or "autograd" in s.lower()][:3]
    for k in ev.kernels:
        key = (k.name[:60], ev.name, " <- ".join(s.split("/")[-1] for s in stack))
        agg[key][0] += 1
        agg[key][1] += k.duration
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
for (kn, op, st), (n, us) in rows[:70]:
    print(f"{n:5d} {us:9.1f}us  {kn:60s} {op:32s} {st}")
print("total launches", sum(v[0] for v in agg.values()))
print("---- copy_/add_/add/mul/fill_ by shapes")
agg2 = collections.defaultdict(int)
for ev in prof.events():
    if ev.kernels and ev.name.startswith("aten::") and ev.name not in ("aten::mm", "aten::miopen_convolution", "aten::convolution_backward"):
        par = ev.cpu_parent
        chain = []
        while par is not None and len(chain) < 4:
            chain.append(par.name)
            par = par.cpu_parent
        agg2[(ev.name, ev.kernels[0].name[:50], str(ev.input_shapes)[:60], " < ".join(chain)[:110])] += 1
for k, n in sorted(agg2.items(), key=lambda kv: -kv[1])[:140]:
    print(n, k)
