"""Partial-sum volume of the deferred parameter gradients of one bench step (what xpt_reduce_partials has to read)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.hip import ops
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.CONV_DTYPE = "bf16"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, None, optimizer)
sink = ops.grad_sink
orig_flush = sink.flush
stats = {}
def flush():
    pend = list(sink.pending)
    by_tag = collections.defaultdict(lambda: [0, 0, 0])
    for (dst, src, off, n, ns, st) in pend:
        key = "n<1k" if n < 1024 else "n<16k" if n < 16384 else "n<256k" if n < 262144 else "n>=256k"
        by_tag[key][0] += 1; by_tag[key][1] += n * ns * 4; by_tag[key][2] += ns
    stats["by"] = {k: (v[0], round(v[1] / 1e6, 1), round(v[2] / max(v[0], 1), 1)) for k, v in by_tag.items()}
    stats["total_mb"] = sum(n * ns * 4 for (_, _, _, n, ns, _) in pend) / 1e6
    stats["jobs"] = len(pend)
    stats["big"] = sorted(((n * ns * 4 / 1e6, n, ns) for (_, _, _, n, ns, _) in pend), reverse=True)[:12]
    return orig_flush()
sink.flush = flush
trainer.run_a_batch(dataset.batches[0]); torch.cuda.synchronize()
print("[sink] jobs", stats["jobs"], "total MB", round(stats["total_mb"], 1))
print("[sink] by output size (jobs, MB, mean splits):", stats["by"])
print("[sink] largest (MB, n, splits):", [(round(a, 1), b, c) for a, b, c in stats["big"]])
