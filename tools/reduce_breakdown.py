"""Where xpt_reduce_partials spends its time: the deferred-gradient jobs of one bench step, re-run class by class
(by output count) with their own job tables and timed with HIP events -> MB, us, GB/s and workgroups per class."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.hip import ops, lib as _lib
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.CONV_DTYPE = "bf16"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, None, optimizer)
sink = ops.grad_sink
orig_flush = sink.flush
seen = {}
def flush():
    seen["pending"] = list(sink.pending)
    return orig_flush()
sink.flush = flush
trainer.run_a_batch(dataset.batches[0]); torch.cuda.synchronize()
pending = seen["pending"]
lib = _lib.load()

def timed(sub, reps=20):
    keep = sink.scratch
    tables = sink._build(lib, sub)
    s = torch.cuda.current_stream()
    def run():
        for jobs, blockmap, nblocks in tables:
            _lib.check(lib.xpt_reduce_partials(jobs.data_ptr(), blockmap.data_ptr(), nblocks, s.cuda_stream), "reduce")
    for _ in range(3): run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): run()
    b.record(); torch.cuda.synchronize()
    sink.scratch = keep
    return a.elapsed_time(b) * 1e3 / reps, sum(t[2] for t in tables), len(tables)

def report(name, sub):
    if not sub: return
    mb = sum(n * ns * 4 for (_, _, _, n, ns, _) in sub) / 1e6
    us, wgs, passes = timed(sub)
    print(f"[reduce] {name:28s} jobs {len(sub):4d}  {mb:7.1f} MB  {us:7.1f} us  {mb / us * 1e3 / 1e3:6.2f} TB/s  wgs {wgs:6d}  passes {passes}", flush=True)

report("all", pending)
edges = [(0, 1024), (1024, 16384), (16384, 262144), (262144, 1 << 40)]
for lo, hi in edges:
    report(f"n in [{lo},{hi})", [p for p in pending if lo <= p[3] < hi])
for lo, hi in [(0, 9), (9, 65), (65, 257), (257, 1 << 30)]:
    report(f"splits in [{lo},{hi})", [p for p in pending if lo <= p[4] < hi])
al = lambda p: p[3] >= 256 and p[3] % 4 == 0 and p[0].data_ptr() % 16 == 0 and (p[1].data_ptr() + 4 * p[2]) % 16 == 0 and p[5] % 4 == 0 and p[4] > 1
report("wide-eligible", [p for p in pending if al(p)])
report("not wide", [p for p in pending if not al(p)])
for p in sorted(pending, key=lambda p: -p[3])[:12]:
    print(f"[job] n {p[3]:8d} splits {p[4]:4d} stride {p[5]:8d} dst%16 {p[0].data_ptr() % 16} src%16 {(p[1].data_ptr() + 4 * p[2]) % 16} wide {al(p)}")
import collections
why = collections.Counter()
for p in pending:
    if al(p): continue
    why["n<256" if p[3] < 256 else "n%4" if p[3] % 4 else "dst" if p[0].data_ptr() % 16 else "src" if (p[1].data_ptr() + 4 * p[2]) % 16 else "stride" if p[5] % 4 else "ns==1"] += 1
print("[not wide]", dict(why))
for fm in (32, 48, 64, 128):
    for grp in (256, 512):
        ops.GradSink.FLAT_MAX, ops.GradSink.GROUP = fm, grp
        report(f"all FLAT_MAX={fm} GROUP={grp}", pending)
