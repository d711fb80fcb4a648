#!/usr/bin/env python3
"""Steady-state time of every C-ABI launch of one training step, WITHOUT a profiler (rocprofv3's kernel trace puts a floor
of ~4.7 us under every dispatch: tools/lab/node_floor5.py; a serial hipGraph chain of tiny kernels costs 1.7 us per node).

One eager step at bench.py's configuration is recorded call by call (function name + arguments of every libxpt_hip.so
entry point); each recorded call is then replayed R times back to back inside one captured hipGraph and timed with HIP
events: its time per launch with hot caches, launch floor included.  Operands are whatever lies at the recorded addresses
by then (the kernels are dense: no data-dependent addressing), results are garbage and the model is not used again.

Usage: tools/hot_replay.py [--repeats 20] [--top 40] > gpurun_out/hot_replay.txt"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--repeats", type=int, default=20)
ap.add_argument("--top", type=int, default=45)
ap.add_argument("--batch", type=int, default=8)
cli = ap.parse_args()

sys.argv = [sys.argv[0], "--mode", "eager", "--batch", str(cli.batch)]
args = bench.parse()
trainer, dataset, mode, _ = bench.build_step(args, 1)
from xpt_mde_2021_amd.hip import lib as _lib, ops  # noqa: E402

for i in range(3):
    trainer.run_a_batch(dataset.batches[i % len(dataset.batches)])
torch.cuda.synchronize()

lib = _lib.load()
calls = []
originals = {}
stream_now = torch.cuda.current_stream().cuda_stream


def recorder(name, fn):
    def wrapped(*a):
        calls.append((name, fn, a))
        return fn(*a)
    return wrapped


for name in _lib.SIGNATURES:
    fn = getattr(lib, name)
    originals[name] = fn
    setattr(lib, name, recorder(name, fn))
trainer.run_a_batch(dataset.batches[0])
torch.cuda.synchronize()
for name, fn in originals.items():
    setattr(lib, name, fn)

# torch.cuda.graph() empties the allocator's cache on entry: the recorded addresses of tensors the step has freed since
# would be unmapped under the replayed kernels.  The cache stays where it is for the rest of this process.
torch.cuda.empty_cache = lambda: None
launches = [(n, f, a) for n, f, a in calls if a and a[-1] == stream_now]
print(f"# recorded {len(calls)} C-ABI calls, {len(launches)} of them launches on the step's stream", flush=True)


def timed(fn, a, repeats):
    def run():
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(repeats):
            fn(*a[:-1], s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * repeats)


rows = []
for idx, (name, fn, a) in enumerate(launches):
    try:
        us = timed(fn, a, cli.repeats)
    except Exception as e:           # noqa: BLE001
        print(f"# {idx} {name}: replay failed: {type(e).__name__}: {e}", flush=True)
        torch.cuda.synchronize()
        continue
    ints = [v for v in a[:-1] if isinstance(v, int) and 0 < v < (1 << 24)]
    rows.append((idx, name, us, ints[:8]))
total = sum(r[2] for r in rows)
print(f"# hot sum over {len(rows)} launches: {total / 1e3:.3f} ms  (launch floor 1.7 us each = {1.7e-3 * len(rows):.3f} ms)")
by = collections.defaultdict(lambda: [0, 0.0])
for _, name, us, _ in rows:
    by[name][0] += 1
    by[name][1] += us
print("\n| entry point | launches | hot ms/step | avg us |\n|---|---|---|---|")
for name, (cnt, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:cli.top]:
    print(f"| `{name}` | {cnt} | {us / 1e3:.3f} | {us / cnt:.1f} |")
print("\n# in launch order: index, entry point, hot us, small integer arguments")
for idx, name, us, ints in rows:
    print(f"{idx:4d} {name:40s} {us:7.1f}  {ints}")
