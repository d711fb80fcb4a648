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
ap.add_argument("--roofline", type=int, default=20, help="per-kernel roofline rows for this many slowest launches (0: none)")
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
    rows.append((idx, name, us, ints[:8], a))
total = sum(r[2] for r in rows)
print(f"# hot sum over {len(rows)} launches: {total / 1e3:.3f} ms  (launch floor 1.7 us each = {1.7e-3 * len(rows):.3f} ms)")
by = collections.defaultdict(lambda: [0, 0.0])
for _, name, us, _, _ in rows:
    by[name][0] += 1
    by[name][1] += us
print("\n| entry point | launches | hot ms/step | avg us |\n|---|---|---|---|")
for name, (cnt, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:cli.top]:
    print(f"| `{name}` | {cnt} | {us / 1e3:.3f} | {us / cnt:.1f} |")
print("\n# in launch order: index, entry point, hot us, small integer arguments")
for idx, name, us, ints, _ in rows:
    print(f"{idx:4d} {name:40s} {us:7.1f}  {ints}")


# ---------------------------------------------------------------------------------------------- per-kernel roofline
# ALGORITHMIC work of a launch from its arguments: bytes = every operand read once + every result written once (bf16 = 2 B,
# fp32 = 4 B, split-K partials as written), FLOPs = 2 x multiply-accumulates.  The binding roof of a launch is the slower of
# bytes / 8 TB/s (HBM) and FLOPs / 2.5 PFLOP/s (dense bf16 MFMA); `frac` = that time / the measured hot time.
HBM, MFMA = 8e12, 2.5e15


def work_of(name, a):
    def i(k):
        v = a[k]
        return int(v.value) if hasattr(v, "value") else int(v)
    if name in ("xpt_conv2d_fwd",):
        B, PH, PW, C, N, KH, KW, OH, OW = i(4), i(5), i(6), i(7), i(9), i(10), i(11), i(15), i(16)
        return (B * PH * PW * C + B * OH * OW * N + N * KH * KW * C) * 2, 2.0 * B * OH * OW * N * KH * KW * C, f"{C}->{N} k{KH} out {OH}x{OW}"
    if name == "xpt_conv2d_fwd_splitk":
        B, PH, PW, C, N, KH, KW, OH, OW = i(4), i(5), i(6), i(7), i(9), i(10), i(11), i(14), i(15)
        return (B * PH * PW * C + B * OH * OW * N + N * KH * KW * C) * 2, 2.0 * B * OH * OW * N * KH * KW * C, f"{C}->{N} k{KH} out {OH}x{OW}"
    if name == "xpt_conv2d_fwd_stream":
        B, PH, PW, C, N, OH, OW = i(4), i(5), i(6), i(7), i(9), i(12), i(13)
        return (B * PH * PW * C + B * OH * OW * N + N * 9 * C) * 2, 2.0 * B * OH * OW * N * 9 * C, f"{C}->{N} k3 out {OH}x{OW}"
    if name in ("xpt_conv2d_bwd_data", "xpt_conv2d_bwd_data_splitk"):
        B, OH, OW, Np, C, KH, KW = i(3), i(4), i(5), i(6), i(8), i(9), i(10)
        IH, IW = (i(14), i(15)) if name == "xpt_conv2d_bwd_data" else (i(13), i(14))
        return (B * OH * OW * Np + B * IH * IW * C + Np * KH * KW * C) * 2, 2.0 * B * OH * OW * Np * KH * KW * C, f"dgrad {Np}->{C} k{KH} in {IH}x{IW}"
    if name == "xpt_conv2d_bwd_data_stream":
        B, OH, OW, Np, C, IH, IW = i(3), i(4), i(5), i(6), i(8), i(11), i(12)
        return (B * OH * OW * Np + B * IH * IW * C + Np * 9 * C) * 2, 2.0 * B * OH * OW * Np * 9 * C, f"dgrad {Np}->{C} k3 in {IH}x{IW}"
    if name == "xpt_conv2d_bwd_weight_partials":
        pf, B, PH, PW, C, N, KH, KW, OH, OW = i(3), i(4), i(5), i(6), i(7), i(10), i(12), i(13), i(17), i(18)
        return (B * PH * PW * C + B * OH * OW * N) * 2 + pf * 4, 2.0 * B * OH * OW * N * KH * KW * C, f"wgrad {C}x{N} k{KH} out {OH}x{OW}"
    if name == "xpt_headconv_fwd":
        B, H, W, C = i(5), i(6), i(7), i(8)
        return B * H * W * (C * 2 + 4), 2.0 * B * H * W * 9 * C, f"head {C}->1 {H}x{W}"
    if name == "xpt_headconv_bwd":
        B, H, W, C = i(7), i(8), i(9), i(10)
        return B * H * W * (C * 2 * 2 + 4) + i(6) * 4, 4.0 * B * H * W * 9 * C, f"head bwd {C} {H}x{W}"
    if name == "xpt_affine_act_bwd_partials":
        rows, C = i(12), i(13)
        return rows * C * 2 * 3, 0.0, f"act bwd [{rows}, {C}]"
    if name == "xpt_adam_step":
        return i(4) * 30, 0.0, f"{i(4)} parameters"
    if name == "xpt_concat_channels":
        return i(5) * i(6) * 2 * 2, 0.0, f"[{i(5)}, {i(6)}]"
    if name == "xpt_pwconv_bn_fwd":
        M, cin, cout = i(10), i(11), i(12)
        return M * (cin + 2 * cout) * 2 + cin * cout * 2, 2.0 * M * cin * cout, f"pw {cin}->{cout} M {M}"
    if name == "xpt_photo_march_ms_fwdbwd":
        n, B, N = i(0), i(13), i(14)
        px = sum(int(a[15][k]) * int(a[16][k]) for k in range(n))
        return B * px * (20 + 12 * N), 0.0, f"one-pass march, {n} scales"
    return None


if cli.roofline:
    print(f"\n# kernel roofline: the {cli.roofline} slowest launches of the step (hot us: replayed back to back, launch floor included)")
    print("| # | entry point | what | algorithmic MB | GFLOP | hot us | HBM floor us | MFMA floor us | bound | frac of roof |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for idx, name, us, ints, a in sorted(rows, key=lambda r: -r[2])[:cli.roofline]:
        try:
            w = work_of(name, a)
        except Exception:            # noqa: BLE001
            w = None
        if w is None:
            print(f"| {idx} | `{name}` | {ints} | n/a | n/a | {us:.1f} | | | | |")
            continue
        nbytes, flops, what = w
        t_hbm, t_mfma = nbytes / HBM * 1e6, flops / MFMA * 1e6
        bound = "hbm" if t_hbm >= t_mfma else "mfma"
        print(f"| {idx} | `{name}` | {what} | {nbytes / 1e6:.1f} | {flops / 1e9:.2f} | {us:.1f} | {t_hbm:.2f} | {t_mfma:.2f} | {bound} | "
              f"{max(t_hbm, t_mfma) / us:.3f} |")
