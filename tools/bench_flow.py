#!/usr/bin/env python3
"""FlowNet branch timings on one MI355X: the correlation cost kernels per pyramid level (HIP events on the launch
stream, algorithmic bytes = both feature maps read once + the cost volume written once, and the mirror for the
backward) and the hipGraph training step of FLOW_NET (flowL2 + flow_reg) and JOINT_NET (cmbL1 + cmbSSIM + smoothe).

    python tools/bench_flow.py [H W B]            (H, W divisible by 64; default 128 384 8)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.config import opts  # noqa: E402
from xpt_mde_2021_amd.hip import lib as _lib, ops  # noqa: E402

H, W, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 384, 8)
N = 4
dev = torch.device("cuda:0")


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3          # us


print(f"correlation cost volume, {B * N} maps, bf16 (the dtype of the training step)")
for level, C in (() if os.environ.get("SKIP_CORR") == "1" else ((2, 32), (3, 64), (4, 96), (5, 128), (6, 196))):
    md = 128 // 2 ** level
    s2 = max(md // 4, 1)
    h, w = H // 2 ** level, W // 2 ** level
    l = torch.randn((B * N, C, h, w), device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    r = torch.randn_like(l)
    out = ops.correlation_cost(l, r, md, s2)
    DD = out.shape[1]
    g = torch.randn_like(out)
    dl, dr = torch.empty_like(l), torch.empty_like(r)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    fwd_us = timed(lambda: lib.xpt_corr_cost_fwd(l.data_ptr(), r.data_ptr(), out.data_ptr(), B * N, h, w, C, md, s2, 1, st))
    bwd_us = timed(lambda: lib.xpt_corr_cost_bwd(l.data_ptr(), r.data_ptr(), g.data_ptr(), dl.data_ptr(), dr.data_ptr(),
                                                 B * N, h, w, C, md, s2, 1, st))
    px = B * N * h * w
    fwd_bytes = px * (2 * C + DD) * 2
    bwd_bytes = px * (2 * C + DD + 2 * C) * 2
    print(f"  level {level}: {h}x{w}x{C} -> {DD} ch | fwd {fwd_us:6.1f} us ({fwd_bytes / fwd_us / 1e3:6.0f} GB/s) | "
          f"bwd {bwd_us:6.1f} us ({bwd_bytes / max(bwd_us, 1e-3) / 1e3:6.0f} GB/s)", flush=True)


def step_time(net_names, loss_weights, label, steps=20):
    from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
    opts.CONV_DTYPE = os.environ.get("DTYPE", "bf16")
    opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = B
    opts.IMAGE_SIZES["kitti_raw"] = (H, W)
    mode = os.environ.get("MODE", "graph")
    opts.TRAIN_MODE = mode
    opts.MIOPEN_FIND = os.environ.get("FIND", "1") == "1"
    dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
    model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, loss_weights, opts.SCALE_WEIGHT_T1,
                                                                  net_names, ckpt_name="__bench_flow__")
    trainer, _ = tv.train_val_factory(mode, model, loss_object, 0, False, None if os.environ.get("NO_AUG") == "1" else aug,
                                      optimizer)
    hist = []
    for i in range(5):
        hist.append(trainer.run_a_batch(dataset.batches[i % len(dataset.batches)])[1].clone())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = trainer.run_a_batch(dataset.batches[i % len(dataset.batches)])
        hist.append(out[1].clone())
    torch.cuda.synchronize()
    print("   losses:", " ".join(f"{float(v):.4f}" for v in hist), flush=True)
    ms = (time.perf_counter() - t0) / steps * 1e3
    params = sum(p.numel() for p in model.trainable_weights()) / 1e6
    print(f"{label}: {ms:.2f} ms/step = {B / ms * 1e3:.0f} snippets/s  ({params:.1f} M parameters, batch {B}, {H}x{W}, "
          f"{opts.CONV_DTYPE}, {mode}{' (PWC-Net trained eagerly)' if getattr(trainer, 'trains_flow_net', False) else ''}{' -> EAGER FALLBACK' if getattr(getattr(trainer, '_graph', None), 'eager_fallback', False) else ''}; loss {float(out[1]):.4f})",
          flush=True)


if os.environ.get("ONLY_CORR") == "1":
    sys.exit(0)
step_time(opts.FLOW_NET, opts.LOSS_FLOW, "FLOW_NET  flowL2 + flow_reg")
if os.environ.get("FLOW_ONLY") != "1":
    step_time(opts.JOINT_NET, {"cmbL1": 5.0, "cmbSSIM": 0.5, "smoothe": 1.0}, "JOINT_NET cmbL1 + cmbSSIM + smoothe")
