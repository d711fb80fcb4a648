#!/usr/bin/env python3
"""Is one training step a pure function of (weights, batch)?  Runs forward+backward of the bench configuration several
times WITHOUT an optimizer step and compares the flat gradient bit for bit inside the process; writes per-parameter
checksums to a JSON file so that two processes can be compared too (tools/determinism_probe.py --compare a.json b.json).

    python tools/determinism_probe.py out.json [repeats] [bf16|fp32] [B H W]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "--compare":
    a, b = (json.load(open(f)) for f in sys.argv[2:4])
    print("loss", a["loss"], b["loss"])
    bad = 0
    for name in a["order"]:
        x, y = a["params"][name], b["params"][name]
        if x != y:
            bad += 1
            if bad <= 40:
                print(f"  differs: {name:70s} {x} vs {y}")
    print(f"{bad} of {len(a['order'])} parameters differ between the two processes")
    sys.exit(0)

import torch  # noqa: E402

from xpt_mde_2021_amd.config import opts  # noqa: E402
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/determinism.json"
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
B, H, W = (int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (8, 128, 416)
opts.CONV_DTYPE = dtype
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = B
opts.IMAGE_SIZES["kitti_raw"] = (H, W)
torch.manual_seed(0)
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1,
                                                              opts.RIGID_NET, ckpt_name="__det__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, None, optimizer)
flat = optimizer.flat
names = {id(p): f"{net}.{n}" for net, m in model.models.items() for n, p in m.named_parameters() if p.requires_grad}
feats = dataset.batches[0]
grads, losses = [], []
for it in range(repeats):
    # dirty the caching allocator between runs: a kernel that reads memory it did not write sees different bytes
    junk = [torch.full((1 << 22,), float("nan") if it % 2 else 1e30, device="cuda") for _ in range(8)]
    del junk
    flat.grad.zero_()
    _, loss, _ = trainer.forward_backward(feats)
    torch.cuda.synchronize()
    grads.append(flat.grad.clone())
    losses.append(float(loss))
print("losses", losses)
first = grads[0]
print("finite:", bool(torch.isfinite(first).all()))
for it in range(1, repeats):
    bad = []
    for p, off in zip(flat.params, flat.offsets):
        a, b = grads[it][off:off + p.numel()], first[off:off + p.numel()]
        if not torch.equal(a, b):
            scale = max(float(b.abs().max()), 1e-30)
            bad.append((names[id(p)], tuple(p.shape), float((a - b).abs().max()) / scale))
    print(f"run {it} vs run 0: {len(bad)} of {len(flat.params)} parameters differ")
    for name, shape, err in bad[:30]:
        print(f"    {name:70s} {shape} rel {err:.3e}")
order = [names[id(p)] for p in flat.params]
sums = {names[id(p)]: float(first[off:off + p.numel()].double().abs().sum()) for p, off in zip(flat.params, flat.offsets)}
json.dump({"loss": losses[0], "order": order, "params": sums}, open(out, "w"))
