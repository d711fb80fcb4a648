import sys, torch
sys.path.insert(0, "/root/repo")
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.CONV_DTYPE = "bf16"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, aug, optimizer)
names = []
for net, m in model.models.items():
    for n, p in m.named_parameters():
        if p.requires_grad: names.append((net + "." + n, p))
flat = optimizer.flat
graph = tv._StepGraph(trainer.forward_backward, state=trainer.optimizer_state)
for i in range(4):
    feats = dataset.batches[i % 4]
    flat.grad.zero_()
    trainer.forward_backward(feats)
    torch.cuda.synchronize()
    g_eager = flat.grad.clone()
    flat.grad.zero_()
    graph(feats)
    torch.cuda.synchronize()
    g_graph = flat.grad.clone()
    flat.grad.zero_()
    bad = ~torch.isfinite(g_graph)
    diff = (g_graph - g_eager).abs()
    rel = diff / (g_eager.abs().max() + 1e-20)
    print("step", i, "eager finite", bool(torch.isfinite(g_eager).all()), "graph bad", int(bad.sum()), "max abs diff", float(diff[~bad].max()), "gmax", float(g_eager.abs().max()))
    worst = []
    for (n, p), off in zip(names, flat.offsets):
        d = diff[off:off + p.numel()]; ge = g_eager[off:off + p.numel()]
        b = bad[off:off + p.numel()]
        sc = float(ge.abs().max()) + 1e-30
        m = float(torch.nan_to_num(d, nan=float("inf")).max())
        worst.append((m / sc, n, tuple(p.shape), int(b.sum()), sc))
    worst.sort(reverse=True)
    for w in worst[:6]: print("   ", w)
    flat.grad.copy_(g_eager)
    optimizer.apply_gradients()
