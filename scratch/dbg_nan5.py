import sys, torch
sys.path.insert(0, "/root/repo")
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.CONV_DTYPE = "bf16"
mode = sys.argv[1]
torch.manual_seed(0)
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, aug, optimizer)
names = []
for net, m in model.models.items():
    for n, p in m.named_parameters():
        if p.requires_grad: names.append((net + "." + n, p))
flat = optimizer.flat
fb = tv._StepGraph(trainer.forward_backward, state=trainer.optimizer_state) if mode == "graph" else trainer.forward_backward
out = {}
for i in range(5):
    feats = dataset.batches[i % 4]
    _, loss, _ = fb(feats)
    torch.cuda.synchronize()
    g = flat.grad.clone()
    big = (~torch.isfinite(g)) | (g.abs() > 1e4)
    print("step", i, "loss", float(loss), "n big", int(big.sum()), "gmax", float(torch.nan_to_num(g).abs().max()), "gsum", float(torch.nan_to_num(g).double().abs().sum()))
    for (n, p), off in zip(names, flat.offsets):
        s = big[off:off + p.numel()]
        if s.any(): print("    ", n, tuple(p.shape), int(s.sum()), g[off:off + p.numel()][s][:3].tolist())
    out[i] = g.cpu()
    optimizer.apply_gradients()
torch.save(out, f"gpurun_out/grads_{mode}.pt")
