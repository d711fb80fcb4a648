import sys, torch
sys.path.insert(0, "/root/repo")
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 2
opts.IMAGE_SIZES["kitti_raw"] = (128, 256)
opts.CONV_DTYPE = "fp32"
torch.manual_seed(0)
weights = {"cmbL1": 5.0, "cmbSSIM": 0.5, "smoothe": 1.0}
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, weights, opts.SCALE_WEIGHT_T1, opts.JOINT_NET, ckpt_name="__test__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, None, optimizer)
feats = dataset.batches[0]
flat = optimizer.flat
flat.grad.zero_()
preds, loss, by = trainer.forward_backward(feats)
torch.cuda.synchronize()
print("loss", float(loss), {k: float(v) for k, v in by.items()})
names = {id(p): f"{net}.{n}" for net, m in model.models.items() for n, p in m.named_parameters()}
tot = {}
for p, off in zip(flat.params, flat.offsets):
    g = flat.grad[off:off + p.numel()]
    net = names[id(p)].split(".")[0]
    tot[net] = tot.get(net, 0.0) + float(g.abs().sum())
print("grad abs sums", tot)
print("pose", preds["pose"][0, 0].tolist())
before = flat.data.clone()
out = trainer.run_a_batch(feats)
torch.cuda.synchronize()
mv = {}
for p, off in zip(flat.params, flat.offsets):
    net = names[id(p)].split(".")[0]
    d = float((flat.data[off:off + p.numel()] - before[off:off + p.numel()]).abs().max())
    mv[net] = max(mv.get(net, 0.0), d)
print("moved", mv, "numel", flat.numel)
offs = [(names[id(p)].split(".")[0], off) for p, off in zip(flat.params, flat.offsets)]
for net in ("depthnet", "posenet", "flownet"):
    o = [x for n, x in offs if n == net]
    print("offsets", net, min(o), max(o))
print("grad after step (should be zeroed)", float(flat.grad.abs().sum()))

trainer_g, _ = tv.train_val_factory("graph", model, loss_object, 0, False, None, optimizer)
for it in range(4):
    before = flat.data.clone()
    out = trainer_g.run_a_batch(feats)
    torch.cuda.synchronize()
    mv = {}
    for p, off in zip(flat.params, flat.offsets):
        net = names[id(p)].split(".")[0]
        d = float((flat.data[off:off + p.numel()] - before[off:off + p.numel()]).abs().max())
        mv[net] = max(mv.get(net, 0.0), d)
    print("graph step", it, "loss", float(out[1]), "moved", mv, "fallback", trainer_g._graph.eager_fallback)

print("---- fresh model, graph trainer first (as the test does)")
torch.manual_seed(0)
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, weights, opts.SCALE_WEIGHT_T1, opts.JOINT_NET, ckpt_name="__test__")
trainer_g, _ = tv.train_val_factory("graph", model, loss_object, 0, False, None, optimizer)
flat = optimizer.flat
names = {id(p): f"{net}.{n}" for net, m in model.models.items() for n, p in m.named_parameters()}
feats = dataset.batches[0]
for it in range(4):
    before = flat.data.clone()
    out = trainer_g.run_a_batch(feats)
    torch.cuda.synchronize()
    mv = {}
    for p, off in zip(flat.params, flat.offsets):
        net = names[id(p)].split(".")[0]
        d = float((flat.data[off:off + p.numel()] - before[off:off + p.numel()]).abs().max())
        mv[net] = max(mv.get(net, 0.0), d)
    print("fresh graph step", it, "loss", float(out[1]), "moved", mv, "fallback", trainer_g._graph.eager_fallback, "pose", out[0]["pose"][0, 0, :3].tolist())

lo, hi = flat.data.data_ptr(), flat.data.data_ptr() + flat.data.numel() * 4
for net, m in model.models.items():
    ps = list(m.parameters())
    inside = sum(1 for q in ps if lo <= q.data_ptr() < hi)
    print("aliasing", net, inside, "of", len(ps), "parameters live in the flat buffer")
