"""Per-shape timing of the 1x1-conv weight gradient: split-K MFMA kernel (several launch plans) vs rocBLAS."""
import sys, collections, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.hip import ops, lib as _lib
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
from xpt_mde_2021_amd.model.build_model import pretrained_nets as pn

shapes = collections.Counter()
orig = ops.conv1x1_weight_grad
def rec(dy2, x2):
    shapes[(dy2.shape[0], dy2.shape[1], x2.shape[1])] += 1
    return orig(dy2, x2)
ops.conv1x1_weight_grad = rec
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.CONV_DTYPE = "bf16"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, None, optimizer)
trainer.run_a_batch(dataset.batches[0])
torch.cuda.synchronize()
ops.conv1x1_weight_grad = orig
print("unique shapes", len(shapes), "calls", sum(shapes.values()))
lib = _lib.load()

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

plans = [(16, 16, 1024, 512), (16, 32, 1024, 512), (16, 16, 1024, 256), (16, 8, 1024, 256), (16, 16, 512, 256), (16, 32, 512, 256)]
tot = collections.defaultdict(float)
print("M cout cin n | rocblas | " + " | ".join(str(p) for p in plans))
for (M, co, ci), n in sorted(shapes.items()):
    dy = torch.randn(M, co, device="cuda").bfloat16(); x = torch.randn(M, ci, device="cuda").bfloat16()
    t_lib = timeit(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))
    row = []
    for pl in plans:
        assert lib.xpt_conv1x1_bwd_weight_tune(*pl) == 0
        t = timeit(lambda: ops.conv1x1_weight_grad(dy, x))
        row.append(t); tot[pl] += t * n
    tot["lib"] += t_lib * n
    print(f"{M:7d} {co:5d} {ci:5d} x{n:3d} | {t_lib:7.1f} | " + " | ".join(f"{t:7.1f}" for t in row), flush=True)
print("total us/step: rocblas %.0f" % tot["lib"], {str(k): round(v) for k, v in tot.items() if k != "lib"})
