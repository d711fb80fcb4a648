"""Per-shape timing of the depthwise-conv and BatchNorm kernels at the bench shapes (B=8, 128x416)."""
import sys, collections, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.hip import ops, lib as _lib
from xpt_mde_2021_amd.model import model_main as mm, train_val as tv

dw_shapes, bn_shapes = collections.Counter(), collections.Counter()
orig_dw, orig_bn = ops._DepthwiseConv.forward, ops._AffineAct.forward
def rec_dw(ctx, x, weight, stride, pt, pb, pl, pr, relu_in):
    dw_shapes[(tuple(x.shape), weight.shape[-1], stride, (pt, pb, pl, pr), bool(relu_in))] += 1
    return orig_dw(ctx, x, weight, stride, pt, pb, pl, pr, relu_in)
def rec_bn(ctx, x, gamma, beta, mean, var, eps, slope, relu_in, residual):
    bn_shapes[(tuple(x.shape), gamma is not None, float(slope), bool(relu_in), residual is not None, str(x.dtype))] += 1
    return orig_bn(ctx, x, gamma, beta, mean, var, eps, slope, relu_in, residual)
ops._DepthwiseConv.forward = staticmethod(rec_dw)
ops._AffineAct.forward = staticmethod(rec_bn)
opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 8
opts.CONV_DTYPE = "bf16"
dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, opts.RIGID_NET, ckpt_name="__dbg__")
trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, None, optimizer)
with torch.no_grad():
    pass
trainer.run_a_batch(dataset.batches[0])
torch.cuda.synchronize()
ops._DepthwiseConv.forward = staticmethod(orig_dw)
ops._AffineAct.forward = staticmethod(orig_bn)

def timeit(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

lib = _lib.load()
class _S:
    def __index__(self): return torch.cuda.current_stream().cuda_stream
tot = collections.defaultdict(float)
print("== depthwise: shape k stride relu n | fwd bwd_data bwd_weight(partials) us | MB moved fwd")
for (shape, k, stride, pad, relu), n in sorted(dw_shapes.items()):
    B, C, H, W = shape
    pt, pb, pl, pr = pad
    x = torch.randn(shape, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, 1, k, k, device="cuda")
    OH, OW = (H + pt + pb - k) // stride + 1, (W + pl + pr - k) // stride + 1
    y = torch.empty((B, C, OH, OW), device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    nchunk = lib.xpt_dwconv_bwd_weight_chunks(B, OH, OW, C, k, stride)
    ws = torch.empty(nchunk * C * k * k, device="cuda")
    f = timeit(lambda: lib.xpt_dwconv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, C, k, stride, pt, pl, OH, OW, int(relu), 1, torch.cuda.current_stream().cuda_stream))
    bd = timeit(lambda: lib.xpt_dwconv_bwd_data(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, H, W, C, k, stride, pt, pl, OH, OW, int(relu), 1, torch.cuda.current_stream().cuda_stream))
    bws = []
    for grp in (0, 4, 8, 16):
        lib.xpt_dwconv_tune(grp)
        nchunk = lib.xpt_dwconv_bwd_weight_chunks(B, OH, OW, C, k, stride)
        ws = torch.empty(nchunk * C * k * k, device="cuda")
        bws.append(timeit(lambda: lib.xpt_dwconv_bwd_weight_partials(x.data_ptr(), dy.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, C, k, stride, pt, pl, OH, OW, int(relu), 1, torch.cuda.current_stream().cuda_stream)))
        tot[f"dw_bw_g{grp}"] += bws[-1] * n
    lib.xpt_dwconv_tune(0)
    bw = bws[0]
    mb = (x.numel() + y.numel()) * 2 / 1e6
    tot["dw_fwd"] += f * n; tot["dw_bd"] += bd * n; tot["dw_bw"] += bw * n
    print(f"{str(shape):22s} k{k} s{stride} r{int(relu)} x{n:3d} | {f:6.1f} {bd:6.1f} | bw grp auto/4/8/16: " + " ".join(f"{t:6.1f}" for t in bws) + f" | {mb:6.2f} MB", flush=True)
print("== affine: shape gamma slope relu_in residual dtype n | fwd bwd(partials) us")
for (shape, has_g, slope, relu, res, dt), n in sorted(bn_shapes.items()):
    B, C, H, W = shape
    dtype = torch.bfloat16 if "bfloat16" in dt else torch.float32
    d = 1 if dtype == torch.bfloat16 else 0
    x = torch.randn(shape, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    gam = torch.rand(C, device="cuda") + .5 if has_g else None
    bet, mean, var = torch.randn(C, device="cuda"), torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + .5
    r = torch.randn_like(x) if res else None
    rows = B * H * W
    nblk = lib.xpt_affine_act_bwd_blocks(rows, C)
    ws = torch.empty(nblk * 2 * C, device="cuda")
    P = lambda t: None if t is None else t.data_ptr()
    f = timeit(lambda: lib.xpt_affine_act_fwd(P(x), P(gam), P(bet), P(mean) if has_g else None, P(var) if has_g else None, 1e-3, P(r), P(y), rows, C, slope, int(relu), d, torch.cuda.current_stream().cuda_stream))
    b = timeit(lambda: lib.xpt_affine_act_bwd_partials(P(x), P(y), P(dy), C, P(gam), P(bet), P(mean) if has_g else None, P(var) if has_g else None, 1e-3, P(dx), P(ws), ws.numel(), rows, C, slope, int(relu), d, torch.cuda.current_stream().cuda_stream))
    tot["bn_fwd"] += f * n; tot["bn_bwd"] += b * n
    print(f"{str(shape):22s} g{int(has_g)} s{slope} r{int(relu)} res{int(res)} {dt[6:]:9s} x{n:3d} | {f:6.1f} {b:6.1f} | blocks {nblk}", flush=True)
print({k: round(v) for k, v in tot.items()})
